"""Per-kernel means of one rocprofv3 --pmc counter (csv output) -> csv 'kernel,counter,dispatches,mean,max'."""
import csv, re, sys
rows = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
        a = rows.setdefault((n, r['Counter_Name']), [])
        a.append(float(r['Counter_Value']))
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'counter', 'dispatches', 'mean', 'max'])
for (n, c), v in sorted(rows.items()):
    w.writerow([n, c, len(v), round(sum(v) / len(v), 1), round(max(v), 1)])
