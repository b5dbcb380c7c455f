"""Per-kernel time inside the timed region of a rocprofv3 kernel-trace database of bench.py:
the window is the last steps*ms_per_step before the last dispatch."""
import sqlite3, sys, json, re
db = sqlite3.connect(sys.argv[1]); log = open(sys.argv[2]).read()
j = json.loads([l for l in log.splitlines() if l.startswith('{"metric"')][-1])
win = j['steps'] * j['ms_per_step'] * 1e6
rows = list(db.execute("select name, start, end from kernels order by start"))
t1 = max(r[2] for r in rows); t0 = t1 - win
agg = {}
prev = ''
for n, s, e in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')
    key = n
    if n.startswith('k_gs<'):   # the layout of a gather-scatter dispatch follows from the kernel before it (scripts/pmc_traffic.py)
        key = n + (' slab-permuted' if re.match(r'k_axhelm3rb?<.*true>|k_axhelm3c<\d+, true|k_rhs<\d+, true', prev) else ' face-grouped' if re.match(r'k_opgradt3<\d+, \d+, true|k_opgradt3[nw]<\d+, true|k_fdm|k_sch|k_q1', prev) else ' natural')
    if not n.startswith('__amd') and not n.startswith('k_cg_final') and not n.startswith('k_cg_post'): prev = n
    if s < t0: continue
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (e - s)
tot = sum(v[1] for v in agg.values())
print('window %.1f ms, kernel time %.1f ms (%.0f%% busy), per step:' % (win / 1e6, tot / 1e6, 100 * tot / win))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print('%-40s %6.1f calls/step %8.1f us avg %7.2f ms/step %5.1f%%' % (n[:40], c / j['steps'], d / c / 1e3, d / j['steps'] / 1e6, 100 * d / tot))
