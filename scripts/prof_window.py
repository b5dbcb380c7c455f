"""Per-kernel time inside the timed region of a rocprofv3 kernel-trace database of bench.py:
the window is the last steps*ms_per_step before the last dispatch."""
import sqlite3, sys, json, re
db = sqlite3.connect(sys.argv[1]); log = open(sys.argv[2]).read()
j = json.loads([l for l in log.splitlines() if l.startswith('{"metric"')][-1])
win = j['steps'] * j['ms_per_step'] * 1e6
rows = list(db.execute("select name, start, end from kernels order by start"))
t1 = max(r[2] for r in rows); t0 = t1 - win
agg = {}
for n, s, e in rows:
    if s < t0: continue
    n = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s)
tot = sum(v[1] for v in agg.values())
print('window %.1f ms, kernel time %.1f ms (%.0f%% busy), per step:' % (win / 1e6, tot / 1e6, 100 * tot / win))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print('%-34s %6.1f calls/step %8.1f us avg %7.2f ms/step %5.1f%%' % (n[:34], c / j['steps'], d / c / 1e3, d / j['steps'] / 1e6, 100 * d / tot))
