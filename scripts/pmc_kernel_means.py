#!/usr/bin/env python3
"""Mean of every collected counter over the dispatches of one kernel:  pmc_kernel_means.py <rocprofv3 -d dir> <regex on the kernel name>
(dispatches whose SQ_WAVE_CYCLES is below 10 % of the maximum -- launches gated off by a converged solve -- are left out)."""
import csv, glob, os, re, sys, collections
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if re.search(sys.argv[2], r["Kernel_Name"]):
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
if not rows:
    sys.exit("no dispatch matches")
key = "SQ_WAVE_CYCLES" if all("SQ_WAVE_CYCLES" in v for v in rows.values()) else None
mx = max(v[key] for v in rows.values()) if key else 0
live = [v for v in rows.values() if not key or v[key] >= 0.1 * mx]
print("kernel /%s/: %d dispatches, %d counted" % (sys.argv[2], len(rows), len(live)))
for c in sorted(live[0]):
    print("%-32s %.4e" % (c, sum(v[c] for v in live) / len(live)))
