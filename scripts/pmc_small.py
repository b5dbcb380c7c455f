#!/usr/bin/env python3
"""Per-kernel means of every counter of one or more `rocprofv3 --pmc` passes (one directory per pass):
   pmc_small.py <dir> [<dir> ...]
Kernels are keyed by their template name; dispatches that return at once (SQ_WAVE_CYCLES or SQ_WAVES below 10 % of the
kernel's maximum: launches gated off by a converged solve) are left out.  SQ_* cycle counters are in quad-cycles summed
over all waves / CUs as rocprofv3 reports them (MI355X_MICROARCH.md, counter table); GRBM_GUI_ACTIVE is the sum over 8 XCDs."""
import collections, csv, glob, os, re, sys

def name(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    return n.split('(')[0][:44]

tab = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    per = collections.defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            per[(f, int(r["Dispatch_Id"]))].setdefault("_k", name(r["Kernel_Name"]))
            per[(f, int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
    gate = "SQ_WAVE_CYCLES" if any("SQ_WAVE_CYCLES" in v for v in per.values()) else "SQ_WAVES"
    mx = collections.defaultdict(float)
    for v in per.values():
        mx[v["_k"]] = max(mx[v["_k"]], v.get(gate, 0.0))
    for v in per.values():
        if v.get(gate, 0.0) < 0.1 * mx[v["_k"]]:
            continue
        for c, x in v.items():
            if c != "_k":
                tab[v["_k"]][c].append(x)
cols = sorted({c for v in tab.values() for c in v})
print("%-44s %6s " % ("kernel", "n") + " ".join("%18s" % c[:18] for c in cols))
def key(kv):
    return -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))
for k, v in sorted(tab.items(), key=key):
    n = max(len(x) for x in v.values())
    print("%-44s %6d " % (k, n) + " ".join("%18.4e" % (sum(v[c]) / len(v[c])) if c in v else "%18s" % "-" for c in cols))
