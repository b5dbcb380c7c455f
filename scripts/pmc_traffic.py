#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), reproducibly.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-units --no-cpu
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-units --no-cpu
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --outdir profiles/r02_pmc --E 10000 --lx1 8 --dim 3

Writes  <outdir>/per_class.csv          one row per (kernel, call-site class): dispatches, gated, median / max of both counters
        <outdir>/traffic_per_launch.json what bench.py reads for `roofline.traffic`

Per-dispatch classification.  The gather-scatter kernel k_gs<3> runs in three element layouts that move different
amounts of memory for the same algorithmic bytes; the layout of a dispatch follows from the kernel dispatched before it
(same stream, in order):  k_axhelm3r<.., true> or k_rhs<.., true> -> x-planes-first (velocity PCG and its right-hand side);
k_opgradt3n<.., true> -> face-grouped (pressure operator, pressure correction);  anything else -> natural (set-up).
--mix-from <timed-region table of scripts/prof_window.py>: the per-launch figure of k_gs is the mean over its call sites weighted with
the calls per step of the TIMED REGION (the PMC run also contains the set-up, whose hundreds of natural-layout and face-grouped calls
would otherwise dominate the mean).
Launches issued after a PCG has converged return at once (device-side done flag): a dispatch whose counter is below 10 %
of its class maximum is counted as "gated" and left out of the statistics.
Corrections (MI355X_MICROARCH.md, section HBM): the counters are in KB; FETCH_SIZE reports half the bytes of wide
coalesced reads on gfx950 and is doubled; WRITE_SIZE is exact:  traffic = (2 * FETCH + WRITE) * 1024 bytes.
Calibrated in round 3 for the gather-scatter's own access shape (scripts/pmc_calibrate.hip, profiles/r03_pmc_calibration.txt):
FETCH_SIZE * 1024 = 0.500 of the bytes for coalesced 8-byte loads as for 16-byte ones, and 0.51 of the bytes of the whole
128-byte lines touched by k_gs's 8- and 16-byte pair accesses; WRITE_SIZE * 1024 = 1.000 of the bytes written.  The factor 2
therefore holds for every kernel here.
"""
import argparse
import csv
import glob
import json
import os
import re
import statistics
import sys

CLASSES = {      # bench.py kernel class -> regex on the demangled kernel name
    "gs": r"^k_gs<3>", "axhelm": r"^k_axhelm3[rc]?<", "opgradt": r"^k_opgradt3(<\d+, \d+|n<\d+), true", "opdiv": r"^k_opdiv3(<\d+, \d+|n<\d+), true",
    "block_dot": r"^k_block_dot<", "axpy_dot": r"^k_block_axpy_dot<", "block_axpy": r"^k_block_axpy$", "cg_vec": r"^k_cg_update<3>", "cg_update": r"^k_cg_update<3>",
    "conv": r"^k_conv3m?<", "fdm": r"^k_fdm_ext(<|_mfma8)",
}


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    depth, out = 0, ""
    for ch in n:                      # cut the argument list, keep template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out += ch
    return out.strip()


def load(path, counter):
    files = [path] if os.path.isfile(path) else sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))
    if not files:
        sys.exit("no *counter_collection.csv under %s" % path)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    return rows


def classify(rows):
    """-> list of (kernel, site, value) with the call-site class of every dispatch"""
    out, prev = [], ""
    for _, k, v in rows:
        site = ""
        if k.startswith("k_gs<"):
            if re.match(r"k_axhelm3rb?<\d+, \d+, true", prev) or re.match(r"k_axhelm3c<\d+, true", prev) or re.match(r"k_rhs<\d+, true", prev):
                site = "slab-permuted (velocity PCG)"
            elif re.match(r"k_opgradt3(<\d+, \d+|n<\d+), true", prev) or re.match(r"k_fdm|k_sch|k_q1", prev):
                site = "face-grouped (pressure operator / Schwarz exchange)"
            else:
                site = "natural"
        out.append((k, site, v))
        if not k.startswith("__amd") and not k.startswith("k_cg_final") and not k.startswith("k_cg_post"):
            prev = k
    return out


def stats(cl):
    groups = {}
    for k, site, v in cl:
        groups.setdefault((k, site), []).append(v)
    res = {}
    for key, vals in groups.items():
        mx = max(vals)
        live = [v for v in vals if v >= 0.1 * mx] if mx > 0 else vals
        res[key] = dict(dispatches=len(vals), gated=len(vals) - len(live), median=statistics.median(live), mean=sum(live) / len(live), max=mx)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--outdir", default="profiles/r03_pmc")
    ap.add_argument("--E", type=int, required=True)
    ap.add_argument("--lx1", type=int, required=True)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--mix-from", default=None)
    a = ap.parse_args()
    fs, ws = stats(classify(load(a.fetch, "FETCH_SIZE"))), stats(classify(load(a.write, "WRITE_SIZE")))
    os.makedirs(a.outdir, exist_ok=True)
    with open(os.path.join(a.outdir, "per_class.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "call_site", "dispatches", "gated", "fetch_kb_median", "fetch_kb_max", "write_kb_median", "write_kb_max",
                    "traffic_bytes_median = (2 fetch + write) 1024"])
        for key in sorted(fs):
            f, wv = fs[key], ws.get(key)
            if wv is None:
                continue
            w.writerow([key[0], key[1], f["dispatches"], f["gated"], round(f["median"], 1), round(f["max"], 1), round(wv["median"], 1),
                        round(wv["max"], 1), round((2 * f["median"] + wv["median"]) * 1024)])
    out = {"_note": "made by scripts/pmc_traffic.py from two rocprofv3 --pmc passes (see its header for the command lines, the "
                    "classification and the corrections); per-launch figures are call-weighted means over the non-gated dispatches "
                    "of the kernel, all call sites together (per call site: per_class.csv)",
           "config": {"E": a.E, "lx1": a.lx1, "dim": a.dim}}
    for cls, rx in CLASSES.items():
        keys = [k for k in fs if re.match(rx, k[0]) and k in ws]
        if not keys:
            continue
        name = max(keys, key=lambda k: fs[k]["dispatches"])[0]
        keys = [k for k in keys if k[0] == name]
        nf = sum(fs[k]["dispatches"] - fs[k]["gated"] for k in keys)
        nw = sum(ws[k]["dispatches"] - ws[k]["gated"] for k in keys)
        fetch = sum(fs[k]["mean"] * (fs[k]["dispatches"] - fs[k]["gated"]) for k in keys) / max(nf, 1)
        write = sum(ws[k]["mean"] * (ws[k]["dispatches"] - ws[k]["gated"]) for k in keys) / max(nw, 1)
        out[cls] = {"kernel": name, "dispatches": nf, "fetch_size_kb": round(fetch, 1), "write_size_kb": round(write, 1),
                    "traffic_bytes": (2 * fetch + write) * 1024,
                    "by_call_site": {k[1] or "all": {"dispatches": fs[k]["dispatches"] - fs[k]["gated"],
                                                     "traffic_bytes": (2 * fs[k]["median"] + ws[k]["median"]) * 1024} for k in keys}}
    if a.mix_from and "gs" in out:   # weights of the call sites of k_gs = calls per step in the timed region
        mix = {}
        for ln in open(a.mix_from):
            mm = re.match(r"k_gs<\d> (\S+)\s+([\d.]+) calls/step", ln)
            if mm:
                mix[mm.group(1)] = float(mm.group(2))
        sites = out["gs"]["by_call_site"]
        num = sum(w * sites[k]["traffic_bytes"] for key, w in mix.items() for k in sites if k.startswith(key))
        den = sum(w for key, w in mix.items() if any(k.startswith(key) for k in sites))
        if den > 0:
            out["gs"]["traffic_bytes_all_dispatches"] = out["gs"]["traffic_bytes"]
            out["gs"]["traffic_bytes"] = num / den
            out["gs"]["timed_region_mix_calls_per_step"] = mix
    with open(os.path.join(a.outdir, "traffic_per_launch.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: (v["kernel"], round(v["traffic_bytes"] / 1e6, 1)) for k, v in out.items() if isinstance(v, dict) and "kernel" in v}, indent=1))


if __name__ == "__main__":
    main()
