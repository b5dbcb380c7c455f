#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of rocprofv3 against known byte counts, for the access shapes of the gather-scatter kernel
(scripts/pmc_calibrate.hip).  Run on the GPU box from the repository root:

    python scripts/pmc_calibrate.py [outdir]         # default gpurun_out/pmc_calib

Builds the program with hipcc, runs it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, as the
guide prescribes), and prints / writes, per kernel, counter bytes (KB x 1024) over known bytes.  The factor found for the 8-byte
pair accesses is what scripts/pmc_traffic.py applies to k_gs (profiles/r03_pmc_calibration.txt holds the run it was read from).
"""
import csv
import glob
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_calib"))
os.makedirs(out, exist_ok=True)
exe = os.path.join(out, "pmc_calibrate")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "scripts", "pmc_calibrate.hip")], check=True)
env = dict(os.environ, TMPDIR="/tmp")
known = {}
r = subprocess.run([exe], capture_output=True, text=True, check=True)
for ln in r.stdout.splitlines():
    p = ln.split()
    if p and p[0] == "KNOWN":
        known[p[1]] = {p[i]: int(p[i + 1]) for i in range(2, len(p), 2)}
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out, counter.lower())
    subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", exe], check=True, cwd="/tmp", env=env,
                   stdout=subprocess.DEVNULL)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                k = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
                res.setdefault(k, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
lines = ["kernel            known_read  known_write  lines_read   FETCH_SIZE*1024   WRITE_SIZE*1024   fetch/read  fetch/lines  write/known"]
for k, kn in known.items():
    f = statistics.median(res.get(k, {}).get("FETCH_SIZE", [0.0])) * 1024
    w = statistics.median(res.get(k, {}).get("WRITE_SIZE", [0.0])) * 1024
    lines.append("%-16s %11d  %11d  %10d  %16.0f  %16.0f  %10s  %11s  %11s" % (
        k, kn["read"], kn["write"], kn.get("lines_read", 0), f, w,
        "%.3f" % (f / kn["read"]) if kn["read"] else "-", "%.3f" % (f / kn["lines_read"]) if kn.get("lines_read") else "-",
        "%.3f" % (w / kn["write"]) if kn["write"] else "-"))
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "calibration.txt"), "w").write(txt + "\n")
