#!/usr/bin/env python3
"""The reference's transient-growth case on the GPU path: examples/back_fstep/transient_growth (bfs.usr:8-21:
exptA_linop(18.0_dp, bf), transient_growth_analysis_fixed_point(exptA, nsv = 4, kdim = 512)) on the reference's own mesh,
boundary tags and base flow (tests/golden/reference_bfs_baseflow.npz), bdf2, Re = 600, tolerances 1e-8 / 1e-6 (bfs.par).
BASELINE.json's config 3 names this case.  The reference publishes no singular values for it; this run records ours.

usage: bfs_transient_growth.py [kdim] [tol] [outdir] [tau]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neklab_amd import host  # noqa: E402
from refdata import load_bfs  # noqa: E402

kdim = int(sys.argv[1]) if len(sys.argv) > 1 else 32
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
outdir = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out"
tau = float(sys.argv[4]) if len(sys.argv) > 4 else 18.0
os.makedirs(outdir, exist_ok=True)
hm, ux, uy, p, re, lxd, _ = load_bfs(with_bcs=True)
ctx = host.Context(0)
gm = host.Mesh(ctx, hm, lxd=lxd)
bf = host.nek_dvector(gm)
bf.set_field(host.VX, ux)
bf.set_field(host.VY, uy)
A = host.exptA_linop(tau, bf, re=re, torder=2, vtol=1e-8, ptol=1e-6, maxit_v=400, maxit_p=4000)   # bfs.par
A.init()
print("E = %d lx1 = %d  info %s" % (hm.E, hm.n, A.info()), flush=True)
# one matvec first: cost of the case
x = host.nek_dvector(gm)
x.rand(True, seed=1)
y = host.nek_dvector(gm)
t0 = time.time()
A.matvec(x, y)
t1 = time.time() - t0
st = A.stats()
print("one direct matvec: %.2f s, %d time steps, %.2f ms per step, %.1f pressure / %.1f velocity iterations per step"
      % (t1, st["steps"], 1e3 * t1 / st["steps"], st["p_iters"] / st["steps"], st["v_iters"] / st["steps"]), flush=True)
t0 = time.time()
S, res, U, V, info = host.transient_growth_analysis_fixed_point(A, 4, kdim, tol=tol, outdir=outdir, seed=1)
dt = time.time() - t0
st = A.stats()
print("svds: info %d, %d matvecs (direct + adjoint), %.1f s, %.2f ms per time step" % (info, st["matvecs"] - 1, dt, 1e3 * (dt + t1) / st["steps"]))
for i, (s, r) in enumerate(zip(S, res)):
    print("sigma_%d = %.10e   gain sigma^2 = %.6e   residual %.2e" % (i + 1, s, s * s, r))
print(open(os.path.join(outdir, "svds_output.txt")).read())
