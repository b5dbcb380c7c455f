#!/usr/bin/env python3
"""The reference's only published number: |mu_1| = 1.0156 +- 1e-4 for exp(tau L), cylinder wake Re = 50,
tau = 1, lx1 = 6, bdf3, kdim = 128, nev = 2 (/root/reference/test/neklabTests.py:43-45, 1cyl.usr:11,20,
1cyl.par).  Runs the GPU path on the reference's own mesh/base-flow data (tests/golden fixture)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neklab_amd import host  # noqa: E402
from refdata import load_cylinder  # noqa: E402

kdim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
outdir = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out"
hm, ux, uy, p, re, lxd, _ = load_cylinder(with_bcs=True)
ctx = host.Context(0)
gm = host.Mesh(ctx, hm, lxd=lxd)
bf = host.nek_dvector(gm)
bf.set_field(host.VX, ux)
bf.set_field(host.VY, uy)
A = host.exptA_linop(1.0, bf, re=re, torder=3, vtol=1e-9, ptol=1e-7, maxit_v=400, maxit_p=4000)   # 1cyl.par
A.init()
print("info", A.info(), flush=True)
t0 = time.time()
eigvals, residuals, eigvecs, mu, nmv = host.linear_stability_analysis_fixed_point(A, kdim, 2, tol=tol, outdir=outdir, seed=1)
dt = time.time() - t0
print("matvecs", nmv, "time %.1f s" % dt, "stats", A.stats())
for m, lam, r in zip(mu, eigvals, residuals):
    print("mu = %.8f %+.8fi  |mu| = %.6f  sigma = %.6f %+.6fi  residual %.2e" % (m.real, m.imag, abs(m), lam.real, lam.imag, r))
print("REFERENCE |mu_1| = 1.0156 +- 1e-4 ; difference %.2e" % (abs(mu[0]) - 1.0156))
