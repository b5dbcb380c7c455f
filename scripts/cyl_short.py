"""Cylinder wake (reference mesh, E = 500-ish, lx1 = 6): a few exptA matvecs + Arnoldi steps, wall time per time step.
Used to see how launch-bound the small 2-D cases are (rocprofv3 --kernel-trace --stats -- python3 scripts/cyl_short.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neklab_amd import host  # noqa: E402
from refdata import load_cylinder  # noqa: E402

hm, ux, uy, p, re, lxd, _ = load_cylinder(with_bcs=True)
ctx = host.Context(0)
gm = host.Mesh(ctx, hm, lxd=lxd)
bf = host.nek_dvector(gm)
bf.set_field(host.VX, ux)
bf.set_field(host.VY, uy)
A = host.exptA_linop(1.0, bf, re=re, torder=3, vtol=1e-9, ptol=1e-7, maxit_v=400, maxit_p=4000)
A.init()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = host.KrylovBasis(gm, m + 1)
B[0].rand(True, seed=1)
H = np.zeros((m + 1, m), order="F")
host.arnoldi_step(A, B, 0, H)
ctx.sync()
s0 = A.stats()
t0 = time.perf_counter()
for k in range(1, m):
    host.arnoldi_step(A, B, k, H)
ctx.sync()
dt = time.perf_counter() - t0
s1 = A.stats()
steps = s1["steps"] - s0["steps"]
print("E %d  matvecs %d  time steps %d  wall %.3f s  %.1f us/time step  p_iters/step %.2f  v_iters/step %.2f" % (
    hm.x.shape[0], m - 1, steps, dt, 1e6 * dt / steps, (s1["p_iters"] - s0["p_iters"]) / steps, (s1["v_iters"] - s0["v_iters"]) / steps))
