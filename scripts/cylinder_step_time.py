#!/usr/bin/env python3
"""Time per time step of the propagator on the reference's own cylinder case (E = 1996, lx1 = 6, 2-D, bdf3, tolerances of 1cyl.par):
the launch-bound regime of the reference's examples.  A few matvecs of 100 + 2 steps; prints microseconds per time step, the
iteration counts and the kernel launches / collective sites per time step (nlg_counters)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neklab_amd import host  # noqa: E402
from refdata import load_cylinder  # noqa: E402

nmv = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hm, ux, uy, p, re, lxd, _ = load_cylinder(with_bcs=True)
ctx = host.Context(0)
gm = host.Mesh(ctx, hm, lxd=lxd)
bf = host.nek_dvector(gm)
bf.set_field(host.VX, ux)
bf.set_field(host.VY, uy)
A = host.exptA_linop(1.0, bf, re=re, torder=3, vtol=1e-9, ptol=1e-7, maxit_v=400, maxit_p=4000)   # 1cyl.par
A.init()
x, y = host.nek_dvector(gm), host.nek_dvector(gm)
x.rand(True, seed=1)
A.matvec(x, y)          # warm-up (first-chunk predictions)
x.assign(y)
ctx.sync()
s0 = A.stats()
l0, c0 = C.c_int64(), C.c_int64()
host.check(ctx.lib.nlg_counters(C.byref(l0), C.byref(c0)))
t0 = time.perf_counter()
for _ in range(nmv):
    A.matvec(x, y)
    x, y = y, x
ctx.sync()
dt = time.perf_counter() - t0
s1 = A.stats()
l1, c1 = C.c_int64(), C.c_int64()
host.check(ctx.lib.nlg_counters(C.byref(l1), C.byref(c1)))
steps = s1["steps"] - s0["steps"]
print("cylinder E=%d lx1=%d: %.1f us per time step over %d steps; %.2f pressure / %.2f velocity iterations per step; "
      "%.1f launches and %.1f collective sites per step"
      % (hm.E, hm.n, 1e6 * dt / steps, steps, (s1["p_iters"] - s0["p_iters"]) / steps, (s1["v_iters"] - s0["v_iters"]) / steps,
         (l1.value - l0.value) / steps, (c1.value - c0.value) / steps))
