#!/usr/bin/env python3
"""The reference's resolvent example on the GPU path: examples/back_fstep/gramian (bfs.usr: a Gaussian actuator in v centred at
(0.6, 1.0), width 0.6; for omega = 0.2, 0.4, ... : R = resolvent_linop(omega, bf); R%matvec(forcing, response); writes omega and
0.5 |response|^2 to amplitude.dat) on the reference's own mesh, boundary tags and base flow (tests/golden/reference_bfs_baseflow.npz),
bdf2, Re = 600, tolerances 1e-8 / 1e-6 (bfs.par).  The reference publishes no amplitudes for it; this run records ours.

usage: bfs_resolvent_gramian.py [number of frequencies, default 15]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neklab_amd import host  # noqa: E402
from refdata import load_bfs  # noqa: E402

nfreq = int(sys.argv[1]) if len(sys.argv) > 1 else 15
hm, ux, uy, p, re, lxd, _ = load_bfs(with_bcs=True)
ctx = host.Context(0)
gm = host.Mesh(ctx, hm, lxd=lxd)
bf = host.nek_dvector(gm)
bf.set_field(host.VX, ux)
bf.set_field(host.VY, uy)
forcing, response = host.nek_zvector(gm), host.nek_zvector(gm)
forcing.zero()
x, y = hm.x.ravel(), hm.y.ravel()
forcing.re.set_field(host.VY, np.exp(-((x - 0.6) ** 2 + (y - 1.0) ** 2) / 0.6 ** 2))      # make_actuator, bfs.usr
print("E = %d lx1 = %d, Re = %g; actuator norm %.6e" % (hm.E, hm.n, re, forcing.norm()), flush=True)
print("# omega   0.5 |R(omega) f|^2   GMRES matvecs   time steps   seconds")
for i in range(1, nfreq + 1):
    omega = 0.2 * i
    R = host.resolvent_linop(omega, bf, re=re, torder=2, vtol=1e-8, ptol=1e-6, maxit_v=400, maxit_p=4000)
    response.zero()
    t0 = time.time()
    R.matvec(forcing, response)
    a = response.norm()
    st = R.last_operator.stats()
    print("%.1f   %.8e   %d   %d   %.1f" % (omega, 0.5 * a * a, R.gmres_matvecs, st["steps"], time.time() - t0), flush=True)
