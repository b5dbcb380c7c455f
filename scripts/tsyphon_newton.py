#!/usr/bin/env python3
"""The reference's temperature-coupled Newton example on the GPU path: examples/thermosyphon/baseflow (tsyphon.usr: nek_system_temp,
newton_fixed_point_iteration(sys, bf, 1e-6, tol_mode = 2) at Ra = 510 from the base flow at Ra = 500; tsyphon.par: nu = 1/5, conductivity 1,
endTime = 1, bdf3; buoyancy ffy = T nu Ra; wall temperature 0.5 (1 + tanh(-20 y)) on the annulus 1 <= r <= 2), against the Newton residuals of
the convergence plot the reference ships with the case (residual.png).  The reference's initial guess BF_Ra500_tsyphon0.f00001 is not shipped:
it is rebuilt here as the left-right symmetric steady state at Ra = 500 (a kick of the case's initial condition, a few maps of time marching,
then Newton at Ra = 500), so only what does not depend on the details of that file can be expected to agree.

    python scripts/tsyphon_newton.py > profiles/r03_tsyphon_newton.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neklab_amd import host  # noqa: E402
from refdata import load_tsyphon  # noqa: E402

hm, d = load_tsyphon()
nu, ra0, ra1 = float(d["nu"]), float(d["rayleigh_guess"]), float(d["rayleigh"])
ctx = host.Context(0)
gm = host.Mesh(ctx, hm, lxd=int(d["lxd"]))
y = hm.y.ravel()
X = host.nek_dvector(gm, 1)
X.set_field(host.THETA, 0.5 * (1.0 + np.tanh(-20.0 * y)))          # useric / userbc of tsyphon.usr
log = lambda s: print(s, flush=True)


def system(ra, X0, tau=1.0, **kw):
    return host.nek_system(tau, X0, re=1.0 / nu, ifheat=1, conductivity=float(d["conductivity"]), rhocp=float(d["rhocp"]),
                           buoy=(0.0, nu * ra, 0.0), maxit_v=400, maxit_p=4000, **kw)


print("annulus 1 <= r <= 2: E = %d, lx1 = %d, lxd = %d; nu = %g, conductivity = %g, buoyancy nu Ra T" % (hm.x.shape[0], hm.n, int(d["lxd"]), nu, float(d["conductivity"])))
# ---- a state with a flow: the case's initial condition (rest, T = wall profile everywhere) kicked for a short time at a fixed time step
t0 = time.time()
A = host.exptA_linop(0.02, X, re=1.0 / nu, ifheat=1, conductivity=float(d["conductivity"]), rhocp=float(d["rhocp"]), buoy=(0.0, nu * ra0, 0.0),
                     dt=1.0e-4, cfl_limit=0.4, vtol=1e-9, ptol=1e-9, maxit_v=400, maxit_p=4000)
A.init()
F = host.nek_dvector(gm, 1)
host.check(ctx.lib.nlg_linop_nonlinear_map(A.h, X.h, F.h))
X.axpby(1.0, F, 1.0)
# ---- time marching at Ra = 500 (dt from the CFL limit of the current state, as the reference's nonlinear_map)
sys0 = system(ra0, X, tau=0.25)
sys0.set_tolerance(1.0e-7)
for k in range(12):
    sys0.eval(X, F)
    X.axpby(1.0, F, 1.0)
    print("  time marching %2d: t = %.2f  |Phi(X) - X| = %.4e  (%d steps per map)" % (k + 1, 0.02 + 0.25 * (k + 1), F.norm(), sys0.nl.info()["nsteps"]), flush=True)
ux = X.get_field(host.VX)
print("left-right symmetry of the marched state: max |u(x, y) + u(-x, y)| not checked pointwise; net circulation proxy sum(u_x y) = %.3e, max |u| = %.3f  (%.0f s)"
      % (float(np.sum(ux * y)), float(max(np.abs(ux).max(), np.abs(X.get_field(host.VY)).max())), time.time() - t0))
# ---- the base flow at Ra = 500: Newton from the marched state
sysA = system(ra0, X)
outA = host.newton_fixed_point_iteration(sysA, X, 1.0e-8, tol_mode=2, kdim=30, log=log)
print("Ra = %g: converged %s after %d Newton iterations, %d GMRES matvecs; residuals %s" % (ra0, outA["converged"], outA["iterations"], outA["gmres_matvecs"], ["%.2e" % r for r in outA["residuals"]]))
# ---- the reference's run: Ra = 510 from that state
sysB = system(ra1, X)
t0 = time.time()
out = host.newton_fixed_point_iteration(sysB, X, float(d["newton_tol"]), tol_mode=2, kdim=30, log=log)
print("Ra = %g: converged %s after %d Newton iterations, %d GMRES matvecs, %.0f s; time steps per map %d (dt = %.2e)"
      % (ra1, out["converged"], out["iterations"], out["gmres_matvecs"], time.time() - t0, sysB.nl.info()["nsteps"], sysB.nl.info()["dt"]))
ref = d["plot_newton_residuals"]
print("\nNewton residual at the start of each step:   this run      reference (read off residual.png)")
for i in range(max(len(out["residuals"]), len(ref))):
    print("  step %d   %s   %s" % (i + 1, "%.3e" % out["residuals"][i] if i < len(out["residuals"]) else "    -    ", "%.1e" % ref[i] if i < len(ref) else "-"))
print("GMRES residuals of Newton step 1 (init, inner steps):  this run %s   reference %s" % (["%.2e" % v for v in out["gmres_residuals"][0][:6]], ["%.2e" % v for v in d["plot_gmres_step1"]]))
print("GMRES inner steps per Newton step: %s" % [len(h) - 1 for h in out["gmres_residuals"]])
