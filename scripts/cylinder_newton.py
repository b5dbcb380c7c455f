"""Newton-Krylov on the reference's cylinder case (Re = 50): the reference's own base flow BF_1cyl0.f00001 must be a
fixed point of the restated nonlinear map, and Newton must return to it from a perturbed state."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from neklab_amd import host
from refdata import load_cylinder

hm, ux, uy, p, re, lxd, _ = load_cylinder(with_bcs=True)
ctx = host.Context(); gm = host.Mesh(ctx, hm, lxd=lxd)
bf = host.nek_dvector(gm); bf.set_field(host.VX, ux); bf.set_field(host.VY, uy)
tau = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sysm = host.nek_system(tau, bf, re=re, maxit_v=400, maxit_p=4000)
F = host.nek_dvector(gm)
for tol in (1e-6, 1e-8):
    sysm.set_tolerance(tol); t0 = time.time(); sysm.eval(bf, F)
    print('tau %g tol %.0e: |F(BF)| = %.3e   (|BF| = %.3e)  %.1fs nsteps %d' % (tau, tol, F.norm(), bf.norm(), time.time() - t0, sysm.nl.info()['nsteps']), flush=True)
# perturbed start: BF + smooth bump in the wake
X = bf.copy()
x, y = hm.x.ravel(), hm.y.ravel()
bump = 0.05 * np.exp(-((x - 3.0) ** 2 + y ** 2) / 2.0) * hm.mask[0].ravel()
X.set_field(host.VX, ux.ravel() + bump)
d0 = X.copy(); d0.sub(bf)
t0 = time.time()
out = host.newton_fixed_point_iteration(sysm, X, 1e-7, tol_mode=2, kdim=int(sys.argv[2]) if len(sys.argv) > 2 else 60, log=lambda s: print(s, flush=True))
d1 = X.copy(); d1.sub(bf)
print(out)
print('distance to the reference base flow: before %.3e  after %.3e   (%.1fs)' % (d0.norm(), d1.norm(), time.time() - t0))
