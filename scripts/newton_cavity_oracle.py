"""Oracle-only try-out of the Newton-Krylov solver on a regularised lid-driven cavity (sizes for the parity test)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM
from oracle.vectors import NekDVector
from oracle.lns import ExptA, LNSConfig
from oracle import krylov as K

nel, n, re, tau = (3, 3), 6, 30.0, 0.4
hm = box_mesh(nel, n, lengths=(1.0, 1.0), deform=0.02)
sem = SEM(hm)
X = NekDVector(sem)
x, y = sem.X[0], sem.X[1]
lid = (16 * x ** 2 * (1 - x) ** 2) * (y > 1 - 1e-9)
X.v[0][...] = lid
cfgn = LNSConfig(re=re, torder=3, tau=tau, cfl_limit=0.4, vtol=1e-9, ptol=1e-9, maxit_v=400, maxit_p=4000)
cfgl = LNSConfig(re=re, torder=3, tau=tau, cfl_limit=0.5, vtol=1e-9, ptol=1e-9, maxit_v=400, maxit_p=4000)
Anl = ExptA(sem, X.v, cfgn); Ajac = ExptA(sem, X.v, cfgl)
def set_tol(t):
    Anl.cfg.vtol = Anl.cfg.ptol = 0.1 * t; Ajac.cfg.vtol = Ajac.cfg.ptol = 0.5 * t
MODE = sys.argv[2] if len(sys.argv) > 2 else 'ref'
def jac_for(Xc):
    if MODE == 'ref':
        Ajac.set_baseflow(Xc.v); return Ajac.matvec
    Ajac.cfg.dt = Anl.dt if 'dt' in MODE else 0.0
    Ajac.set_baseflow(Xc.v)
    def mv(v):
        w = v.copy()
        if 'nohist' in MODE: w.clear_rst_fields()
        return Ajac.matvec(w)
    return mv
t0 = time.time()
out = K.newton(Anl.nonlinear_map, jac_for, set_tol, X, 1e-8, tol_mode=int(sys.argv[1]) if len(sys.argv) > 1 else 1, log=print)
print(out, 'time %.1f' % (time.time() - t0), 'nsteps', Anl.nsteps, Ajac.nsteps, 'max|u|', np.abs(X.v[0]).max(), np.abs(X.v[1]).max())
