"""Oracle-only check of the Boussinesq coupling: Rayleigh-Benard onset, rigid-rigid, Ra_c = 1707.762 at k = 3.117
(Chandrasekhar 1961; quoted at /root/reference/examples/rayBen/baseflow/rayBen.par:9)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM
from oracle.vectors import NekDVector
from oracle.lns import ExptA, LNSConfig
from oracle.krylov import eigs
nel, n = (3, 3), 8
hm = box_mesh(nel, n, lengths=(2 * np.pi / 3.117, 1.0), periodic=(True, False), deform=0.0)
sem = SEM(hm)
U = [np.zeros(sem.shape1), np.zeros(sem.shape1)]
Theta = 1.0 - sem.X[1]
for Ra in (1600.0, 1707.762, 1800.0):
    cfg = LNSConfig(re=1.0, torder=3, tau=0.1, dt=0.005, vtol=1e-11, ptol=1e-11, maxit_v=2000, maxit_p=4000, ifheat=True,
                    conductivity=1.0, rhocp=1.0, buoy=(0.0, Ra, 0.0))
    A = ExptA(sem, U, cfg, Theta)
    v = NekDVector(sem, nscal=1); v.rand(True, seed=1) if 'seed' in NekDVector.rand.__code__.co_varnames else v.rand(True)
    t0 = time.time()
    lam, vecs, res, nmv = eigs(A.matvec, v, nev=1, kdim=16, tol=1e-8)
    print('Ra %.3f  mu = %s  growth rate = %.5f   res %.1e  nmv %d  %.0fs  stats %s' % (Ra, lam[0], np.log(abs(lam[0])) / cfg.tau, res[0], nmv, time.time() - t0, A.stats), flush=True)
