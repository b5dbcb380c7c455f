import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from neklab_amd import host
from test_gpu_linop import setup_case, load_pair
from oracle.krylov import eigs as o_eigs
ctx = host.Context(0)
hm, sem, gm, oA, gA, rng = setup_case(ctx, 2, tau=1.0, re=10.0)
ov, gv = load_pair(sem, gm, rng)
nev, kdim = 3, 14
X = [host.nek_dvector(gm) for _ in range(nev)]
mu, res, info = host.eigs(gA, X, kdim=kdim, tol=1e-9, x0=gv, write_intermediate=False, max_restarts=8)
olam, ovecs, ores, onmv = o_eigs(oA.matvec, ov, nev, kdim, tol=1e-9, max_restarts=8)
print(mu, res, info); print(olam, ores, onmv)
def flat(v): return np.concatenate([v.get_field(i) for i in range(2)]) if hasattr(v,'get_field') else np.concatenate([a.ravel() for a in v.v])
Bo=np.stack([flat(v) for v in ovecs],axis=1); Bg=np.stack([flat(v) for v in X],axis=1)
for q in range(nev):
    c,*_=np.linalg.lstsq(Bo,Bg[:,q],rcond=None); print(q,'resid',np.abs(Bg[:,q]-Bo@c).max(), c, 'norms', np.linalg.norm(Bg[:,q]), np.linalg.norm(Bo[:,q]))
# check eigen-residuals directly: A x - lam x for oracle & gpu vectors (real ones)
for q in range(nev):
    y=host.nek_dvector(gm); gA.matvec(X[q],y); 
    print('gpu vec',q,'|Ax|/|x|', y.norm()/X[q].norm())
