"""Prototype 5 (round 4): what would an INTERMEDIATE level buy?  The reference's cases run Nek5000's
`semg_xxt` (Schwarz multigrid over several polynomial levels + coarse solve); this build has two levels (weighted overlapping
Schwarz on the extended element + trilinear vertex space).  Before writing kernels for a third level, its BEST case is priced
here on the explicit sparse E of a small deformed mesh: the mid-level space is the span of the tensor Legendre modes of degree
<= q per element (discontinuous, like the pressure space itself), its Galerkin operator is inverted EXACTLY, and it is
combined additively -- what the kernels could do -- and multiplicatively (one extra application of E per iteration).
Any real mid-level smoother does worse than the exact solve.  Iterations of PCG to 1e-7, mean-free right-hand side.

    python docs/prototypes/precond_proto5.py
"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'docs/prototypes')
import numpy as np
from numpy.polynomial import legendre as npl
from precond_proto2 import build_E, pcg
from precond_proto3 import q1
import precond_proto4 as p4


def modal_space(sem, q):
    """R (n_p x E (q+1)^3): per element the tensor Legendre modes of degree <= q evaluated at the GL pressure points"""
    n2, E_, dim = sem.n2, sem.E, sem.dim
    z = sem.z2                       # GL points of the pressure mesh in [-1, 1]
    P1 = np.stack([npl.legval(z, np.eye(q + 1)[k]) for k in range(q + 1)], 1)        # (n2, q+1)
    if dim == 3:
        loc = np.einsum('za,yb,xc->zyxabc', P1, P1, P1).reshape(n2 ** 3, (q + 1) ** 3)
    else:
        loc = np.einsum('ya,xb->yxab', P1, P1).reshape(n2 ** 2, (q + 1) ** 2)
    import scipy.sparse as sp
    return sp.block_diag([loc] * E_).tocsr()


def run(nel, n, deform):
    t0 = time.time()
    d = p4.run(nel, n, deform, only_build=True)
    sem, local1w = d['sem'], d['local1w']
    A = build_E(sem)
    rng = np.random.default_rng(0)
    u = [sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1)) for i in range(sem.dim)]
    b = sem.opdiv(u).ravel()
    R, vg, nvert = q1(sem)
    Aci = np.linalg.pinv((R.T @ A @ R).toarray(), hermitian=True, rcond=1e-10)
    coarse = lambda r: R @ (Aci @ (R.T @ r))
    res = {}
    base = lambda r: local1w(r) + coarse(r)
    res['two-level (shipped structure, exact vertex solve)'] = pcg(A, b, base, 1e-7)[1]
    for q in (1, 2, 3):
        R2 = modal_space(sem, q)
        A2i = np.linalg.pinv((R2.T @ A @ R2).toarray(), hermitian=True, rcond=1e-10)
        mid = lambda r, R2=R2, A2i=A2i: R2 @ (A2i @ (R2.T @ r))
        res['+ exact degree-%d level, additive' % q] = pcg(A, b, lambda r: base(r) + mid(r), 1e-7)[1]
        # symmetric multiplicative: mid, then the two-level sum on the updated residual, then mid again
        def mult(r, mid=mid):
            z1 = mid(r)
            z2 = z1 + base(r - A @ z1)
            return z2 + mid(r - A @ z2)
        res['+ exact degree-%d level, multiplicative (2 extra E)' % q] = pcg(A, b, mult, 1e-7)[1]
    # Chebyshev-free alternative: two sweeps of the two-level sum (symmetrised Richardson, damping 0.7): 1 extra E per iteration
    def two_sweeps(r):
        z1 = 0.7 * base(r)
        return z1 + 0.7 * base(r - A @ z1)
    res['two damped sweeps of the two-level sum (1 extra E)'] = pcg(A, b, two_sweeps, 1e-7)[1]
    print('mesh %s lx1 = %d (E = %d, %d pressure dofs), %.0f s' % (nel, n, sem.E, A.shape[0], time.time() - t0))
    for k, v in res.items():
        print('  %-62s %3d iterations' % (k, v))


if __name__ == '__main__':
    run((4, 4, 4), 8, 0.05)
    run((6, 5, 4), 6, 0.05)
