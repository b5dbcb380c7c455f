"""Prototype 2 (numpy/scipy, algebraic): what stronger pressure preconditioners would buy.
Builds E as an explicit sparse matrix on a small mesh and compares PCG iteration counts of
  bj+P0   : exact element-block Jacobi + exact piecewise-constant coarse grid   (upper bound of the shipped FDM + P0)
  bj+Q1   : ... + trilinear vertex coarse grid
  as+P0/Q1: additive Schwarz with one layer of face-neighbour overlap (exact sub-solves)
"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp, scipy.linalg as sl
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM

def build_E(sem):
    E_, n, n2, dim = sem.E, sem.n, sem.n2, sem.dim
    nv, npr = n ** dim, n2 ** dim
    G = [np.zeros((E_, nv, npr)) for _ in range(dim)]
    for q in range(npr):
        p = np.zeros((E_, npr)); p[:, q] = 1.0
        g = sem.opgradt(p.reshape(sem.shape2))
        for i in range(dim):
            G[i][:, :, q] = g[i].reshape(E_, nv)
    glo = sem.glo.reshape(E_, nv)
    _, glo = np.unique(glo, return_inverse=True); glo = glo.reshape(E_, nv)
    ng = glo.max() + 1
    Emat = None
    for i in range(dim):
        rows = np.repeat(glo[:, :, None], npr, axis=2).ravel()
        cols = (np.arange(E_)[:, None, None] * npr + np.arange(npr)[None, None, :] + np.zeros((1, nv, 1), dtype=int)).ravel()
        M = sp.csr_matrix((G[i].ravel(), (rows, cols)), shape=(ng, E_ * npr))
        w = np.zeros(ng); w[glo.ravel()] = (sem.mask[i] * sem.binvm1).ravel()
        T = (M.T @ sp.diags(w) @ M)
        Emat = T if Emat is None else Emat + T
    return Emat.tocsr()

def pcg(A, b, prec, tol, maxit=3000):
    P = lambda a: a - a.mean()
    x = np.zeros_like(b); r = P(b.copy()); z = P(prec(r)); p = z.copy(); rz = r @ z
    r0 = np.linalg.norm(r)
    for it in range(maxit):
        if np.linalg.norm(r) < tol * r0: return x, it
        w = P(A @ p); a = rz / (p @ w); x += a * p; r -= a * w
        z = P(prec(r)); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn
    return x, maxit

def run(nel, n, deform):
    t0 = time.time()
    hm = box_mesh(nel, n, deform=deform); sem = SEM(hm)
    E_, n2, dim = sem.E, sem.n2, sem.dim; npr = n2 ** dim
    A = build_E(sem)
    # sanity
    rng = np.random.default_rng(0)
    p = rng.standard_normal(sem.shape2)
    assert np.allclose(A @ p.ravel(), sem.cdabdtp(p).ravel(), atol=1e-9 * np.abs(p).max() * abs(A).max())
    u = [sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1)) for i in range(dim)]
    b = sem.opdiv(u).ravel()
    # element neighbours through faces: share >= n^(dim-1) global nodes
    nv = n ** dim
    glo = sem.glo.reshape(E_, nv)
    lidx = np.arange(npr).reshape((n2,) * dim)
    # element-lattice neighbours from the box structure
    ne = nel
    eid = np.arange(E_).reshape(ne[::-1])
    doms_bj, doms_as = [], []
    for e in range(E_):
        own = e * npr + np.arange(npr)
        doms_bj.append(own)
        ext = [own]
        ez = np.unravel_index(e, ne[::-1])     # (z,y,x) order
        for ax in range(dim):
            for s in (-1, 1):
                q = list(ez); q[ax] += s
                if q[ax] < 0 or q[ax] >= ne[::-1][ax]: continue
                nb = eid[tuple(q)]
                sl_ = [slice(None)] * dim
                sl_[ax] = (n2 - 1) if s == -1 else 0        # neighbour's layer adjacent to the shared face
                ext.append(nb * npr + lidx[tuple(sl_)].ravel())
        doms_as.append(np.concatenate(ext))
    def make_schwarz(doms):
        inv = []
        for d in doms:
            S = A[d][:, d].toarray()
            inv.append(np.linalg.pinv(S, hermitian=True, rcond=1e-12))
        def apply(r):
            z = np.zeros_like(r)
            for d, Si in zip(doms, inv): z[d] += Si @ r[d]
            return z
        return apply
    bj = make_schwarz(doms_bj); as_ = make_schwarz(doms_as)
    # coarse spaces
    R0 = sp.csr_matrix((np.ones(E_ * npr), (np.arange(E_ * npr), np.repeat(np.arange(E_), npr))), shape=(E_ * npr, E_))
    # Q1: vertices
    cidx = np.array([[(0 if (c >> a) & 1 == 0 else n - 1) for a in range(dim)] for c in range(2 ** dim)])   # (x,y,z) bits
    lin = np.arange(nv).reshape((n,) * dim)
    vg = np.array([[glo[e, lin[tuple(ci[::-1])]] for ci in cidx] for e in range(E_)])        # (E, 2^dim)
    _, vg = np.unique(vg, return_inverse=True); vg = vg.reshape(E_, -1); nvert = vg.max() + 1
    z2 = sem.z2; h0 = (1 - z2) / 2; h1 = (1 + z2) / 2
    rows, cols, vals = [], [], []
    for c in range(2 ** dim):
        wgt = 1.0
        fs = []
        for a in range(dim):
            fs.append(h1 if (c >> a) & 1 else h0)
        if dim == 3: wv = fs[2][:, None, None] * fs[1][None, :, None] * fs[0][None, None, :]
        else: wv = fs[1][:, None] * fs[0][None, :]
        for e in range(E_):
            rows.append(e * npr + np.arange(npr)); cols.append(np.full(npr, vg[e, c])); vals.append(wv.ravel())
    R1 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(E_ * npr, nvert))
    def coarse(R):
        Ac = (R.T @ A @ R).toarray(); Aci = np.linalg.pinv(Ac, hermitian=True, rcond=1e-10)
        return lambda r: R @ (Aci @ (R.T @ r))
    c0 = coarse(R0); c1 = coarse(R1); c01 = coarse(sp.hstack([R0, R1]).tocsr())
    res = {}
    for name, pr in (('bj', bj), ('bj+P0', lambda r: bj(r) + c0(r)), ('bj+Q1', lambda r: bj(r) + c1(r)),
                     ('bj+P0Q1', lambda r: bj(r) + c01(r)),
                     ('as', as_), ('as+P0', lambda r: as_(r) + c0(r)), ('as+Q1', lambda r: as_(r) + c1(r)),
                     ('as+P0Q1', lambda r: as_(r) + c01(r))):
        x, it = pcg(A, b, pr, 1e-7); res[name] = it
    print(nel, n, 'E=%d' % E_, res, 'time %.0f' % (time.time() - t0), flush=True)

if __name__ == '__main__':
    run((3, 3, 3), 6, 0.05)
    run((4, 4, 4), 6, 0.05)
    run((4, 4, 4), 8, 0.05)
    run((6, 6, 6), 6, 0.05)
