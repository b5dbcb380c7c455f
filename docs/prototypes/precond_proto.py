"""Prototype (numpy): iteration counts of PCG on E with (a) Jacobi, (b) element-wise FDM block-Jacobi,
(c) FDM + piecewise-constant coarse grid, additive."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, scipy.linalg as sl
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM

def build(nel, n, deform):
    hm = box_mesh(nel, n, deform=deform)
    return hm, SEM(hm)

def fdm_setup(sem, hm):
    n, n2, dim, E = sem.n, sem.n2, sem.dim, sem.E
    w1, w2 = sem.w1, sem.w2
    Dh = sem.D12 * w2[:, None]           # (n2 x n) diag(w2) D12
    Ih = sem.I12 * w2[:, None]
    X = sem.X
    S = np.zeros((E, dim, n2, n2)); lam = np.zeros((E, dim, n2))
    mult = sem.mult
    for e in range(E):
        for d in range(dim):
            ax = dim - d   # axis index in (E, z, y, x) arrays: x -> last
            # face-centre coordinates
            def face(side):
                idx = [e] + [slice(None)] * dim
                idx[ax] = 0 if side == 0 else n - 1
                return np.array([X[c][tuple(idx)].mean() for c in range(dim)])
            l = np.linalg.norm(face(1) - face(0))
            binv = 1.0 / ((l / 2) * w1)
            for side in (0, 1):
                idx = [e] + [n // 2] * dim
                idx[ax] = 0 if side == 0 else n - 1
                m = sem.mask[d][tuple(idx)]
                mu = mult[tuple(idx)]
                k = 0 if side == 0 else n - 1
                binv[k] = 0.0 if m == 0 else binv[k] / mu
            A = Dh @ (binv[:, None] * Dh.T)
            B = (l / 2) ** 2 * Ih @ (binv[:, None] * Ih.T)
            lam_d, S_d = sl.eigh(A, B)
            S[e, d] = S_d; lam[e, d] = np.maximum(lam_d, 0)
    return S, lam

def fdm_apply(sem, S, lam, r):
    dim, n2 = sem.dim, sem.n2
    if dim == 3:
        t = np.einsum('eza,eyb,exc,ezyx->eabc', S[:, 2], S[:, 1], S[:, 0], r, optimize=True)   # S^T r ; index order z,y,x
        den = lam[:, 2][:, :, None, None] + lam[:, 1][:, None, :, None] + lam[:, 0][:, None, None, :]
        t = np.where(den > 1e-12 * den.max(), t / np.where(den > 0, den, 1), 0.0)
        return np.einsum('eza,eyb,exc,eabc->ezyx', S[:, 2], S[:, 1], S[:, 0], t, optimize=True)
    t = np.einsum('eya,exb,eyx->eab', S[:, 1], S[:, 0], r, optimize=True)
    den = lam[:, 1][:, :, None] + lam[:, 0][:, None, :]
    t = np.where(den > 1e-12 * den.max(), t / np.where(den > 0, den, 1), 0.0)
    return np.einsum('eya,exb,eab->eyx', S[:, 1], S[:, 0], t, optimize=True)

def coarse_setup(sem):
    E = sem.E
    ones = np.ones(sem.shape2)
    G = sem.opgradt(ones)                       # local (unassembled) D^T 1_e
    Ac = np.zeros((E, E))
    glo = sem.glo.reshape(E, -1)
    for i in range(sem.dim):
        c = (sem.mask[i] * sem.binvm1).reshape(E, -1)
        g = G[i].reshape(E, -1)
        # assemble via global nodes: M (nodes x E) sparse -> Ac += M^T diag(c) M
        import scipy.sparse as sp
        rows = glo.ravel(); cols = np.repeat(np.arange(E), glo.shape[1])
        M = sp.csr_matrix((g.ravel(), (rows, cols)), shape=(rows.max() + 1, E))
        cg = np.zeros(rows.max() + 1); cg[rows] = c.ravel()
        Ac += (M.T @ sp.diags(cg) @ M).toarray()
    return Ac

def pcg(sem, b, prec, tol, maxit=5000, proj=True):
    P = (lambda a: a - a.mean()) if proj else (lambda a: a)
    x = np.zeros_like(b); r = P(b.copy()); z = P(prec(r)); p = z.copy(); rz = np.sum(r * z)
    r0 = np.sqrt(np.sum(r * r / sem.bm2))
    for it in range(maxit):
        if np.sqrt(np.sum(r * r / sem.bm2)) < tol * r0: return x, it
        w = P(sem.cdabdtp(p)); a = rz / np.sum(p * w); x += a * p; r -= a * w
        z = P(prec(r)); rzn = np.sum(r * z); p = z + (rzn / rz) * p; rz = rzn
    return x, maxit

for nel, n, deform in [((4,4,4), 8, 0.05), ((6,6,6), 8, 0.05), ((8,8,8), 8, 0.05)]:
    t0 = time.time(); hm, sem = build(nel, n, deform)
    rng = np.random.default_rng(0)
    u = [sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1)) for i in range(sem.dim)]
    b = sem.opdiv(u)                              # a divergence-like right-hand side
    ed = sem.e_diag(); S, lam = fdm_setup(sem, hm); Ac = coarse_setup(sem); Acp = np.linalg.pinv(Ac)
    def jac(r): return r / ed
    def fdm(r): return fdm_apply(sem, S, lam, r)
    def two(r):
        rc = r.reshape(sem.E, -1).sum(1); c = Acp @ (rc - rc.mean())
        return fdm_apply(sem, S, lam, r) + c.reshape((sem.E,) + (1,) * sem.dim)
    res = {}
    for name, pr in (('jacobi', jac), ('fdm', fdm), ('fdm+coarse', two)):
        x, it = pcg(sem, b, pr, 1e-7); res[name] = it
    print(nel, n, 'E=%d' % sem.E, res, 'time %.1f' % (time.time() - t0), flush=True)

# ---------------------------------------------------------------------------------------------------
# part 2: replace the exact coarse solve by one symmetric V-cycle (Jacobi smoothing on the element-level
# operator + exact solve on greedy aggregates of elements)
def greedy_aggregates(Ac):
    E = Ac.shape[0]
    agg = -np.ones(E, dtype=int); na = 0
    nbrs = [np.nonzero(Ac[e])[0] for e in range(E)]
    for e in range(E):
        if agg[e] >= 0: continue
        if all(agg[q] < 0 for q in nbrs[e]):
            for q in nbrs[e]: agg[q] = na
            na += 1
    for e in range(E):            # leftovers join the neighbouring aggregate with the strongest coupling
        if agg[e] < 0:
            cand = [q for q in nbrs[e] if agg[q] >= 0 and q != e]
            if cand:
                q = max(cand, key=lambda q: abs(Ac[e, q])); agg[e] = agg[q]
            else:
                agg[e] = na; na += 1
    return agg, na

print('--- V-cycle coarse solver')
for nel, n, deform in [((6,6,6), 8, 0.05), ((8,8,8), 8, 0.05), ((12,10,8), 8, 0.05)]:
    t0 = time.time(); hm, sem = build(nel, n, deform)
    rng = np.random.default_rng(0)
    u = [sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1)) for i in range(sem.dim)]
    b = sem.opdiv(u)
    S, lam = fdm_setup(sem, hm); Ac = coarse_setup(sem)
    agg, na = greedy_aggregates(Ac)
    R1 = np.zeros((na, sem.E)); R1[agg, np.arange(sem.E)] = 1.0
    Acc = R1 @ Ac @ R1.T; Accp = np.linalg.pinv(Acc)
    dinv = 1.0 / np.diag(Ac)
    res = {}
    for nu, om in ((1, 0.7), (2, 0.7), (1, 0.9)):
        def vcycle(bc):
            x = np.zeros_like(bc)
            for _ in range(nu): x = x + om * dinv * (bc - Ac @ x)
            rr = bc - Ac @ x
            x = x + R1.T @ (Accp @ (R1 @ rr))
            for _ in range(nu): x = x + om * dinv * (bc - Ac @ x)
            return x
        def two(r):
            rc = r.reshape(sem.E, -1).sum(1); c = vcycle(rc - rc.mean())
            return fdm_apply(sem, S, lam, r) + c.reshape((sem.E,) + (1,) * sem.dim)
        x, it = pcg(sem, b, two, 1e-7); res[(nu, om)] = it
    print(nel, 'E=%d aggregates=%d' % (sem.E, na), res, 'time %.1f' % (time.time() - t0), flush=True)
