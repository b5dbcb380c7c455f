"""Prototype 3: Q1 (trilinear vertex) Galerkin coarse space for E, exact vs one V-cycle (damped Jacobi + greedy vertex
aggregates), with exact element-block Jacobi as the local part.  Explicit sparse E (see precond_proto2.py)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import numpy as np, scipy.sparse as sp
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM
from precond_proto2 import build_E, pcg

def q1(sem):
    E_, n, n2, dim = sem.E, sem.n, sem.n2, sem.dim; npr, nv = n2 ** dim, n ** dim
    glo = sem.glo.reshape(E_, nv)
    cidx = np.array([[(0 if (c >> a) & 1 == 0 else n - 1) for a in range(dim)] for c in range(2 ** dim)])
    lin = np.arange(nv).reshape((n,) * dim)
    vg = np.array([[glo[e, lin[tuple(ci[::-1])]] for ci in cidx] for e in range(E_)])
    _, vg = np.unique(vg, return_inverse=True); vg = vg.reshape(E_, -1); nvert = vg.max() + 1
    z2 = sem.z2; h0 = (1 - z2) / 2; h1 = (1 + z2) / 2
    rows, cols, vals = [], [], []
    for c in range(2 ** dim):
        fs = [h1 if (c >> a) & 1 else h0 for a in range(dim)]
        wv = fs[2][:, None, None] * fs[1][None, :, None] * fs[0][None, None, :] if dim == 3 else fs[1][:, None] * fs[0][None, :]
        for e in range(E_):
            rows.append(e * npr + np.arange(npr)); cols.append(np.full(npr, vg[e, c])); vals.append(wv.ravel())
    R = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(E_ * npr, nvert))
    return R, vg, nvert

def aggregates(vg, nvert):
    nb = [set() for _ in range(nvert)]
    for row in vg:
        for a in row: nb[a].update(row)
    agg = -np.ones(nvert, dtype=int); na = 0
    for v in range(nvert):
        if agg[v] >= 0: continue
        if all(agg[q] < 0 for q in nb[v]):
            for q in nb[v]: agg[q] = na
            na += 1
    return agg, na, nb

def run(nel, n, deform):
    t0 = time.time(); hm = box_mesh(nel, n, deform=deform); sem = SEM(hm)
    E_, n2, dim = sem.E, sem.n2, sem.dim; npr = n2 ** dim
    A = build_E(sem); rng = np.random.default_rng(0)
    u = [sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1)) for i in range(dim)]
    b = sem.opdiv(u).ravel()
    inv = [np.linalg.pinv(A[e * npr:(e + 1) * npr][:, e * npr:(e + 1) * npr].toarray(), hermitian=True, rcond=1e-12) for e in range(E_)]
    def bj(r): return np.concatenate([inv[e] @ r[e * npr:(e + 1) * npr] for e in range(E_)])
    R, vg, nvert = q1(sem)
    Ac = (R.T @ A @ R).toarray()
    Aci = np.linalg.pinv(Ac, hermitian=True, rcond=1e-10)
    res = {}
    x, it = pcg(A, b, lambda r: bj(r) + R @ (Aci @ (R.T @ r)), 1e-7); res['exact'] = it
    agg, na, nb = aggregates(vg, nvert)
    # leftovers: join strongest-coupled aggregated neighbour
    for v in range(nvert):
        if agg[v] < 0:
            cand = [q for q in nb[v] if agg[q] >= 0]
            q = max(cand, key=lambda q: abs(Ac[v, q])); agg[v] = agg[q]
    R1 = np.zeros((na, nvert)); R1[agg, np.arange(nvert)] = 1.0
    Acc = R1 @ Ac @ R1.T; Accp = np.linalg.pinv(Acc, hermitian=True, rcond=1e-10)
    dinv = 1.0 / np.diag(Ac)
    for nu, om in ((1, 0.7), (1, 0.5), (2, 0.6), (1, 0.9)):
        def vcycle(bc):
            x = np.zeros_like(bc)
            for _ in range(nu): x = x + om * dinv * (bc - Ac @ x)
            x = x + R1.T @ (Accp @ (R1 @ (bc - Ac @ x)))
            for _ in range(nu): x = x + om * dinv * (bc - Ac @ x)
            return x
        x, it = pcg(A, b, lambda r: bj(r) + R @ vcycle(R.T @ r), 1e-7); res[(nu, om)] = it
    for om in (0.5, 0.7, 1.0):
        def addc(bc): return om * dinv * bc + R1.T @ (Accp @ (R1 @ bc))
        x, it = pcg(A, b, lambda r: bj(r) + R @ addc(R.T @ r), 1e-7); res[('add', om)] = it
    def post(bc):       # coarse first, then one Jacobi sweep on the residual, symmetrised by a mirrored pre-sweep is the V-cycle; this is the half-cost variant
        x = R1.T @ (Accp @ (R1 @ bc)); return x + 0.7 * dinv * (bc - Ac @ x)
    x, it = pcg(A, b, lambda r: bj(r) + R @ post(R.T @ r), 1e-7); res['post-only(nonsym)'] = it
    print(nel, n, 'E=%d nvert=%d na=%d nnz/row=%.0f' % (E_, nvert, na, np.count_nonzero(np.abs(Ac) > 1e-14) / nvert), res, 'time %.0f' % (time.time() - t0), flush=True)

if __name__ == '__main__':
    run((8, 8, 8), 6, 0.05)
    run((16, 8, 8), 6, 0.05)
