"""Prototype 4: overlapping additive Schwarz with FDM local solves on the extended (n2+2)^3 grid (one ghost layer from
each face neighbour) + Q1 coarse (exact), vs the non-overlapping FDM.  Explicit sparse E, structured lattice of
deformed elements (orientation trivial here; the device version uses the velocity gather-scatter for the exchange)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import numpy as np, scipy.sparse as sp, scipy.linalg as sl
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM
from precond_proto2 import build_E, pcg
from precond_proto3 import q1

def lengths(sem):
    """l[e, d] = distance between the centres of the two faces of element e normal to direction d."""
    n, dim, E_ = sem.n, sem.dim, sem.E
    L = np.zeros((E_, dim))
    for d in range(dim):
        ax = dim - d
        def face(side):
            idx = [slice(None)] * (dim + 1); idx[ax] = 0 if side == 0 else n - 1
            return np.stack([sem.X[c][tuple(idx)].reshape(E_, -1).mean(1) for c in range(dim)], 1)
        L[:, d] = np.linalg.norm(face(1) - face(0), axis=1)
    return L

FAR_LUMP = False
def run(nel, n, deform, only_build=False):
    t0 = time.time(); hm = box_mesh(nel, n, deform=deform); sem = SEM(hm)
    E_, n2, dim = sem.E, sem.n2, sem.dim; npr = n2 ** dim; m = n2 + 2
    A = build_E(sem); rng = np.random.default_rng(0)
    u = [sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1)) for i in range(dim)]
    b = sem.opdiv(u).ravel()
    R, vg, nvert = q1(sem)
    Ac = (R.T @ A @ R).toarray(); Aci = np.linalg.pinv(Ac, hermitian=True, rcond=1e-10)
    coarse = lambda r: R @ (Aci @ (R.T @ r))
    L = lengths(sem)
    w1, w2 = sem.w1, sem.w2
    Dh = sem.D12 * w2[:, None]; Ih = sem.I12 * w2[:, None]
    ne = nel[::-1]                       # lattice shape in (z, y, x) order
    eid = np.arange(E_).reshape(ne)
    def nbr(e, d, s):
        ez = list(np.unravel_index(e, ne)); ax = dim - 1 - d
        ez[ax] += s
        if ez[ax] < 0 or ez[ax] >= ne[ax]: return -1
        return eid[tuple(ez)]
    # ---- 1-D operators: non-overlapping (as shipped) and extended
    S0 = np.zeros((E_, dim, n2, n2)); lam0 = np.zeros((E_, dim, n2))
    S1 = np.zeros((E_, dim, m, m)); lam1 = np.zeros((E_, dim, m))
    for e in range(E_):
        for d in range(dim):
            l = L[e, d]; eL, eR = nbr(e, d, -1), nbr(e, d, 1)
            # shipped: single element, neighbours' mass lumped through the multiplicity, Dirichlet wall -> 0
            bi = 1.0 / ((l / 2) * w1)
            bi[0] = 0.0 if eL < 0 else bi[0] / 2
            bi[-1] = 0.0 if eR < 0 else bi[-1] / 2
            A1 = Dh @ (bi[:, None] * Dh.T); B1 = (l / 2) ** 2 * Ih @ (bi[:, None] * Ih.T)
            lam_d, S_d = sl.eigh(A1, B1); S0[e, d] = S_d; lam0[e, d] = np.maximum(lam_d, 0)
            # extended: line of up to three elements
            els = [(eL, L[eL, d] if eL >= 0 else None), (e, l), (eR, L[eR, d] if eR >= 0 else None)]
            nvl = 3 * n - 2
            Bl = np.zeros(nvl); Dl = np.zeros((3 * n2, nvl)); Il = np.zeros((3 * n2, nvl))
            for q, (ee, ll) in enumerate(els):
                if ee < 0: continue
                sl_v = slice(q * (n - 1), q * (n - 1) + n)
                Bl[sl_v] += (ll / 2) * w1
                Dl[q * n2:(q + 1) * n2, sl_v] = Dh
                Il[q * n2:(q + 1) * n2, sl_v] = (ll / 2) * Ih
            binv = np.where(Bl > 0, 1.0 / np.where(Bl > 0, Bl, 1), 0.0)
            # far ends of the line: wall -> Dirichlet (0), otherwise lump the next element (multiplicity 2)
            first = 0 if eL >= 0 else n - 1
            last = nvl - 1 if eR >= 0 else 2 * (n - 1)
            for (end, ee, s) in ((first, eL if eL >= 0 else e, -1), (last, eR if eR >= 0 else e, 1)):
                if FAR_LUMP and ee != e:
                    binv[end] = binv[end] / 2          # far end of a neighbour: always treated as interior
                else:
                    binv[end] = 0.0 if nbr(ee, d, s) < 0 else binv[end] / 2
            Af = Dl @ (binv[:, None] * Dl.T); Bf = Il @ (binv[:, None] * Il.T)
            idx = np.arange(n2 - 1, 2 * n2 + 1)
            Ae = Af[np.ix_(idx, idx)]; Be = Bf[np.ix_(idx, idx)]
            for g, ee in ((0, eL), (m - 1, eR)):
                if ee < 0:
                    Ae[g, :] = 0; Ae[:, g] = 0; Be[g, :] = 0; Be[:, g] = 0; Ae[g, g] = 1.0; Be[g, g] = 1.0
            lam_d, S_d = sl.eigh(Ae, Be); S1[e, d] = S_d; lam1[e, d] = np.maximum(lam_d, 0)
    def fdm(S, lam, r):      # r: (E, k, k, k) in (z, y, x)
        t = np.einsum('eza,eyb,exc,ezyx->eabc', S[:, 2], S[:, 1], S[:, 0], r, optimize=True)
        den = lam[:, 2][:, :, None, None] + lam[:, 1][:, None, :, None] + lam[:, 0][:, None, None, :]
        t = np.where(den > 1e-12 * den.max(), t / np.where(den > 0, den, 1), 0.0)
        return np.einsum('eza,eyb,exc,eabc->ezyx', S[:, 2], S[:, 1], S[:, 0], t, optimize=True)
    def local0(r): return fdm(S0, lam0, r.reshape(sem.shape2)).ravel()
    def local1(r): return local1_core(r)
    def local_count():
        c = np.ones(sem.shape2)
        for e in range(E_):
            for d in range(dim):
                ax = dim - 1 - d
                for s_ in (-1, 1):
                    q = nbr(e, d, s_)
                    if q < 0: continue
                    dst = [slice(None)] * dim; dst[ax] = (n2 - 1) if s_ == -1 else 0
                    c[(q,) + tuple(dst)] += 1
        return c
    def local1_core(r, wt=None):
        r = r.reshape(sem.shape2)
        ext = np.zeros((E_, m, m, m)); ext[:, 1:-1, 1:-1, 1:-1] = r
        for e in range(E_):
            for d in range(dim):
                ax = dim - 1 - d
                for s in (-1, 1):
                    q = nbr(e, d, s)
                    if q < 0: continue
                    src = [slice(None)] * dim; src[ax] = (n2 - 1) if s == -1 else 0
                    dst = [slice(1, -1)] * dim; dst[ax] = 0 if s == -1 else m - 1
                    ext[(e,) + tuple(dst)] = r[(q,) + tuple(src)]
        z = fdm(S1, lam1, ext)
        out = z[:, 1:-1, 1:-1, 1:-1].copy()
        for e in range(E_):
            for d in range(dim):
                ax = dim - 1 - d
                for s in (-1, 1):
                    q = nbr(e, d, s)
                    if q < 0: continue
                    src = [slice(1, -1)] * dim; src[ax] = 0 if s == -1 else m - 1      # my ghost layer = q's boundary layer
                    dst = [slice(None)] * dim; dst[ax] = (n2 - 1) if s == -1 else 0
                    out[(q,) + tuple(dst)] += z[(e,) + tuple(src)]
        return out.ravel()
    cnt = local_count()
    sq = 1.0 / np.sqrt(cnt)
    def local1w(r):
        return sq.ravel() * local1_core(sq.ravel() * r, wt=sq)
    if only_build:
        return dict(sem=sem, local0=local0, local1=local1, local1w=local1w)
    res = {}
    for name, pr in (('fdm+Q1', lambda r: local0(r) + coarse(r)), ('fdm_ext+Q1', lambda r: local1(r) + coarse(r))):
        x, it = pcg(A, b, pr, 1e-7); res[name] = it
    # symmetry check of the overlapping operator
    a1, a2 = rng.standard_normal(E_ * npr), rng.standard_normal(E_ * npr)
    res['sym'] = float(abs(a1 @ local1(a2) - a2 @ local1(a1)) / abs(a1 @ local1(a2)))
    print(nel, n, 'E=%d' % E_, res, 'time %.0f' % (time.time() - t0), flush=True)

if __name__ == '__main__':
    for FAR_LUMP in (False, True):
        print('FAR_LUMP', FAR_LUMP)
        run((4, 4, 4), 6, 0.05)
        run((3, 3, 3), 8, 0.05)
