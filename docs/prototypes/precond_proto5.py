"""Prototype 5: how accurate must the coarse solve be for the overlapping local solves?  exact vs additive
(omega D^-1 + aggregate solve) vs multiplicative V-cycle, aggregates = 2x2x2 vertices (element-based greedy)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import numpy as np
import precond_proto4 as P4
from precond_proto2 import build_E, pcg
from precond_proto3 import q1

def run(nel, n):
    d = P4.run(nel, n, 0.05, only_build=True)
    sem, l0, l1, l1w = d['sem'], d['local0'], d['local1'], d['local1w']
    A = build_E(sem); rng = np.random.default_rng(0)
    u = [sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1)) for i in range(sem.dim)]
    b = sem.opdiv(u).ravel()
    R, vg, nvert = q1(sem)
    Ac = (R.T @ A @ R).toarray(); Aci = np.linalg.pinv(Ac, hermitian=True, rcond=1e-10)
    agg = -np.ones(nvert, dtype=int); na = 0
    for row in vg:
        if all(agg[v] < 0 for v in row):
            agg[list(row)] = na; na += 1
    for v in range(nvert):
        if agg[v] < 0:
            cand = [w for w in np.nonzero(Ac[v])[0] if agg[w] >= 0 and w != v]
            agg[v] = agg[max(cand, key=lambda w: abs(Ac[v, w]))]
    P = np.zeros((nvert, na)); P[np.arange(nvert), agg] = 1.0
    Acc = np.linalg.pinv(P.T @ Ac @ P, hermitian=True, rcond=1e-10); dinv = 1.0 / np.diag(Ac)
    def exact(bc): return Aci @ bc
    def add(om):
        return lambda bc: om * dinv * bc + P @ (Acc @ (P.T @ bc))
    def vcyc(om):
        def f(bc):
            x = om * dinv * bc
            x = x + P @ (Acc @ (P.T @ (bc - Ac @ x)))
            return x + om * dinv * (bc - Ac @ x)
        return f
    res = {'nvert': nvert, 'na': na}
    for lname, loc in (('fdm_ext', l1), ('fdm_extw', l1w)):
        for cname, cs in (('exact', exact), ('add0.7', add(0.7)), ('add1.0', add(1.0)), ('vcyc0.7', vcyc(0.7))):
            x, it = pcg(A, b, lambda r: loc(r) + R @ cs(R.T @ r), 1e-7); res[lname + '+' + cname] = it
    print(nel, n, res, flush=True)

if __name__ == '__main__':
    run((6, 6, 6), 6)
