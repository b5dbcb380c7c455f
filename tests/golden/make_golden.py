#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the CPU oracle (fixed seeds).

The reference cannot be built or imported here (LightKrylov and Nek5000 are absent and
un-pinned), so these vectors pin the ORACLE against itself over time and pin the HIP path against the
oracle on the GPU box, where /root/reference and the oracle's heavier cases are not re-run.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from neklab_amd.mesh import box_mesh  # noqa: E402
from oracle.krylov import arnoldi_step, eigs  # noqa: E402
from oracle.lns import ExptA, LNSConfig  # noqa: E402
from oracle.sem import SEM  # noqa: E402
from oracle.vectors import NekDVector  # noqa: E402

CASES = {
    "2d": dict(nel=(3, 2), n=6, lengths=(3.0, 2.0), periodic=(True, False), deform=0.04),
    "3d": dict(nel=(2, 2, 2), n=5, lengths=(2.0, 2.0, 2.0), periodic=(False, False, True), deform=0.04),
    # the instantiation the headline benchmark runs (lx1 = 8, 3-D: k_axhelm3r<8,4>, face-grouped k_opdiv3 / k_opgradt3<8,3>,
    # k_fdm_ext<8,1>, k_conv3<8,12>, ...): propagator and Arnoldi only, the operators are covered by the cases above
    "3d_n8": dict(nel=(3, 3, 2), n=8, lengths=(3.0, 3.0, 2.0), periodic=(True, False, False), deform=0.04, ops=False),
}


def build(case):
    c = CASES[case]
    hm = box_mesh(c["nel"], c["n"], lengths=c["lengths"], periodic=c["periodic"], deform=c["deform"])
    return hm, SEM(hm)


def base_flow(sem):
    return [sem.mask[i] * ((1.0 if i == 0 else 0.3) * np.cos(sem.X[1] * (i + 1)) * np.sin(0.5 * sem.X[0] + i)) for i in range(sem.dim)]


def lns_cfg():
    return dict(re=30.0, torder=3, tau=0.03, dt=0.01, vtol=1e-13, ptol=1e-13, fixed_iters_v=40, fixed_iters_p=500)


def generate(case):
    hm, sem = build(case)
    dim = sem.dim
    rng = np.random.default_rng(20260101)
    out = {}
    if not CASES[case].get("ops", True):
        return generate_propagator(sem, out)
    u = [rng.standard_normal(sem.shape1) for _ in range(dim)]
    w = [rng.standard_normal(sem.shape1) for _ in range(dim)]
    p = rng.standard_normal(sem.shape2)
    q = rng.standard_normal(sem.shape2)
    out["in_u"] = np.stack(u)
    out["in_w"] = np.stack(w)
    out["in_p"] = p
    out["in_q"] = q
    out["bm1"] = sem.bm1
    out["binvm1"] = sem.binvm1
    out["vmult"] = sem.vmult
    out["bm2"] = sem.bm2
    out["axhelm"] = np.stack([sem.axhelm_local(u[i], 0.7, 3.0) for i in range(dim)])
    out["gs"] = np.stack([sem.gs(u[i]) for i in range(dim)])
    out["opdiv"] = sem.opdiv(u)
    out["opgradt"] = np.stack(sem.opgradt(p))
    out["cdabdtp"] = sem.cdabdtp(p)
    out["ediag"] = sem.e_diag()
    out["hdiag"] = sem.gs(sem.helm_diag_local(0.02, 30.0))
    out["conv_dir"] = np.stack(sem.lns_conv_weak(w, u, adjoint=False))
    out["conv_adj"] = np.stack(sem.lns_conv_weak(w, u, adjoint=True))
    out["cfl"] = np.array(sem.compute_cfl(w, 0.01))
    # vector space
    a, b = NekDVector(sem, 1), NekDVector(sem, 1)
    for i in range(dim):
        a.v[i][...] = u[i]
        b.v[i][...] = w[i]
    a.pr[...] = p
    b.pr[...] = q
    a.theta[0][...] = rng.standard_normal(sem.shape1)
    b.theta[0][...] = rng.standard_normal(sem.shape1)
    out["in_ta"] = a.theta[0].copy()
    out["in_tb"] = b.theta[0].copy()
    out["dot"] = np.array(a.dot(b))
    out["size"] = np.array(a.get_size())
    a.save_rst(b, 1)
    a.axpby(0.3, b, -1.7, consistent_rst=False)     # literal reading of real_vectors.f90:188-192
    a.scal(1.0 / 3.0)
    out["axpby_v"] = np.stack(a.v)
    out["axpby_pr"] = a.pr
    out["axpby_theta"] = a.theta[0]
    out["axpby_rst1_v"] = np.stack(a.v_rst[0])
    return generate_propagator(sem, out, case)


def generate_propagator(sem, out, case=""):
    # exptA matvec, direct and adjoint, plus chained matvec using the restart history
    U = base_flow(sem)
    out["baseflow"] = np.stack(U)
    A = ExptA(sem, U, LNSConfig(**lns_cfg()))
    x = NekDVector(sem)
    x.rand(ifnorm=True, seed=5)
    out["mv_in_v"] = np.stack(x.v)
    y = A.matvec(x)
    out["mv_out_v"] = np.stack(y.v)
    out["mv_out_pr"] = y.pr
    out["mv_out_rst2_v"] = np.stack(y.v_rst[1])
    y2 = A.matvec(y)
    out["mv2_out_v"] = np.stack(y2.v)
    z = A.rmatvec(x)
    out["rmv_out_v"] = np.stack(z.v)
    if case == "":
        # three Arnoldi steps (matvec + CGS2): Hessenberg matrix and the last basis vector
        V, H = [x, None, None, None], np.zeros((4, 3))
        for k in range(3):
            arnoldi_step(A.matvec, V, H, k)
        out["arnoldi_H"] = H
        out["arnoldi_v3"] = np.stack(V[3].v)
    # eigs: a converged leading pair (2-D only; tau = 1 separates the spectrum, 15 matvecs)
    if case == "2d":
        cfg = lns_cfg()
        cfg.update(tau=1.0, dt=0.025, re=10.0)
        A2 = ExptA(sem, U, LNSConfig(**cfg))
        lam, vecs, res, nmv = eigs(A2.matvec, x, nev=2, kdim=12, tol=1e-9, max_restarts=6)
        out["eigs_lam"] = lam
        out["eigs_res"] = res
        out["eigs_nmv"] = np.array(nmv)
        out["eigs_vec0"] = np.stack(vecs[0].v)
        out["eigs_vec1"] = np.stack(vecs[1].v)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for case in (sys.argv[1:] or CASES):     # no argument: every case
        data = generate(case)
        path = os.path.join(here, "golden_%s.npz" % case)
        np.savez_compressed(path, **data)
        print(path, os.path.getsize(path), "bytes")
