"""GPU parity of the propagator and the Arnoldi step against the oracle at lx1 = 8 in 3-D: the kernel instantiations the
headline benchmark is made of (k_axhelm3r<8,4> with the fused direction update and (p, w) sums, k_cg_update<3>, face-grouped
k_opdiv3 / k_opgradt3<8,3>, k_fdm_ext<8,1>, k_sch_finish<8>, k_conv3<8,12>, the done-flag gating, the pressure projection),
inside the time stepper rather than operator by operator.

Two modes.  Fixed iteration counts with the Jacobi preconditioner (pprecond = 1): the iteration is the oracle's own, so
the results agree to rounding (1e-11).  Tolerance mode at 1e-13 with the two-level Schwarz preconditioner (pprecond = 0, 2),
with and without the residual projection: a different iteration for the same discrete problem, agreement at the level the
solves are converged to (1e-9).  Direct and adjoint.  Plus the committed fixture tests/golden/golden_3d_n8.npz.
"""
import os
import sys

import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.krylov import arnoldi_step as o_arnoldi_step
from oracle.lns import ExptA, LNSConfig
from oracle.sem import SEM
from oracle.vectors import NekDVector

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

pytestmark = pytest.mark.gpu

NEL, N, LEN, PER = (3, 3, 2), 8, (3.0, 3.0, 2.0), (True, False, False)
_cache = {}


def mesh():
    if "hm" not in _cache:
        hm = box_mesh(NEL, N, lengths=LEN, periodic=PER, deform=0.04)
        _cache["hm"], _cache["sem"] = hm, SEM(hm)
    return _cache["hm"], _cache["sem"]


def base_flow(sem):
    U = [sem.mask[i] * sem.dsavg(np.sin(sem.X[0] * (i + 1)) * np.cos(sem.X[1]) * np.cos(0.5 * sem.X[2] + i)) for i in range(3)]
    U[0] = U[0] + sem.mask[0]
    return U


def start_vector(sem):
    ov = NekDVector(sem)
    ov.rand(ifnorm=True, seed=3)
    ov.pr[...] = 0.01 * np.random.default_rng(5).standard_normal(sem.shape2)
    return ov


def oracle_case(fixed, adjoint):
    """(operator, start vector, result of one matvec, result of a second, chained one) -- computed once per mode"""
    key = ("o", fixed, adjoint)
    if key not in _cache:
        hm, sem = mesh()
        kw = dict(re=50.0, torder=3, tau=0.03, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=4000)
        if fixed:
            kw.update(fixed_iters_v=40, fixed_iters_p=600)
        oA = ExptA(sem, base_flow(sem), LNSConfig(**kw))
        ov = start_vector(sem)
        o1 = oA.matvec(ov, adjoint=adjoint)
        o2 = oA.matvec(o1, adjoint=adjoint)
        _cache[key] = (oA, kw, ov, o1, o2)
    return _cache[key]


def gpu_case(ctx, kw, pprecond, pproj):
    hm, sem = mesh()
    gm = host.Mesh(ctx, hm)
    gb = host.nek_dvector(gm)
    for i, u in enumerate(base_flow(sem)):
        gb.set_field(i, u)
    gA = host.exptA_linop(kw["tau"], gb, pprecond=pprecond, pproj=pproj, **{k: v for k, v in kw.items() if k != "tau"})
    gA.init()
    return gm, gA


def upload(gm, ov):
    gv = host.nek_dvector(gm)
    for i in range(3):
        gv.set_field(i, ov.v[i])
    gv.set_field(host.PR, ov.pr)
    return gv


def cmp_vec(gv, ov, tol, what):
    sc = max(np.abs(a).max() for a in ov.v)
    for i in range(3):
        err = np.max(np.abs(gv.get_field(i) - ov.v[i].ravel()))
        assert err < tol * sc, "%s v%d err %.3e (scale %.3e)" % (what, i, err, sc)
    errp = np.max(np.abs(gv.get_field(host.PR) - ov.pr.ravel()))
    assert errp < 10 * tol * max(np.abs(ov.pr).max(), sc), "%s pr err %.3e" % (what, errp)


@pytest.mark.parametrize("adjoint", [False, True])
def test_matvec_n8_fixed_iterations(gpu_ctx, adjoint):
    oA, kw, ov, o1, o2 = oracle_case(True, adjoint)
    gm, gA = gpu_case(gpu_ctx, kw, pprecond=1, pproj=0)
    info = gA.info()
    assert info["nsteps"] == oA.nsteps and abs(info["dt"] - oA.dt) < 1e-15
    gv, g1, g2 = upload(gm, ov), host.nek_dvector(gm), host.nek_dvector(gm)
    mv = gA.rmatvec if adjoint else gA.matvec
    mv(gv, g1)
    cmp_vec(g1, o1, 1e-11, "matvec")
    assert g1.nrst == o1.nrst == 2
    for r in (1, 2):
        for i in range(3):
            assert np.max(np.abs(g1.get_field(i, r) - o1.v_rst[r - 1][i].ravel())) < 1e-11 * np.abs(o1.v[i]).max()
    mv(g1, g2)                         # replays the restart history (exponential_propagator.f90:44)
    cmp_vec(g2, o2, 1e-10, "matvec2")


@pytest.mark.parametrize("adjoint", [False, True])
@pytest.mark.parametrize("pprecond,pproj", [(0, 1), (0, 0), (2, 1), (1, 1)])
def test_matvec_n8_tolerance_mode(gpu_ctx, adjoint, pprecond, pproj):
    oA, kw, ov, o1, o2 = oracle_case(False, adjoint)
    gm, gA = gpu_case(gpu_ctx, kw, pprecond, pproj)
    gv, g1, g2 = upload(gm, ov), host.nek_dvector(gm), host.nek_dvector(gm)
    mv = gA.rmatvec if adjoint else gA.matvec
    mv(gv, g1)
    cmp_vec(g1, o1, 1e-9, "matvec")
    for r in (1, 2):
        for i in range(3):
            assert np.max(np.abs(g1.get_field(i, r) - o1.v_rst[r - 1][i].ravel())) < 1e-9 * np.abs(o1.v[i]).max()
    mv(g1, g2)
    cmp_vec(g2, o2, 1e-9, "matvec2")
    st = gA.stats()
    assert st["steps"] == 2 * (oA.nsteps + 2)
    if pprecond != 1:                  # the two-level preconditioner must actually be at work
        assert st["p_iters"] / st["steps"] < 100, st     # Jacobi needs ~385 per step at this tolerance (1e-13)


@pytest.mark.parametrize("fixed", [True, False])
def test_arnoldi_n8(gpu_ctx, fixed):
    """Three Arnoldi steps (matvec + CGS2 + normalisation): Hessenberg matrix and basis against the oracle."""
    oA, kw, ov, _, _ = oracle_case(fixed, False)
    gm, gA = gpu_case(gpu_ctx, kw, pprecond=1 if fixed else 0, pproj=1)
    m = 3
    B = host.KrylovBasis(gm, m + 1)
    B[0].assign(upload(gm, ov))
    H, oH = np.zeros((m + 1, m), order="F"), np.zeros((m + 1, m))
    oV = [ov.copy()] + [None] * m
    for k in range(m):
        host.arnoldi_step(gA, B, k, H)
        o_arnoldi_step(oA.matvec, oV, oH, k)
    tol = 1e-10 if fixed else 1e-9
    assert np.max(np.abs(H - oH)) < tol * np.max(np.abs(oH)), np.max(np.abs(H - oH))
    for k in range(m + 1):
        cmp_vec(B[k], oV[k], 10 * tol, "basis %d" % k)
    G = np.array([[B[i].dot(B[j]) for j in range(m + 1)] for i in range(m + 1)])
    assert np.max(np.abs(G - np.eye(m + 1))) < 1e-13


def test_golden_n8(gpu_ctx):
    """The committed fixture (made by tests/golden/make_golden.py 3d_n8): default preconditioner of the library."""
    g = np.load(os.path.join(HERE, "golden", "golden_3d_n8.npz"))
    c = mg.CASES["3d_n8"]
    hm = box_mesh(c["nel"], c["n"], lengths=c["lengths"], periodic=c["periodic"], deform=c["deform"])
    gm = host.Mesh(gpu_ctx, hm)
    bf = host.nek_dvector(gm)
    for i in range(3):
        bf.set_field(i, g["baseflow"][i])
    cfg = mg.lns_cfg()
    tau = cfg.pop("tau")
    for pprecond in (1, 0):
        A = host.exptA_linop(tau, bf, pprecond=pprecond, **cfg)
        A.init()
        x, y, y2, z = (host.nek_dvector(gm) for _ in range(4))
        for i in range(3):
            x.set_field(i, g["mv_in_v"][i])
        A.matvec(x, y)
        sc = np.abs(g["mv_out_v"]).max()
        tol = 1e-10 if pprecond == 1 else 1e-9
        assert max(np.abs(y.get_field(i) - g["mv_out_v"][i].ravel()).max() for i in range(3)) < tol * sc
        assert np.abs(y.get_field(host.PR) - g["mv_out_pr"].ravel()).max() < 10 * tol * max(sc, np.abs(g["mv_out_pr"]).max())
        assert max(np.abs(y.get_field(i, 2) - g["mv_out_rst2_v"][i].ravel()).max() for i in range(3)) < tol * sc
        A.matvec(y, y2)
        assert max(np.abs(y2.get_field(i) - g["mv2_out_v"][i].ravel()).max() for i in range(3)) < 10 * tol * sc
        A.rmatvec(x, z)
        assert max(np.abs(z.get_field(i) - g["rmv_out_v"][i].ravel()).max() for i in range(3)) < tol * sc
        B = host.KrylovBasis(gm, 4)
        B[0].assign(x)
        H = np.zeros((4, 3), order="F")
        for k in range(3):
            host.arnoldi_step(A, B, k, H)
        assert np.max(np.abs(H - g["arnoldi_H"])) < 10 * tol * np.max(np.abs(g["arnoldi_H"]))
        assert max(np.abs(B[3].get_field(i) - g["arnoldi_v3"][i].ravel()).max() for i in range(3)) < 100 * tol * np.abs(g["arnoldi_v3"]).max()
