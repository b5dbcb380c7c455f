"""The lx1 > 8 kernel instantiations (BASELINE.json configs 4 and 5: lx1 = 10, 12) inside the time stepper, against the
oracle: k_axhelm3c<10|12> (register columns, block barriers), the 128-thread k_opgradt3 / k_opdiv3<N, 1>, the multi-wave
k_fdm_ext<N, 1, 2|3>, the dynamic-LDS k_conv3<10, 15> / <12, 18>, natural-layout velocity PCG.  Tolerance mode with the
two-level Schwarz preconditioner (1e-9) and fixed iteration counts with Jacobi (1e-11), direct and adjoint."""
import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.krylov import arnoldi_step as o_arnoldi_step
from oracle.lns import ExptA, LNSConfig
from oracle.sem import SEM
from oracle.vectors import NekDVector

pytestmark = pytest.mark.gpu


def case(ctx, n, fixed, pprecond):
    nel = (2, 2, 2) if n == 10 else (2, 2, 1)
    hm = box_mesh(nel, n, lengths=(2.0, 2.0, float(nel[2])), periodic=(True, False, False), deform=0.04)
    sem, gm = SEM(hm), host.Mesh(ctx, hm)
    U = [sem.mask[i] * sem.dsavg(np.sin(sem.X[0] * (i + 1)) * np.cos(sem.X[1]) * np.cos(0.5 * sem.X[2] + i)) for i in range(3)]
    U[0] = U[0] + sem.mask[0]
    kw = dict(re=50.0, torder=3, tau=0.02, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=6000)
    if fixed:
        kw.update(fixed_iters_v=40, fixed_iters_p=900)
    oA = ExptA(sem, U, LNSConfig(**kw))
    gb = host.nek_dvector(gm)
    for i in range(3):
        gb.set_field(i, U[i])
    gA = host.exptA_linop(kw["tau"], gb, pprecond=pprecond, **{k: v for k, v in kw.items() if k != "tau"})
    gA.init()
    ov = NekDVector(sem)
    ov.rand(ifnorm=True, seed=3)
    ov.pr[...] = 0.01 * np.random.default_rng(5).standard_normal(sem.shape2)
    gv = host.nek_dvector(gm)
    for i in range(3):
        gv.set_field(i, ov.v[i])
    gv.set_field(host.PR, ov.pr)
    return sem, gm, oA, gA, ov, gv


def cmp_vec(gv, ov, tol, what):
    sc = max(np.abs(a).max() for a in ov.v)
    for i in range(3):
        err = np.max(np.abs(gv.get_field(i) - ov.v[i].ravel()))
        assert err < tol * sc, "%s v%d err %.3e (scale %.3e)" % (what, i, err, sc)
    errp = np.max(np.abs(gv.get_field(host.PR) - ov.pr.ravel()))
    assert errp < 10 * tol * max(np.abs(ov.pr).max(), sc), "%s pr err %.3e" % (what, errp)


@pytest.mark.parametrize("n", [9, 10, 12])
@pytest.mark.parametrize("fixed,pprecond,adjoint", [(False, 0, False), (True, 1, True)])
def test_matvec_large_lx1(gpu_ctx, n, fixed, pprecond, adjoint):
    sem, gm, oA, gA, ov, gv = case(gpu_ctx, n, fixed, pprecond)
    g1, g2 = host.nek_dvector(gm), host.nek_dvector(gm)
    mv = gA.rmatvec if adjoint else gA.matvec
    mv(gv, g1)
    o1 = oA.matvec(ov, adjoint=adjoint)
    tol = 1e-11 if fixed else 1e-9
    cmp_vec(g1, o1, tol, "matvec")
    for i in range(3):
        assert np.max(np.abs(g1.get_field(i, 2) - o1.v_rst[1][i].ravel())) < tol * np.abs(o1.v[i]).max()
    mv(g1, g2)                                   # replays the restart history
    o2 = oA.matvec(o1, adjoint=adjoint)
    cmp_vec(g2, o2, 10 * tol, "matvec2")
    if pprecond == 0:
        st = gA.stats()
        assert st["p_iters"] / st["steps"] < 150, st


def test_arnoldi_lx1_10(gpu_ctx):
    sem, gm, oA, gA, ov, gv = case(gpu_ctx, 10, False, 0)
    m = 2
    B = host.KrylovBasis(gm, m + 1)
    B[0].assign(gv)
    H, oH = np.zeros((m + 1, m), order="F"), np.zeros((m + 1, m))
    oV = [ov.copy()] + [None] * m
    for k in range(m):
        host.arnoldi_step(gA, B, k, H)
        o_arnoldi_step(oA.matvec, oV, oH, k)
    assert np.max(np.abs(H - oH)) < 1e-9 * np.max(np.abs(oH))
    for k in range(m + 1):
        cmp_vec(B[k], oV[k], 1e-8, "basis %d" % k)
