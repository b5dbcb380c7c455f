"""GPU parity of the exponential propagator, the Arnoldi step and eigs against the CPU oracle."""
import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.krylov import arnoldi_step as o_arnoldi_step
from oracle.krylov import eigs as o_eigs
from oracle.lns import ExptA, LNSConfig
from oracle.sem import SEM
from oracle.vectors import NekDVector

pytestmark = pytest.mark.gpu


def setup_case(ctx, dim, n=6, torder=3, fixed=True, tau=0.05, re=50.0, deform=0.04, pprecond=0, pproj=1):
    if dim == 2:
        hm = box_mesh((4, 3), n, lengths=(4.0, 2.0), periodic=(True, False), deform=deform)
    else:
        hm = box_mesh((3, 2, 2), n, lengths=(3.0, 2.0, 2.0), periodic=(True, False, False), deform=deform)
    sem = SEM(hm)
    gm = host.Mesh(ctx, hm)
    rng = np.random.default_rng(5)
    # smooth-ish divergence-free-agnostic base flow: C0, masked
    ob = NekDVector(sem)
    for i in range(dim):
        ob.v[i][...] = sem.mask[i] * sem.dsavg(np.sin(sem.X[0] * (i + 1)) * np.cos(sem.X[1]) + 0.3 * rng.standard_normal(sem.shape1) * 0)
    ob.v[0][...] += sem.mask[0] * 1.0
    gb = host.nek_dvector(gm)
    for i in range(dim):
        gb.set_field(i, ob.v[i])
    kw = dict(re=re, torder=torder, tau=tau, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=4000)
    if fixed:
        kw.update(fixed_iters_v=30, fixed_iters_p=600)   # converged: unconverged CG amplifies rounding differences
    ocfg = LNSConfig(**kw)
    oA = ExptA(sem, ob.v, ocfg)
    gA = host.exptA_linop(tau, gb, pprecond=pprecond, pproj=pproj, **{k: v for k, v in kw.items() if k != "tau"})
    gA.init()
    return hm, sem, gm, oA, gA, rng


def load_pair(sem, gm, rng):
    ov = NekDVector(sem)
    ov.rand(ifnorm=True, seed=3)
    ov.pr[...] = 0.01 * rng.standard_normal(sem.shape2)
    gv = host.nek_dvector(gm)
    for i in range(sem.dim):
        gv.set_field(i, ov.v[i])
    gv.set_field(host.PR, ov.pr)
    return ov, gv


def cmp_vec(gv, ov, tol, what=""):
    sc = max(np.abs(a).max() for a in ov.v)
    for i in range(len(ov.v)):
        err = np.max(np.abs(gv.get_field(i) - ov.v[i].ravel()))
        assert err < tol * sc, "%s v%d err %.3e (scale %.3e)" % (what, i, err, sc)
    scp = max(np.abs(ov.pr).max(), 1e-300)
    errp = np.max(np.abs(gv.get_field(host.PR) - ov.pr.ravel()))
    assert errp < 10 * tol * max(scp, sc), "%s pr err %.3e (scale %.3e)" % (what, errp, scp)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("adjoint", [False, True])
def test_matvec_fixed_iterations(gpu_ctx, dim, adjoint):
    """Identical discrete operator, identical solver (Jacobi-PCG as in the oracle) -> agreement to rounding."""
    hm, sem, gm, oA, gA, rng = setup_case(gpu_ctx, dim, pprecond=1)
    info = gA.info()
    assert info["nsteps"] == oA.nsteps and abs(info["dt"] - oA.dt) < 1e-15
    ov, gv = load_pair(sem, gm, rng)
    gout = host.nek_dvector(gm)
    (gA.rmatvec if adjoint else gA.matvec)(gv, gout)
    oout = oA.matvec(ov, adjoint=adjoint)
    cmp_vec(gout, oout, 1e-11, "matvec")
    assert gout.nrst == oout.nrst == 2
    for r in (1, 2):
        for i in range(dim):
            assert np.max(np.abs(gout.get_field(i, r) - oout.v_rst[r - 1][i].ravel())) < 1e-11 * np.abs(oout.v[i]).max()
    # second matvec replays the restart history (exponential_propagator.f90:44)
    gout2 = host.nek_dvector(gm)
    (gA.rmatvec if adjoint else gA.matvec)(gout, gout2)
    oout2 = oA.matvec(oout, adjoint=adjoint)
    cmp_vec(gout2, oout2, 1e-10, "matvec2")
    st = gA.stats()
    # (iteration counts agree up to the rounding-level stopping floor)
    assert st["steps"] == oA.stats["steps"]
    assert abs(st["p_iters"] - oA.stats["p_iters"]) <= 0.02 * oA.stats["p_iters"] + 2
    assert abs(st["v_iters"] - oA.stats["v_iters"]) <= 0.02 * oA.stats["v_iters"] + 2


@pytest.mark.parametrize("dim", [2, 3])
def test_two_level_pressure_preconditioner(gpu_ctx, dim):
    """The FDM + coarse-grid preconditioner changes the iteration count, not the answer."""
    hm, sem, gm, oA, gJ, rng = setup_case(gpu_ctx, dim, fixed=False, pprecond=1)
    _, _, _, _, g2, _ = setup_case(gpu_ctx, dim, fixed=False, pprecond=0)
    ov, gv = load_pair(sem, gm, rng)
    gv2 = host.nek_dvector(g2.mesh)
    for i in range(dim):
        gv2.set_field(i, ov.v[i])
    gv2.set_field(host.PR, ov.pr)
    oJ, o2 = host.nek_dvector(gm), host.nek_dvector(g2.mesh)
    gJ.matvec(gv, oJ)
    g2.matvec(gv2, o2)
    sc = max(np.abs(oJ.get_field(i)).max() for i in range(dim))
    for i in range(dim):
        assert np.max(np.abs(oJ.get_field(i) - o2.get_field(i))) < 1e-9 * sc
    assert g2.stats()["p_iters"] < 0.5 * gJ.stats()["p_iters"]
    if dim == 3:
        # the variant without overlap (pprecond = 2): same answer, more iterations than with overlap
        _, _, _, _, g3, _ = setup_case(gpu_ctx, dim, fixed=False, pprecond=2)
        gv3, o3 = host.nek_dvector(g3.mesh), host.nek_dvector(g3.mesh)
        for i in range(dim):
            gv3.set_field(i, ov.v[i])
        gv3.set_field(host.PR, ov.pr)
        g3.matvec(gv3, o3)
        for i in range(dim):
            assert np.max(np.abs(oJ.get_field(i) - o3.get_field(i))) < 1e-9 * sc
        assert g2.stats()["p_iters"] <= g3.stats()["p_iters"] < 0.5 * gJ.stats()["p_iters"]


@pytest.mark.parametrize("overlap", [0, 1])
def test_pressure_preconditioner_is_symmetric_positive(gpu_ctx, overlap):
    """PCG needs M symmetric positive (semi-)definite: checked on the operator itself (nlg_op_pprec), with and without
    the layer of face overlap, on a deformed 3-D mesh with more vertices than the exact coarse solve handles."""
    hm = box_mesh((5, 4, 3), 6, deform=0.05)
    gm = host.Mesh(gpu_ctx, hm)
    lib = gpu_ctx.lib
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal(gm.lpn), rng.standard_normal(gm.lpn)
    va, vb, out = host.nek_dvector(gm), host.nek_dvector(gm), host.nek_dvector(gm)
    va.set_field(host.PR, a)
    vb.set_field(host.PR, b)
    for with_coarse in (0, 1):
        host.check(lib.nlg_op_pprec(gm.h, va.h, out.h, overlap, with_coarse))
        Ma = out.get_field(host.PR)
        host.check(lib.nlg_op_pprec(gm.h, vb.h, out.h, overlap, with_coarse))
        Mb = out.get_field(host.PR)
        assert abs(b @ Ma - a @ Mb) < 1e-11 * abs(b @ Ma) + 1e-13 * np.linalg.norm(Ma) * np.linalg.norm(b)
        assert a @ Ma > 0 and b @ Mb > 0
    with pytest.raises(host.NlgError):
        host.check(lib.nlg_op_pprec(gm.h, va.h, va.h, overlap, 1))     # in == out



def test_matvec_tolerance_mode(gpu_ctx):
    """Tolerance-terminated solves: agreement at the level of the tolerances."""
    # undeformed mesh: the GL(lx2) quadrature of the divergence is then exact, the constant is exactly in
    # the null space of E and `ortho` is consistent, so the result must be discretely solenoidal
    hm, sem, gm, oA, gA, rng = setup_case(gpu_ctx, 2, fixed=False, deform=0.0)
    ov, gv = load_pair(sem, gm, rng)
    gout = host.nek_dvector(gm)
    gA.matvec(gv, gout)
    oout = oA.matvec(ov)
    cmp_vec(gout, oout, 1e-9, "matvec-tol")
    # incompressibility of the result
    div = sem.opdiv([gout.get_field(i).reshape(sem.shape1) for i in range(2)]) / sem.bm2
    assert np.sqrt(np.sum(div ** 2 * sem.bm2) / sem.volvm2) < 1e-10


def test_matvec_errors(gpu_ctx):
    hm, sem, gm, oA, gA, rng = setup_case(gpu_ctx, 2)
    ov, gv = load_pair(sem, gm, rng)
    with pytest.raises(host.NlgError):
        gA.matvec(gv, gv)                      # intent(in)/intent(out) aliasing
    with pytest.raises(TypeError):
        gA.matvec(gv, object())                # type_error
    other = host.Mesh(gpu_ctx, box_mesh((2, 2), 6))
    with pytest.raises(host.NlgError):
        gA.matvec(gv, host.nek_dvector(other))
    hb = host.nek_dvector(gm)
    with pytest.raises(host.NlgError):
        host.exptA_linop(1.0, hb, fixed_iters_v=1).matvec(gv, host.nek_dvector(gm))   # init() not called


@pytest.mark.parametrize("dim", [2, 3])
def test_arnoldi_steps(gpu_ctx, dim):
    hm, sem, gm, oA, gA, rng = setup_case(gpu_ctx, dim, tau=0.03)
    m = 5
    ov, gv = load_pair(sem, gm, rng)
    B = host.KrylovBasis(gm, m + 1)
    B[0].assign(gv)
    H = np.zeros((m + 1, m), order="F")
    oV = [None] * (m + 1)
    oV[0] = ov
    oH = np.zeros((m + 1, m))
    for k in range(m):
        host.arnoldi_step(gA, B, k, H)
        o_arnoldi_step(oA.matvec, oV, oH, k)
    assert np.max(np.abs(H - oH)) < 1e-10 * np.max(np.abs(oH))
    for k in range(m + 1):
        cmp_vec(B[k], oV[k], 1e-9, "basis %d" % k)
    # orthonormality in the mass inner product
    G = np.array([[B[i].dot(B[j]) for j in range(m + 1)] for i in range(m + 1)])
    assert np.max(np.abs(G - np.eye(m + 1))) < 1e-13


def test_eigs_against_oracle(gpu_ctx):
    """Ritz values within 1e-10 relative, Ritz vectors within 1e-6 (BASELINE.json north_star), on a
    converged, well-separated leading pair; complex pairs are compared as 2-D invariant subspaces
    because the phase of a complex eigenvector is a convention of the dense eigen-solver."""
    hm, sem, gm, oA, gA, rng = setup_case(gpu_ctx, 2, tau=1.0, re=10.0)
    ov, gv = load_pair(sem, gm, rng)
    nev, kdim = 3, 14
    X = [host.nek_dvector(gm) for _ in range(nev)]
    import os
    import tempfile
    log = os.path.join(tempfile.mkdtemp(), "eigs_output.txt")
    mu, res, info = host.eigs(gA, X, kdim=kdim, tol=1e-9, x0=gv, logfile=log, max_restarts=8)
    olam, ovecs, ores, onmv = o_eigs(oA.matvec, ov, nev, kdim, tol=1e-9, max_restarts=8)
    assert info == onmv
    assert np.max(np.abs(mu - olam) / np.abs(olam)) < 1e-10
    assert np.all(res[:nev] < 1e-8) and np.max(np.abs(res - ores)) < 1e-9

    def flat(v):
        return np.concatenate([v.get_field(i) for i in range(2)]) if hasattr(v, "get_field") else np.concatenate([a.ravel() for a in v.v])

    # The converged Ritz vectors span the same invariant subspace (the split of a subspace into
    # individual vectors -- phase of a complex pair, two nearly equal real eigenvalues -- is a convention
    # of the dense eigen-solver, the subspace is not).
    ncmp = nev - 1 if (abs(olam[nev - 1].imag) > 0 and not np.isclose(olam[nev - 1], np.conj(olam[nev - 2]))) else nev
    Bo = np.stack([flat(v) for v in ovecs[:ncmp]], axis=1)
    for q in range(ncmp):
        a = flat(X[q])
        coef, *_ = np.linalg.lstsq(Bo, a, rcond=None)
        assert np.max(np.abs(a - Bo @ coef)) < 1e-6 * np.max(np.abs(a))
    # eigs_output.txt in the format test/lib/neklabTestCase.py:425-449 parses
    rows = [ln.split() for ln in open(log) if not ln.startswith("#")]
    assert len(rows) >= nev and all(len(r) == 6 and r[5] in ("T", "F") for r in rows)
    assert abs(float(rows[0][3]) - abs(mu[0])) < 1e-12


def test_no_history_mode_matches_oracle(gpu_ctx):
    """cfg.no_history (the memory plan for bases that do not fit with lorder - 1 history copies per vector): vectors of lorder = 1, every
    matvec starts impulsively, no history steps -- against the oracle twin (LNSConfig.no_history): matvec fields, Hessenberg matrix
    of a 6-step Arnoldi run and its Ritz values to the north_star tolerance 1e-10; and against the reference's protocol on the same
    vectors: a different start-up, hence a slightly different propagator."""
    hm = box_mesh((4, 3), 6, lengths=(4.0, 2.0), periodic=(True, False), deform=0.04)
    sem, gm = SEM(hm), host.Mesh(gpu_ctx, hm)
    U = [sem.mask[0] * (1.0 + np.sin(sem.X[0]) * np.cos(sem.X[1])), sem.mask[1] * np.sin(2 * sem.X[0]) * np.cos(sem.X[1])]
    kw = dict(re=10.0, torder=3, tau=0.3, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=4000)
    oA = ExptA(sem, U, LNSConfig(no_history=True, **kw))
    gb = host.nek_dvector(gm, 0, 1)
    for i in range(2):
        gb.set_field(i, U[i])
    gA = host.exptA_linop(kw["tau"], gb, no_history=1, **{k: v for k, v in kw.items() if k != "tau"})
    gA.init()
    ov = NekDVector(sem)
    ov.rand(ifnorm=True, seed=3)
    m = 6
    B = host.KrylovBasis(gm, m + 1, 0, 1)                # lorder = 1: a third of the memory
    for i in range(2):
        B[0].set_field(i, ov.v[i])
    H, oH = np.zeros((m + 1, m), order="F"), np.zeros((m + 1, m))
    oV = [ov] + [None] * m
    for k in range(m):
        host.arnoldi_step(gA, B, k, H)
        o_arnoldi_step(oA.matvec, oV, oH, k)
    assert B[m].nrst == 0 and not oV[m].has_rst_fields()
    assert np.max(np.abs(H - oH)) < 1e-10 * np.max(np.abs(oH))
    ev, oev = np.sort_complex(np.linalg.eigvals(H[:m])), np.sort_complex(np.linalg.eigvals(oH[:m]))
    assert np.max(np.abs(ev - oev)) < 1e-10 * np.max(np.abs(oev))
    cmp_vec(B[m], oV[m], 1e-9, "last basis vector")
    # lorder = 1 vectors are refused by the reference's protocol
    gA3 = host.exptA_linop(kw["tau"], gb, **{k: v for k, v in kw.items() if k != "tau"})
    gA3.init()
    with pytest.raises(host.NlgError):
        gA3.matvec(B[0], B[1])


def test_pressure_residual_projection(gpu_ctx):
    """residualProj (1cyl.par:23): the projection onto the previous increments changes the starting guess of the
    pressure solve, not the answer; it saves iterations once the matvec has several time steps."""
    outs, iters = [], []
    for pproj in (0, 1):
        hm, sem, gm, oA, gA, rng = setup_case(gpu_ctx, 3, fixed=False, tau=0.3, pproj=pproj)
        ov, gv = load_pair(sem, gm, rng)
        out = host.nek_dvector(gm)
        gA.matvec(gv, out)
        outs.append([out.get_field(i) for i in range(3)])
        iters.append(gA.stats()["p_iters"])
        assert gA.stats()["steps"] >= 8
    sc = max(np.abs(a).max() for a in outs[0])
    for a, b in zip(*outs):
        assert np.max(np.abs(a - b)) < 1e-9 * sc
    assert iters[1] < iters[0]
