"""Block orthogonalisation and block Arnoldi (BASELINE.json config 5 names a block-Arnoldi run; the reference advances
several perturbations together through Nek5000's lpert / npert, src/neklab_nek_setup.f90:39-247) on the GPU against the
oracle twin, and against the single-vector path: the same spectrum from a block Krylov space."""
import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.krylov import block_arnoldi_step as o_block_step
from oracle.krylov import block_cgs2 as o_block_cgs2
from oracle.krylov import cgs2_step
from oracle.lns import ExptA, LNSConfig
from oracle.sem import SEM
from oracle.vectors import NekDVector

pytestmark = pytest.mark.gpu


def pair(sem, gm, seed, with_history):
    ov = NekDVector(sem)
    ov.rand(ifnorm=True, seed=seed)
    ov.pr[...] = 0.1 * np.random.default_rng(seed).standard_normal(sem.shape2)
    if with_history:
        for r in (1, 2):
            h = NekDVector(sem)
            h.rand(ifnorm=True, seed=100 * r + seed)
            ov.save_rst(h, r)
    return ov


def upload(gv, ov, dim):
    for i in range(dim):
        gv.set_field(i, ov.v[i])
    gv.set_field(host.PR, ov.pr)
    for r in range(ov.nrst):
        tmp = host.nek_dvector(gv.mesh)
        for i in range(dim):
            tmp.set_field(i, ov.v_rst[r][i])
        tmp.set_field(host.PR, ov.pr_rst[r])
        gv.save_rst(tmp, r + 1)


@pytest.mark.parametrize("dim,n", [(2, 6), (3, 5), (3, 8)])
@pytest.mark.parametrize("s", [2, 3, 4])
def test_block_cgs2_matches_oracle(gpu_ctx, dim, n, s):
    nel = (4, 3) if dim == 2 else (3, 2, 2)
    hm = box_mesh(nel, n, periodic=(True,) + (False,) * (dim - 1), deform=0.04)
    sem, gm = SEM(hm), host.Mesh(gpu_ctx, hm)
    k = 9                                          # not a multiple of the kernel's tile of 8 basis vectors
    B = host.KrylovBasis(gm, k + s)
    oV = []
    for j in range(k):                              # an orthonormal basis with history, built by the single-vector path on both sides
        ov = pair(sem, gm, 10 + j, with_history=True)
        upload(B[j], ov, dim)
        B.cgs2(j, B[j])
        if j:
            cgs2_step(oV, ov)
        ov.scal(1.0 / ov.norm())
        oV.append(ov)
    for v in range(s):
        ov = pair(sem, gm, 50 + v, with_history=True)
        upload(B[k + v], ov, dim)
        oV.append(ov)
    coef = B.block_cgs2(k, s)
    ocoef = o_block_cgs2(oV, k, s)
    assert np.max(np.abs(coef - ocoef)) < 1e-11 * np.max(np.abs(ocoef)), np.max(np.abs(coef - ocoef))
    assert np.allclose(np.tril(coef[k:], -1), 0.0)                     # R upper triangular
    sc = max(np.abs(a).max() for a in oV[k].v)
    for v in range(s):
        for i in range(dim):
            assert np.max(np.abs(B[k + v].get_field(i) - oV[k + v].v[i].ravel())) < 1e-10 * sc
            assert np.max(np.abs(B[k + v].get_field(i, 2) - oV[k + v].v_rst[1][i].ravel())) < 1e-10 * sc     # history follows
        assert np.max(np.abs(B[k + v].get_field(host.PR) - oV[k + v].pr.ravel())) < 1e-10 * max(sc, np.abs(oV[k + v].pr).max())
    G = np.array([[B[i].dot(B[j]) for j in range(k + s)] for i in range(k + s)])
    assert np.max(np.abs(G - np.eye(k + s))) < 1e-13
    # s = 1 falls back to the single-vector path with the same output convention
    w = pair(sem, gm, 99, with_history=False)
    B2 = host.KrylovBasis(gm, 3)
    for j in range(2):
        upload(B2[j], oV[j], dim)
    upload(B2[2], w, dim)
    c1 = B2.block_cgs2(2, 1)
    assert abs(c1[0, 0] - oV[0].dot(w)) < 1e-12 and c1[2, 0] > 0
    with pytest.raises(host.NlgError):
        B.block_cgs2(k, 5)


def test_block_cgs2_deflates_a_dependent_column(gpu_ctx):
    """A block whose third column is a combination of a basis vector and the first new column (a block Krylov space that has reached
    an invariant subspace; a drawn column next to x0): not an error -- the column is deflated (zero vector, zero diagonal entry of R,
    its coefficients on the others as for any column), the other columns come out orthonormal, the rank is reported, and the oracle
    twin does the same.  A block in which only one column carries a restart history leaves zeros, not stale memory, in the others."""
    hm = box_mesh((3, 2, 2), 5, periodic=(True, False, False), deform=0.04)
    sem, gm = SEM(hm), host.Mesh(gpu_ctx, hm)
    k, s = 4, 3
    B = host.KrylovBasis(gm, k + s)
    oV = []
    for j in range(k):
        ov = pair(sem, gm, 10 + j, with_history=False)
        upload(B[j], ov, 3)
        B.cgs2(j, B[j])
        if j:
            cgs2_step(oV, ov)
        ov.scal(1.0 / ov.norm())
        oV.append(ov)
    new = [pair(sem, gm, 60, with_history=True), pair(sem, gm, 61, with_history=False)]
    dep = new[0].copy()
    dep.nrst = 0
    dep.scal(0.7)
    dep.axpby(0.4, oV[1], 1.0)                       # 0.7 w_0 + 0.4 v_1: in the span of the basis and the first new column
    new.append(dep)
    # stale data in the history slots of a column that declares none: must not leak into the result
    junk = host.nek_dvector(gm)
    junk.rand(False, seed=5)
    for v in range(s):
        B[k + v].save_rst(junk, 1)
        B[k + v].save_rst(junk, 2)
        B[k + v].clear_rst_fields()
        upload(B[k + v], new[v], 3)
        oV.append(new[v])
    coef = B.block_cgs2(k, s)
    ocoef = o_block_cgs2(oV, k, s)
    assert B.last_block_rank() == 2
    assert coef[k + 2, 2] == 0.0 and ocoef[k + 2, 2] == 0.0
    assert np.max(np.abs(coef - ocoef)) < 1e-11 * np.max(np.abs(ocoef)), np.max(np.abs(coef - ocoef))
    assert B[k + 2].norm() == 0.0 and abs(coef[1, 2] - (0.4 + 0.7 * coef[1, 0])) < 1e-12 and abs(coef[k, 2] - 0.7 * coef[k, 0]) < 1e-12
    G = np.array([[B[i].dot(B[j]) for j in range(k + 2)] for i in range(k + 2)])
    assert np.max(np.abs(G - np.eye(k + 2))) < 1e-13
    # history: column k carried one, the others get the zero history blocks they declared (combined consistently), not `junk`
    for v in range(2):
        for r in (1, 2):
            d = max(np.max(np.abs(B[k + v].get_field(i, r) - oV[k + v].v_rst[r - 1][i].ravel())) for i in range(3))
            assert d < 1e-10, (v, r, d)


def _basis_of(sem, gm, k, s):
    B = host.KrylovBasis(gm, k + s)
    oV = []
    for j in range(k):
        ov = pair(sem, gm, 10 + j, with_history=False)
        upload(B[j], ov, 3)
        B.cgs2(j, B[j])
        if j:
            cgs2_step(oV, ov)
        ov.scal(1.0 / ov.norm())
        oV.append(ov)
    return B, oV


def test_block_cgs2_keeps_columns_of_large_norm(gpu_ctx):
    """ADVICE round 3 (vec.hip deflation test): the pivot of the SECOND CholQR round must be compared with the column's norm at entry
    to that round, not with the norm it had before the first -- healthy columns of norm 1e8 / 1e-8 are kept, like columns of norm 1."""
    hm = box_mesh((3, 2, 2), 5, periodic=(True, False, False), deform=0.04)
    sem, gm = SEM(hm), host.Mesh(gpu_ctx, hm)
    k, s = 3, 3
    B, oV = _basis_of(sem, gm, k, s)
    new = [pair(sem, gm, 60 + v, with_history=False) for v in range(s)]
    for v, scale in enumerate((1e8, 1.0, 1e-8)):
        new[v].scal(scale)
        upload(B[k + v], new[v], 3)
        oV.append(new[v])
    coef = B.block_cgs2(k, s)
    ocoef = o_block_cgs2(oV, k, s)
    assert B.last_block_rank() == 3
    assert np.all(np.abs(coef - ocoef) < 1e-10 * np.max(np.abs(ocoef), axis=0)), np.max(np.abs(coef - ocoef))
    G = np.array([[B[i].dot(B[j]) for j in range(k + s)] for i in range(k + s)])
    assert np.max(np.abs(G - np.eye(k + s))) < 1e-12


def test_block_cgs2_deflates_a_block_that_lies_in_the_span_of_the_basis(gpu_ctx):
    """The documented 'invariant subspace reached' case: every column of the new block is a combination of basis vectors.  What the two
    projections leave is rounding noise whose columns are NOT mutually dependent, so only the comparison with the norm before
    the projections (|w|^2 = |w - V h|^2 + |h|^2) can see it: rank 0, zero vectors, the coefficients on the basis returned."""
    hm = box_mesh((3, 2, 2), 5, periodic=(True, False, False), deform=0.04)
    sem, gm = SEM(hm), host.Mesh(gpu_ctx, hm)
    k, s = 4, 2
    B, oV = _basis_of(sem, gm, k, s)
    rng = np.random.default_rng(3)
    C = rng.standard_normal((k, s))
    for v in range(s):
        w = oV[0].copy()
        w.scal(C[0, v])
        for j in range(1, k):
            w.axpby(C[j, v], oV[j], 1.0)
        upload(B[k + v], w, 3)
        oV.append(w)
    coef = B.block_cgs2(k, s)
    ocoef = o_block_cgs2(oV, k, s)
    assert B.last_block_rank() == 0
    assert np.max(np.abs(coef[:k] - C)) < 1e-12 and np.max(np.abs(ocoef[:k] - C)) < 1e-12
    assert np.all(np.diag(coef[k:]) == 0.0) and np.all(np.diag(ocoef[k:]) == 0.0)
    assert B[k].norm() == 0.0 and B[k + 1].norm() == 0.0


def test_block_arnoldi_matches_oracle_and_single_vector_spectrum(gpu_ctx):
    hm = box_mesh((4, 3), 6, lengths=(4.0, 2.0), periodic=(True, False), deform=0.04)
    sem, gm = SEM(hm), host.Mesh(gpu_ctx, hm)
    U = [sem.mask[0] * (1.0 + np.sin(sem.X[0]) * np.cos(sem.X[1])), sem.mask[1] * np.sin(2 * sem.X[0]) * np.cos(sem.X[1])]
    kw = dict(re=10.0, torder=3, tau=1.0, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=4000)
    oA = ExptA(sem, U, LNSConfig(**kw))
    gb = host.nek_dvector(gm)
    for i in range(2):
        gb.set_field(i, U[i])
    gA = host.exptA_linop(1.0, gb, **{k: v for k, v in kw.items() if k != "tau"})
    gA.init()
    s, nblk = 2, 2
    m = s * nblk
    B = host.KrylovBasis(gm, m + s)
    oV = [pair(sem, gm, 3 + v, with_history=False) for v in range(s)] + [None] * m
    for v in range(s):
        upload(B[v], oV[v], 2)
    B.block_cgs2(0, s)
    o_block_cgs2(oV, 0, s)
    H, oH = np.zeros((m + s, m), order="F"), np.zeros((m + s, m))
    for j in range(nblk):
        host.block_arnoldi_step(gA, B, j * s, s, H)
        o_block_step(oA.matvec, oV, oH, j * s, s)
    assert np.max(np.abs(H - oH)) < 1e-9 * np.max(np.abs(oH)), np.max(np.abs(H - oH))
    # the block Arnoldi relation A V_m = V_{m+s} H on the device vectors themselves
    for c in range(m):
        w = host.nek_dvector(gm)
        gA.matvec(B[c], w)
        r = w.copy()
        for i in range(m + s):
            r.axpby(-H[i, c], B[i], 1.0)
        assert r.norm() < 1e-9 * w.norm()
    # a longer block run finds the leading eigenvalue of the single-vector Arnoldi
    m2 = 24
    B2 = host.KrylovBasis(gm, m2 + s)
    for v in range(s):                       # start from IMAGES of random vectors: every column then carries a restart
        x = host.nek_dvector(gm)             # history and the whole block Krylov space belongs to one linear map
        x.rand(True, seed=21 + v)            # (otherwise the converged Ritz values depend on the start vectors)
        gA.matvec(x, B2[v])
    B2.block_cgs2(0, s)
    H2 = np.zeros((m2 + s, m2), order="F")
    for j in range(m2 // s):
        host.block_arnoldi_step(gA, B2, j * s, s, H2)
    lam = np.linalg.eigvals(H2[:m2, :m2])
    lead = lam[np.argmax(np.abs(lam))]
    X = [host.nek_dvector(gm) for _ in range(2)]
    mu, res, info = host.eigs(gA, X, kdim=24, tol=1e-9, write_intermediate=False, seed=1, warm_start=True)
    assert abs(abs(lead) - abs(mu[0])) < 1e-7 * abs(mu[0]), (lead, mu[0])


def test_eigs_block_mode_and_warm_start(gpu_ctx):
    """nlg_eigs options: block_size = 2 (block Arnoldi inside eigs) finds the eigenvalues of the single-vector run started
    the same way; warm_start removes the dependence of the converged Ritz values on the start vector that the history-
    free first Krylov column causes: two seeds agree to 1e-8 warm, and differ by more than 1e-5 cold."""
    hm = box_mesh((4, 3), 6, lengths=(4.0, 2.0), periodic=(True, False), deform=0.04)
    gm = host.Mesh(gpu_ctx, hm)
    U = [hm.mask[0] * (1.0 + np.sin(hm.x) * np.cos(hm.y)), hm.mask[1] * np.sin(2 * hm.x) * np.cos(hm.y)]
    gb = host.nek_dvector(gm)
    for i in range(2):
        gb.set_field(i, U[i])
    gA = host.exptA_linop(1.0, gb, re=10.0, torder=3, vtol=1e-12, ptol=1e-12, maxit_v=400, maxit_p=4000)
    gA.init()

    def lead(**kw):
        X = [host.nek_dvector(gm) for _ in range(2)]
        mu, res, info = host.eigs(gA, X, kdim=28, tol=1e-9, write_intermediate=False, **kw)
        assert res[0] < 1e-9
        return mu[0], info, X

    cold = [lead(seed=sd)[0] for sd in (1, 2)]
    warm = [lead(seed=sd, warm_start=True)[0] for sd in (1, 2)]
    assert abs(warm[0] - warm[1]) < 1e-8 * abs(warm[0]), warm
    assert abs(cold[0] - cold[1]) > 1e-5 * abs(cold[0]), cold
    assert abs(cold[0] - warm[0]) < 1e-2 * abs(warm[0])
    mu_b, info_b, Xb = lead(seed=1, warm_start=True, block_size=2)
    assert info_b % 2 == 0                                    # two warm-up matvecs + an even number of block matvecs
    assert abs(mu_b - warm[0]) < 1e-8 * abs(warm[0]), (mu_b, warm[0])
    # the Ritz vector of the block run is an eigenvector: residual of the matvec itself
    w = host.nek_dvector(gm)
    gA.matvec(Xb[0], w)
    if abs(mu_b.imag) < 1e-12:
        w.axpby(-mu_b.real, Xb[0], 1.0)
        assert w.norm() < 1e-6 * Xb[0].norm()


@pytest.mark.parametrize("dim,n,s", [(2, 6, 2), (3, 8, 3), (3, 5, 4), (3, 10, 2)])   # (lx1 = 10: lanes as consecutive blocks of k_conv3s)
@pytest.mark.parametrize("adjoint", [False, True])
def test_matvec_block_equals_single_matvecs(gpu_ctx, dim, n, s, adjoint):
    """nlg_linop_matvec_block: s vectors advanced together in lockstep PCGs give what s single matvecs give -- with
    restart-history replay on some lanes and not on others, tolerance-terminated solves that converge at different
    iteration counts per lane, direct and adjoint."""
    nel = (4, 3) if dim == 2 else (3, 2, 2)
    hm = box_mesh(nel, n, periodic=(True,) + (False,) * (dim - 1), deform=0.04)
    gm = host.Mesh(gpu_ctx, hm)
    X = [hm.x, hm.y] + ([hm.z] if dim == 3 else [])
    gb = host.nek_dvector(gm)
    gb.set_field(0, hm.mask[0] * (1.0 + 0.5 * np.sin(X[0]) * np.cos(X[1])))
    gb.set_field(1, hm.mask[1] * 0.3 * np.sin(2 * X[0]))
    A = host.exptA_linop(0.05, gb, re=40.0, dt=0.01, torder=3, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=4000)
    A.init()
    vin = []
    for v in range(s):
        x = host.nek_dvector(gm)
        x.rand(True, seed=40 + v)
        x.scal(10.0 ** (-2 * v))                     # very different magnitudes: different iteration counts per lane
        if v % 2 == 1:                               # odd lanes carry a restart history (the image of a first matvec)
            y = host.nek_dvector(gm)
            (A.rmatvec if adjoint else A.matvec)(x, y)
            x = y
        vin.append(x)
    single = [host.nek_dvector(gm) for _ in range(s)]
    for v in range(s):
        (A.rmatvec if adjoint else A.matvec)(vin[v], single[v])
    st0 = A.stats()
    blk = [host.nek_dvector(gm) for _ in range(s)]
    A.matvec_block(vin, blk, transpose=adjoint)
    st1 = A.stats()
    assert st1["matvecs"] - st0["matvecs"] == s
    for v in range(s):
        sc = max(np.abs(single[v].get_field(i)).max() for i in range(dim))
        for r in range(3):
            for i in range(dim):
                assert np.max(np.abs(blk[v].get_field(i, r) - single[v].get_field(i, r))) < 1e-11 * sc, (v, r, i)
            assert np.max(np.abs(blk[v].get_field(host.PR, r) - single[v].get_field(host.PR, r))) < 1e-9 * max(sc, np.abs(single[v].get_field(host.PR, r)).max())
        assert blk[v].nrst == 2
    with pytest.raises(host.NlgError):
        A.matvec_block(vin[:2], [blk[0], blk[0]])
    with pytest.raises(host.NlgError):
        A.matvec_block(vin[:1], vin[:1])
