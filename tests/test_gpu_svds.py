"""svds / transient growth: the reference's back_fstep case runs
transient_growth_analysis_fixed_point -> svds(exptA, U, S, V, ...) (src/neklab_analysis.f90:107-156)."""
import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.krylov import svds as o_svds
from test_gpu_linop import load_pair, setup_case

pytestmark = pytest.mark.gpu


def flat(v, dim):
    return np.concatenate([v.get_field(i) for i in range(dim)]) if hasattr(v, "get_field") else np.concatenate([a.ravel() for a in v.v])


@pytest.mark.parametrize("dim", [2, 3])
def test_svds_against_oracle(gpu_ctx, dim, tmp_path):
    hm, sem, gm, oA, gA, rng = setup_case(gpu_ctx, dim, tau=0.3, re=15.0)
    ov, gv = load_pair(sem, gm, rng)
    nsv, kdim = 2, 10
    U = [host.nek_dvector(gm) for _ in range(nsv)]
    V = [host.nek_dvector(gm) for _ in range(nsv)]
    S, res, info = host.svds(gA, U, V, kdim=kdim, tol=1e-7, u0=gv, logfile=str(tmp_path / "svds_output.txt"))
    oS, oU, oV, ores, onmv = o_svds(oA.matvec, oA.rmatvec, ov, nsv, kdim, tol=1e-7)
    assert info == onmv
    assert np.max(np.abs(S - oS) / oS) < 1e-10                    # singular values, north_star tolerance
    assert np.max(np.abs(res - ores)) < 1e-9
    # optimal perturbation / response of the leading triplet: defined up to a common sign
    for gvec, ovec in ((V[0], oV[0]), (U[0], oU[0])):
        a, b = flat(gvec, dim), flat(ovec, dim)
        s = np.sign(a @ b)
        assert np.max(np.abs(a - s * b)) < 1e-6 * np.max(np.abs(b))
    # defining property ||A v_1|| = sigma_1, A v_1 parallel to u_1.  It holds only approximately: rmatvec is the
    # CONTINUOUS adjoint propagator (as in the reference, exponential_propagator.f90:62-107), not the transpose of
    # the discrete matvec, so the bidiagonal recurrence that svds records is itself approximate (measured 0.3 %).
    w = host.nek_dvector(gm)
    gA.matvec(V[0], w)
    assert abs(w.norm() - S[0]) < 2e-2 * S[0]
    assert abs(abs(w.dot(U[0])) - S[0]) < 2e-2 * S[0]
    assert abs(V[0].norm() - 1.0) < 1e-10 and abs(U[0].norm() - 1.0) < 1e-10
    rows = [ln.split() for ln in open(tmp_path / "svds_output.txt") if not ln.startswith("#")]
    assert abs(float(rows[0][1]) - S[0]) < 1e-12 and rows[0][3] in ("T", "F")


def test_transient_growth_driver_outputs(gpu_ctx, tmp_path):
    hm, sem, gm, oA, gA, rng = setup_case(gpu_ctx, 2, tau=0.2, re=15.0)
    S, res, U, V, info = host.transient_growth_analysis_fixed_point(gA, 2, 8, tol=1e-6, outdir=str(tmp_path), seed=2)
    assert S[0] >= S[1] > 0 and info % 2 == 0
    vals = [float(x) for x in open(tmp_path / "singular_spectrum.dat").read().split()]
    assert np.allclose(vals, S)


def test_outpost_dnek_round_trip(gpu_ctx, tmp_path):
    """outpost_dnek (neklab_utils.f90:305-333): field files readable by the reader; pressure on mesh 1 equals the
    oracle's interpolation + direct-stiffness average; coordinates only in the first file."""
    from neklab_amd import nekio
    from oracle.sem import SEM, interp_matrix
    hm = box_mesh((3, 2, 2), 6, deform=0.03)
    sem = SEM(hm)
    gm = host.Mesh(gpu_ctx, hm)
    rng = np.random.default_rng(2)
    vecs = []
    for _ in range(2):
        v = host.nek_dvector(gm)
        for c in range(3):
            v.set_field(c, rng.standard_normal(gm.lvn))
        v.set_field(host.PR, rng.standard_normal(gm.lpn))
        vecs.append(v)
    paths = host.outpost_dnek(vecs, "dir", session="case", outdir=str(tmp_path))
    assert [p.split("/")[-1] for p in paths] == ["dircase0.f00001", "dircase0.f00002"]
    d0, d1 = nekio.read_fld(paths[0]), nekio.read_fld(paths[1])
    assert "x" in d0 and "x" not in d1
    assert np.array_equal(d0["x"].ravel(), hm.x.ravel())
    assert np.array_equal(d1["uy"].ravel(), vecs[1].get_field(1))
    I21 = interp_matrix(sem.z2, sem.z1)
    p = vecs[0].get_field(host.PR).reshape(sem.shape2)
    for ax in (1, 2, 3):
        p = np.moveaxis(np.tensordot(I21, p, axes=([1], [ax])), 0, ax)
    ref = sem.dsavg(p)
    assert np.max(np.abs(d0["p"].ravel() - ref.ravel())) < 1e-12 * np.abs(ref).max()
