"""GPU parity: every kernel of the hot path against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM
from oracle.vectors import NekDVector

pytestmark = pytest.mark.gpu

CASES = [
    dict(nel=(4, 3), n=6, periodic=(True, False)),
    dict(nel=(3, 4), n=8, periodic=(False, False)),
    dict(nel=(3, 2, 2), n=6, periodic=(True, False, False)),
    dict(nel=(3, 3, 2), n=8, periodic=(False, False, True)),
    dict(nel=(2, 2, 2), n=5, periodic=(False, False, False)),
    dict(nel=(2, 2, 2), n=10, periodic=(False, False, True)),     # lx1 > 8: the LDS-cube Helmholtz kernel, generic convection,
    dict(nel=(2, 1, 2), n=12, periodic=(False, False, False)),    # NC = 1 pressure kernels, non-overlapping local solves
    dict(nel=(2, 2, 2), n=9, periodic=(True, False, False)),      # odd sizes: lx1 = 9 (in-place pressure kernels, dynamic-LDS convection)
    dict(nel=(2, 2, 2), n=7, periodic=(False, True, False)),      # ... and lx1 = 7 (the 256-thread kernels of lx1 <= 7)
]


def make(ctx, case, seed=0):
    hm = box_mesh(case["nel"], case["n"], periodic=case["periodic"], deform=0.05)
    sem = SEM(hm)
    gm = host.Mesh(ctx, hm)
    return hm, sem, gm


def rand_vec(sem, gm, rng, nscal=0):
    ov = NekDVector(sem, nscal)
    gv = host.nek_dvector(gm, nscal)
    for i in range(sem.dim):
        ov.v[i][...] = rng.standard_normal(sem.shape1)
        gv.set_field(i, ov.v[i])
    ov.pr[...] = rng.standard_normal(sem.shape2)
    gv.set_field(host.PR, ov.pr)
    for m in range(nscal):
        ov.theta[m][...] = rng.standard_normal(sem.shape1)
        gv.set_field(host.THETA + m, ov.theta[m])
    return ov, gv


def relerr(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c["nel"])) + "_n%d" % c["n"])
def test_mesh_geometry(gpu_ctx, case):
    hm, sem, gm = make(gpu_ctx, case)
    assert relerr(gm.get("bm1"), sem.bm1) < 1e-13
    assert relerr(gm.get("jac"), sem.jac) < 1e-13
    assert relerr(gm.get("vmult"), sem.vmult) < 1e-14
    assert relerr(gm.get("binvm1"), sem.binvm1) < 1e-13
    assert relerr(gm.get("bm2", 2), sem.bm2) < 1e-13
    names = {2: ["g11", "g12", "g22"], 3: ["g11", "g12", "g13", "g22", "g23", "g33"]}[sem.dim]
    pairs = {2: [(0, 0), (0, 1), (1, 1)], 3: [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]}[sem.dim]
    gmax = max(np.abs(sem.G[i][j]).max() for i, j in pairs)
    for nm, (i, j) in zip(names, pairs):
        assert np.max(np.abs(gm.get(nm) - sem.G[i][j].ravel())) < 1e-13 * gmax
    for j in range(sem.dim):
        for i in range(sem.dim):
            assert np.max(np.abs(gm.get("rst2w%d%d" % (j + 1, i + 1), 2) - sem.rst2w[j][i].ravel())) < 1e-13 * np.abs(sem.rst2w[j][j]).max()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c["nel"])) + "_n%d" % c["n"])
def test_operators(gpu_ctx, case):
    hm, sem, gm = make(gpu_ctx, case)
    rng = np.random.default_rng(1)
    ov, gv = rand_vec(sem, gm, rng)
    out = host.nek_dvector(gm)
    lib = gm.lib
    dim = sem.dim
    # local Helmholtz
    host.check(lib.nlg_op_helmholtz(gm.h, gv.h, out.h, 0.7, 3.0, 0))
    for i in range(dim):
        assert relerr(out.get_field(i), sem.axhelm_local(ov.v[i], 0.7, 3.0)) < 1e-13
    # assembled + masked
    host.check(lib.nlg_op_helmholtz(gm.h, gv.h, out.h, 0.7, 3.0, 1))
    for i in range(dim):
        assert relerr(out.get_field(i), sem.mask[i] * sem.gs(sem.axhelm_local(ov.v[i], 0.7, 3.0))) < 1e-13
    # dssum
    tmp = gv.copy()
    host.check(lib.nlg_op_dssum(gm.h, tmp.h))
    for i in range(dim):
        assert relerr(tmp.get_field(i), sem.gs(ov.v[i])) < 1e-14
    # divergence, gradient transpose, consistent Poisson operator
    host.check(lib.nlg_op_opdiv(gm.h, gv.h, out.h))
    assert relerr(out.get_field(host.PR), sem.opdiv(ov.v)) < 1e-13
    host.check(lib.nlg_op_opgradt(gm.h, gv.h, out.h))
    ref = sem.opgradt(ov.pr)
    for i in range(dim):
        assert relerr(out.get_field(i), ref[i]) < 1e-13
    host.check(lib.nlg_op_cdabdtp(gm.h, gv.h, out.h))
    assert relerr(out.get_field(host.PR), sem.cdabdtp(ov.pr)) < 1e-12
    # convection direct + adjoint
    ob, gb = rand_vec(sem, gm, rng)
    for adj in (0, 1):
        host.check(lib.nlg_op_conv(gm.h, gb.h, gv.h, out.h, adj))
        ref = sem.lns_conv_weak(ob.v, ov.v, adjoint=bool(adj))
        sc = max(np.abs(r).max() for r in ref)
        for i in range(dim):
            assert np.max(np.abs(out.get_field(i) - ref[i].ravel())) < 1e-12 * sc
    # CFL
    import ctypes as C
    c = C.c_double()
    host.check(lib.nlg_op_cfl(gm.h, gb.h, 0.01, C.byref(c)))
    assert abs(c.value - sem.compute_cfl(ob.v, 0.01)) < 1e-12 * c.value


@pytest.mark.parametrize("case", [CASES[0], CASES[3]], ids=["2d", "3d"])
def test_vector_space(gpu_ctx, case):
    hm, sem, gm = make(gpu_ctx, case)
    rng = np.random.default_rng(2)
    oa, ga = rand_vec(sem, gm, rng, nscal=1)
    ob, gb = rand_vec(sem, gm, rng, nscal=1)
    # dot / norm / size
    assert abs(ga.dot(gb) - oa.dot(ob)) < 1e-13 * abs(oa.norm() * ob.norm())
    assert abs(ga.norm() - oa.norm()) < 1e-13 * oa.norm()
    assert ga.get_size() == oa.get_size()
    # restart history + axpby quirk, bit-exact
    ga.save_rst(gb, 1); oa.save_rst(ob, 1)
    ga.save_rst(ga, 2); oa.save_rst(oa, 2)
    assert ga.nrst == 2 and ga.has_rst_fields()
    with pytest.raises(host.NlgError):
        ga.save_rst(gb, 3)
    # both treatments of the history in axpby (include/neklab_gpu.h: nlg_set_axpby_rst_consistent), bit-exact
    for mode in (0, 1):
        host.check(gm.lib.nlg_set_axpby_rst_consistent(mode))
        g2, o2 = ga.copy(), oa.copy()
        g2.axpby(0.25, gb, 2.0); o2.axpby(0.25, ob, 2.0, consistent_rst=bool(mode))
        for r in (1, 2):
            assert np.array_equal(g2.get_field(0, r), o2.v_rst[r - 1][0].ravel())
    host.check(gm.lib.nlg_set_axpby_rst_consistent(1))
    gb.save_rst(ga, 1); ob.save_rst(oa, 1)
    ga.axpby(0.3, gb, -1.7); oa.axpby(0.3, ob, -1.7)
    ga.scal(1.0 / 3.0); oa.scal(1.0 / 3.0)
    for i in range(sem.dim):
        assert np.array_equal(ga.get_field(i), oa.v[i].ravel())
        for r in (1, 2):
            assert np.array_equal(ga.get_field(i, r), oa.v_rst[r - 1][i].ravel())
    assert np.array_equal(ga.get_field(host.PR), oa.pr.ravel())
    assert np.array_equal(ga.get_field(host.PR, 2), oa.pr_rst[1].ravel())
    assert np.array_equal(ga.get_field(host.THETA), oa.theta[0].ravel())
    tmp = host.nek_dvector(gm, 1)
    ga.get_rst(tmp, 1)
    assert np.array_equal(tmp.get_field(0), oa.v_rst[0][0].ravel())
    ga.clear_rst_fields()
    assert not ga.has_rst_fields()
    ga.zero()
    assert ga.norm() == 0.0 and ga.nrst == 0


def test_rand_properties(gpu_ctx):
    hm, sem, gm = make(gpu_ctx, CASES[3])
    v = host.nek_dvector(gm)
    v.rand(ifnorm=True, seed=7)
    assert abs(v.norm() - 1.0) < 1e-13
    w = host.nek_dvector(gm)
    w.rand(ifnorm=True, seed=7)
    for i in range(sem.dim):
        a = v.get_field(i).reshape(sem.shape1)
        assert np.array_equal(a.ravel(), w.get_field(i))             # deterministic per seed
        assert np.allclose(sem.gs(a) * sem.vmult, a, rtol=0, atol=1e-14 * np.abs(a).max())   # C0
        assert np.all(a[sem.mask[i] == 0] == 0)                      # Dirichlet
        assert np.std(a) > 0
    w.rand(ifnorm=True, seed=8)


def test_block_kernels(gpu_ctx):
    hm, sem, gm = make(gpu_ctx, CASES[3])
    rng = np.random.default_rng(3)
    k = 11
    B = host.KrylovBasis(gm, k + 1)
    ovs = []
    for j in range(k):
        ov, gv = rand_vec(sem, gm, rng)
        B[j].assign(gv)
        ovs.append(ov)
    ow, gw = rand_vec(sem, gm, rng)
    h = B.block_dot(k, gw)
    href = np.array([o.dot(ow) for o in ovs])
    assert np.max(np.abs(h - href)) < 1e-13 * np.max(np.abs(href))
    B.block_axpy(k, h, gw)
    for hj, o in zip(href, ovs):
        ow.axpby(-hj, o, 1.0)
    for i in range(sem.dim):
        assert relerr(gw.get_field(i), ow.v[i]) < 1e-13
    assert relerr(gw.get_field(host.PR), ow.pr) < 1e-13
    B.close()
