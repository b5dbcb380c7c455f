"""Parity at BASELINE.json's full size (3-D, E = 10 000, N = 7) through size-independent properties: the oracle cannot
run at this size in seconds, so the kernels are checked against identities of the discretisation instead
(transposes, symmetry, idempotence, linearity, orthonormality), on the bench's own mesh and base flow."""
import ctypes as C

import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(gpu_ctx):
    hm = box_mesh((25, 20, 20), 8, deform=0.05)
    gm = host.Mesh(gpu_ctx, hm)
    return hm, gm


def vec(gm, rng, fields=(0, 1, 2, host.PR)):
    v = host.nek_dvector(gm)
    for f in fields:
        v.set_field(f, rng.standard_normal(gm.lpn if f == host.PR else gm.lvn))
    return v


def vdot(a, b, fields):
    return sum(float(a.get_field(f) @ b.get_field(f)) for f in fields)


def test_divergence_and_gradient_are_transposes(gpu_ctx, big):
    hm, gm = big
    lib, rng = gpu_ctx.lib, np.random.default_rng(0)
    u, p = vec(gm, rng), vec(gm, rng)
    Du, Dtp = host.nek_dvector(gm), host.nek_dvector(gm)
    host.check(lib.nlg_op_opdiv(gm.h, u.h, Du.h))
    host.check(lib.nlg_op_opgradt(gm.h, p.h, Dtp.h))
    lhs = vdot(p, Du, (host.PR,))
    rhs = vdot(Dtp, u, (0, 1, 2))
    assert abs(lhs - rhs) < 1e-12 * abs(lhs)


def test_consistent_poisson_operator_symmetric_semidefinite(gpu_ctx, big):
    hm, gm = big
    lib, rng = gpu_ctx.lib, np.random.default_rng(1)
    a, b = vec(gm, rng, (host.PR,)), vec(gm, rng, (host.PR,))
    Ea, Eb = host.nek_dvector(gm), host.nek_dvector(gm)
    host.check(lib.nlg_op_cdabdtp(gm.h, a.h, Ea.h))
    host.check(lib.nlg_op_cdabdtp(gm.h, b.h, Eb.h))
    ab, ba = vdot(b, Ea, (host.PR,)), vdot(a, Eb, (host.PR,))
    assert abs(ab - ba) < 1e-11 * abs(ab) + 1e-14 * np.sqrt(vdot(Ea, Ea, (host.PR,)) * vdot(b, b, (host.PR,)))
    assert vdot(a, Ea, (host.PR,)) > 0
    # the constant is (nearly: GL quadrature on deformed elements) in the null space: |E 1| << |E a| / |a|
    one = host.nek_dvector(gm)
    one.set_field(host.PR, np.ones(gm.lpn))
    E1 = host.nek_dvector(gm)
    host.check(lib.nlg_op_cdabdtp(gm.h, one.h, E1.h))
    r1 = np.sqrt(vdot(E1, E1, (host.PR,)) / gm.lpn)
    ra = np.sqrt(vdot(Ea, Ea, (host.PR,)) / vdot(a, a, (host.PR,)))
    assert r1 < 1e-6 * ra


@pytest.mark.parametrize("overlap", [0, 1])
def test_pressure_preconditioner_symmetric_positive(gpu_ctx, big, overlap):
    hm, gm = big
    lib, rng = gpu_ctx.lib, np.random.default_rng(2)
    a, b = vec(gm, rng, (host.PR,)), vec(gm, rng, (host.PR,))
    Ma, Mb = host.nek_dvector(gm), host.nek_dvector(gm)
    host.check(lib.nlg_op_pprec(gm.h, a.h, Ma.h, overlap, 1))
    host.check(lib.nlg_op_pprec(gm.h, b.h, Mb.h, overlap, 1))
    ab, ba = vdot(b, Ma, (host.PR,)), vdot(a, Mb, (host.PR,))
    assert abs(ab - ba) < 1e-11 * max(abs(ab), np.sqrt(vdot(Ma, Ma, (host.PR,)) * vdot(b, b, (host.PR,))) * 1e-3)
    assert vdot(a, Ma, (host.PR,)) > 0 and vdot(b, Mb, (host.PR,)) > 0


def test_helmholtz_symmetric_and_dssum_average_idempotent(gpu_ctx, big):
    hm, gm = big
    lib, rng = gpu_ctx.lib, np.random.default_rng(3)
    a, b = vec(gm, rng), vec(gm, rng)
    # make both continuous: u <- vmult * dssum(u)
    vm = gm.get("vmult")
    for v in (a, b):
        host.check(lib.nlg_op_dssum(gm.h, v.h))
        for f in range(3):
            v.set_field(f, v.get_field(f) * vm)
    a2 = a.copy()
    host.check(lib.nlg_op_dssum(gm.h, a2.h))
    for f in range(3):
        assert np.max(np.abs(a2.get_field(f) * vm - a.get_field(f))) < 1e-13 * np.abs(a.get_field(f)).max()   # idempotent
    Ha, Hb = host.nek_dvector(gm), host.nek_dvector(gm)
    host.check(lib.nlg_op_helmholtz(gm.h, a.h, Ha.h, 0.01, 92.0, 0))       # element-local: (b, H_L a) is the global form
    host.check(lib.nlg_op_helmholtz(gm.h, b.h, Hb.h, 0.01, 92.0, 0))
    ab, ba = vdot(b, Ha, (0, 1, 2)), vdot(a, Hb, (0, 1, 2))
    assert abs(ab - ba) < 1e-12 * abs(ab)
    assert vdot(a, Ha, (0, 1, 2)) > 0


def test_matvec_linear_and_arnoldi_orthonormal(gpu_ctx, big):
    hm, gm = big
    lib, rng = gpu_ctx.lib, np.random.default_rng(4)
    bf = host.nek_dvector(gm)
    L = hm.lengths
    ph = [2 * np.pi * c / l for c, l in zip((hm.x, hm.y, hm.z), L)]
    U = [np.sin(ph[1]) * np.cos(ph[2]), 0.5 * np.sin(ph[2]) * np.cos(ph[0]), 0.5 * np.sin(ph[0]) * np.cos(ph[1])]
    for i in range(3):
        bf.set_field(i, U[i] * hm.mask[i])
    A = host.exptA_linop(0.03, bf, re=100.0, torder=3, vtol=1e-11, ptol=1e-10, maxit_v=400, maxit_p=4000)
    A.init()
    x, y = host.nek_dvector(gm), host.nek_dvector(gm)
    x.rand(True, seed=1)
    y.rand(True, seed=2)
    z = x.copy()
    z.axpby(-0.7, y, 1.3)                     # z = 1.3 x - 0.7 y
    Ax, Ay, Az = host.nek_dvector(gm), host.nek_dvector(gm), host.nek_dvector(gm)
    A.matvec(x, Ax)
    A.matvec(y, Ay)
    A.matvec(z, Az)
    Ax.axpby(-0.7, Ay, 1.3)
    Ax.axpby(-1.0, Az, 1.0)                   # 1.3 A x - 0.7 A y - A z
    assert Ax.norm() < 1e-7 * Az.norm()       # linear up to the solver tolerances
    assert A.stats()["p_iters"] / A.stats()["steps"] < 40
    # three Arnoldi steps: the basis stays orthonormal to rounding, H is upper Hessenberg with positive subdiagonal
    m = 3
    B = host.KrylovBasis(gm, m + 1)
    v0 = B[0]
    v0.rand(True, seed=5)
    H = np.zeros((m + 2, m + 1), order="F")
    for k in range(m):
        host.arnoldi_step(A, B, k, H)
    G = np.array([B.block_dot(m + 1, B[j]) for j in range(m + 1)])
    assert np.max(np.abs(G - np.eye(m + 1))) < 1e-12
    assert all(H[k + 1, k] > 0 for k in range(m)) and abs(H[2, 0]) == 0.0
