"""BASELINE.json configs 4 and 5 at (per-GPU) full size through size-independent properties, as tests/test_gpu_fullsize.py does
for the headline config: the oracle cannot run these sizes in seconds, so the lx1 = 10 / 12 kernel instantiations are checked
inside the time stepper against identities of the path (linearity of the propagator, orthonormality and the Arnoldi relation of
the factorisation, lockstep block propagator = single propagator).

config 4: 3-D thermosyphon-like Boussinesq case, E = 20 000, N = 9 (lx1 = 10), velocity + temperature state, m = 128
          (reference: examples/thermosyphon, `exptA_linop_temp` of tsyphon.usr:13,52) -- here m is cut to what the check needs.
config 5: E = 200 000, N = 11 (lx1 = 12), m = 256 block-Arnoldi on 8 GPUs = 25 000 elements per GPU; here one GPU's share.
"""
import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh

pytestmark = pytest.mark.gpu


def swirl(hm):
    ph = [2 * np.pi * c / l for c, l in zip((hm.x, hm.y, hm.z), hm.lengths)]
    return [np.sin(ph[1]) * np.cos(ph[2]), 0.5 * np.sin(ph[2]) * np.cos(ph[0]), 0.5 * np.sin(ph[0]) * np.cos(ph[1])]


def test_config4_boussinesq_lx1_10_full_size(gpu_ctx):
    hm = box_mesh((25, 40, 20), 10, deform=0.04)
    assert hm.x.shape[0] == 20000
    gm = host.Mesh(gpu_ctx, hm)
    bf = host.nek_dvector(gm, 1)
    U = swirl(hm)
    for i in range(3):
        bf.set_field(i, U[i] * hm.mask[i])
    bf.set_field(host.THETA, 1.0 - hm.y / hm.lengths[1])
    A = host.exptA_linop(0.02, bf, re=100.0, torder=3, vtol=1e-11, ptol=1e-10, maxit_v=400, maxit_p=4000, ifheat=1,
                         conductivity=0.01, rhocp=1.0, buoy=(0.0, 1.0, 0.0))
    A.init()
    x, y = host.nek_dvector(gm, 1), host.nek_dvector(gm, 1)
    x.rand(True, seed=1)
    y.rand(True, seed=2)
    z = x.copy()
    z.axpby(-0.7, y, 1.3)
    Ax, Ay, Az = (host.nek_dvector(gm, 1) for _ in range(3))
    A.matvec(x, Ax)
    A.matvec(y, Ay)
    A.matvec(z, Az)
    # the temperature is part of the state and of the inner product (real_vectors.f90:217-224 with ifto)
    assert np.max(np.abs(Az.get_field(host.THETA))) > 0
    Ax.axpby(-0.7, Ay, 1.3)
    Ax.axpby(-1.0, Az, 1.0)
    assert Ax.norm() < 1e-7 * Az.norm()       # linear up to the solver tolerances
    st = A.stats()
    assert st["p_iters"] / st["steps"] < 60, st
    # the adjoint propagator runs the same kernels with the transposed coupling: <A x, y> = <x, A^T y> up to the
    # continuous-adjoint discretisation error (tests/test_gpu_heat.py states it at small size; here only finiteness + scale)
    ATy = host.nek_dvector(gm, 1)
    A.rmatvec(y, ATy)
    assert np.isfinite(ATy.norm()) and 0.1 < ATy.norm() / Ay.norm() < 10.0
    # three Arnoldi steps
    m = 3
    B = host.KrylovBasis(gm, m + 1, 1)
    B[0].rand(True, seed=5)
    H = np.zeros((m + 2, m + 1), order="F")
    for k in range(m):
        host.arnoldi_step(A, B, k, H)
    G = np.array([B.block_dot(m + 1, B[j]) for j in range(m + 1)])
    assert np.max(np.abs(G - np.eye(m + 1))) < 1e-12
    assert all(H[k + 1, k] > 0 for k in range(m))


def test_config5_block_arnoldi_lx1_12_per_gpu_share(gpu_ctx):
    hm = box_mesh((25, 25, 40), 12, deform=0.04)
    assert hm.x.shape[0] == 25000
    gm = host.Mesh(gpu_ctx, hm)
    bf = host.nek_dvector(gm)
    U = swirl(hm)
    for i in range(3):
        bf.set_field(i, U[i] * hm.mask[i])
    A = host.exptA_linop(0.01, bf, re=100.0, torder=3, vtol=1e-11, ptol=1e-10, maxit_v=400, maxit_p=4000)
    A.init()
    s, nblk = 4, 2
    m = s * nblk
    B = host.KrylovBasis(gm, m + s)
    for v in range(s):
        B[v].rand(True, seed=20 + v)
    B.block_cgs2(0, s)
    # lockstep block propagator against the single propagator on the first block (the same kernels with lane loops)
    single = host.nek_dvector(gm)
    A.matvec(B[1], single)
    H = np.zeros((m + s, m), order="F")
    for j in range(nblk):
        host.block_arnoldi_step(A, B, j * s, s, H)
    G = np.array([B.block_dot(m + s, B[j]) for j in range(m + s)])
    assert np.max(np.abs(G - np.eye(m + s))) < 1e-11, np.max(np.abs(G - np.eye(m + s)))
    # band Hessenberg: nothing below the s-th subdiagonal, positive diagonal of every R block
    for c in range(m):
        assert np.all(H[c + s + 1:, c] == 0.0)
        assert H[c + s, c] > 0
    # column 1 of the block Arnoldi relation: A v_1 = V_{m+s} H[:, 1]
    r = single.copy()
    for i in range(m + s):
        if H[i, 1] != 0.0:
            r.axpby(-H[i, 1], B[i], 1.0)
    assert r.norm() < 1e-7 * single.norm(), r.norm() / single.norm()
    st = A.stats()
    assert st["p_iters"] / st["steps"] < 60, st
