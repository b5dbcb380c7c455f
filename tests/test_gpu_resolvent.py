"""Resolvent operator by time stepping (src/linops/resolvent.f90): forced integration and the full
R(omega) f against the oracle twins, plus the physics: the real part is the periodic state the forced flow settles to."""
import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle import krylov as K
from oracle.lns import ExptA, LNSConfig
from oracle.sem import SEM
from oracle.vectors import NekDVector

pytestmark = pytest.mark.gpu


def case(ctx, dim):
    if dim == 2:
        hm = box_mesh((3, 3), 6, lengths=(2.0, 1.0), periodic=(True, False), deform=0.03)
    else:
        hm = box_mesh((2, 2, 2), 6, lengths=(2.0, 1.0, 1.0), periodic=(True, False, True), deform=0.03)
    sem = SEM(hm)
    gm = host.Mesh(ctx, hm)
    U = [sem.mask[0] * (4 * sem.X[1] * (1 - sem.X[1]))] + [np.zeros(sem.shape1) for _ in range(dim - 1)]
    gb = host.nek_dvector(gm)
    gb.set_field(0, U[0])
    rng = np.random.default_rng(1)
    fz = host.nek_zvector(gm)
    fre, fim = [], []
    for i in range(dim):
        a = sem.mask[i] * sem.dsavg(np.sin(np.pi * sem.X[0] + i) * np.sin(np.pi * sem.X[1]) + 0.1 * rng.standard_normal(sem.shape1))
        b = sem.mask[i] * sem.dsavg(np.cos(np.pi * sem.X[0]) * np.sin(2 * np.pi * sem.X[1]))
        fre.append(a)
        fim.append(b)
        fz.re.set_field(i, a)
        fz.im.set_field(i, b)
    return hm, sem, gm, U, gb, fz, fre, fim


@pytest.mark.parametrize("dim,adjoint", [(2, False), (2, True), (3, False)])
def test_resolvent_matches_oracle(gpu_ctx, dim, adjoint):
    hm, sem, gm, U, gb, fz, fre, fim = case(gpu_ctx, dim)
    omega, re = 8.0, 20.0
    kw = dict(re=re, torder=3, vtol=1e-12, ptol=1e-12, maxit_v=400, maxit_p=4000)
    tau = 2 * np.pi / omega
    oA = ExptA(sem, U, LNSConfig(tau=tau, **kw))
    # forced integration alone
    gA = host.exptA_linop(tau, gb, **kw)
    gA.init()
    assert gA.info()["nsteps"] == oA.nsteps
    gout = host.nek_dvector(gm)
    host.integrate_forced(gA, None, fz.re, fz.im, omega, adjoint, gout)
    ob = oA.integrate_forced(None, fre, fim, omega, adjoint)
    sc = max(np.abs(a).max() for a in ob.v)
    for i in range(dim):
        assert np.max(np.abs(gout.get_field(i).reshape(sem.shape1) - ob.v[i])) < 1e-9 * sc
    with pytest.raises(host.NlgError):
        host.integrate_forced(gA, None, fz.re, fz.im, omega, adjoint, fz.re)
    # the full operator
    R = host.resolvent_linop(omega, gb, **kw)
    out = host.nek_zvector(gm)
    (R.rmatvec if adjoint else R.matvec)(fz, out)
    ox, oy, _, _ = K.resolvent_apply(oA, fre, fim, omega, adjoint)
    sx = max(np.abs(a).max() for a in ox.v)
    for i in range(dim):
        assert np.max(np.abs(out.re.get_field(i).reshape(sem.shape1) - ox.v[i])) < 2e-5 * sx     # both GMRES stop at rtol 1e-6
        assert np.max(np.abs(out.im.get_field(i).reshape(sem.shape1) - oy.v[i])) < 2e-5 * sx


def test_resolvent_is_the_periodic_response(gpu_ctx):
    """Independent of the GMRES: a stable flow forced periodically from rest converges to Re[x exp(i omega t)]; its
    state after k whole periods tends to Re x, a quarter period later to the vector the operator returns as Im."""
    hm, sem, gm, U, gb, fz, fre, fim = case(gpu_ctx, 2)
    omega, re = 8.0, 20.0
    kw = dict(re=re, torder=3, vtol=1e-11, ptol=1e-11, maxit_v=400, maxit_p=4000)
    R = host.resolvent_linop(omega, gb, **kw)
    out = host.nek_zvector(gm)
    R.matvec(fz, out)
    assert R.gmres_matvecs > 3
    A = host.exptA_linop(2 * np.pi / omega, gb, **kw)
    A.init()
    state, nxt = host.nek_dvector(gm), host.nek_dvector(gm)
    host.integrate_forced(A, None, fz.re, fz.im, omega, False, state)
    for _ in range(11):                        # the slowest mode decays like exp(-pi^2 t / Re): 12 periods ~ 1e-2
        state.clear_rst_fields()
        host.integrate_forced(A, state, fz.re, fz.im, omega, False, nxt)
        state, nxt = nxt, state
    d = state.copy()
    d.sub(out.re)
    assert d.norm() < 3e-2 * out.re.norm(), (d.norm(), out.re.norm())
    A.tau = 2 * np.pi / omega / 4
    q = host.nek_dvector(gm)
    host.integrate_forced(A, out.re, fz.re, fz.im, omega, False, q)
    d = q.copy()
    d.sub(out.im)
    assert d.norm() < 1e-8 * max(out.im.norm(), 1e-30)
    # zvector algebra used by the callers of the operator
    z = host.nek_zvector(gm)
    z.axpby(2.0 - 1.0j, out, 0.0)
    assert abs(z.norm() - abs(2.0 - 1.0j) * out.norm()) < 1e-12 * out.norm()
    assert abs(out.dot(z) - (2.0 - 1.0j) * out.norm() ** 2) < 1e-10 * out.norm() ** 2
