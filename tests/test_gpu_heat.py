"""Boussinesq (temperature) coupling of the propagator: matvec against the oracle, and the classical known answer the
reference's Rayleigh-Benard case quotes -- onset at Ra_c = 1707.762 for the wavenumber 3.117, rigid-rigid
(Chandrasekhar 1961, Table III; examples/rayBen/baseflow/rayBen.par:9)."""
import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.lns import ExptA, LNSConfig
from oracle.sem import SEM
from oracle.vectors import NekDVector

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,n", [(2, 6), (3, 6), (3, 8), (3, 9), (3, 10)])   # lx1 = 8 ... 10: the fused scalar-transport kernel k_conv3s_scalar (lx1 = 6, 2-D: generic path)
def test_boussinesq_matvec_matches_oracle(gpu_ctx, dim, n):
    if dim == 2:
        hm = box_mesh((3, 2), n, lengths=(2.0, 1.0), periodic=(True, False), deform=0.03)
    else:
        hm = box_mesh((2, 2, 2), n, lengths=(2.0, 1.0, 1.0), periodic=(True, False, True), deform=0.03)
    sem = SEM(hm)
    gm = host.Mesh(gpu_ctx, hm)
    U = [sem.mask[0] * (4 * sem.X[1] * (1 - sem.X[1]))] + [np.zeros(sem.shape1) for _ in range(dim - 1)]
    Theta = 1.0 - sem.X[1] + 0.1 * np.sin(np.pi * sem.X[0]) * np.sin(np.pi * sem.X[1])
    gb = host.nek_dvector(gm, 1)
    gb.set_field(0, U[0])
    gb.set_field(host.THETA, Theta)
    buoy = (0.0, 50.0, 0.0)
    kw = dict(re=5.0, torder=3, vtol=1e-13, ptol=1e-13, maxit_v=600, maxit_p=4000, dt=0.01)
    heat = dict(ifheat=True, conductivity=0.3, rhocp=1.5, buoy=buoy)
    oA = ExptA(sem, U, LNSConfig(tau=0.05, **kw, **heat), Theta)
    gA = host.exptA_linop(0.05, gb, **kw, **{**heat, "ifheat": 1})
    gA.init()
    rng = np.random.default_rng(0)
    ov, gv = NekDVector(sem, 1), host.nek_dvector(gm, 1)
    for i in range(dim):
        ov.v[i][...] = sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1))
        gv.set_field(i, ov.v[i])
    ov.theta[0][...] = sem.tmask * sem.dsavg(rng.standard_normal(sem.shape1))
    gv.set_field(host.THETA, ov.theta[0])
    gout = host.nek_dvector(gm, 1)
    gA.matvec(gv, gout)
    oout = oA.matvec(ov)
    sc = max(np.abs(a).max() for a in oout.v)
    for i in range(dim):
        assert np.max(np.abs(gout.get_field(i).reshape(sem.shape1) - oout.v[i])) < 1e-9 * sc
    st = np.abs(oout.theta[0]).max()
    assert np.max(np.abs(gout.get_field(host.THETA).reshape(sem.shape1) - oout.theta[0])) < 1e-9 * st
    # second application: the restart history carries the temperature as well
    g2 = host.nek_dvector(gm, 1)
    gA.matvec(gout, g2)
    o2 = oA.matvec(oout)
    assert np.max(np.abs(g2.get_field(host.THETA).reshape(sem.shape1) - o2.theta[0])) < 1e-8 * np.abs(o2.theta[0]).max()
    # a vector without the scalar is refused
    with pytest.raises(host.NlgError):
        gA.matvec(host.nek_dvector(gm), host.nek_dvector(gm))
    # adjoint of the coupled operator (exponential_propagator_temp.f90:62-107): against the oracle twin ...
    ga = host.nek_dvector(gm, 1)
    gA.rmatvec(gv, ga)
    oa = oA.matvec(ov, adjoint=True)
    for i in range(dim):
        assert np.max(np.abs(ga.get_field(i).reshape(sem.shape1) - oa.v[i])) < 1e-9 * max(np.abs(a).max() for a in oa.v)
    assert np.max(np.abs(ga.get_field(host.THETA).reshape(sem.shape1) - oa.theta[0])) < 1e-9 * np.abs(oa.theta[0]).max()
    ga2 = host.nek_dvector(gm, 1)
    gA.rmatvec(ga, ga2)
    oa2 = oA.matvec(oa, adjoint=True)
    assert np.max(np.abs(ga2.get_field(host.THETA).reshape(sem.shape1) - oa2.theta[0])) < 1e-8 * np.abs(oa2.theta[0]).max()
    # ... and as an adjoint: <A x, y> = <x, A+ y> in the velocity + temperature inner product, up to the splitting error
    # of the continuous adjoint (the same level as for the velocity-only operator, tests/test_gpu_svds.py)
    gy = host.nek_dvector(gm, 1)
    gy.rand(True, seed=5)
    gAty = host.nek_dvector(gm, 1)
    gA.rmatvec(gy, gAty)
    lhs, rhs = gout.dot(gy), gv.dot(gAty)
    # (rough random fields, five time steps from an impulsive start: the inner products themselves are small against
    #  |A x| |y|, which is the scale the splitting error lives on; measured 0.03 of it)
    assert abs(lhs - rhs) < 6e-2 * gout.norm() * gy.norm(), (lhs, rhs, gout.norm() * gy.norm())


def test_rayleigh_benard_onset(gpu_ctx, tmp_path):
    """Conduction state Theta = 1 - y, U = 0, Pr = 1, box of one critical wavelength 2 pi / 3.117: the growth rate of the
    leading mode changes sign at Ra_c = 1707.762.  Measured with dt = 0.005 (first-order splitting of the coupling):
    sigma(1600) = -0.825, sigma(1707.762) = +0.071, sigma(1800) = +0.822, i.e. a zero crossing at Ra = 1699 (0.5 %)."""
    hm = box_mesh((3, 3), 8, lengths=(2 * np.pi / 3.117, 1.0), periodic=(True, False), deform=0.0)
    gm = host.Mesh(gpu_ctx, hm)
    bf = host.nek_dvector(gm, 1)
    bf.set_field(host.THETA, 1.0 - hm.y)
    sig = {}
    for Ra in (1600.0, 1800.0):
        A = host.exptA_linop(0.1, bf, re=1.0, torder=3, dt=0.005, vtol=1e-11, ptol=1e-11, maxit_v=2000, maxit_p=4000,
                             ifheat=1, conductivity=1.0, rhocp=1.0, buoy=(0.0, Ra, 0.0))
        A.init()
        eigvecs = [host.nek_dvector(gm, 1, 3)]
        mu, res, info = host.eigs(A, eigvecs, kdim=16, tol=1e-8, logfile=str(tmp_path / "eigs.txt"), seed=1)
        assert res[0] < 1e-7 and abs(mu[0].imag) < 1e-9            # onset is stationary (exchange of stabilities)
        sig[Ra] = np.log(abs(mu[0])) / 0.1
        # the marginal mode is a pair of counter-rotating rolls: vertical velocity and temperature in phase
        v, th = eigvecs[0].get_field(1), eigvecs[0].get_field(host.THETA)
        assert abs(np.corrcoef(v, th)[0, 1]) > 0.95
    assert sig[1600.0] < 0 < sig[1800.0]
    ra_c = 1600.0 - sig[1600.0] * 200.0 / (sig[1800.0] - sig[1600.0])
    assert abs(ra_c - 1707.762) < 0.01 * 1707.762, (sig, ra_c)


def test_nonlinear_map_with_temperature_matches_oracle(gpu_ctx):
    """nonlinear_map of the temperature-coupled system (nek_system_temp of examples/thermosyphon/baseflow/tsyphon.usr:12,38):
    full convection of velocity and temperature, buoyancy, F(X) = Phi_tau(X) - X including the scalar."""
    hm = box_mesh((3, 2), 6, lengths=(2.0, 1.0), periodic=(True, False), deform=0.03)
    sem = SEM(hm)
    gm = host.Mesh(gpu_ctx, hm)
    rng = np.random.default_rng(1)
    oX, gX = NekDVector(sem, 1), host.nek_dvector(gm, 1)
    for i in range(2):
        oX.v[i][...] = sem.mask[i] * sem.dsavg(0.5 * np.sin(np.pi * sem.X[0] + i) * np.sin(np.pi * sem.X[1]))
        gX.set_field(i, oX.v[i])
    oX.theta[0][...] = 1.0 - sem.X[1] + 0.2 * sem.tmask * sem.dsavg(np.sin(np.pi * sem.X[0]) * np.sin(np.pi * sem.X[1]))
    gX.set_field(host.THETA, oX.theta[0])
    kw = dict(re=5.0, torder=3, vtol=1e-13, ptol=1e-13, maxit_v=600, maxit_p=4000, cfl_limit=0.4)
    heat = dict(conductivity=0.3, rhocp=1.0, buoy=(0.0, 20.0, 0.0))
    oA = ExptA(sem, oX.v, LNSConfig(tau=0.1, ifheat=True, **kw, **heat), oX.theta[0])
    oF = oA.nonlinear_map(oX)
    gA = host.exptA_linop(0.1, gX, ifheat=1, **kw, **heat)
    gA.init()
    gF = host.nek_dvector(gm, 1)
    host.check(gpu_ctx.lib.nlg_linop_nonlinear_map(gA.h, gX.h, gF.h))
    assert gA.info()["nsteps"] == oA.nsteps
    sc = max(np.abs(a).max() for a in oF.v)
    for i in range(2):
        assert np.max(np.abs(gF.get_field(i).reshape(sem.shape1) - oF.v[i])) < 1e-9 * sc
    assert np.max(np.abs(gF.get_field(host.THETA).reshape(sem.shape1) - oF.theta[0])) < 1e-9 * np.abs(oF.theta[0]).max()
    # Dirichlet values of the temperature stay: F vanishes on the walls
    assert np.max(np.abs(gF.get_field(host.THETA).reshape(sem.shape1) * (1 - sem.tmask))) == 0.0


@pytest.mark.parametrize("dim,n,s", [(2, 6, 2), (3, 8, 3), (3, 10, 2)])
@pytest.mark.parametrize("adjoint", [False, True])
def test_boussinesq_matvec_block_equals_single_matvecs(gpu_ctx, dim, n, s, adjoint):
    """Lane-batched propagator with the temperature coupling (the reference's coupled operator
    exponential_propagator_temp.f90:15-60 / :62-107): s vectors advanced together -- scalar right-hand side, operator, gather-scatter
    and the scalar's PCG as ONE launch for all lanes -- give what s single matvecs give, velocity, pressure, temperature and their
    restart histories, with very different magnitudes per lane (different iteration counts) and a restart history on the odd lanes."""
    if dim == 2:
        hm = box_mesh((3, 2), n, lengths=(2.0, 1.0), periodic=(True, False), deform=0.03)
    else:
        hm = box_mesh((2, 2, 2), n, lengths=(2.0, 1.0, 1.0), periodic=(True, False, True), deform=0.03)
    sem = SEM(hm)
    gm = host.Mesh(gpu_ctx, hm)
    gb = host.nek_dvector(gm, 1)
    gb.set_field(0, sem.mask[0] * (4 * sem.X[1] * (1 - sem.X[1])))
    gb.set_field(host.THETA, 1.0 - sem.X[1] + 0.1 * np.sin(np.pi * sem.X[0]) * np.sin(np.pi * sem.X[1]))
    A = host.exptA_linop(0.05, gb, re=5.0, torder=3, vtol=1e-13, ptol=1e-13, maxit_v=600, maxit_p=4000, dt=0.01,
                         ifheat=1, conductivity=0.3, rhocp=1.5, buoy=(0.0, 50.0, 0.0))
    A.init()
    mv = A.rmatvec if adjoint else A.matvec
    vin = []
    for v in range(s):
        x = host.nek_dvector(gm, 1)
        x.rand(True, seed=70 + v)
        x.scal(10.0 ** (-2 * v))
        if v % 2 == 1:
            y = host.nek_dvector(gm, 1)
            mv(x, y)
            x = y
        vin.append(x)
    single = [host.nek_dvector(gm, 1) for _ in range(s)]
    for v in range(s):
        mv(vin[v], single[v])
    blk = [host.nek_dvector(gm, 1) for _ in range(s)]
    A.matvec_block(vin, blk, transpose=adjoint)
    for v in range(s):
        sc = max(np.abs(single[v].get_field(i)).max() for i in range(dim))
        st = np.abs(single[v].get_field(host.THETA)).max()
        for r in range(3):
            for i in range(dim):
                assert np.max(np.abs(blk[v].get_field(i, r) - single[v].get_field(i, r))) < 1e-11 * sc, (v, r, i)
            assert np.max(np.abs(blk[v].get_field(host.THETA, r) - single[v].get_field(host.THETA, r))) < 1e-11 * max(st, sc), (v, r)
            assert np.max(np.abs(blk[v].get_field(host.PR, r) - single[v].get_field(host.PR, r))) < 1e-9 * max(sc, np.abs(single[v].get_field(host.PR, r)).max())
        assert blk[v].nrst == 2
    with pytest.raises(host.NlgError):       # a vector without the scalar is refused by the block path too
        A.matvec_block([host.nek_dvector(gm)], [host.nek_dvector(gm)])
