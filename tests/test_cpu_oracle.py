"""CPU suite: the oracle against its golden vectors and against independent mathematical facts."""
import os

import numpy as np
import pytest

from neklab_amd.mesh import box_mesh, gll_points
from oracle.sem import SEM, gl, gll
from oracle.vectors import NekDVector

HERE = os.path.dirname(os.path.abspath(__file__))
import sys
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402


def golden(case):
    return np.load(os.path.join(HERE, "golden", "golden_%s.npz" % case))


def close(a, b, tol=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b)) <= tol * max(np.max(np.abs(b)), 1e-300)


def test_quadrature_rules():
    for n in range(3, 14):
        x, w = gll(n)
        assert abs(w.sum() - 2.0) < 1e-14 and np.allclose(x, gll_points(n), atol=1e-15)
        # GLL integrates degree 2n-3 exactly
        for d in range(0, 2 * n - 2, 2):
            assert abs(np.sum(w * x ** d) - 2.0 / (d + 1)) < 1e-13
        xg, wg = gl(n)
        for d in range(0, 2 * n, 2):
            assert abs(np.sum(wg * xg ** d) - 2.0 / (d + 1)) < 1e-13


@pytest.mark.parametrize("case", ["2d", "3d"])
def test_operators_match_golden(case):
    g = golden(case)
    hm, sem = mg.build(case)
    dim = sem.dim
    u = [g["in_u"][i] for i in range(dim)]
    w = [g["in_w"][i] for i in range(dim)]
    assert close(sem.bm1, g["bm1"]) and close(sem.binvm1, g["binvm1"]) and close(sem.vmult, g["vmult"])
    assert close(np.stack([sem.axhelm_local(u[i], 0.7, 3.0) for i in range(dim)]), g["axhelm"])
    assert close(np.stack([sem.gs(u[i]) for i in range(dim)]), g["gs"])
    assert close(sem.opdiv(u), g["opdiv"])
    assert close(np.stack(sem.opgradt(g["in_p"])), g["opgradt"])
    assert close(sem.cdabdtp(g["in_p"]), g["cdabdtp"])
    assert close(sem.e_diag(), g["ediag"])
    assert close(np.stack(sem.lns_conv_weak(w, u)), g["conv_dir"])
    assert close(np.stack(sem.lns_conv_weak(w, u, adjoint=True)), g["conv_adj"])
    assert abs(sem.compute_cfl(w, 0.01) - float(g["cfl"])) < 1e-12 * float(g["cfl"])


@pytest.mark.parametrize("case", ["2d", "3d"])
def test_operator_identities(case):
    """Independent facts: symmetry, adjointness, exactness on polynomials, partition of unity."""
    hm, sem = mg.build(case)
    dim = sem.dim
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(sem.shape1), rng.standard_normal(sem.shape1)
    assert abs(np.sum(v * sem.axhelm_local(u, 1.0, 0.3)) - np.sum(u * sem.axhelm_local(v, 1.0, 0.3))) < 1e-11
    p = rng.standard_normal(sem.shape2)
    w = [rng.standard_normal(sem.shape1) for _ in range(dim)]
    lhs = np.sum(p * sem.opdiv(w))
    rhs = sum(np.sum(w[i] * gi) for i, gi in enumerate(sem.opgradt(p)))
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)
    # E symmetric positive semi-definite
    q = rng.standard_normal(sem.shape2)
    assert abs(np.sum(q * sem.cdabdtp(p)) - np.sum(p * sem.cdabdtp(q))) < 1e-10
    assert np.sum(p * sem.cdabdtp(p)) > 0
    # exact diagonal entries
    ed = sem.e_diag()
    e = np.zeros(sem.shape2)
    idx = (0,) + (1,) * dim
    e[idx] = 1.0
    assert abs(sem.cdabdtp(e)[idx] - ed[idx]) < 1e-12 * ed[idx]
    hd = sem.helm_diag_local(0.5, 2.0)
    e1 = np.zeros(sem.shape1)
    idx1 = (1,) + (2,) * dim
    e1[idx1] = 1.0
    assert abs(sem.axhelm_local(e1, 0.5, 2.0)[idx1] - hd[idx1]) < 1e-12 * hd[idx1]
    # gradient of a linear function, Laplacian of a constant, volume
    g = sem.gradm1(2.0 * sem.X[0] - sem.X[1])
    assert np.max(np.abs(g[0] - 2.0)) < 1e-11 and np.max(np.abs(g[1] + 1.0)) < 1e-11
    assert np.max(np.abs(sem.axhelm_local(np.ones(sem.shape1), 1.0, 0.0))) < 1e-11
    # gather-scatter: multiplicity and idempotence of the averaging
    assert np.all(sem.mult >= 1) and close(sem.dsavg(sem.dsavg(u)), sem.dsavg(u), 1e-14)


def test_vector_space_matches_golden_and_reference_semantics():
    g = golden("2d")
    hm, sem = mg.build("2d")
    a, b = NekDVector(sem, 1), NekDVector(sem, 1)
    for i in range(2):
        a.v[i][...] = g["in_u"][i]
        b.v[i][...] = g["in_w"][i]
    a.pr[...] = g["in_p"]
    b.pr[...] = g["in_q"]
    a.theta[0][...] = g["in_ta"]
    b.theta[0][...] = g["in_tb"]
    assert abs(a.dot(b) - float(g["dot"])) < 1e-13 * abs(float(g["dot"]))
    assert a.get_size() == int(g["size"]) == 3 * sem.lvn + sem.lpn
    # pressure is NOT part of the inner product (real_vectors.f90:217-224)
    c = a.copy()
    c.pr[...] = 0.0
    assert c.dot(b) == a.dot(b)
    a.save_rst(b, 1)
    with pytest.raises(ValueError):
        a.save_rst(b, 3)                                  # irst == torder is an error (:264-267)
    a.axpby(0.3, b, -1.7, consistent_rst=False)
    a.scal(1.0 / 3.0)
    assert np.array_equal(np.stack(a.v), g["axpby_v"]) and np.array_equal(a.pr, g["axpby_pr"])
    assert np.array_equal(a.theta[0], g["axpby_theta"])
    # the history slot received alpha * vec's MAIN field (real_vectors.f90:188-192)
    expect = (b.v[0] * (-1.7) + 0.3 * b.v[0]) * (1.0 / 3.0)
    assert np.allclose(a.v_rst[0][0], g["axpby_rst1_v"][0]) and np.allclose(a.v_rst[0][0], expect)
    # default (consistent) treatment: the slot receives alpha * vec's slot
    c2, d2 = NekDVector(sem, 1), NekDVector(sem, 1)
    c2.v[0][...] = 1.0
    d2.v[0][...] = 2.0
    c2.save_rst(c2, 1)
    d2.save_rst(c2, 1)
    c2.axpby(0.5, d2, 1.0)
    assert np.allclose(c2.v[0], 2.0) and np.allclose(c2.v_rst[0][0], 1.5)
    a.zero()
    assert a.nrst == 0 and a.norm() == 0.0


def test_rand_is_continuous_masked_and_partition_independent():
    hm, sem = mg.build("3d")
    v = NekDVector(sem)
    v.rand(ifnorm=True, seed=9)
    assert abs(v.norm() - 1.0) < 1e-13
    for i in range(3):
        assert close(sem.dsavg(v.v[i]), v.v[i], 1e-13) and np.all(v.v[i][sem.mask[i] == 0] == 0)
    # the raw noise of an element depends only on its GLOBAL id: a sub-mesh reproduces it
    sub = hm.take(np.array([5, 2]))
    ssem = SEM(sub)
    raw_full = v.raw_noise(9, hm.elem_gid, 1)
    raw_sub = NekDVector(ssem).raw_noise(9, sub.elem_gid, 1)
    assert np.array_equal(raw_sub[0], raw_full[5]) and np.array_equal(raw_sub[1], raw_full[2])


def test_matvec_and_eigs_match_golden_2d():
    from oracle.krylov import eigs
    from oracle.lns import ExptA, LNSConfig
    g = golden("2d")
    hm, sem = mg.build("2d")
    U = mg.base_flow(sem)
    assert close(np.stack(U), g["baseflow"])
    A = ExptA(sem, U, LNSConfig(**mg.lns_cfg()))
    x = NekDVector(sem)
    for i in range(2):
        x.v[i][...] = g["mv_in_v"][i]
    y = A.matvec(x)
    assert close(np.stack(y.v), g["mv_out_v"], 1e-10) and close(y.pr, g["mv_out_pr"], 1e-9) and y.nrst == 2
    y2 = A.matvec(y)
    assert close(np.stack(y2.v), g["mv2_out_v"], 1e-10)
    # protocol: without restart history the first steps run at reduced order -> a different answer
    y.clear_rst_fields()
    y2b = A.matvec(y)
    assert not close(np.stack(y2b.v), g["mv2_out_v"], 1e-8)


@pytest.mark.slow
def test_eigs_matches_golden_2d():
    from oracle.krylov import eigs
    from oracle.lns import ExptA, LNSConfig
    g = golden("2d")
    hm, sem = mg.build("2d")
    U = mg.base_flow(sem)
    x = NekDVector(sem)
    for i in range(2):
        x.v[i][...] = g["mv_in_v"][i]
    cfg = mg.lns_cfg()
    cfg.update(tau=1.0, dt=0.025, re=10.0)
    A2 = ExptA(sem, U, LNSConfig(**cfg))
    lam, vecs, res, nmv = eigs(A2.matvec, x, nev=2, kdim=12, tol=1e-9, max_restarts=6)
    assert nmv == int(g["eigs_nmv"]) and np.max(np.abs(lam - g["eigs_lam"]) / np.abs(g["eigs_lam"])) < 1e-10
    assert np.all(res < 1e-9)


def test_dt_rule_matches_reference_formula():
    """neklab_nek_setup.f90:195-198: dt = cfl/ctarg ; nsteps = ceiling(tau/dt) ; dt = tau/nsteps."""
    from oracle.lns import dt_rule
    dt, ns = dt_rule(1.0, 23.7, 0.5)
    assert ns == int(np.ceil(1.0 / (0.5 / 23.7))) and abs(dt * ns - 1.0) < 1e-15


def test_krylov_on_known_spectrum():
    """eigs on a diagonal operator with known eigenvalues: leading pair, restart path included."""
    from oracle.krylov import eigs

    class V:
        def __init__(self, a):
            self.a = np.array(a, dtype=float)
        def copy(self):
            return V(self.a)
        def zero(self):
            self.a[:] = 0
        def dot(self, o):
            return float(self.a @ o.a)
        def norm(self):
            return float(np.sqrt(self.a @ self.a))
        def scal(self, s):
            self.a *= s
        def axpby(self, al, o, be):
            self.a = al * o.a + be * self.a

    d = np.concatenate([[1.5, -1.2, 1.1], np.linspace(0.0, 0.9, 57)])
    lam, vecs, res, nmv = eigs(lambda v: V(d * v.a), V(np.ones(60)), nev=2, kdim=12, tol=1e-10, max_restarts=30)
    assert np.allclose(np.sort(np.abs(lam))[::-1], [1.5, 1.2], atol=1e-9) and np.all(res < 1e-10)
    assert abs(abs(vecs[0].a[0]) - 1.0) < 1e-8


def test_oracle_gmres_and_nonlinear_map():
    """Oracle twins of the Newton-Krylov row: GMRES on exp(tau L) - I reproduces its own Arnoldi residual estimate with
    history-free Krylov vectors; the nonlinear map reduces to the linearised one for small amplitudes."""
    from oracle import krylov as K
    from oracle.lns import ExptA, LNSConfig
    hm = box_mesh((3, 2), 5, lengths=(1.0, 1.0), deform=0.02)
    sem = SEM(hm)
    rng = np.random.default_rng(0)
    U = [sem.mask[i] * sem.dsavg(np.sin(2 * sem.X[0] + i) * np.cos(sem.X[1])) for i in range(2)]
    cfg = LNSConfig(re=20.0, torder=2, tau=0.05, dt=0.01, vtol=1e-13, ptol=1e-13, maxit_v=300, maxit_p=3000)
    A = ExptA(sem, U, cfg)
    b = NekDVector(sem)
    for i in range(2):
        b.v[i][...] = sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1))

    def mv(v):                      # the operator GMRES sees: impulsive start, no history replay
        w = v.copy()
        w.clear_rst_fields()
        return A.matvec(w)

    x, res, nmv = K.gmres(mv, b, atol=1e-9, kdim=25)
    r = mv(x)
    r.axpby(-1.0, x, 1.0)
    r.axpby(-1.0, b, 1.0)
    assert res <= 1e-9 and r.norm() < 1e-7 * b.norm(), (res, r.norm())
    # about the state of rest (a fixed point, so the frozen-base-flow Jacobian is the exact derivative):
    # F(eps v) / eps -> (exp(tau L(0)) - I) v, the nonlinear term being O(eps)
    zero = [np.zeros(sem.shape1) for _ in range(2)]
    An = ExptA(sem, zero, cfg)
    eps = 1e-4
    Xp = b.copy()
    Xp.scal(eps)
    F1 = An.nonlinear_map(Xp)
    F1.scal(1.0 / eps)
    A.set_baseflow(zero)
    Jb = mv(b)
    Jb.axpby(-1.0, b, 1.0)
    d = F1.copy()
    d.axpby(-1.0, Jb, 1.0)
    assert d.norm() < 1e-3 * Jb.norm(), (d.norm(), Jb.norm())


def test_c_port_matches_numpy_restatement():
    """oracle/c/sem_cpu.c (the C + OpenMP port timed by bench.py's cpu_baseline) against oracle/sem.py."""
    import subprocess
    from oracle.cport import CPort, load
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if load() is None:
        subprocess.run(["make", "-C", os.path.join(root, "oracle", "c")], check=True, capture_output=True)
    hm = box_mesh((3, 2, 2), 6, periodic=(True, False, False), deform=0.05)
    sem = SEM(hm)
    cp = CPort(sem)
    rng = np.random.default_rng(0)
    u = [rng.standard_normal(sem.shape1) for _ in range(3)]
    U = [rng.standard_normal(sem.shape1) for _ in range(3)]
    p = rng.standard_normal(sem.shape2)

    def err(a, b):
        return np.max(np.abs(a - b)) / np.max(np.abs(b))

    assert err(cp.axhelm_local(u[0], 0.3, 2.0), sem.axhelm_local(u[0], 0.3, 2.0)) < 1e-13
    assert err(cp.gs(u[1].copy()), sem.gs(u[1])) < 1e-14
    assert max(err(a, b) for a, b in zip(cp.opgradt(p), sem.opgradt(p))) < 1e-13
    assert err(cp.opdiv(u), sem.opdiv(u)) < 1e-13
    assert err(cp.cdabdtp(p), sem.cdabdtp(p)) < 1e-13
    assert max(err(a, b) for a, b in zip(cp.lns_conv_weak(U, u), sem.lns_conv_weak(U, u))) < 1e-13
    assert abs(cp.glsc3(u[0], U[0], np.ascontiguousarray(sem.bm1)) - sem.glsc3(u[0], U[0])) < 1e-12 * abs(sem.glsc3(u[0], U[0])) + 1e-12
    y = u[2].copy()
    cp.axpby(0.3, u[0], 0.5, y)
    assert err(y, 0.5 * u[2] + 0.3 * u[0]) < 1e-15


def test_c_port_time_step_matches_the_numpy_time_step():
    """oracle/cpu_step.py CStep.advance -- the whole time step on the C + OpenMP port, both PCG solvers included, the thing bench.py's
    cpu_baseline times end to end -- against oracle/lns.py ExptA.advance: four steps (bdf1, bdf2, bdf3, bdf3) from the same state give the
    same velocity and pressure and the same iteration counts, with tolerance-terminated and with fixed-count solves."""
    import subprocess
    from oracle.cport import load
    from oracle.cpu_step import CStep
    from oracle.lns import ExptA, LNSConfig
    from oracle.vectors import NekDVector
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle", "c")], check=True, capture_output=True)   # (rebuilds when the source is newer)
    assert load() is not None
    hm = box_mesh((3, 2, 2), 6, periodic=(True, False, False), deform=0.05)
    sem = SEM(hm)
    U = [sem.mask[0] * (1.0 + 0.3 * np.cos(sem.X[1])), sem.mask[1] * 0.2 * np.sin(sem.X[0]), sem.mask[2] * 0.1 * np.sin(sem.X[1])]
    rng = np.random.default_rng(3)
    u0 = [sem.mask[i] * sem.dsavg(rng.standard_normal(sem.shape1)) for i in range(3)]
    p0 = rng.standard_normal(sem.shape2)
    for fixed in (0, 7):
        cfg = LNSConfig(tau=0.04, re=40.0, torder=3, dt=0.01, vtol=1e-11, ptol=1e-10, maxit_v=200, maxit_p=3000, fixed_iters_v=fixed, fixed_iters_p=3 * fixed)
        A = ExptA(sem, U, cfg)
        x = NekDVector(sem)
        for i in range(3):
            x.v[i][...] = u0[i]
        x.pr[...] = p0
        A._reset_state(x, False)
        B = CStep(sem, U, cfg)
        B.reset(u0, p0)
        for step in range(4):
            A.advance()
            B.advance()
            sc = max(np.abs(a).max() for a in A.u)
            assert max(np.abs(a - b).max() for a, b in zip(A.u, B.u)) < 1e-9 * sc, (fixed, step)
            assert np.abs(A.p - B.p).max() < 1e-8 * np.abs(A.p).max(), (fixed, step)
        assert A.stats["v_iters"] == B.stats["v_iters"], (A.stats, B.stats)
        assert abs(A.stats["p_iters"] - B.stats["p_iters"]) <= (0 if fixed else 2), (A.stats, B.stats)


def test_matvec_matches_golden_3d_n8():
    """The lx1 = 8, 3-D fixture the GPU box checks the benchmark's kernel instantiations against (first matvec only here:
    the whole fixture takes half a minute of oracle time, tests/golden/make_golden.py 3d_n8)."""
    from oracle.lns import ExptA, LNSConfig
    from oracle.vectors import NekDVector
    g = golden("3d_n8")
    hm, sem = mg.build("3d_n8")
    A = ExptA(sem, list(g["baseflow"]), LNSConfig(**mg.lns_cfg()))
    x = NekDVector(sem)
    for i in range(3):
        x.v[i][...] = g["mv_in_v"][i]
    y = A.matvec(x)
    assert max(np.abs(y.v[i] - g["mv_out_v"][i]).max() for i in range(3)) < 1e-13 * np.abs(g["mv_out_v"]).max()
    assert max(np.abs(y.v_rst[1][i] - g["mv_out_rst2_v"][i]).max() for i in range(3)) < 1e-13 * np.abs(g["mv_out_v"]).max()


def test_block_cgs2_deflation_is_scale_free_and_sees_the_span_of_the_basis():
    """The oracle twin of nlg_basis_block_cgs2's two deflation tests: columns of norm 1e8 / 1e-8 are kept;
    a block in span(V) is deflated although the rounding noise the projections leave is not mutually dependent."""
    from oracle.krylov import block_cgs2, cgs2_step
    sem = SEM(box_mesh((3, 2), 5, periodic=(True, False), deform=0.04))
    def basis(k):
        V = []
        for j in range(k):
            v = NekDVector(sem)
            v.rand(ifnorm=True, seed=20 + j)
            if j:
                cgs2_step(V, v)
            v.scal(1.0 / v.norm())
            V.append(v)
        return V
    V = basis(3)
    for v, scale in enumerate((1e8, 1.0, 1e-8)):
        w = NekDVector(sem)
        w.rand(ifnorm=True, seed=50 + v)
        w.scal(scale)
        V.append(w)
    coef = block_cgs2(V, 3, 3)
    assert np.all(np.diag(coef[3:]) > 0.0)
    G = np.array([[a.dot(b) for b in V] for a in V])
    assert np.max(np.abs(G - np.eye(6))) < 1e-12
    V = basis(4)
    C = np.random.default_rng(1).standard_normal((4, 2))
    for v in range(2):
        w = V[0].copy()
        w.scal(C[0, v])
        for j in range(1, 4):
            w.axpby(C[j, v], V[j], 1.0)
        V.append(w)
    coef = block_cgs2(V, 4, 2)
    assert np.max(np.abs(coef[:4] - C)) < 1e-12 and np.all(np.diag(coef[4:]) == 0.0)
    assert V[4].norm() == 0.0 and V[5].norm() == 0.0


def test_single_reduction_pcg_twin_matches_the_standard_pcg():
    """oracle/lns.py pcg_helm_single_reduction (the twin of run_pcg's Chronopoulos-Gear branch, csrc/lns.hip cg_post_logic mode 4) against
    pcg_helm on the Helmholtz problem of a time step: same solution to rounding, same iteration count (the merged reduction changes when the
    convergence is NOTICED, not the iterates)."""
    from oracle.lns import ExptA, LNSConfig
    sem = SEM(box_mesh((3, 2, 2), 5, periodic=(True, False, False), deform=0.04))
    U = [sem.mask[i] * (1.0 if i == 0 else 0.0) * np.cos(sem.X[1]) for i in range(3)]
    A = ExptA(sem, U, LNSConfig(tau=0.02, re=40.0, torder=3, dt=0.01, vtol=1e-12, ptol=1e-12, maxit_v=200, maxit_p=2000))
    rng = np.random.default_rng(2)
    b = [sem.mask[i] * sem.gs(sem.bm1 * rng.standard_normal(sem.shape1)) for i in range(3)]
    h2 = 11.0 / 6.0 / 0.01
    it0 = A.stats["v_iters"]
    x = A.pcg_helm(b, h2)
    its = A.stats["v_iters"] - it0
    y, its_sr = A.pcg_helm_single_reduction(b, h2)
    assert its_sr == its and its > 3
    sc = max(np.abs(a).max() for a in x)
    assert max(np.abs(a - c).max() for a, c in zip(x, y)) < 1e-12 * sc
