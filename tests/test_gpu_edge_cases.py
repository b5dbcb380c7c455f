"""Edge cases of the C ABI: smallest / largest supported sizes, degenerate meshes, argument validation."""
import ctypes as C

import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("nel,n,periodic", [
    ((1, 1, 1), 4, (False, False, False)),     # one element, smallest lx1: no shared dofs at all
    ((1, 1, 2), 5, (True, True, False)),       # periodic directions one element wide: a dof meets itself
    ((2, 1, 1), 12, (False, False, False)),    # largest lx1 (one velocity component per pass in the kernels)
    ((2, 2), 10, (True, False)),               # 2-D, lx1 = 10
    ((3, 1), 4, (False, True)),
])
def test_operator_parity_at_size_limits(gpu_ctx, nel, n, periodic):
    hm = box_mesh(nel, n, periodic=periodic, deform=0.03)
    sem = SEM(hm)
    gm = host.Mesh(gpu_ctx, hm)
    dim = sem.dim
    rng = np.random.default_rng(0)
    v, out = host.nek_dvector(gm), host.nek_dvector(gm)
    u = [rng.standard_normal(sem.shape1) for _ in range(dim)]
    p = rng.standard_normal(sem.shape2)
    for i in range(dim):
        v.set_field(i, u[i])
    v.set_field(host.PR, p)
    lib = gm.lib
    assert rel(gm.get("vmult"), sem.vmult) < 1e-14
    host.check(lib.nlg_op_helmholtz(gm.h, v.h, out.h, 0.3, 2.0, 1))
    for i in range(dim):
        assert rel(out.get_field(i), sem.mask[i] * sem.gs(sem.axhelm_local(u[i], 0.3, 2.0))) < 1e-12
    host.check(lib.nlg_op_cdabdtp(gm.h, v.h, out.h))
    assert rel(out.get_field(host.PR), sem.cdabdtp(p)) < 1e-11
    host.check(lib.nlg_op_opdiv(gm.h, v.h, out.h))
    assert rel(out.get_field(host.PR), sem.opdiv(u)) < 1e-12
    host.check(lib.nlg_op_conv(gm.h, v.h, v.h, out.h, 1))
    ref = sem.lns_conv_weak(u, u, adjoint=True)
    sc = max(np.abs(r).max() for r in ref)
    for i in range(dim):
        assert np.max(np.abs(out.get_field(i) - ref[i].ravel())) < 1e-11 * sc


def test_outflow_mesh_matvec_runs_and_is_finite(gpu_ctx):
    """has_outflow = 1: no pressure null space, no projection; a short propagator must stay finite and contract."""
    hm = box_mesh((4, 3), 6, lengths=(4.0, 2.0), deform=0.02, outflow_xmax=True)
    gm = host.Mesh(gpu_ctx, hm)
    bf = host.nek_dvector(gm)
    bf.set_field(0, hm.mask[0] * (1.0 - (hm.y - 1.0) ** 2))
    A = host.exptA_linop(0.05, bf, re=20.0, dt=0.01)
    A.init()
    x, y = host.nek_dvector(gm), host.nek_dvector(gm)
    x.rand(True, seed=3)
    A.matvec(x, y)
    assert np.isfinite(y.norm()) and 0.0 < y.norm() < 1.5


def test_argument_validation(gpu_ctx):
    lib = host._lib.load()
    hm = box_mesh((2, 2), 6)
    gm = host.Mesh(gpu_ctx, hm)
    v = host.nek_dvector(gm)
    # unsupported sizes are refused with a message, nothing is silently emulated
    bad = box_mesh((2, 2), 6)
    bad.n = 11
    with pytest.raises(host.NlgError, match="unsupported"):
        host.Mesh(gpu_ctx, bad)
    with pytest.raises(host.NlgError):
        v.set_field(host.VZ, np.zeros(gm.lvn))                 # no z component on a 2-D mesh
    with pytest.raises(host.NlgError):
        v.set_field(host.VX, np.zeros(gm.lvn - 1))             # wrong length
    with pytest.raises(host.NlgError):
        v.get_field(host.THETA)                                # no scalar allocated
    with pytest.raises(host.NlgError):
        host.nek_dvector(gm, nscal=99)
    with pytest.raises(host.NlgError):
        host.KrylovBasis(gm, 0)
    B = host.KrylovBasis(gm, 3)
    with pytest.raises(host.NlgError):
        B.block_dot(4, v)                                      # k > nvec
    with pytest.raises(host.NlgError):
        B[5]
    bf = host.nek_dvector(gm)
    with pytest.raises(host.NlgError):
        host.exptA_linop(-1.0, bf)                             # tau must be positive
    with pytest.raises(host.NlgError):
        host.exptA_linop(1.0, bf, torder=4)
    A = host.exptA_linop(1.0, bf)                              # zero base flow: the CFL rule cannot give a dt
    with pytest.raises(host.NlgError, match="CFL"):
        A.init()
    X = [host.nek_dvector(gm)]
    A2 = host.exptA_linop(0.02, bf, dt=0.01)
    A2.init()
    with pytest.raises(host.NlgError):
        host.eigs(A2, X, kdim=1)                               # kdim must exceed nev
    z = host.nek_dvector(gm)
    with pytest.raises(host.NlgError, match="zero norm"):
        host.eigs(A2, X, kdim=4, x0=z)                         # zero start vector
    # a Jacobian of the wrong sign is caught at mesh creation
    flipped = box_mesh((2, 2), 6)
    flipped.x = -flipped.x
    with pytest.raises(host.NlgError, match="Jacobian"):
        host.Mesh(gpu_ctx, flipped)
    # NULL handles
    assert lib.nlg_vec_zero(None) != 0 and b"NULL" in lib.nlg_last_error()
    out = C.c_double()
    assert lib.nlg_vec_dot(None, None, C.byref(out)) != 0


def test_scalar_fields_in_the_basis(gpu_ctx):
    """nscal > 0 (temperature in the inner product, real_vectors.f90:220): block kernels with 4 components."""
    hm = box_mesh((2, 2, 2), 6, deform=0.03)
    sem = SEM(hm)
    gm = host.Mesh(gpu_ctx, hm)
    rng = np.random.default_rng(1)
    k = 5
    B = host.KrylovBasis(gm, k + 1, nscal=1)
    cols = []
    for j in range(k):
        f = [rng.standard_normal(sem.shape1) for _ in range(4)]
        for i in range(3):
            B[j].set_field(i, f[i])
        B[j].set_field(host.THETA, f[3])
        cols.append(f)
    w = host.nek_dvector(gm, nscal=1)
    fw = [rng.standard_normal(sem.shape1) for _ in range(4)]
    for i in range(3):
        w.set_field(i, fw[i])
    w.set_field(host.THETA, fw[3])
    h = B.block_dot(k, w)
    ref = np.array([sum(sem.glsc3(c[i], fw[i]) for i in range(4)) for c in cols])
    assert np.max(np.abs(h - ref)) < 1e-13 * np.max(np.abs(ref))
    for j in range(k):                                     # orthonormalise the columns, then project w
        B.cgs2(j, B[j])
    hh, beta = B.cgs2(k, w)
    assert beta > 0 and abs(w.norm() - 1.0) < 1e-13
    assert np.max(np.abs(B.block_dot(k, w))) < 1e-13


@pytest.mark.parametrize("k", [23, 24, 40, 64, 65, 100, 128, 131])
def test_cgs2_fused_sweep_matches_separate_kernels(gpu_ctx, k):
    """CGS2 (LightKrylov's double Gram-Schmidt) through `nlg_basis_cgs2` -- which for 24 <= k <= 64 fuses the
    first subtraction with the second projection (k_block_axpy_dot) -- against the same four passes made with the
    separate block_dot / block_axpy entry points: coefficients, the orthogonalised vector (velocity, pressure and the
    restart-history blocks) and the norm.  k = 23 takes the unfused path and pins the comparison itself; 64 < k <= 128 (round 4) takes
    the two-tile sweep k_block_axpy_dot2 (second half of a point's basis values parked in LDS), above 128 the last 128 vectors are fused."""
    hm = box_mesh((3, 3, 2), 6, deform=0.03)
    gm = host.Mesh(gpu_ctx, hm)
    B = host.KrylovBasis(gm, k + 1)
    hist = host.nek_dvector(gm)
    for j in range(k):                                     # orthonormal columns with restart history
        v = B[j]
        v.rand(False, seed=300 + j)
        for irst in (1, 2):
            hist.rand(False, seed=1000 + 10 * j + irst)
            v.save_rst(hist, irst)
        B.cgs2(j, v)
    w1, w2 = host.nek_dvector(gm), host.nek_dvector(gm)
    w1.rand(False, seed=7)
    for irst in (1, 2):
        hist.rand(False, seed=5000 + irst)
        w1.save_rst(hist, irst)
    w2.assign(w1)
    # reference: four separate passes; the history receives the sum of both coefficient sets
    h1 = B.block_dot(k, w2)
    B.block_axpy(k, h1, w2)
    h2 = B.block_dot(k, w2)
    B.block_axpy(k, h2, w2)
    nrm = w2.norm()
    w2.scal(1.0 / nrm)
    h, beta = B.cgs2(k, w1)
    assert np.max(np.abs(h - (h1 + h2))) < 1e-12 * np.max(np.abs(h1))
    assert abs(beta - nrm) < 1e-12 * nrm
    for f in range(4):
        a, b = w1.get_field(f), w2.get_field(f)
        assert np.max(np.abs(a - b)) < 1e-12 * max(np.max(np.abs(b)), 1e-300), f
    ra, rb = host.nek_dvector(gm), host.nek_dvector(gm)
    for irst in (1, 2):
        w1.get_rst(ra, irst)
        w2.get_rst(rb, irst)
        for f in range(4):
            a, b = ra.get_field(f), rb.get_field(f)
            assert np.max(np.abs(a - b)) < 1e-11 * max(np.max(np.abs(b)), 1e-300), (irst, f)
    assert np.max(np.abs(B.block_dot(k, w1))) < 1e-12


def test_sampled_kernel_timing(gpu_ctx):
    """nlg_prof_sample: only every stride-th launch of an enabled class carries a pair of events (bench.py times every 8th launch of
    the dominant class, because timing every launch costs 12 us of stream time per launch); counts and totals follow."""
    import ctypes as C
    hm = box_mesh((2, 2, 2), 6)
    gm = host.Mesh(gpu_ctx, hm)
    lib = gpu_ctx.lib
    v, out = host.nek_dvector(gm), host.nek_dvector(gm)
    v.rand(False, seed=3)

    def timed(stride, calls):
        host.check(lib.nlg_prof_enable(gpu_ctx.h, -1))
        host.check(lib.nlg_prof_sample(gpu_ctx.h, stride))
        host.check(lib.nlg_prof_reset(gpu_ctx.h))
        for _ in range(calls):
            host.check(lib.nlg_op_helmholtz(gm.h, v.h, out.h, 0.3, 2.0, 0))
        cnt, ms = C.c_int64(), C.c_double()
        host.check(lib.nlg_prof_get(gpu_ctx.h, b"axhelm", C.byref(cnt), C.byref(ms)))
        host.check(lib.nlg_prof_enable(gpu_ctx.h, 0))
        host.check(lib.nlg_prof_sample(gpu_ctx.h, 1))
        return cnt.value, ms.value

    n1, t1 = timed(1, 8)
    n4, t4 = timed(4, 8)
    assert n1 == 8 and n4 == 2 and t1 > 0 and t4 > 0
    assert host.check(lib.nlg_prof_sample(gpu_ctx.h, 1)) is None
    assert lib.nlg_prof_sample(gpu_ctx.h, 0) != 0          # stride < 1 is an error


def test_small_mesh_pressure_kernels_agree_with_the_one_wave_kernels():
    """Below NLG_SMALL_E local elements the pressure operator runs three waves per element (k_opgradt3w / k_opdiv3w, the strong-scaling
    variants).  The same propagator application in two fresh processes -- variants on (default) and off (NLG_SMALL_E=0) -- must give the
    same fields and, within a couple of iterations, the same pressure iteration count: the first build of k_opgradt3w let wave 0 write the
    updated PCG direction in place while the other two waves were still reading it, which no parity test saw (the solve still converged,
    to the same answer) and which doubled the iteration count of a 4-rank rehearsal."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from neklab_amd import host
from neklab_amd.mesh import box_mesh
hm = box_mesh((9, 8, 8), 8, deform=0.05)
ctx = host.Context(0); gm = host.Mesh(ctx, hm)
X = [hm.x, hm.y, hm.z]
gb = host.nek_dvector(gm)
gb.set_field(0, hm.mask[0] * np.sin(X[1]) * np.cos(X[2])); gb.set_field(1, hm.mask[1] * 0.5 * np.sin(X[2]) * np.cos(X[0]))
A = host.exptA_linop(0.05, gb, re=100.0, dt=0.01, torder=3, vtol=1e-10, ptol=1e-9, maxit_v=200, maxit_p=2000); A.init()
v = host.nek_dvector(gm); v.rand(True, seed=5); w = host.nek_dvector(gm)
A.matvec(v, w)
st = A.stats()
print("RESULT", st["p_iters"], st["v_iters"], " ".join("%%.15e" %% float(np.sum(np.abs(w.get_field(i)) * (1.0 + 0.001 * (np.arange(w.get_field(i).size) %% 977)))) for i in range(3)))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for tag, val in (("on", None), ("off", "0")):
        env = dict(os.environ)
        env.pop("NLG_SMALL_E", None)
        if val is not None:
            env["NLG_SMALL_E"] = val
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        ln = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][-1].split()
        out[tag] = (int(ln[1]), int(ln[2]), [float(x) for x in ln[3:]])
    (pa, va, ca), (pb, vb, cb) = out["on"], out["off"]
    assert va == vb and abs(pa - pb) <= max(2, pb // 50), out
    for a, b in zip(ca, cb):
        assert abs(a - b) < 1e-7 * max(abs(b), 1e-30), out      # (solver tolerances 1e-10 / 1e-9: the two builds round differently)


def test_single_reduction_pcg_matches_the_two_reduction_pcg():
    """NLG_PCG_SINGLE_RED=1 (the default with several ranks): the velocity and scalar solves run Chronopoulos & Gear's PCG -- one reduction per
    iteration carrying (w, u), (r, u) and |r|^2, p and s = A p as recurrences (csrc/lns.hip cg_post_logic mode 4; oracle twin
    oracle/lns.py pcg_helm_single_reduction).  The same propagator application in two processes, switch on and off: the same fields to the
    solver tolerance, the same Helmholtz iteration counts (the convergence is noticed one operator application later, the iterates
    are the same), direct and with the temperature coupling."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from neklab_amd import host
from neklab_amd.mesh import box_mesh
heat = len(sys.argv) > 1 and sys.argv[1] == "heat"
hm = box_mesh((4, 3, 3), 8, deform=0.05)
ctx = host.Context(0); gm = host.Mesh(ctx, hm)
X = [hm.x, hm.y, hm.z]
gb = host.nek_dvector(gm, 1 if heat else 0)
gb.set_field(0, hm.mask[0] * np.sin(X[1]) * np.cos(X[2])); gb.set_field(1, hm.mask[1] * 0.5 * np.sin(X[2]) * np.cos(X[0]))
kw = dict(ifheat=1, conductivity=0.3, rhocp=1.5, buoy=(0.0, 5.0, 0.0)) if heat else {}
if heat: gb.set_field(host.THETA, 1.0 - X[1] / hm.lengths[1])
A = host.exptA_linop(0.05, gb, re=100.0, dt=0.01, torder=3, vtol=1e-11, ptol=1e-10, maxit_v=200, maxit_p=2000, **kw); A.init()
v = host.nek_dvector(gm, 1 if heat else 0); v.rand(True, seed=5); w = host.nek_dvector(gm, 1 if heat else 0)
A.matvec(v, w)
st = A.stats()
fields = list(range(3)) + ([host.THETA] if heat else [])
print("RESULT", st["v_iters"], st["p_iters"], " ".join("%%.15e" %% float(np.sqrt(np.sum(w.get_field(i) ** 2))) for i in fields),
      " ".join("%%.15e" %% float(np.sum(np.abs(w.get_field(i)) * (1.0 + 0.001 * (np.arange(w.get_field(i).size) %% 977)))) for i in fields))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("plain", "heat"):
        out = {}
        for tag, val in (("two", "0"), ("one", "1")):
            env = dict(os.environ)
            env["NLG_PCG_SINGLE_RED"] = val
            r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            ln = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][-1].split()
            out[tag] = (int(ln[1]), int(ln[2]), [float(x) for x in ln[3:]])
        (va, pa, ca), (vb, pb, cb) = out["two"], out["one"]
        assert abs(va - vb) <= max(1, va // 50) and abs(pa - pb) <= max(2, pa // 50), (mode, out)
        for a, b in zip(ca, cb):
            assert abs(a - b) < 1e-8 * max(abs(a), 1e-30), (mode, out)


def test_deferred_solution_update_of_the_velocity_pcg_is_bit_identical():
    """The velocity PCG does not update x every iteration: the operator kernel stores direction i into slot i mod PH of a ring, the scalar
    logic keeps alpha_i, and the consumer of x assembles x = sum alpha_i p_i in iteration order (csrc/lns.hip k_add_hist; k_x_flush when a
    solve outlasts the ring).  Same additions on the same operands: the matvec must give the same BITS with the ring switched off
    (NLG_PCG_DEFER_X=0, x += alpha p inside k_cg_update), at its default depth, at depth 3 (several flushes and wrap-arounds per solve) and
    at depth 1 (in place, a flush every iteration) -- 3-D single vector, 2-D, and a block of three lanes that converge at different counts.
    The pressure PCG does the same where the gradient kernel performs the direction update (NLG_PCG_DEFER_XP; the update kernel of the
    preconditioner then streams neither x nor p, pres_solve assembles x)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from neklab_amd import host
from neklab_amd.mesh import box_mesh
mode = sys.argv[1]
dim = 2 if mode == "2d" else 3
hm = box_mesh((5, 4) if dim == 2 else (4, 3, 2), 7 if dim == 2 else 8, periodic=(True,) + (False,) * (dim - 1), deform=0.04)
ctx = host.Context(0); gm = host.Mesh(ctx, hm)
X = [hm.x, hm.y] + ([hm.z] if dim == 3 else [])
heat = mode == "heat"
gb = host.nek_dvector(gm, 1 if heat else 0)
gb.set_field(0, hm.mask[0] * (1.0 + 0.5 * np.sin(X[0]) * np.cos(X[1]))); gb.set_field(1, hm.mask[1] * 0.3 * np.sin(2 * X[0]))
kw = dict(ifheat=1, conductivity=0.3, rhocp=1.5, buoy=(0.0, 5.0, 0.0)) if heat else {}
if heat: gb.set_field(host.THETA, 1.0 - X[1] / hm.lengths[1])
A = host.exptA_linop(0.05, gb, re=40.0, dt=0.01, torder=3, vtol=1e-12, ptol=1e-11, maxit_v=400, maxit_p=4000, **kw); A.init()
s = 3 if mode == "block" else 1
vin = []
for v in range(s):
    x = host.nek_dvector(gm, 1 if heat else 0); x.rand(True, seed=40 + v); x.scal(10.0 ** (-3 * v)); vin.append(x)
out = [host.nek_dvector(gm, 1 if heat else 0) for _ in range(s)]
if s == 1: A.matvec(vin[0], out[0])
else: A.matvec_block(vin, out)
st = A.stats()
words = []
for w in out:
    for i in list(range(dim)) + ([host.THETA] if heat else []):
        f = w.get_field(i).ravel()
        words.append(float(np.sqrt(np.sum(f * f))).hex()); words.append(float(np.sum(f * (1.0 + 0.001 * (np.arange(f.size) %% 977)))).hex())
print("RESULT", st["v_iters"], st["p_iters"], " ".join(words))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("3d", "2d", "block", "heat"):
        out = {}
        for depth in ("0", "", "3", "1", "16"):      # ("": the defaults -- velocity ring 16, pressure ring off)
            env = dict(os.environ)
            env.pop("NLG_PCG_DEFER_X", None)
            env.pop("NLG_PCG_DEFER_XP", None)
            if depth:
                env["NLG_PCG_DEFER_X"] = depth
                env["NLG_PCG_DEFER_XP"] = depth      # the pressure PCG keeps its directions in a ring of its own (3-D, lx1 = 8 .. 10)
            r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            out[depth] = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][-1]
        assert int(out["0"].split()[1]) >= 8, out["0"]          # the solves do outlast a ring of depth 3
        for depth in ("", "3", "1", "16"):
            assert out[depth] == out["0"], (mode, depth, out)


def test_compact_divergence_weights_are_bit_identical():
    """k_opdiv3n reads the weights mask_i * binvm1 of the consistent Poisson operator as ONE array (binvm1) and one byte per point (bit i =
    mask_i) instead of three arrays (csrc/sem.hip sem_opdiv_lanes; NLG_OPDIV_MASKB=0 = the three arrays).  Same products: a matvec gives the
    same BITS either way, also where the three masks differ (a free-slip plane: tangential components free, normal component fixed).  The
    Jacobi preconditioner mask_i / diag(H) of the velocity PCG is read the same way (1 / diag and the mask bytes, NLG_PC_MASKB=0 = three arrays)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from neklab_amd import host
from neklab_amd.mesh import box_mesh
hm = box_mesh((4, 3, 2), 8, periodic=(True, False, False), deform=0.0)
ymin = np.abs(hm.y - hm.y.min()) < 1e-9
zwall = (np.abs(hm.z - hm.z.min()) < 1e-9) | (np.abs(hm.z - hm.z.max()) < 1e-9)
for c in (0, 2):
    hm.mask[c][ymin & ~zwall] = 1.0          # free-slip on y = ymin: u and w free, v fixed
assert not np.array_equal(hm.mask[0], hm.mask[1])
ctx = host.Context(0); gm = host.Mesh(ctx, hm)
gb = host.nek_dvector(gm)
gb.set_field(0, hm.mask[0] * (1.0 + 0.5 * np.sin(hm.x) * np.cos(hm.y))); gb.set_field(1, hm.mask[1] * 0.3 * np.sin(2 * hm.x))
A = host.exptA_linop(0.03, gb, re=40.0, dt=0.01, torder=3, vtol=1e-12, ptol=1e-11, maxit_v=400, maxit_p=4000); A.init()
x = host.nek_dvector(gm); x.rand(True, seed=7); w = host.nek_dvector(gm)
A.matvec(x, w)
st = A.stats()
words = []
for i in range(4):
    f = w.get_field(i).ravel()
    words.append(float(np.sqrt(np.sum(f * f))).hex()); words.append(float(np.sum(f * (1.0 + 0.001 * (np.arange(f.size) %% 977)))).hex())
print("RESULT", st["v_iters"], st["p_iters"], " ".join(words))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for val in ("00", "10", "01", "11"):     # (divergence weights, Jacobi preconditioner of the velocity PCG: k_cg_init / k_cg_update, NLG_PC_MASKB)
        env = dict(os.environ)
        env["NLG_OPDIV_MASKB"] = val[0]
        env["NLG_PC_MASKB"] = val[1]
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[val] = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][-1]
    assert out["00"] == out["10"] == out["01"] == out["11"], out
