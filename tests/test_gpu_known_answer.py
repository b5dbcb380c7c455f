"""The reference's only published number, on the reference's own mesh and base flow, through the GPU path.

/root/reference/test/neklabTests.py:43-45 asserts |mu_1| = 1.0156 +- 1e-4 for the leading eigenvalue of
exp(tau L), cylinder wake at Re = 50, tau = 1, lx1 = 6, bdf3, kdim = 128, nev = 2 (1cyl.usr:11,20; 1cyl.par).
Mesh coordinates, boundary conditions and base flow come from the reference's data files (fixture made by
tests/golden/make_reference_fixture.py); everything else is this repository's HIP path.

What round 2 established (profiles/r02_cylinder_*.txt):
* the discrete eigenvalue of the bdf3 propagator on this mesh is |mu_1| = 1.0157265 -- independent of the start vector
  (three seeds agree to 1e-7), of dt (1.015728 at dt / 2), of the solver tolerances, of lxd (6 .. 12), of the geometry
  source (float32-quantised coordinates of the field file or the double-precision rebuild from 1cyl.re2), and of the
  pressure projection -- PROVIDED every Krylov vector, the first included, carries a restart history (`warm_start`);
* with the reference's protocol the start vector has no history, its matvec starts impulsively at bdf1, and that rank-one
  inconsistency of the Arnoldi relation moves the converged Ritz value by up to +-8e-5 depending on the (compiler-RNG)
  start vector: 1.015705 .. 1.015865 over eight seeds, each converged to residual < 1e-8;
* with the literal reading of real_vectors.f90:188-192 the scatter is 2.4e-3 (1.0170 .. 1.0194).
The printed reference value 1.0156 is 1.27e-4 below the discrete eigenvalue and 1.05e-4 below the lowest sample of the
reference's own protocol: outside the +-1e-4 window by a quarter of its width, inside the scatter that the protocol itself
produces.  The assertions below state exactly that.
"""
import numpy as np
import pytest

from neklab_amd import host
from refdata import load_cylinder

pytestmark = pytest.mark.gpu


def _cylinder_operator(gpu_ctx):
    hm, ux, uy, p, re, lxd, _ = load_cylinder(with_bcs=True)
    gm = host.Mesh(gpu_ctx, hm, lxd=lxd)
    bf = host.nek_dvector(gm)
    bf.set_field(host.VX, ux)
    bf.set_field(host.VY, uy)
    A = host.exptA_linop(1.0, bf, re=re, torder=3, vtol=1e-9, ptol=1e-7, maxit_v=400, maxit_p=4000)
    A.init()
    return gm, A


def test_cylinder_re50_leading_eigenvalue(gpu_ctx, tmp_path):
    """The reference's protocol (start vector without history), as in round 1."""
    gm, A = _cylinder_operator(gpu_ctx)
    info = A.info()
    assert info["nsteps"] == 100 and abs(info["cfl"] - 0.5) < 0.01        # dt rule, neklab_nek_setup.f90:195-198
    eigvals, residuals, eigvecs, mu, nmv = host.linear_stability_analysis_fixed_point(
        A, 128, 2, tol=1e-6, outdir=str(tmp_path), seed=1)
    assert residuals[0] < 1e-6
    assert abs(abs(mu[0]) - 1.0156) < 3e-4, abs(mu[0])                    # published value +- (its window + the protocol's scatter)
    assert abs(abs(mu[0]) - 1.0157265) < 1.7e-4, abs(mu[0])               # the scatter of this protocol around the discrete eigenvalue
    assert abs(mu[0].imag) > 0.6 and np.isclose(mu[0], np.conj(mu[1]))    # oscillatory wake mode, St ~ 0.12
    assert abs(eigvals[0].real - np.log(1.0156)) < 3e-4                   # growth rate log|mu|/tau
    # outputs the reference's tooling reads
    rows = [ln.split() for ln in open(tmp_path / "eigs_output.txt") if not ln.startswith("#")]
    conv = [r for r in rows if r[5] == "T"]
    assert abs(float(conv[0][3]) - abs(mu[0])) < 1e-12                    # get_converged_eigs_data()['lambda_1']['modulus']
    spec = np.load(tmp_path / "dir_eigenspectrum.npy")
    assert spec.shape == (2, 3)


def test_cylinder_re50_discrete_eigenvalue_is_start_vector_independent(gpu_ctx, tmp_path):
    """With every Krylov vector carrying a restart history (warm_start) the leading Ritz value no longer depends on the
    start vector: two seeds, |mu_1| = 1.0157265 +- 2e-6 each -- 1.27e-4 above the printed reference value."""
    gm, A = _cylinder_operator(gpu_ctx)
    vals = []
    for seed in (2, 11):
        eigvals, residuals, eigvecs, mu, nmv = host.linear_stability_analysis_fixed_point(
            A, 128, 2, tol=1e-7, outdir=str(tmp_path), seed=seed, warm_start=True)
        assert residuals[0] < 1e-7
        vals.append(abs(mu[0]))
    assert abs(vals[0] - vals[1]) < 2e-6, vals
    assert abs(vals[0] - 1.0157265) < 2e-6, vals
    assert 1.0e-4 < abs(vals[0] - 1.0156) < 1.5e-4


def test_cylinder_base_flow_is_fixed_point_and_newton_returns_to_it(gpu_ctx):
    """Newton-Krylov row pinned on reference data: the reference's own Re = 50 base flow
    (BF_1cyl0.f00001, an UNSTABLE steady state -- time stepping leaves it, which is why the reference computes it by
    Newton) is a fixed point of the restated nonlinear map, and Newton + GMRES returns to it from a perturbed state.
    Measured (profiles/r01_cylinder_newton.log): |F(BF)| = 1.56e-5 at |BF| = 46.2 (tau = 1, 125 steps); from a wake
    perturbation of norm 8.9e-2 Newton ends 2.7e-5 from the reference field."""
    hm, ux, uy, p, re, lxd, _ = load_cylinder(with_bcs=True)
    gm = host.Mesh(gpu_ctx, hm, lxd=lxd)
    bf = host.nek_dvector(gm)
    bf.set_field(host.VX, ux)
    bf.set_field(host.VY, uy)
    sysm = host.nek_system(1.0, bf, re=re, maxit_v=400, maxit_p=4000)
    F = host.nek_dvector(gm)
    sysm.set_tolerance(1e-8)
    sysm.eval(bf, F)
    assert sysm.nl.info()["nsteps"] == 125                     # CFL limit 0.4 of the nonlinear set-up (fixed_point.f90:15)
    assert F.norm() < 3e-5 and F.norm() < 1e-6 * bf.norm()
    X = bf.copy()
    x, y = hm.x.ravel(), hm.y.ravel()
    X.set_field(host.VX, ux.ravel() + 0.02 * np.exp(-((x - 3.0) ** 2 + y ** 2) / 2.0) * hm.mask[0].ravel())
    d0 = X.copy()
    d0.sub(bf)
    out = host.newton_fixed_point_iteration(sysm, X, 1e-6, tol_mode=2, kdim=60)
    assert out["converged"] and out["iterations"] <= 10, out   # the count depends on where the loose early solves stall (|F| = 0 plateaus)
    d1 = X.copy()
    d1.sub(bf)
    assert d0.norm() > 3e-2 and d1.norm() < 1e-4, (d0.norm(), d1.norm())
    sysm.set_tolerance(1e-8)                                    # the claim is checked with tighter solves than Newton used
    sysm.eval(X, F)
    assert F.norm() < 1e-5


def test_poiseuille_re7500_orr_sommerfeld(gpu_ctx, tmp_path):
    """Second known answer, independent of Nek5000: plane Poiseuille flow at Re = 7500, alpha = 1 (the reference's
    examples/poiseuille/stability/direct, poiseuille.par:31, base flow 1 - y^2 at poiseuille.usr:140).  The leading
    eigenvalue of exp(tau L), tau = 1, must be exp(-i c) with the Orr-Sommerfeld phase speed c = 0.24989146 +
    0.00223497 i (computed by the Chebyshev collocation solver of scripts/poiseuille_oracle.py; Orszag 1971 gives
    c = 0.24989154 + 0.00223498 i): mu = 0.97110724 - 0.24785212 i, |mu| = 1.00223747.
    Measured: |mu - mu_OS| = 2.5e-5 on 10 x 12 elements, lx1 = 8 (profiles/r01_poiseuille_orr_sommerfeld_gpu.log)."""
    from neklab_amd.mesh import box_mesh
    mu_os = np.exp(-1j * (0.24989146 + 0.00223497j))
    hm = box_mesh((10, 12), 8, lengths=(2 * np.pi, 2.0), periodic=(True, False), deform=0.0, origin=(0.0, -1.0))
    gm = host.Mesh(gpu_ctx, hm)
    bf = host.nek_dvector(gm)
    bf.set_field(host.VX, 1.0 - hm.y ** 2)
    A = host.exptA_linop(1.0, bf, re=7500.0, torder=3, vtol=1e-11, ptol=1e-10, maxit_p=4000)
    A.init()
    eigvals, residuals, eigvecs, mu, nmv = host.linear_stability_analysis_fixed_point(
        A, 160, 2, tol=1e-6, outdir=str(tmp_path), seed=1)
    m = mu[0] if mu[0].imag < 0 else np.conj(mu[0])
    assert residuals[0] < 1e-6
    assert abs(m - mu_os) < 1e-4, (m, mu_os)
    assert abs(abs(m) - 1.00223747) < 5e-5          # a weakly UNSTABLE mode: growth rate 2.2e-3 resolved to 2 %
    st = A.stats()
    assert st["p_iters"] / st["steps"] < 60          # overlapping Schwarz + vertex coarse space + projection in 2-D
