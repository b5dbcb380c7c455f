"""The Fortran 2008 shim (neklab_amd/fortran) driven like LightKrylov would drive it, on the GPU (drivers and the LightKrylov
stand-in: tests/fortran)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "tests", "fortran")

pytestmark = pytest.mark.gpu


def test_fortran_shim_arnoldi_matches_c_abi(gpu_ctx):
    subprocess.run(["make", "-s", "-C", FDIR], check=True)
    exe = os.path.join(FDIR, "_build", "arnoldi_driver")
    hm = box_mesh((3, 2), 6, lengths=(3.0, 2.0), periodic=(True, False), deform=0.03)
    kdim, tau, re, dt = 3, 0.03, 30.0, 0.01
    bf = [hm.mask[0] * np.cos(hm.y), 0.2 * hm.mask[1] * np.sin(hm.x)]
    gm = host.Mesh(gpu_ctx, hm)
    x0 = host.nek_dvector(gm)
    x0.rand(True, seed=4)
    v0 = [x0.get_field(i) for i in range(2)]
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "mesh.bin"), "wb") as f:
        np.array([2, 6, hm.E, kdim], dtype=np.int32).tofile(f)
        np.array([tau, re, dt], dtype=np.float64).tofile(f)
        for a in (hm.x, hm.y):
            a.astype(np.float64).tofile(f)
        hm.glo_num.astype(np.int64).tofile(f)
        for a in (hm.mask[0], hm.mask[1], bf[0], bf[1], v0[0], v0[1]):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    r = subprocess.run([exe], cwd=tmp, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    H = np.zeros((kdim + 1, kdim))
    alias = size = move = realloc = trim = None
    for ln in r.stdout.splitlines():
        p = ln.split()
        if p and p[0] == "H":
            H[int(p[1]) - 1, int(p[2]) - 1] = float(p[3])
        elif p and p[0] == "ALIAS":
            alias = float(p[1])
        elif p and p[0] == "SIZE":
            size = int(p[1])
        elif p and p[0] == "MOVE":
            move = [float(v) for v in p[1:]]
        elif p and p[0] == "REALLOC":
            realloc = [float(v) for v in p[1:]]
        elif p and p[0] == "TRIM":
            trim = [float(v) for v in p[1:5]] + [p[5]]
    # same computation through the C ABI's block path
    gb = host.nek_dvector(gm)
    for i in range(2):
        gb.set_field(i, bf[i])
    A = host.exptA_linop(tau, gb, re=re, dt=dt, torder=3, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=4000)
    A.init()
    B = host.KrylovBasis(gm, kdim + 1)
    B[0].assign(x0)
    Href = np.zeros((kdim + 1, kdim), order="F")
    for k in range(kdim):
        host.arnoldi_step(A, B, k, Href)
    assert np.max(np.abs(H - Href)) < 1e-9 * np.max(np.abs(Href))
    assert abs(alias - 1.0) < 1e-12          # wrk = X(1); wrk%scal(2) must not touch X(1)
    assert size == x0.get_size()
    # move_alloc keeps the handles; Z = [Z, extra] moves the elements (their old selves are finalised after the temporary was
    # built): values survive, the moved element adopts its released handle, the copy of a live vector gets a clone
    n1, n2 = move[0], move[1]
    assert abs(n1 - 1.0) < 1e-12 and abs(n2 - 3.0) < 1e-12 and abs(move[2] - n1) < 1e-12 and abs(move[3] - n2) < 1e-12
    assert realloc[0] == 3 and abs(realloc[1] - 2.0 * n1) < 1e-12 and abs(realloc[2] - n2) < 1e-12
    assert abs(realloc[3] - 10.0) < 1e-12 and abs(realloc[4] - 5.0) < 1e-12
    # a moved element read once, then nlg_vec_pool_trim, then modified: its handle must have left the pool with the read
    # (owner-by-address copies re-adopt, other copies pin) -- ADVICE round 3, neklab_vectors.f90:87
    assert trim[0] == 4 and abs(trim[1] - n2) < 1e-12 and abs(trim[2] - 2.0 * n2) < 1e-12 and abs(trim[3] - n2 / 3.0) < 1e-12 and trim[4] in ("T", "F"), trim


@pytest.mark.parametrize("device_eigs", [0, 1])
def test_fortran_stability_driver_is_the_reference_call_sequence(gpu_ctx, device_eigs):
    """tests/fortran/stability_driver.f90 = the userchk of 1cyl.usr:13-24 + linear_stability_analysis_fixed_point
    (neklab_analysis.f90:77-93) with the reference's module names: `use neklab`, nek2vec / vec2nek, the positional
    constructor exptA_linop(tau, bf), init(), eigs, log(mu)/tau, save_eigenspectrum, outpost_dnek.  device_eigs = 0 runs
    LightKrylov's loop structure (stand-in) through the type-bound procedures, 1 the device block path (nlg_eigs)."""
    subprocess.run(["make", "-s", "-C", FDIR], check=True)
    exe = os.path.join(FDIR, "_build", "stability_driver")
    hm = box_mesh((4, 3), 6, lengths=(4.0, 2.0), periodic=(True, False), deform=0.04)
    kdim, nev, tau, re, vtol, ptol = 24, 2, 1.0, 10.0, 1e-11, 1e-10
    bfv = [hm.mask[0] * (1.0 + np.sin(hm.x) * np.cos(hm.y)), hm.mask[1] * np.sin(2 * hm.x) * np.cos(hm.y)]
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "case.bin"), "wb") as f:
        np.array([2, 6, hm.E, kdim, nev, device_eigs], dtype=np.int32).tofile(f)
        np.array([tau, re, vtol, ptol], dtype=np.float64).tofile(f)
        for a in (hm.x, hm.y):
            a.astype(np.float64).tofile(f)
        hm.glo_num.astype(np.int64).tofile(f)
        for a in (hm.mask[0], hm.mask[1], bfv[0], bfv[1]):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    r = subprocess.run([exe], cwd=tmp, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.split()}
    # the same analysis through the Python mirror of the C ABI
    gm = host.Mesh(gpu_ctx, hm)
    gb = host.nek_dvector(gm)
    for i in range(2):
        gb.set_field(i, bfv[i])
    A = host.exptA_linop(tau, gb, re=re, torder=3, vtol=vtol, ptol=ptol, maxit_v=400, maxit_p=4000)
    A.init()
    x0 = host.nek_dvector(gm)
    x0.rand(True, seed=1)                      # the shim's first rand() draws seed 1
    X = [host.nek_dvector(gm) for _ in range(nev)]
    mu, res, info = host.eigs(A, X, kdim=kdim, x0=x0, write_intermediate=False)
    lam = np.log(mu.astype(complex)) / tau
    spec = np.load(os.path.join(tmp, "dir_eigenspectrum.npy"))
    assert spec.shape == (nev, 3)
    flam = spec[:, 0] + 1j * spec[:, 1]
    assert np.all(spec[:, 2] < 1e-6)
    for a in flam:                             # same continuous-time eigenvalues (pair order / conjugate aside)
        assert min(abs(a - b) for b in np.concatenate([lam, np.conj(lam)])) < 1e-7 * max(1.0, abs(a)), (flam, lam)
    rows = [ln.split() for ln in open(os.path.join(tmp, "eigs_output.txt")) if not ln.startswith("#")]
    conv = [r_ for r_ in rows if r_[5] == "T"]
    assert len(conv) >= nev and abs(float(conv[0][3]) - abs(mu[0])) < 1e-7
    assert int(out["NSTEPS"][0]) == A.info()["nsteps"] and int(out["SIZE"][0]) == gb.get_size()
    assert abs(float(out["BFNORM"][0]) - gb.norm()) < 1e-12 * gb.norm()
    assert abs(float(out["OPBFNORM"][0]) - gb.norm()) < 1e-12 * gb.norm()
    c = [float(v) for v in out["COPIES"]]
    assert abs(c[0] - gb.norm()) < 1e-12 * c[0] and abs(c[1] - 2 * c[0]) < 1e-12 * c[0] and abs(c[2] - 3 * c[0]) < 1e-12 * c[0]
    # eigenvector field files: dir<session>0.f00001 carries the coordinates, both carry velocity and pressure
    from neklab_amd import nekio
    f1 = nekio.read_fld(os.path.join(tmp, "dirneklab0.f00001"))
    f2 = nekio.read_fld(os.path.join(tmp, "dirneklab0.f00002"))
    assert "x" in f1 and "x" not in f2 and "ux" in f2 and "p" in f2
    assert np.max(np.abs(f1["x"] - hm.x)) < 1e-14
    v = np.concatenate([f1["ux"].ravel(), f1["uy"].ravel()])
    # a Ritz vector of the converged pair: inside the span of the Python path's pair
    Bm = np.stack([np.concatenate([X[q].get_field(0), X[q].get_field(1)]) for q in range(nev)], axis=1)
    coef, *_ = np.linalg.lstsq(Bm, v, rcond=None)
    assert np.max(np.abs(v - Bm @ coef)) < 1e-5 * np.max(np.abs(v))


def test_outpost_matches_python_writer(gpu_ctx, tmp_path):
    """nlg_vec_outpost (the C-ABI writer behind the shim's outpost_dnek) against host.outpost_dnek / nekio.write_fld,
    which reproduces the reference's own field file byte for byte (tests/test_cpu_nekio.py)."""
    for nel, n in (((3, 2), 6), ((2, 2, 2), 5)):
        hm = box_mesh(nel, n, deform=0.04)
        gm = host.Mesh(gpu_ctx, hm)
        v = host.nek_dvector(gm)
        v.rand(True, seed=2)
        v.set_field(host.PR, np.random.default_rng(1).standard_normal(gm.lpn))
        for with_coords in (1, 0):
            pa = str(tmp_path / ("c%d_%d.f00001" % (hm.dim, with_coords)))
            host.check(gm.lib.nlg_vec_outpost(v.h, pa.encode(), with_coords, 0.0, 1 if with_coords else 2))
            pb = host.outpost_dnek(v, "ref", "x%d%d" % (hm.dim, with_coords), str(tmp_path), first_index=1 if with_coords else 2)[0]
            a, b = open(pa, "rb").read(), open(pb, "rb").read()
            from neklab_amd import nekio
            fa, fb = nekio.read_fld(pa), nekio.read_fld(pb)
            assert len(a) == len(b) - (0 if with_coords else hm.dim * 8 * gm.lvn + (hm.dim * hm.E * 8 if hm.dim == 3 else 0))
            assert a[:10] == b[:10] and a[132:136] == b[132:136]              # "#std 8 .." and the endian tag
            if with_coords:                    # header (incl. field code XUP, time, step) and everything up to the pressure
                assert a[:132] == b[:132]
                assert np.array_equal(fa["x"], fb["x"]) and np.array_equal(fa["y"], fb["y"])
            else:                              # host.outpost_dnek writes the coordinates into the first file of a call
                assert "x" not in fa
            for key in ("ux", "uy") + (("uz",) if hm.dim == 3 else ()):
                assert np.array_equal(fa[key], fb[key])
            # pressure on the velocity mesh: interpolated on the device here, with numpy there
            assert np.max(np.abs(fa["p"] - fb["p"])) < 1e-13 * np.max(np.abs(fb["p"]))


def test_fortran_thermosyphon_call_sequence(gpu_ctx):
    """tests/fortran/tsyphon_driver.f90 = the userchk of examples/thermosyphon/baseflow/tsyphon.usr:29-70 with the reference's names:
    nek_system_temp / nek_jacobian_temp, newton_fixed_point_iteration(sys, bf, tol, tol_mode = 2), exptA_linop_temp(tau, bf),
    eigs through LightKrylov's loop structure (stand-in).  A heated box below the onset of convection: Newton must return to the
    conduction state from a perturbed guess, and the spectrum of the temperature-coupled propagator about it must be the one the
    C ABI gives through the Python mirror."""
    subprocess.run(["make", "-s", "-C", FDIR], check=True)
    exe = os.path.join(FDIR, "_build", "tsyphon_driver")
    # (walls all around: in a box periodic in x the leading modes come in cos / sin pairs, which LightKrylov's loop -- the stand-in
    #  does not restart -- separates only slowly; undeformed elements: the hydrostatic pressure of the conduction state, quadratic in
    #  y, is then in the pressure space and the state at rest is an exact discrete fixed point)
    hm = box_mesh((4, 3), 6, lengths=(2.0, 1.0), periodic=(False, False), deform=0.0)
    # (tau = 2: the two leading multipliers exp(tau lambda) are then 7 % apart instead of 0.7 %, which an unrestarted Arnoldi needs)
    kdim, nev, tau, re, vtol, ptol = 40, 2, 2.0, 1.0, 1e-11, 1e-11
    cond, rhocp, buoy, endtime, tol = 1.0, 1.0, (0.0, 500.0, 0.0), 0.2, 1e-8
    k = np.pi / 2.0
    T0 = 1.0 - hm.y
    guess_t = T0 + 0.05 * hm.tmask * np.sin(np.pi * hm.y) * np.cos(k * hm.x)
    guess_u = [1e-3 * hm.mask[0] * np.sin(k * hm.x) * np.cos(np.pi * hm.y), 1e-3 * hm.mask[1] * np.cos(k * hm.x) * np.sin(np.pi * hm.y)]
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "case.bin"), "wb") as f:
        np.array([2, 6, hm.E, kdim, nev, 0], dtype=np.int32).tofile(f)
        np.array([tau, re, vtol, ptol, cond, rhocp, *buoy, endtime, tol], dtype=np.float64).tofile(f)
        for a in (hm.x, hm.y):
            a.astype(np.float64).tofile(f)
        hm.glo_num.astype(np.int64).tofile(f)
        for a in (hm.mask[0], hm.mask[1], hm.tmask, guess_u[0], guess_u[1], guess_t):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    r = subprocess.run([exe], cwd=tmp, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.split()}
    assert float(out["FNORM"][0]) < 10 * tol, out
    umax, vmax, tmax = (float(v) for v in out["BFMAX"])
    assert umax < 1e-6 and vmax < 1e-6 and abs(tmax - 1.0) < 1e-6, out          # the conduction state
    from neklab_amd import nekio
    bf = nekio.read_fld(os.path.join(tmp, "BF_neklab0.f00001"))
    assert np.max(np.abs(bf["t"] - T0)) < 1e-6
    assert os.path.exists(os.path.join(tmp, "nwtneklab0.f00001"))
    # the same spectrum through the Python mirror, about the exact conduction state
    gm = host.Mesh(gpu_ctx, hm)
    gb = host.nek_dvector(gm, 1)
    gb.set_field(host.THETA, T0)
    A = host.exptA_linop(tau, gb, re=re, torder=3, vtol=vtol, ptol=ptol, maxit_v=600, maxit_p=4000, ifheat=1, conductivity=cond, rhocp=rhocp,
                         buoy=buoy, dt=0.02)
    A.init()
    X = [host.nek_dvector(gm, 1) for _ in range(nev)]
    # the same start vector as LightKrylov's loop draws in the driver (the shim's first rand(): seed 1) -- the leading Ritz values of
    # a run depend on the start vector through its missing restart history, here (10 time steps per matvec) at the per-cent level
    x0 = host.nek_dvector(gm, 1)
    x0.rand(True, seed=1)
    mu, res, info = host.eigs(A, X, kdim=kdim, x0=x0, write_intermediate=False)
    lam = np.log(mu.astype(complex)) / tau
    spec = np.load(os.path.join(tmp, "dir_eigenspectrum.npy"))
    flam = spec[:, 0] + 1j * spec[:, 1]
    assert spec.shape == (nev, 3) and spec[0, 2] < 1e-6, spec
    assert lam[0].real < 0 and abs(flam[0] - lam[0]) < 1e-4 * abs(lam[0]), (flam, lam)


@pytest.mark.parametrize("adjoint", [0, 1])
def test_fortran_resolvent_linop(gpu_ctx, adjoint):
    """tests/fortran/resolvent_driver.f90: `resolvent_linop(omega, bf)` on `nek_zvector`s (neklab_linops.f90:198-205, resolvent.f90) through
    the shim -- forced period from rest, GMRES(64) on I - exp(T L) through the type-bound procedures, quarter-period continuation --
    against the Python mirror of the same C entry points."""
    subprocess.run(["make", "-s", "-C", FDIR], check=True)
    exe = os.path.join(FDIR, "_build", "resolvent_driver")
    hm = box_mesh((3, 3), 6, lengths=(2.0, 1.0), periodic=(True, False), deform=0.03)
    omega, re, vtol, ptol = 8.0, 20.0, 1e-12, 1e-12
    U = [hm.mask[0] * (4 * hm.y * (1 - hm.y)), np.zeros_like(hm.x)]
    gm = host.Mesh(gpu_ctx, hm)
    tmpv = host.nek_dvector(gm)
    fre, fim = [], []
    for i in range(2):      # C0, masked forcing fields: drawn on the device, the same arrays go to both paths
        tmpv.zero()
        tmpv.rand(True, seed=30 + i)
        fre.append(tmpv.get_field(i).copy())
        tmpv.zero()
        tmpv.rand(True, seed=40 + i)
        fim.append(tmpv.get_field(i).copy())
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "case.bin"), "wb") as f:
        np.array([2, 6, hm.E, adjoint], dtype=np.int32).tofile(f)
        np.array([omega, re, vtol, ptol], dtype=np.float64).tofile(f)
        for a in (hm.x, hm.y):
            a.astype(np.float64).tofile(f)
        hm.glo_num.astype(np.int64).tofile(f)
        for a in (hm.mask[0], hm.mask[1], U[0], U[1], fre[0], fre[1], fim[0], fim[1]):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    r = subprocess.run([exe], cwd=tmp, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    o = np.fromfile(os.path.join(tmp, "response.bin"), dtype=np.float64).reshape(4, -1)
    gb = host.nek_dvector(gm)
    gb.set_field(0, U[0])
    fz, out = host.nek_zvector(gm), host.nek_zvector(gm)
    for i in range(2):
        fz.re.set_field(i, fre[i])
        fz.im.set_field(i, fim[i])
    R = host.resolvent_linop(omega, gb, re=re, torder=3, vtol=vtol, ptol=ptol, maxit_v=400, maxit_p=4000)
    (R.rmatvec if adjoint else R.matvec)(fz, out)
    sc = max(np.abs(out.re.get_field(i)).max() for i in range(2))
    # both paths solve (I - exp(T L)) x = b to rtol 1e-6 with their own GMRES: agreement at that level
    for i in range(2):
        assert np.max(np.abs(o[i] - out.re.get_field(i))) < 2e-5 * sc, (i, np.max(np.abs(o[i] - out.re.get_field(i))) / sc)
        assert np.max(np.abs(o[2 + i] - out.im.get_field(i))) < 2e-5 * sc
    qn = float([ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("QNORM")][0])
    assert abs(qn - out.norm()) < 1e-5 * qn
