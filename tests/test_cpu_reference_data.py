"""The oracle's operators against DATA produced by the reference's own Nek5000 discretisation.

The fixture is the base-flow field file of the reference's only integration test (cylinder, Re = 50, curved
mesh, E = 1996, lx1 = 6): a steady Navier-Stokes solution in Nek5000's Pn-Pn-2 discretisation.  If the oracle
restates that discretisation correctly, then with the ORACLE's operators
  * the discrete divergence D u on the lx2 = lx1 - 2 Gauss mesh vanishes to the accuracy of the stored solution
    (the collocated divergence on the velocity mesh does not: it is 1e7 times larger),
  * the steady momentum residual  (U.grad)U + nu A U - D^T p  (dealiased convection on lxd = 9 points, pressure
    taken back to the Gauss mesh) vanishes at all nodes off the domain boundary to the solver tolerance.
This pins GLL/GL nodes, geometry factors, the divergence / gradient pair, the Helmholtz stiffness, the dealiased
convective operator, the mass matrix and the gather-scatter of oracle/sem.py against the reference.
"""
import numpy as np

from oracle.sem import SEM
from refdata import load_cylinder


def test_reference_base_flow_is_discretely_solenoidal_and_steady():
    hm, ux, uy, p, re, lxd, interior = load_cylinder()
    sem = SEM(hm, lxd=lxd)
    u = [ux.reshape(sem.shape1), uy.reshape(sem.shape1)]
    # divergence on the pressure mesh
    div = sem.opdiv(u) / sem.bm2
    l2 = np.sqrt(np.sum(div ** 2 * sem.bm2) / sem.volvm2)
    g = [sem.gradm1(a) for a in u]
    col = g[0][0] + g[1][1]
    l2col = np.sqrt(np.sum(col ** 2 * sem.bm1) / sem.volvm1)
    assert l2 < 1e-10 and np.abs(div).max() < 1e-8, (l2, np.abs(div).max())
    assert l2col > 1e5 * l2                       # the check discriminates: a different divergence would fail
    # steady momentum residual
    p2 = sem.to_mesh2(p.reshape(sem.shape1))      # Nek writes the pressure interpolated to mesh 1
    N = sem.lns_conv_weak(u, u)                   # = 2 (U.grad) U
    gp = sem.opgradt(p2)
    inter = interior.reshape(sem.shape1)
    for i in range(2):
        r = sem.gs(0.5 * N[i] + (1.0 / re) * sem.axhelm_local(u[i], 1.0, 0.0) - gp[i]) * sem.binvm1
        scale = np.abs(sem.gs(0.5 * N[i]) * sem.binvm1)[inter].max()
        assert scale > 0.5
        assert np.sqrt(np.mean(r[inter] ** 2)) < 1e-6 * scale and np.abs(r[inter]).max() < 1e-5 * scale
    # without dealiasing-consistent quadrature (convection collocated on the GLL mesh) the residual is far larger
    conv_col = u[0] * g[0][0] + u[1] * g[0][1]
    r_col = sem.gs(sem.bm1 * conv_col + (1.0 / re) * sem.axhelm_local(u[0], 1.0, 0.0) - gp[0]) * sem.binvm1
    assert np.sqrt(np.mean(r_col[inter] ** 2)) > 100 * 1e-6


def test_bfs_base_flow_divergence_and_steady_residual():
    """The base flow of the reference's transient-growth case (examples/back_fstep/transient_growth/BF_bfs0.f00001, E = 2760,
    lx1 = 6, Re = 600: `viscosity = -600`, bfs.par:31) on its own mesh (80 elements with midside-node edges, coordinates to
    float32).  That case was run with Nek5000's explicit filter (`filtering = explicit`, weight 0.01, bfs.par:16-18), which acts on
    the velocity AFTER the pressure projection of every step: the stored field is a fixed point of "step, then filter", so neither
    the discrete divergence nor the steady residual vanish to solver tolerance -- they are filter-sized.  Measured with the oracle's
    operators: divergence L2 1.8e-7 / max 2.8e-6 (collocated divergence: 7.5e-3 / 26), momentum residual rms 1.1e-6 / max 1.9e-5
    where the terms are O(1).  The bounds below are those measurements with a factor ~3."""
    from refdata import load_bfs
    hm, ux, uy, p, re, lxd, interior = load_bfs()
    sem = SEM(hm, lxd=lxd)
    u = [ux.reshape(sem.shape1), uy.reshape(sem.shape1)]
    div = sem.opdiv(u) / sem.bm2
    l2 = np.sqrt(np.sum(div ** 2 * sem.bm2) / sem.volvm2)
    g = [sem.gradm1(a) for a in u]
    col = g[0][0] + g[1][1]
    l2col = np.sqrt(np.sum(col ** 2 * sem.bm1) / sem.volvm1)
    assert l2 < 6e-7 and np.abs(div).max() < 1e-5, (l2, np.abs(div).max())
    assert l2col > 1e4 * l2 and np.abs(col).max() > 1e6 * np.abs(div).max()
    p2 = sem.to_mesh2(p.reshape(sem.shape1))
    N = sem.lns_conv_weak(u, u)
    gp = sem.opgradt(p2)
    inter = interior.reshape(sem.shape1)
    for i in range(2):
        r = sem.gs(0.5 * N[i] + (1.0 / re) * sem.axhelm_local(u[i], 1.0, 0.0) - gp[i]) * sem.binvm1
        scale = np.abs(sem.gs(0.5 * N[i]) * sem.binvm1)[inter].max()
        assert scale > 0.5
        assert np.sqrt(np.mean(r[inter] ** 2)) < 4e-6 * scale and np.abs(r[inter]).max() < 6e-5 * scale
    # the check discriminates: at Re = 500 instead of 600 the residual is three orders of magnitude larger
    r = sem.gs(0.5 * N[0] + (1.0 / 500.0) * sem.axhelm_local(u[0], 1.0, 0.0) - gp[0]) * sem.binvm1
    assert np.sqrt(np.mean(r[inter] ** 2)) > 1e-4


def test_rayben_field_file_conduction_state_and_hydrostatic_balance():
    """examples/rayBen/baseflow/BF_rayBen0.f00001 (fields X U P T, 10 x 4 box, lx1 = 10).  What the file holds: the conduction
    profile T = 2 (1 - y) to 1.5e-6, velocities of 8e-5 (solver noise; not an eigenmode of the coupled operator: Rayleigh-quotient
    residual 0.9), and a pressure that is NOT the hydrostatic pressure of that temperature -- its weak gradient has no correlation
    with the buoyancy bm1 T (3e-3), it changes sign twice across the layer.  (The case's userchk is commented out and names types
    that do not exist, rayBen.usr:29-71; the file is a by-product of a Newton run, whose pressure component is a Krylov
    combination.)  The balance D^T p = buoy theta can therefore not be read off the file.  Pinned instead:
      (a) the file's T is discretely harmonic under the oracle's scalar Helmholtz operator at lx1 = 10: the assembled weak
          Laplacian is 1.1e-7 where the element-local fluxes it sums are 6.6e-2 (float32 coordinates);
      (b) with the file's T, the case's buoyancy ffy = Ra Pr T (rayBen.usr:98, userParam05/06 of rayBen.par) and the analytic
          hydrostatic pressure p = Ra Pr (2 y - y^2) + const on the Gauss mesh, the oracle's temperature-coupled NONLINEAR step
          stays at rest and keeps that pressure; with the opposite sign of the buoyancy it does not: sign and scale of the buoyancy
          term against the weak gradient D^T p of the `ifheat` operators, on the reference's mesh and temperature field."""
    from oracle.lns import ExptA, LNSConfig
    from oracle.vectors import NekDVector
    from refdata import load_rayben
    hm, ux, uy, p, t, lxd, pr, ra = load_rayben()
    sem = SEM(hm, lxd=lxd)
    T = t.reshape(sem.shape1)
    Y = sem.X[1]
    assert np.abs(T - 2.0 * (1.0 - Y)).max() < 5e-6
    loc = sem.axhelm_local(T, 1.0, 0.0)                     # element-local fluxes: O(6e-2); assembled they cancel
    assert np.abs(sem.tmask * sem.gs(loc)).max() < 1e-5 * np.abs(loc).max(), (np.abs(sem.tmask * sem.gs(loc)).max(), np.abs(loc).max())
    # the file's pressure is not hydrostatic (documented above)
    gp = sem.opgradt(sem.to_mesh2(p.reshape(sem.shape1)))
    a, b = sem.mask[1] * sem.gs(gp[1]), sem.mask[1] * sem.gs(sem.bm1 * T)
    assert abs(np.sum(a * b)) < 0.05 * np.sqrt(np.sum(a * a) * np.sum(b * b))
    # (b) the exact hydrostatic pressure on the Gauss mesh as the state's pressure: at rest and staying there
    y2 = sem.to_mesh2(Y)
    ph = ra * pr * (2.0 * y2 - y2 ** 2)
    ph = ph - np.sum(ph * sem.bm2) / sem.volvm2
    out = {}
    for sgn in (1.0, -1.0):
        cfg = LNSConfig(re=1.0 / pr, torder=1, tau=2e-3, dt=1e-3, vtol=1e-13, ptol=1e-13, maxit_v=2000, maxit_p=6000, ifheat=True,
                        conductivity=1.0, rhocp=1.0, buoy=(0.0, sgn * ra * pr, 0.0))
        X = NekDVector(sem, 1)
        X.theta[0][...] = T
        X.pr[...] = ph
        A = ExptA(sem, X.v, cfg, X.theta[0])
        F = A.nonlinear_map(X)
        ps = A.p - np.sum(A.p * sem.bm2) / sem.volvm2
        out[sgn] = (max(np.abs(F.v[i]).max() for i in range(2)), np.abs(ps - ph).max() / np.abs(ph).max(), np.abs(F.theta[0]).max())
    # measured: 2.9e-6 (the 1.5e-6 noise of the file's T times Ra Pr dt) / 4.5e-7 / 6e-8; with the buoyancy sign flipped 1.0e-2 / 1.9
    assert out[1.0][0] < 2e-5 and out[1.0][1] < 5e-6 and out[1.0][2] < 1e-6, out
    assert out[-1.0][0] > 100 * out[1.0][0] and out[-1.0][1] > 1.0, out


def test_re40_newton_guess_fixture():
    """examples/cylinder/newton/Re40_fixed_point/BF.fld (the initial guess of the reference's Newton-Krylov example: a DNS snapshot
    at Re = 40, fp32) on the mesh of the stability case: the fixture carries the fields in the element order of
    reference_cyl_baseflow.npz and the numbers read off the reference's convergence plot.  With the oracle's operators: boundary
    values of the case (inflow u = (1, 0), no slip on the cylinder), discretely solenoidal to single precision.  The run itself
    (Newton residuals 9.0e-3, 1.33e-4, 1.3e-6 reproduced) is the GPU test test_gpu_newton_re40_against_the_reference_convergence_plot."""
    from refdata import load_cylinder_re40_guess
    hm, ux50, _, _, _, lxd, _ = load_cylinder(with_bcs=True)
    g = load_cylinder_re40_guess()
    assert g["ux"].shape == ux50.shape and float(g["re"]) == 40.0 and float(g["tau"]) == 1.0
    assert len(g["plot_gmres_step1"]) == 21 and list(g["plot_gmres_inner_steps"]) == [20, 18, 2]
    # the digitised figure (tests/golden/digitize_reference_plots.py): its own error bars and internal consistency -- a GMRES residual
    # history decreases monotonically, and the "init" residual of Newton step k is the Newton residual of step k read on the other axes
    assert float(g["plot_rel_err"]) < 0.01 and np.max(np.abs(g["plot_cross_check"])) < 0.015
    for key in ("plot_gmres_step1", "plot_gmres_step2", "plot_gmres_step3"):
        assert np.all(np.diff(g[key]) < 0.0), key
    assert abs(g["plot_gmres_step1"][0] / g["plot_newton_residuals"][0] - 1.0) < 0.015
    assert abs(g["plot_gmres_step1"][-1] / 1e-6 - 1.0) < 0.05 and abs(g["plot_gmres_step2"][-1] / 1e-6 - 1.0) < 0.05   # both stop at the tolerance 1e-6
    sem = SEM(hm, lxd=lxd)
    u = [g["ux"].reshape(sem.shape1), g["uy"].reshape(sem.shape1)]
    x, y = hm.x.reshape(sem.shape1), hm.y.reshape(sem.shape1)
    inflow = np.abs(x + 16.0) < 1e-6
    wall = np.abs(np.hypot(x, y) - 0.5) < 1e-5
    assert inflow.sum() > 50 and wall.sum() > 50
    assert np.abs(u[0][inflow] - 1.0).max() < 1e-6 and np.abs(u[1][inflow]).max() < 1e-6
    assert np.abs(u[0][wall]).max() < 1e-6 and np.abs(u[1][wall]).max() < 1e-6
    div = sem.opdiv(u) / sem.bm2
    l2 = np.sqrt(np.sum(div ** 2 * sem.bm2) / sem.volvm2)
    gcol = sem.gradm1(u[0])[0] + sem.gradm1(u[1])[1]
    l2col = np.sqrt(np.sum(gcol ** 2 * sem.bm1) / sem.volvm1)
    assert l2 < 1e-6 and l2col > 100 * l2, (l2, l2col)     # fp32 field: divergence at single-precision level


def test_tsyphon_mesh_fixture():
    """examples/thermosyphon/baseflow/tsyphon.re2 (genbox + circular-arc sides): the annulus 1 <= r <= 2 with 8 x 32 elements, lx1 = 8 --
    closed ring, walls at r = 1, 2.  57 x 224 distinct points; every element edge on a wall is an arc of that circle."""
    from refdata import load_tsyphon
    hm, d = load_tsyphon()
    assert hm.x.shape == (256, 64) and len(np.unique(hm.glo_num)) == 57 * 224
    r = np.hypot(hm.x, hm.y)
    assert abs(r.min() - 1.0) < 1e-12 and abs(r.max() - 2.0) < 1e-12
    assert int((hm.mask[0] == 0).sum()) == 64 * 8 and np.array_equal(hm.mask[0], hm.tmask)
    sem = SEM(hm, lxd=int(d["lxd"]))
    assert abs(np.sum(sem.bm1) - np.pi * 3.0) < 1e-9          # area of the annulus from the curved-element mass matrix
