"""The reference's Newton-Krylov example on the GPU path: examples/cylinder/newton/Re40_fixed_point (1cyl.usr: load BF.fld,
newton_fixed_point_iteration(sys, bf, tol = 1e-6); 1cyl.par: Re = 40, endTime = 1.0, bdf3), against the convergence history the
reference publishes for that very run (residual.png next to the case: Newton residuals 9.0e-3, 1.33e-4, 1.3e-6; GMRES inner steps
20, 18, 2 at a constant tolerance of 1e-6) -- tests/golden/reference_cyl_re40_guess.npz holds the guess and the numbers read off the plot.

    python scripts/cylinder_newton_re40.py > profiles/r03_cylinder_newton_re40.txt
"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from neklab_amd import host
from refdata import load_cylinder, load_cylinder_re40_guess

hm, _, _, _, _, lxd, _ = load_cylinder(with_bcs=True)
g = load_cylinder_re40_guess()
ctx = host.Context(); gm = host.Mesh(ctx, hm, lxd=lxd)
X = host.nek_dvector(gm); X.set_field(host.VX, g["ux"]); X.set_field(host.VY, g["uy"]); X.set_field(host.PR, host.pressure_from_mesh1(gm, g['p']))   # load_fld reads the pressure too (XUP file)
tau, tol = float(g["tau"]), float(g["newton_tol"])
sysm = host.nek_system(tau, X, re=float(g["re"]), maxit_v=400, maxit_p=4000)
for a in sys.argv[1:]:
    if a.startswith("jactol="):      # Jacobian solver tolerance as a multiple of the scheduler's (this build 0.5; the reference's param(22) chain gives 0.05)
        f = float(a.split("=")[1])
        sysm.set_tolerance = lambda t, f=f: (host.check(ctx.lib.nlg_linop_set_tolerances(sysm.nl.h, 0.1 * t, 0.1 * t)), host.check(ctx.lib.nlg_linop_set_tolerances(sysm.jac.h, f * t, f * t)))
        print("Jacobian tolerance = %g x scheduler tolerance" % f)
    if a == "jaccfl=0.4":            # Jacobian products on the time step of the nonlinear map
        sysm.jac = host.exptA_linop(tau, X, re=float(g["re"]), torder=3, cfl_limit=0.4, maxit_v=400, maxit_p=4000); sysm.jac.init()
        print("Jacobian products with cfl_limit 0.4 (the nonlinear map's time step)")
print("mesh: E = %d, lx1 = %d, lxd = %d; Re = %g, tau = %g, Newton tolerance %.0e (constant solver-tolerance scheduler)" % (hm.x.shape[0], hm.n, lxd, float(g["re"]), tau, tol))
t0 = time.time()
replay = len(sys.argv) > 1 and sys.argv[1].startswith("replay")
if len(sys.argv) > 1 and sys.argv[1] == "replay-literal":     # ... with the literal reading of the history update in axpby (include/neklab_gpu.h)
    host.check(ctx.lib.nlg_set_axpby_rst_consistent(0))
    print("axpby: literal history update")   # jac_exptA_matvec replays the restart history of its argument (fixed_point.f90:73)
print("GMRES: Krylov vectors %s their restart history" % ("replay" if replay else "are stripped of"))
out = host.newton_fixed_point_iteration(sysm, X, tol, tol_mode=1, kdim=30, log=lambda s: print(s, flush=True), replay_history=replay)
print("converged %s after %d Newton iterations, %d GMRES matvecs, %d evaluations of the nonlinear map, %.1f s; time steps per map: %d (dt = %.5f)"
      % (out["converged"], out["iterations"], out["gmres_matvecs"], out["evals"], time.time() - t0, sysm.nl.info()["nsteps"], sysm.nl.info()["dt"]))
ref_n, ref_k = g["plot_newton_residuals"], g["plot_gmres_inner_steps"]
print("reference numbers: digitised from residual.png (tests/golden/digitize_reference_plots.py), one-sigma %.2f %%" % (100 * float(g["plot_rel_err"])))
print("\nNewton residual at the start of each step        this run        reference (read off residual.png)")
for i, r in enumerate(out["residuals"]):
    print("  step %d   %.4e    %s" % (i + 1, r, "%.4e  %+.2f %%" % (ref_n[i], 100 * (r / ref_n[i] - 1)) if i < len(ref_n) else "-"))
print("GMRES inner steps per Newton step:  this run %s   reference %s" % ([len(h) - 1 for h in out["gmres_residuals"]], list(ref_k)))
for s, key in ((0, "plot_gmres_step1"), (1, "plot_gmres_step2"), (2, "plot_gmres_step3")):
    if s < len(out["gmres_residuals"]):
        print("\nGMRES residuals of Newton step %d (init, then after every inner step):   this run / reference" % (s + 1))
        h, ref = out["gmres_residuals"][s], g[key]
        for k in range(max(len(h), len(ref))):
            print("  %2d   %s   %s   %s" % (k, "%.4e" % h[k] if k < len(h) else "    -     ", "%.4e" % ref[k] if k < len(ref) else "-",
                                       "%+.2f %%" % (100 * (h[k] / ref[k] - 1)) if k < min(len(h), len(ref)) else ""))
