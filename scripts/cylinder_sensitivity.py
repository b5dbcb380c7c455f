"""Which ingredient carries the gap to the reference's only published number?

/root/reference/test/neklabTests.py:43-45 asserts |mu_1| = 1.0156 +- 1e-4 (cylinder, Re = 50, tau = 1, lx1 = 6, lxd = 9, bdf3,
kdim = 128, nev = 2).  The default GPU path gives 1.015780, independent of dt and of the solver tolerances (round 1).  This
script varies ONE ingredient at a time -- the restart-history protocol of the Krylov vectors, lxd, the outflow boundary,
the tolerances, the time order -- with a host-side Arnoldi loop over the device Arnoldi step so that the protocol between
matvecs can be changed, and prints |mu_1| at the first iteration whose residual passes `tol` (what LightKrylov flags "T")
and at the end.  Output: profiles/r02_cylinder_sensitivity.txt.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from neklab_amd import host  # noqa: E402
from refdata import load_cylinder  # noqa: E402

ctx = host.Context(0)
KDIM = 128


def leading(H, k):
    lam, Y = np.linalg.eig(H[:k, :k])
    res = np.abs(H[k, k - 1] * Y[k - 1, :])
    o = np.argsort(-np.abs(lam))
    return lam[o], res[o]


THRESHOLDS = (1e-6, 1e-7, 1.5e-8, 1e-9, 1e-10)


def run(name, history="consistent", lxd=None, tags=("v", "W"), tol=1e-6, zero_pr=False, seed=1, deep=False, geometry="fld", **kw):
    hm, ux, uy, p, re, lxd0, _ = load_cylinder(with_bcs=True, dirichlet_tags=tags, geometry=geometry)
    gm = host.Mesh(ctx, hm, lxd=lxd if lxd else lxd0)
    bf = host.nek_dvector(gm)
    bf.set_field(0, ux)
    bf.set_field(1, uy)
    host.check(gm.lib.nlg_set_axpby_rst_consistent(0 if history == "literal" else 1))
    cfg = dict(re=re, torder=3, vtol=1e-9, ptol=1e-7, maxit_v=400, maxit_p=4000)
    cfg.update(kw)
    A = host.exptA_linop(1.0, bf, **cfg)
    A.init()
    B = host.KrylovBasis(gm, KDIM + 1)
    B[0].rand(True, seed=seed)
    if history == "warm":
        # start from the IMAGE of the random vector: it carries a restart history like every later Krylov vector, so all
        # columns of the Arnoldi relation come from one and the same linear map (see the header of the seeds table)
        w = host.nek_dvector(gm)
        A.matvec(B[0], w)
        w.scal(1.0 / w.norm())
        B[0].assign(w)
    H = np.zeros((KDIM + 2, KDIM + 1), order="F")
    t0 = time.time()
    first, last = None, None
    at = {}
    for k in range(KDIM):
        if zero_pr:
            B[k].set_field(host.PR, np.zeros(gm.lpn))
        host.arnoldi_step(A, B, k, H)
        if history == "none":
            B[k + 1].clear_rst_fields()          # every matvec starts impulsively (BDF1/EXT1, then 2, then 3)
        lam, res = leading(H, k + 1)
        last = (k + 1, lam[0], res[0])
        if res[0] < tol and first is None:
            first = last
        for th in THRESHOLDS:
            if res[0] < th and th not in at:
                at[th] = (k + 1, abs(lam[0]))
        if deep:
            if res[0] < 1e-11:
                break
        elif res[0] < 1e-9 or (first is not None and k + 1 >= first[0] + 12):
            break
    host.check(gm.lib.nlg_set_axpby_rst_consistent(1))
    st = A.stats()
    f = first or last
    line = "%-44s dt=%.5f  first<%.0e: k=%3d |mu|=%.6f  | end: k=%3d |mu|=%.6f arg=%.6f res=%.1e  p_it/step=%.1f  %.0fs" % (
        name, A.info()["dt"], tol, f[0], abs(f[1]), last[0], abs(last[1]), abs(np.angle(last[1])), last[2],
        st["p_iters"] / st["steps"], time.time() - t0)
    if deep:      # |mu_1| at the iteration where the residual first passes each threshold (1.5e-8 = LightKrylov's rtol_dp)
        line = "%-30s " % name + "  ".join("res<%.1e: k=%3d %.6f" % (th, at[th][0], at[th][1]) if th in at else "res<%.1e: --" % th
                                            for th in THRESHOLDS) + "  | end k=%d res=%.1e |mu|=%.7f" % (last[0], last[2], abs(last[1]))
    print(line, flush=True)
    return line


if __name__ == "__main__":
    which = sys.argv[1:] or ["all"]
    V = [
        ("default: consistent history", dict()),
        ("no history replay (impulsive start)", dict(history="none")),
        ("literal real_vectors.f90:188-192", dict(history="literal")),
        ("no history, cfl 0.25", dict(history="none", cfl_limit=0.25)),
        ("lxd = 8", dict(lxd=8)),
        ("lxd = 12", dict(lxd=12)),
        ("lxd = 6 (no dealiasing)", dict(lxd=6)),
        ("outflow 'O' -> Dirichlet 'v'", dict(tags=("v", "W", "O"))),
        ("loose tolerances 1e-7 / 1e-5", dict(vtol=1e-7, ptol=1e-5)),
        ("tight tolerances 1e-11 / 1e-10", dict(vtol=1e-11, ptol=1e-10)),
        ("pressure of the Krylov vector zeroed", dict(zero_pr=True)),
        ("no history, pressure zeroed", dict(history="none", zero_pr=True)),
        ("torder 2, consistent", dict(torder=2)),
        ("torder 2, no history", dict(torder=2, history="none")),
        ("other start vector (seed 7)", dict(seed=7)),
        ("no pressure projection", dict(pproj=0)),
    ]
    lines = ["reference: |mu_1| = 1.0156 +- 1e-4  (test/neklabTests.py:43-45)"]
    print(lines[0], flush=True)
    if which == ["seeds"]:      # scatter over the start vector (the reference draws it with the compiler's random_number)
        for sd in (1, 2, 3, 5, 7, 11, 13, 17):
            lines.append(run("start vector seed %d" % sd, seed=sd, deep=True))
    if which == ["geometry"]:   # float32-quantised coordinates of the field file vs the double-precision rebuild from 1cyl.re2
        for geo in ("fld", "re2"):
            for hist in ("warm", "none"):
                lines.append(run("geometry %s, %s start" % (geo, hist), history=hist, geometry=geo, deep=True))
        open(os.path.join(ROOT, "gpurun_out", "r02_cylinder_geometry.txt"), "w").write("\n".join(lines) + "\n")
        sys.exit(0)
    if which == ["protocol"]:   # is the scatter the missing history of the start vector?
        for hist in ("warm", "none", "literal"):
            for sd in (1, 2, 11):
                lines.append(run("%s, seed %d" % (hist, sd), history=hist, seed=sd, deep=True))
        open(os.path.join(ROOT, "gpurun_out", "r02_cylinder_protocol.txt"), "w").write("\n".join(lines) + "\n")
        sys.exit(0)
    if which == ["seeds"]:
        open(os.path.join(ROOT, "gpurun_out", "r02_cylinder_seeds.txt"), "w").write("\n".join(lines) + "\n")
        sys.exit(0)
    for i, (name, kw) in enumerate(V):
        if which != ["all"] and str(i) not in which:
            continue
        try:
            lines.append(run(name, **kw))
        except Exception as exc:      # a variant the kernels are not built for must not lose the others
            lines.append("%-44s FAILED: %r" % (name, exc))
            print(lines[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "r02_cylinder_sensitivity.txt"), "w").write("\n".join(lines) + "\n")
