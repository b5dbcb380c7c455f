import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from neklab_amd import host
from refdata import load_cylinder
hm, ux, uy, p, re, lxd, _ = load_cylinder(with_bcs=True)
ctx = host.Context(0); gm = host.Mesh(ctx, hm, lxd=lxd)
bf = host.nek_dvector(gm); bf.set_field(0, ux); bf.set_field(1, uy)
def run(name, consistent=0, **kw):
    host.check(gm.lib.nlg_set_axpby_rst_consistent(consistent))
    cfg = dict(re=re, torder=3, vtol=1e-9, ptol=1e-7, maxit_v=400, maxit_p=4000); cfg.update(kw)
    A = host.exptA_linop(1.0, bf, **cfg); A.init()
    t0 = time.time()
    ev, res, vecs, mu, nmv = host.linear_stability_analysis_fixed_point(A, 128, 2, tol=1e-6, outdir="gpurun_out", seed=1)
    print("%-28s dt=%.5f |mu|=%.6f sigma=%.6f%+.6fi res=%.1e nmv=%d t=%.0fs" % (name, A.info()["dt"], abs(mu[0]), ev[0].real, ev[0].imag, res[0], nmv, time.time()-t0), flush=True)
run("default (reference quirk)")
run("consistent rst", consistent=1)
run("cfl 0.25", cfl_limit=0.25)
run("cfl 0.25 consistent", consistent=1, cfl_limit=0.25)
run("tight tol", vtol=1e-12, ptol=1e-11)
run("torder 2", torder=2)
