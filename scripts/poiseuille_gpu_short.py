import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from neklab_amd import host
from neklab_amd.mesh import box_mesh
MU_OS = np.exp(-1j * (0.24989146 + 0.00223497j))
ctx = host.Context(0)
for nel, n, cfl in [((10, 12), 8, 0.5), ((12, 16), 10, 0.25)]:
    hm = box_mesh(nel, n, lengths=(2 * np.pi, 2.0), periodic=(True, False), deform=0.0, origin=(0.0, -1.0))
    gm = host.Mesh(ctx, hm)
    bf = host.nek_dvector(gm)
    bf.set_field(0, (1.0 - hm.y ** 2))
    A = host.exptA_linop(1.0, bf, re=7500.0, torder=3, vtol=1e-11, ptol=1e-10, cfl_limit=cfl, maxit_p=4000)
    A.init()
    t0 = time.time()
    ev, res, vecs, mu, nmv = host.linear_stability_analysis_fixed_point(A, 160, 2, tol=1e-7, outdir="/tmp", seed=1)
    m = mu[0] if mu[0].imag < 0 else np.conj(mu[0])
    print("E=%dx%d n=%d dt=%.4f  mu=%.7f%+.7fi |mu|=%.7f  |mu-mu_OS|=%.2e res=%.1e nmv=%d %.0fs stats %s"
          % (nel[0], nel[1], n, A.info()["dt"], m.real, m.imag, abs(m), abs(m - MU_OS), res[0], nmv, time.time() - t0, A.stats()), flush=True)
