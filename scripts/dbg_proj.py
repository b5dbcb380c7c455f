import sys; sys.path.insert(0, '.')
import numpy as np
from neklab_amd import host
from neklab_amd.mesh import box_mesh
ctx = host.Context(0)
hm = box_mesh((10, 12), 8, lengths=(2 * np.pi, 2.0), periodic=(True, False), deform=0.0, origin=(0.0, -1.0))
gm = host.Mesh(ctx, hm)
bf = host.nek_dvector(gm); bf.set_field(0, 1.0 - hm.y ** 2)
A = host.exptA_proj_linop(1.0, bf, 2.0, idir=1, re=7500.0, torder=3, vtol=1e-11, ptol=1e-10, maxit_p=4000)
A.init()
v = host.nek_dvector(gm); v.rand(True, seed=1)
w = host.nek_dvector(gm)
for it in range(12):
    A.matvec(v, w)
    nrm = w.norm()
    # wavenumber content of w: project again and compare
    p = w.copy(); A.proj(p); d = p.copy(); d.sub(w)
    print(it, 'growth', nrm, ' |Pw - w|/|w| = %.2e' % (d.norm() / nrm), 'nrst', w.nrst if hasattr(w, 'nrst') else None, flush=True)
    w.scal(1.0 / nrm)
    v, w = w, v
