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
ev, res, vecs, mu, nmv = host.linear_stability_analysis_fixed_point(A, int(sys.argv[1]), 4, tol=1e-6, outdir="gpurun_out", seed=1)
print('nmv', nmv); 
for m_, r in zip(mu, res): print(m_, abs(m_), r)
# residual check of the leading pair by hand: A v - Re(mu) v + Im(mu) w
v, w = vecs[0], vecs[1]
Av = host.nek_dvector(gm); A.matvec(v, Av)
r = Av.copy(); r.axpby(-mu[0].real, v, 1.0); r.axpby(mu[0].imag, w, 1.0)
print('true residual |A v - (a v - b w)| / |v| =', r.norm() / v.norm())
pv = v.copy(); A.proj(pv); d = pv.copy(); d.sub(v); print('|Pv - v|/|v|', d.norm() / v.norm())
