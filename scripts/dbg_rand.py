import numpy as np, sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT)
from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM
from oracle.vectors import NekDVector, u01
ctx=host.Context(0)
hm = box_mesh((3,3,2),8,periodic=(True,False,False),deform=0.04)
sem,gm=SEM(hm),host.Mesh(ctx,hm)
gv,ov=host.nek_dvector(gm),NekDVector(sem)
host.check(gm.lib.nlg_vec_rand_noise(gv.h, 11))
got=gv.get_field(0)
n,dim,E=8,3,hm.E; npts=n**3
ijk=np.arange(npts); ix=(ijk%n)+1; iy=((ijk//n)%n)+1; iz=((ijk//(n*n))%n)+1
ieg=(hm.elem_gid+1)[:,None]
gpt=(ieg-1)*npts+ijk[None,:]
base=(gpt.astype(np.uint64)*np.uint64(8)+np.uint64(0))*np.uint64(4)
fc=[u01(11,base+np.uint64(c))*1e4 for c in range(3)]
xl=[sem.X[d].reshape(E,npts) for d in range(3)]
gx=[gm.get("x%d"%d) if False else None for d in range(3)]
r2=fc[0]*(ieg+xl[0]*np.sin(xl[1]))+fc[1]*ix*iy+fc[2]*ix
def fin(r):
    r=1e3*np.sin(r); r=1e3*np.sin(r); return np.cos(r)
cands={
 'oracle': fin(fc[0]*(ieg+xl[2]*np.sin(r2))+fc[1]*iz*ix+fc[2]*iz),
 '2d only': fin(r2),
 'z->y': fin(fc[0]*(ieg+xl[1]*np.sin(r2))+fc[1]*iz*ix+fc[2]*iz),
 'iz=1': fin(fc[0]*(ieg+xl[2]*np.sin(r2))+fc[1]*1*ix+fc[2]*1),
}
for k,v in cands.items():
    d=np.abs(got-v.ravel()); print(k, 'max',d.max(),'median',np.median(d),'frac<1e-6',np.mean(d<1e-6))

r3=fc[0]*(ieg+xl[2]*np.sin(r2))+fc[1]*iz*ix+fc[2]*iz
d=np.abs(got-cands['oracle'].ravel())
bad=np.nonzero(d>1e-5)[0]
print('bad points',len(bad),'of',d.size)
s2=np.sin(r2).ravel()
for i in bad[:12]: print(i, 'got %.6f want %.6f'%(got[i], cands['oracle'].ravel()[i]), 'r2 %.6f r3 %.6f sin(r2) %.17g'%(r2.ravel()[i], r3.ravel()[i], s2[i]), 'fc', [f.ravel()[i] for f in fc])
print('r2 range of bad', np.abs(r2.ravel()[bad]).min(), np.abs(r2.ravel()[bad]).max(), ' all', np.abs(r2).min(), np.abs(r2).max())
print('r3 range of bad', np.abs(r3.ravel()[bad]).min(), np.abs(r3.ravel()[bad]).max(), ' all', np.abs(r3).min(), np.abs(r3).max())
