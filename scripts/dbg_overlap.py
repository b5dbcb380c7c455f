"""GPU overlapping local solves vs the numpy prototype (scripts/precond_proto4.py) on the same small mesh."""
import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import numpy as np
from neklab_amd import host
from neklab_amd.mesh import box_mesh
from oracle.sem import SEM

nel, n = (4, 4, 4), 8
hm = box_mesh(nel, n, deform=0.05)
ctx = host.Context(); gm = host.Mesh(ctx, hm); lib = ctx.lib
rng = np.random.default_rng(1)
sem = SEM(hm)
r = rng.standard_normal(sem.shape2)
vin = host.nek_dvector(gm); vout = host.nek_dvector(gm)
vin.set_field(host.PR, r.ravel())
res = {}
for ov in (0, 1):
    host.check(lib.nlg_op_pprec(gm.h, vin.h, vout.h, ov, 0))
    res[ov] = vout.get_field(host.PR).reshape(sem.shape2).copy()
np.save('gpurun_out/dbg_overlap_gpu.npy', np.stack([r, res[0], res[1]]))
print('gpu |z0|', np.linalg.norm(res[0]), '|z1|', np.linalg.norm(res[1]), '|z1-z0|', np.linalg.norm(res[1] - res[0]))
a = rng.standard_normal(sem.shape2)
vin.set_field(host.PR, a.ravel()); host.check(lib.nlg_op_pprec(gm.h, vin.h, vout.h, 1, 0)); Ma = vout.get_field(host.PR).reshape(sem.shape2).copy()
print('symmetry', abs(np.sum(r * Ma) - np.sum(a * res[1])) / abs(np.sum(r * Ma)))
