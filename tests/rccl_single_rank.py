"""Run by tests/test_gpu_rccl.py with NLG_FORCE_COMM=1: a one-rank RCCL communicator exercises the
all-reduce / all-gather code paths of the library on a one-GPU box and must not change any result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neklab_amd import host  # noqa: E402
from neklab_amd.mesh import box_mesh  # noqa: E402

assert os.environ.get("NLG_FORCE_COMM") == "1"
ctx = host.Context(0)
ctx.comm_init(0, 1, host.Context.unique_id())
hm = box_mesh((3, 2, 2), 6, periodic=(True, False, False), deform=0.04)
gm = host.Mesh(ctx, hm)
v, w = host.nek_dvector(gm), host.nek_dvector(gm)
v.rand(True, seed=1)
w.rand(True, seed=2)
B = host.KrylovBasis(gm, 4)
B[0].assign(v)
B[1].assign(w)
h = B.block_dot(2, v)
bf = host.nek_dvector(gm)
bf.set_field(0, hm.mask[0] * np.cos(hm.y))
A = host.exptA_linop(0.02, bf, re=30.0, dt=0.01, vtol=1e-12, ptol=1e-12)
A.init()
out = host.nek_dvector(gm)
A.matvec(v, out)
print("RESULT %.17e %.17e %.17e %.17e" % (v.dot(w), h[0], h[1], out.norm()))
