"""Workload for the MFMA counters: the base-flow side of the convective term (sem_conv_setup -> k_interp4_mfma<8,12>) at
the benchmark size, E = 10 000, lx1 = 8: what the nonlinear map of the Newton-Krylov solver runs once per time step.
    rocprofv3 --kernel-trace --stats -d out -o mfma -- python3 scripts/mfma_profile.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out2 -- python3 scripts/mfma_profile.py
NLG_MFMA=0 selects the generic tensor kernel for the comparison."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neklab_amd import host  # noqa: E402
from neklab_amd.mesh import box_mesh  # noqa: E402

ctx = host.Context(0)
hm = box_mesh((25, 20, 20), 8, deform=0.05)
gm = host.Mesh(ctx, hm)
bf = host.nek_dvector(gm)
bf.rand(False, seed=1)
A = host.exptA_linop(0.01, bf, re=100.0, dt=0.005)
A.init()
for _ in range(int(os.environ.get("REPS", "6"))):
    host.check(gm.lib.nlg_linop_set_baseflow(A.h, bf.h))
ctx.sync()
print("done")
