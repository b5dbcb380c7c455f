import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, pytest
import test_gpu_ops as T
from neklab_amd import host
ctx = host.Context(0)
for n in (9, 10, 12):
    case = dict(nel=(2, 2, 2), n=n, periodic=(False, False, True))
    T.test_operators.__wrapped__(ctx, case) if hasattr(T.test_operators, '__wrapped__') else T.test_operators(ctx, case)
    print('operators ok n =', n, flush=True)
import test_gpu_linop as L
hm, sem, gm, oA, gA, rng = L.setup_case(ctx, 3, n=10, fixed=False)
ov, gv = L.load_pair(sem, gm, rng)
out = host.nek_dvector(gm); gA.matvec(gv, out); oo = oA.matvec(ov)
sc = max(np.abs(a).max() for a in oo.v)
print('matvec n=10 maxdiff', max(np.max(np.abs(out.get_field(i).reshape(sem.shape1) - oo.v[i])) for i in range(3)) / sc, gA.stats())
