"""The HIP kernels against the reference's own Nek5000-generated base flow (see test_cpu_reference_data.py)."""
import numpy as np
import pytest

from neklab_amd import host
from refdata import load_cylinder

pytestmark = pytest.mark.gpu


def test_gpu_operators_on_reference_base_flow(gpu_ctx):
    hm, ux, uy, p, re, lxd, interior = load_cylinder()
    gm = host.Mesh(gpu_ctx, hm, lxd=lxd)
    lib = gm.lib
    n, E = hm.n, hm.E
    bm2 = gm.get("bm2", 2)
    binv = gm.get("binvm1")
    U = host.nek_dvector(gm)
    U.set_field(host.VX, ux)
    U.set_field(host.VY, uy)
    out = host.nek_dvector(gm)
    # discrete divergence of the reference solution: ~1e-11
    host.check(lib.nlg_op_opdiv(gm.h, U.h, out.h))
    div = out.get_field(host.PR) / bm2
    l2 = np.sqrt(np.sum(div ** 2 * bm2) / np.sum(bm2))
    assert l2 < 1e-10 and np.abs(div).max() < 1e-8, (l2, np.abs(div).max())
    # steady momentum residual through the GPU operators
    from oracle.sem import gl, gll, interp_matrix
    I12 = interp_matrix(gll(n)[0], gl(n - 2)[0])
    p2 = np.einsum("by,ax,eyx->eba", I12, I12, p.reshape(E, n, n))
    U.set_field(host.PR, p2)
    conv, hel, gpt = host.nek_dvector(gm), host.nek_dvector(gm), host.nek_dvector(gm)
    host.check(lib.nlg_op_conv(gm.h, U.h, U.h, conv.h, 0))              # 2 (U.grad) U, weak, dealiased
    host.check(lib.nlg_op_helmholtz(gm.h, U.h, hel.h, 1.0 / re, 0.0, 0))
    host.check(lib.nlg_op_opgradt(gm.h, U.h, gpt.h))
    res = host.nek_dvector(gm)
    for i in range(2):
        res.set_field(i, 0.5 * conv.get_field(i) + hel.get_field(i) - gpt.get_field(i))
    host.check(lib.nlg_op_dssum(gm.h, res.h))
    inter = interior.ravel()
    for i in range(2):
        r = res.get_field(i) * binv
        assert np.sqrt(np.mean(r[inter] ** 2)) < 1e-6 and np.abs(r[inter]).max() < 1e-5
