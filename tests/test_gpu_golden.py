"""GPU parity against the committed golden vectors (tests/golden/*.npz, made by make_golden.py)."""
import os
import sys

import numpy as np
import pytest

from neklab_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("case", ["2d", "3d"])
def test_golden_operators_and_matvec(gpu_ctx, case):
    g = np.load(os.path.join(HERE, "golden", "golden_%s.npz" % case))
    c = mg.CASES[case]
    from neklab_amd.mesh import box_mesh
    hm = box_mesh(c["nel"], c["n"], lengths=c["lengths"], periodic=c["periodic"], deform=c["deform"])
    gm = host.Mesh(gpu_ctx, hm)
    dim = hm.dim
    lib = gm.lib
    assert rel(gm.get("bm1"), g["bm1"]) < 1e-13 and rel(gm.get("binvm1"), g["binvm1"]) < 1e-13
    assert rel(gm.get("ediag", 2), g["ediag"]) < 1e-12 and rel(gm.get("hdiag:0.02:30.0"), g["hdiag"]) < 1e-12
    vin, w, out = host.nek_dvector(gm), host.nek_dvector(gm), host.nek_dvector(gm)
    for i in range(dim):
        vin.set_field(i, g["in_u"][i])
        w.set_field(i, g["in_w"][i])
    vin.set_field(host.PR, g["in_p"])
    host.check(lib.nlg_op_helmholtz(gm.h, vin.h, out.h, 0.7, 3.0, 0))
    assert max(rel(out.get_field(i), g["axhelm"][i]) for i in range(dim)) < 1e-13
    tmp = vin.copy()
    host.check(lib.nlg_op_dssum(gm.h, tmp.h))
    assert max(rel(tmp.get_field(i), g["gs"][i]) for i in range(dim)) < 1e-14
    host.check(lib.nlg_op_opdiv(gm.h, vin.h, out.h))
    assert rel(out.get_field(host.PR), g["opdiv"]) < 1e-13
    host.check(lib.nlg_op_opgradt(gm.h, vin.h, out.h))
    assert max(rel(out.get_field(i), g["opgradt"][i]) for i in range(dim)) < 1e-13
    host.check(lib.nlg_op_cdabdtp(gm.h, vin.h, out.h))
    assert rel(out.get_field(host.PR), g["cdabdtp"]) < 1e-12
    for adj, key in ((0, "conv_dir"), (1, "conv_adj")):
        host.check(lib.nlg_op_conv(gm.h, w.h, vin.h, out.h, adj))
        sc = np.abs(g[key]).max()
        assert max(np.abs(out.get_field(i) - g[key][i].ravel()).max() for i in range(dim)) < 1e-12 * sc
    # exptA
    bf = host.nek_dvector(gm)
    for i in range(dim):
        bf.set_field(i, g["baseflow"][i])
    cfg = mg.lns_cfg()
    tau = cfg.pop("tau")
    A = host.exptA_linop(tau, bf, **cfg)
    A.init()
    x, y, y2, z = (host.nek_dvector(gm) for _ in range(4))
    for i in range(dim):
        x.set_field(i, g["mv_in_v"][i])
    A.matvec(x, y)
    sc = np.abs(g["mv_out_v"]).max()
    assert max(np.abs(y.get_field(i) - g["mv_out_v"][i].ravel()).max() for i in range(dim)) < 1e-10 * sc
    assert max(np.abs(y.get_field(i, 2) - g["mv_out_rst2_v"][i].ravel()).max() for i in range(dim)) < 1e-10 * sc
    A.matvec(y, y2)
    assert max(np.abs(y2.get_field(i) - g["mv2_out_v"][i].ravel()).max() for i in range(dim)) < 1e-9 * sc
    A.rmatvec(x, z)
    assert max(np.abs(z.get_field(i) - g["rmv_out_v"][i].ravel()).max() for i in range(dim)) < 1e-10 * sc
    # eigs: Ritz values to 1e-10 relative, Ritz vectors to 1e-6 (BASELINE.json north_star)
    if case != "2d":
        return
    cfg = mg.lns_cfg()
    cfg.update(dt=0.025, re=10.0)
    cfg.pop("tau")
    A2 = host.exptA_linop(1.0, bf, **cfg)
    A2.init()
    X = [host.nek_dvector(gm) for _ in range(2)]
    mu, res, info = host.eigs(A2, X, kdim=12, tol=1e-9, x0=x, write_intermediate=False, max_restarts=6)
    assert info == int(g["eigs_nmv"])
    assert np.max(np.abs(mu - g["eigs_lam"]) / np.abs(g["eigs_lam"])) < 1e-10
    assert np.all(res < 1e-9)
    # leading eigenvalue is real: eigenvector defined up to sign
    a = np.concatenate([X[0].get_field(i) for i in range(dim)])
    b = g["eigs_vec0"].reshape(dim, -1).ravel()
    s = np.sign(a @ b)
    assert np.max(np.abs(a - s * b)) < 1e-6 * np.abs(b).max()
    # continuous-time eigenvalue as the driver reports it (neklab_analysis.f90:84)
    assert abs(np.log(mu[0]) / 1.0 - np.log(g["eigs_lam"][0])) < 1e-9
