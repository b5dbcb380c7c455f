"""The Fortran 2008 shim (neklab_amd/fortran) driven like LightKrylov would drive it, on the GPU."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from neklab_amd import host
from neklab_amd.mesh import box_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "neklab_amd", "fortran")

pytestmark = pytest.mark.gpu


def test_fortran_shim_arnoldi_matches_c_abi(gpu_ctx):
    subprocess.run(["make", "-s", "-C", FDIR], check=True)
    exe = os.path.join(FDIR, "_build", "arnoldi_driver")
    hm = box_mesh((3, 2), 6, lengths=(3.0, 2.0), periodic=(True, False), deform=0.03)
    kdim, tau, re, dt = 3, 0.03, 30.0, 0.01
    bf = [hm.mask[0] * np.cos(hm.y), 0.2 * hm.mask[1] * np.sin(hm.x)]
    gm = host.Mesh(gpu_ctx, hm)
    x0 = host.nek_dvector(gm)
    x0.rand(True, seed=4)
    v0 = [x0.get_field(i) for i in range(2)]
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "mesh.bin"), "wb") as f:
        np.array([2, 6, hm.E, kdim], dtype=np.int32).tofile(f)
        np.array([tau, re, dt], dtype=np.float64).tofile(f)
        for a in (hm.x, hm.y):
            a.astype(np.float64).tofile(f)
        hm.glo_num.astype(np.int64).tofile(f)
        for a in (hm.mask[0], hm.mask[1], bf[0], bf[1], v0[0], v0[1]):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)
    r = subprocess.run([exe], cwd=tmp, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    H = np.zeros((kdim + 1, kdim))
    alias = size = None
    for ln in r.stdout.splitlines():
        p = ln.split()
        if p and p[0] == "H":
            H[int(p[1]) - 1, int(p[2]) - 1] = float(p[3])
        elif p and p[0] == "ALIAS":
            alias = float(p[1])
        elif p and p[0] == "SIZE":
            size = int(p[1])
    # same computation through the C ABI's block path
    gb = host.nek_dvector(gm)
    for i in range(2):
        gb.set_field(i, bf[i])
    A = host.exptA_linop(tau, gb, re=re, dt=dt, torder=3, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=4000)
    A.init()
    B = host.KrylovBasis(gm, kdim + 1)
    B[0].assign(x0)
    Href = np.zeros((kdim + 1, kdim), order="F")
    for k in range(kdim):
        host.arnoldi_step(A, B, k, Href)
    assert np.max(np.abs(H - Href)) < 1e-9 * np.max(np.abs(Href))
    assert abs(alias - 1.0) < 1e-12          # wrk = X(1); wrk%scal(2) must not touch X(1)
    assert size == x0.get_size()
