"""CPU suite: the C-ABI library loads, exports every declared symbol, and fails loudly without a GPU."""
import ctypes as C
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from neklab_amd import build
    build.build_library()
    from neklab_amd import _lib
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "neklab_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nlg_[a-zA-Z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from neklab_amd import _lib
    names = declared_symbols()
    assert len(names) >= 60
    for nm in names:
        assert hasattr(lib, nm), "library lacks %s" % nm
        assert nm in _lib.SIGNATURES, "ctypes table lacks %s" % nm
    assert set(_lib.SIGNATURES) <= set(names)


def test_no_cpu_fallback_without_gpu(lib):
    # (do not import torch here: its bundled HIP runtime next to the system one the library links
    #  aborts at interpreter exit on a GPU-less machine)
    if os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK):
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.nlg_ctx_create(0, C.byref(h))
    assert rc != 0 and b"no HIP device" in lib.nlg_last_error()
    from neklab_amd import host
    with pytest.raises(host.NlgError):
        host.Context(0)


def test_dense_eig_host_helper(lib):
    from neklab_amd._lib import dptr
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 30, 64):
        A = np.asfortranarray(rng.standard_normal((n, n)))
        wr, wi, vr = np.zeros(n), np.zeros(n), np.zeros((n, n), order="F")
        assert lib.nlg_dense_eig(n, dptr(A), n, dptr(wr), dptr(wi), dptr(vr), n) == 0
        lam = wr + 1j * wi
        ref = np.linalg.eigvals(A)
        assert max(np.min(np.abs(ref - l)) for l in lam) < 1e-10 * max(1.0, np.abs(ref).max())
        j = 0
        while j < n:
            if wi[j] > 0:
                v = vr[:, j] + 1j * vr[:, j + 1]
                j += 2
            else:
                v = vr[:, j].astype(complex)
                j += 1
            k = j - 1 if wi[j - 1] == 0 else j - 2
            assert np.linalg.norm(A @ v - lam[k] * v) < 1e-9 * max(1.0, np.abs(ref).max())
            assert abs(np.linalg.norm(v) - 1.0) < 1e-12


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "neklab_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".f90")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt and f.endswith(".py") and "import" in txt and re.search(r"import.*oracle", txt):
                    bad.append(f)
    assert not bad, bad


def test_mesh_generator_properties():
    from neklab_amd.mesh import box_mesh, partition_elements
    m = box_mesh((3, 2, 2), 5, periodic=(True, False, False), deform=0.05)
    assert m.E == 12 and m.x.shape == (12, 125) and m.glo_num.min() == 0
    # periodic direction: first and last x-plane share labels
    nuniq = len(np.unique(m.glo_num))
    assert nuniq == (3 * 4) * (2 * 4 + 1) * (2 * 4 + 1)
    parts = partition_elements(m.E, 5)
    assert sum(len(p) for p in parts) == 12 and max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    # ragged / degenerate inputs
    with pytest.raises(AssertionError):
        box_mesh((3,), 5)
    # recursive coordinate bisection: a partition into sub-boxes; every part generated on its own equals its share of the full mesh
    from neklab_amd.mesh import rcb_boxes
    boxes = rcb_boxes((25, 20, 20), 8)
    sizes = [int(np.prod([b - a for a, b in bx])) for bx in boxes]
    assert sum(sizes) == 10000 and max(sizes) <= 1.04 * 1250 and sorted(set(sizes)) == [1200, 1300]
    full = box_mesh((5, 4, 3), 4, periodic=(True, False, False), deform=0.05)
    seen = []
    for bx in rcb_boxes((5, 4, 3), 3):
        part = box_mesh((5, 4, 3), 4, periodic=(True, False, False), deform=0.05, ranges=bx)
        g = part.elem_gid
        assert np.array_equal(part.x, full.x[g]) and np.array_equal(part.glo_num, full.glo_num[g]) and np.array_equal(part.mask[2], full.mask[2][g])
        seen += list(g)
    assert sorted(seen) == list(range(full.E))
    with pytest.raises(ValueError):
        rcb_boxes((2, 1), 3)


def test_fortran_shim_compiles_against_the_abstract_types():
    """amdflang builds the ISO_C_BINDING shim (neklab_amd/fortran) against the LightKrylov abstract-type stand-in, and the three
    driver programs (tests/fortran)."""
    import shutil
    fdir = os.path.join(ROOT, "neklab_amd", "fortran")
    tdir = os.path.join(ROOT, "tests", "fortran")
    if not (shutil.which("amdflang") or os.path.exists("/opt/rocm/bin/amdflang")):
        pytest.skip("no Fortran compiler")
    from neklab_amd import build
    build.build_library()
    r = subprocess.run(["make", "-s", "-C", tdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for exe in ("arnoldi_driver", "stability_driver", "tsyphon_driver", "resolvent_driver"):
        assert os.path.exists(os.path.join(tdir, "_build", exe))
    # nothing of the scaffolding lives in the product directory
    assert not os.path.exists(os.path.join(fdir, "lightkrylov_stub.f90")) and not glob.glob(os.path.join(fdir, "*driver*.f90"))
    # the modules carry the reference's names (src/neklab_analysis.f90:16-18 `use`s them unchanged)
    for mod in ("neklab_vectors", "neklab_linops", "neklab_utils", "neklab_systems", "neklab_analysis", "neklab"):
        assert re.search(r"^\s*module\s+%s\b" % mod, open(os.path.join(fdir, mod + ".f90")).read(), re.M | re.I), mod
    # every C symbol the shim binds is declared in the header
    txt = open(os.path.join(fdir, "neklab_gpu_capi.f90")).read()
    bound = set(re.findall(r'name="(nlg_[a-z0-9_A-Z]+)"', txt))
    assert len(bound) > 40 and bound <= set(declared_symbols()), bound - set(declared_symbols())
