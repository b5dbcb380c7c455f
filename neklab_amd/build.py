"""Build recipe: hipcc (gfx950 only) -> neklab_amd/libneklab_gpu.so, in-tree.

`python -m neklab_amd.build` or `build_library()`.  Objects are cached under neklab_amd/csrc/_obj and
rebuilt when the source or a header is newer.  There is no CPU build of the product.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libneklab_gpu.so")
SOURCES = ["ctx.hip", "shm_transport.hip", "vec.hip", "sem.hip", "halo.hip", "pprec.hip", "lns.hip", "krylov.hip", "dense_eig.cpp"]
HEADERS = [os.path.join(CSRC, "internal.h"), os.path.join(ROOT, "include", "neklab_gpu.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=on",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X build needs ROCm (no CPU fallback exists)")


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        stale = force or _newer(s, o) or any(_newer(h, o) for h in HEADERS)
        if stale:
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB) or any(_newer(o, LIB) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
            + ["-L/opt/rocm/lib", "-lrccl", "-lrocsolver", "-lrocblas", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
