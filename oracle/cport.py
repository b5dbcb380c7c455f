"""ORACLE-side loader of the C + OpenMP CPU port (oracle/c/sem_cpu.c).  Test / measurement infrastructure only:
tests check it against the numpy restatement, bench.py's cpu_baseline leg times it.  3-D meshes."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "c", "libsem_cpu.so")
        if not os.path.exists(path):
            return None
        _LIB = C.CDLL(path)
        _LIB.nl_threads.restype = C.c_int
        _LIB.nl_glsc3.restype = C.c_double
        _LIB.nl_cgvec.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pp(arrs):
    return (C.POINTER(C.c_double) * len(arrs))(*[_p(a) for a in arrs])


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class CPort:
    """The operators of oracle/sem.py on the arrays of one SEM object, computed by the C port."""

    def __init__(self, sem):
        lib = load()
        if lib is None:
            raise RuntimeError("oracle/c/libsem_cpu.so is not built (make -C oracle/c)")
        if sem.dim != 3:
            raise ValueError("the C port covers 3-D meshes")
        self.lib, self.sem = lib, sem
        s = sem
        self.E, self.n, self.n2, self.nd = s.E, s.n, s.n2, s.nd
        self.D, self.I12, self.D12, self.Jd, self.DJd = _c(s.D), _c(s.I12), _c(s.D12), _c(s.Jd), _c(s.DJd)
        self.G = [_c(s.G[i][j]) for (i, j) in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))]
        self.bm1 = _c(s.bm1)
        self.rst2w = [_c(s.rst2w[j][i]) for j in range(3) for i in range(3)]
        self.rstdw = [_c(s.rstdw[j][i]) for j in range(3) for i in range(3)]
        self.mbinv = [_c(s.mask[i] * s.binvm1) for i in range(3)]
        self.mask = [_c(m) for m in s.mask]
        # gather-scatter groups (copies of a shared label), CSR
        glo = np.asarray(s.glo).ravel()
        order = np.argsort(glo, kind="stable")
        gs_sorted = glo[order]
        start = np.flatnonzero(np.r_[True, gs_sorted[1:] != gs_sorted[:-1]])
        cnt = np.diff(np.r_[start, glo.size])
        keep = cnt > 1
        self.off = np.r_[0, np.cumsum(cnt[keep])].astype(np.int64)
        self.idx = np.concatenate([order[a:a + c] for a, c in zip(start[keep], cnt[keep])]).astype(np.int64) if keep.any() else np.zeros(0, np.int64)
        self.ngroups = int(keep.sum())

    def threads(self):
        return int(self.lib.nl_threads())

    def set_threads(self, n):
        self.lib.nl_set_threads(int(n))

    def gs(self, f):
        self.lib.nl_gs(C.c_long(self.ngroups), self.off.ctypes.data_as(C.POINTER(C.c_long)), self.idx.ctypes.data_as(C.POINTER(C.c_long)), _p(f))
        return f

    def axhelm_local(self, u, h1, h2):
        u = _c(u)
        w = np.empty_like(u)
        self.lib.nl_axhelm(C.c_long(self.E), self.n, _p(self.D), _pp(self.G), _p(self.bm1), _p(u), _p(w), C.c_double(h1), C.c_double(h2))
        return w

    def helm(self, u, h1, h2):
        return [self.mask[i] * self.gs(self.axhelm_local(u[i], h1, h2)) for i in range(3)]

    def opgradt(self, p):
        p = _c(p)
        w = [np.empty(self.sem.shape1) for _ in range(3)]
        self.lib.nl_opgradt(C.c_long(self.E), self.n, self.n2, _p(self.I12), _p(self.D12), _pp(self.rst2w), _p(p), _pp(w))
        return w

    def opdiv(self, u):
        u = [_c(a) for a in u]
        out = np.empty(self.sem.shape2)
        self.lib.nl_opdiv(C.c_long(self.E), self.n, self.n2, _p(self.I12), _p(self.D12), _pp(self.rst2w), _pp(u), _p(out))
        return out

    def cdabdtp(self, p):
        w = self.opgradt(p)
        w = [self.mbinv[i] * self.gs(w[i]) for i in range(3)]
        return self.opdiv(w)

    def lns_conv_weak(self, U, u):
        U, u = [_c(a) for a in U], [_c(a) for a in u]
        out = [np.empty(self.sem.shape1) for _ in range(3)]
        self.lib.nl_conv(C.c_long(self.E), self.n, self.nd, _p(self.Jd), _p(self.DJd), _pp(self.rstdw), _pp(U), _pp(u), _pp(out))
        return out

    def glsc3(self, a, b, w):
        return float(self.lib.nl_glsc3(C.c_long(a.size), _p(a), _p(b), _p(w)))

    def axpby(self, alpha, x, beta, y):
        self.lib.nl_axpby(C.c_long(y.size), C.c_double(alpha), _p(x), C.c_double(beta), _p(y))

    def cgvec(self, x, r, z, p, w, m, wt):
        return float(self.lib.nl_cgvec(C.c_long(x.size), _p(x), _p(r), _p(z), _p(p), _p(w), _p(m), _p(wt)))
