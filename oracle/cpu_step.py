"""ORACLE-side: one REAL time step of the restated integrator on the C + OpenMP port (test / measurement infrastructure,
never the product).

`CStep.advance` is `oracle/lns.py ExptA.advance` (3-D, direct, no scalar) with every array pass executed by
oracle/c/sem_cpu.c: the dealiased convective term, the BDF / EXT right-hand side, the velocity solve (`nl_pcg_helm`, the
Jacobi-PCG of `pcg_helm`), the pressure solve on the mean-free subspace (`nl_pcg_E`, the Jacobi-PCG of `pcg_E`) and the two
corrections.  tests/test_cpu_oracle.py checks it against the numpy twin step by step; bench.py's cpu_baseline times it.
Reference structure: nek_advance as driven by /root/reference/src/linops/exponential_propagator.f90:39-46.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .cport import CPort, _c, _p, _pp
from .lns import BDF, EXT, ExptA, LNSConfig


class CStep:
    def __init__(self, sem, baseflow, cfg: LNSConfig, threads=None):
        if sem.dim != 3 or cfg.ifheat:
            raise ValueError("CStep covers the 3-D velocity-pressure step")
        self.sem, self.cfg = sem, cfg
        self.cp = CPort(sem)
        if threads:
            self.cp.set_threads(threads)
        self.lib = self.cp.lib
        self.lib.nl_sum.restype = C.c_double
        self.lib.nl_pcg_helm.restype = C.c_int
        self.lib.nl_pcg_E.restype = C.c_int
        twin = ExptA(sem, baseflow, cfg)          # set-up only: dt / nsteps rule, diagonal of E
        self.twin = twin
        self.dt, self.nsteps, self.nu = twin.dt, twin.nsteps, twin.nu
        self.U = [_c(a) for a in twin.U]
        self.ediag_inv = _c(twin.ediag_inv)
        self.bm1, self.bm2 = _c(sem.bm1), _c(sem.bm2)
        self.mask = [_c(m * np.ones(sem.shape1)) for m in sem.mask]
        self.mbinv = self.cp.mbinv
        self.vmult = _c(sem.vmult * np.ones(sem.shape1))
        self.wnorm = _c(sem.binvm1 * sem.vmult / sem.volvm1)
        self.n1, self.n2 = int(np.prod(sem.shape1)), int(np.prod(sem.shape2))
        self._hd = {}
        z1 = lambda: np.zeros(sem.shape1)
        self.wk = {k: [z1() for _ in range(3)] for k in ("x", "z", "p", "w", "res", "gp", "hu", "uh", "g2")}
        self.wk2 = {k: np.zeros(sem.shape2) for k in ("x", "z", "p", "w", "r")}
        self.stats = {"v_iters": 0, "p_iters": 0, "steps": 0}

    def hdiag_inv(self, h2):
        if h2 not in self._hd:
            self._hd[h2] = _c(self.twin.hdiag_inv(h2))
        return self._hd[h2]

    def reset(self, u, p):
        s = self.sem
        self.u = [_c(np.array(a, dtype=float).reshape(s.shape1)) for a in u]
        self.p = _c(np.array(p, dtype=float).reshape(s.shape2))
        self.ulag = [[np.zeros(s.shape1) for _ in range(3)] for _ in range(2)]
        self.flag = [[np.zeros(s.shape1) for _ in range(3)] for _ in range(2)]
        self.istep = 0

    # ---- small helpers over the C loops ----
    def _lincomb(self, xs, cs, scale, y, acc):
        cs = np.ascontiguousarray(cs, dtype=np.float64)
        self.lib.nl_lincomb(C.c_long(y.size), len(xs), _pp(xs), _p(cs), _p(scale) if scale is not None else None, _p(y), int(acc))

    def _ortho(self, a):
        if not self.sem.has_outflow:
            self.lib.nl_shift(C.c_long(a.size), _p(a), C.c_double(self.lib.nl_sum(C.c_long(a.size), _p(a)) / a.size))

    def advance(self):
        s, cfg, cp, lib = self.sem, self.cfg, self.cp, self.lib
        dt = self.dt
        self.istep += 1
        self._ortho(self.p)
        k = min(self.istep, cfg.torder)
        b0, bd = BDF[k]
        ab = EXT[k]
        N = cp.lns_conv_weak(self.U, self.u)
        F = N
        for i in range(3):
            cp.axpby(0.0, F[i], -1.0, F[i])                      # F = -N
        hist_f = [F] + self.flag
        hist_u = [self.u] + self.ulag
        rhs = self.wk["g2"]
        for i in range(3):
            self._lincomb([hist_f[j][i] for j in range(k)], [ab[j] for j in range(k)], None, rhs[i], False)
            self._lincomb([hist_u[j][i] for j in range(k)], [bd[j] / dt for j in range(k)], self.bm1, rhs[i], True)
        # shift histories (the oldest buffers are recycled)
        old_u = self.ulag[1]
        self.flag = [F, self.flag[0]]
        for i in range(3):
            np.copyto(old_u[i], self.u[i])
        self.ulag = [old_u, self.ulag[0]]
        # tentative velocity, residual form with lagged pressure
        h2 = b0 / dt
        gp = self.wk["gp"]
        lib.nl_opgradt(C.c_long(s.E), s.n, s.n2, _p(cp.I12), _p(cp.D12), _pp(cp.rst2w), _p(self.p), _pp(gp))
        res, hu = self.wk["res"], self.wk["hu"]
        for i in range(3):
            lib.nl_axhelm(C.c_long(s.E), s.n, _p(cp.D), _pp(cp.G), _p(cp.bm1), _p(self.u[i]), _p(hu[i]), C.c_double(self.nu), C.c_double(h2))
            lib.nl_residual(C.c_long(self.n1), _p(self._ones1()), _p(rhs[i]), _p(gp[i]), _p(hu[i]), _p(res[i]))
            cp.gs(res[i])
            lib.nl_residual(C.c_long(self.n1), _p(self.mask[i]), _p(res[i]), None, None, _p(res[i]))
        x, z, p, w = self.wk["x"], self.wk["z"], self.wk["p"], self.wk["w"]
        itv = lib.nl_pcg_helm(C.c_long(s.E), s.n, _p(cp.D), _pp(cp.G), _p(cp.bm1), C.c_long(cp.ngroups), cp.off.ctypes.data_as(C.POINTER(C.c_long)),
                              cp.idx.ctypes.data_as(C.POINTER(C.c_long)), _pp(self.mask), _p(self.hdiag_inv(h2)), _p(self.vmult), _p(self.wnorm),
                              C.c_double(self.nu), C.c_double(h2), _pp(res), _pp(x), _pp(z), _pp(p), _pp(w), C.c_double(cfg.vtol ** 2),
                              int(cfg.maxit_v), int(cfg.fixed_iters_v))
        uh = self.wk["uh"]
        for i in range(3):
            lib.nl_add_scaled(C.c_long(self.n1), _p(self.u[i]), C.c_double(1.0), None, _p(x[i]), _p(uh[i]))
        # pressure correction
        rp = self.wk2["r"]
        lib.nl_opdiv(C.c_long(s.E), s.n, s.n2, _p(cp.I12), _p(cp.D12), _pp(cp.rst2w), _pp(uh), _p(rp))
        cp.axpby(0.0, rp, -(b0 / dt), rp)
        self._ortho(rp)
        scale = dt / b0
        itp = lib.nl_pcg_E(C.c_long(s.E), s.n, s.n2, _p(cp.I12), _p(cp.D12), _pp(cp.rst2w), C.c_long(cp.ngroups),
                           cp.off.ctypes.data_as(C.POINTER(C.c_long)), cp.idx.ctypes.data_as(C.POINTER(C.c_long)), _pp(self.mbinv), _p(self.ediag_inv),
                           _p(self.bm2), C.c_double(s.volvm2), int(not s.has_outflow), _p(rp), _p(self.wk2["x"]), _p(self.wk2["z"]), _p(self.wk2["p"]),
                           _p(self.wk2["w"]), _pp(w), C.c_double((cfg.ptol / scale) ** 2), int(cfg.maxit_p), int(cfg.fixed_iters_p))
        dp = self.wk2["x"]
        cp.axpby(1.0, dp, 1.0, self.p)
        lib.nl_opgradt(C.c_long(s.E), s.n, s.n2, _p(cp.I12), _p(cp.D12), _pp(cp.rst2w), _p(dp), _pp(gp))
        for i in range(3):
            cp.gs(gp[i])
            lib.nl_add_scaled(C.c_long(self.n1), _p(uh[i]), C.c_double(dt / b0), _p(self.mbinv[i]), _p(gp[i]), _p(self.u[i]))
        self.stats["v_iters"] += int(itv)
        self.stats["p_iters"] += int(itp)
        self.stats["steps"] += 1
        return int(itv), int(itp)

    def _ones1(self):
        if not hasattr(self, "_one"):
            self._one = np.ones(self.sem.shape1)
        return self._one
