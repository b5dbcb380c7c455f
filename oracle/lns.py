"""ORACLE (test infrastructure only) -- exponential propagator exp(tau L) of the linearised
Navier-Stokes operator, restated in numpy.  Never imported by the product.

In-tree protocol followed (what the reference DOES pin):
* matvec protocol: base flow in, solver set-up, IC in, `nsteps` x advance with restart-history replay
  for `istep <= nrst`, result out, `nrst` extra steps to refill the history
  (/root/reference/src/linops/exponential_propagator.f90:15-60, :109-142),
* `dt`/`nsteps` rule from tau and CFL (/root/reference/src/neklab_nek_setup.f90:195-200, cfl_limit 0.5
  at exponential_propagator.f90:9-12),
* operator terms of L (/root/reference/src/linops/neklab_linops.f90:268-426).

The time integrator itself (`nek_advance`) is Nek5000 code (absent, un-pinned); no fixture pins it step by
step.  End to end it is pinned by the reference's published eigenvalue (cylinder Re = 50: 1.015780 here vs
1.0156 +- 1e-4, tests/test_gpu_known_answer.py) and by the Orr-Sommerfeld value for Poiseuille flow
(DESIGN.md section 2).  It is restated from the published Pn-Pn-2 BDFk/EXTk splitting (Fischer 1997; Deville-Fischer-Mund ch. 6):
tentative Helmholtz solve in residual form with the lagged pressure, consistent-Poisson pressure
correction, velocity update.  Solver details that Nek5000 hides (one joint Jacobi-PCG over the
velocity components, Jacobi-PCG on E) are this project's own and are documented in DESIGN.md.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .vectors import NekDVector

BDF = {1: (1.0, (1.0,)), 2: (1.5, (2.0, -0.5)), 3: (11.0 / 6.0, (3.0, -1.5, 1.0 / 3.0))}
EXT = {1: (1.0,), 2: (2.0, -1.0), 3: (3.0, -3.0, 1.0)}
FLOOR2 = 1e-28   # relative floor on |r|^2 below which PCG stops (same constant as csrc/lns.hip kFloor2)


@dataclass
class LNSConfig:
    re: float = 100.0           # viscosity = 1/re  (1cyl.par: viscosity = -50 => Re = 50)
    torder: int = 3             # |param(27)|, 1cyl.par: timeStepper = bdf3
    tau: float = 1.0
    cfl_limit: float = 0.5      # exponential_propagator.f90:12
    vtol: float = 1e-9          # 1cyl.par [VELOCITY] residualTol
    ptol: float = 1e-7          # 1cyl.par [PRESSURE] residualTol
    maxit_v: int = 200
    maxit_p: int = 2000
    fixed_iters_v: int = 0      # >0: run exactly this many PCG iterations (parity mode)
    fixed_iters_p: int = 0
    dt: float = 0.0             # >0: skip the CFL rule and use nsteps = ceil(tau/dt)
    # Boussinesq coupling (Nek5000 [TEMPERATURE] block + the buoyancy written in the case's userf, e.g.
    # examples/rayBen/baseflow/rayBen.par:39-45, rayBen.usr:77-103): rhocp (d/dt + U.grad) theta + rhocp u.grad Theta =
    # conductivity lap theta ; momentum forcing buoy_i * theta
    ifheat: bool = False
    conductivity: float = 1.0
    rhocp: float = 1.0
    buoy: tuple = (0.0, 0.0, 0.0)
    no_history: bool = False    # no restart history: impulsive start of every matvec, no history steps (nlg_exptA_config.no_history)


def dt_rule(tau, cfl_at_unit_dt, cfl_limit):
    """reference: neklab_nek_setup.f90:195-198."""
    dt = cfl_limit / cfl_at_unit_dt
    nsteps = int(math.ceil(tau / dt))
    return tau / nsteps, nsteps


class ExptA:
    def __init__(self, sem, baseflow, cfg: LNSConfig, baseflow_theta=None):
        self.sem = sem
        self.cfg = cfg
        self.U = [sem.f1(a).copy() for a in baseflow]
        self.Theta = None if baseflow_theta is None else sem.f1(baseflow_theta).copy()
        if cfg.ifheat and self.Theta is None:
            raise ValueError("ifheat needs the base temperature")
        if cfg.dt > 0:
            self.nsteps = int(math.ceil(cfg.tau / cfg.dt - 1e-12))
            self.dt = cfg.tau / self.nsteps
        else:
            c1 = sem.compute_cfl(self.U, 1.0)
            self.dt, self.nsteps = dt_rule(cfg.tau, c1, cfg.cfl_limit)
        self.nu = 1.0 / cfg.re
        self.ediag_inv = 1.0 / sem.e_diag()
        self._hdiag = {}
        self.stats = {"v_iters": 0, "p_iters": 0, "steps": 0}

    # ---------------- solvers ----------------
    def hdiag_inv(self, h2):
        key = float(h2)
        if key not in self._hdiag:
            d = self.sem.gs(self.sem.helm_diag_local(self.nu, h2))
            self._hdiag[key] = 1.0 / d
        return self._hdiag[key]

    def helm_apply(self, p, h2):
        s = self.sem
        return [s.mask[i] * s.gs(s.axhelm_local(p[i], self.nu, h2)) for i in range(s.dim)]

    def pcg_helm(self, b, h2):
        s, cfg = self.sem, self.cfg
        dim = s.dim
        minv = self.hdiag_inv(h2)
        wnorm = s.binvm1 * s.vmult / s.volvm1
        x = [np.zeros(s.shape1) for _ in range(dim)]
        r = [bi.copy() for bi in b]
        z = [s.mask[i] * minv * r[i] for i in range(dim)]
        p = [zi.copy() for zi in z]
        rz = sum(np.sum(r[i] * z[i] * s.vmult) for i in range(dim))
        it = 0
        maxit = cfg.fixed_iters_v if cfg.fixed_iters_v > 0 else cfg.maxit_v
        rn2_0 = None
        while it < maxit:
            rn2 = sum(np.sum(r[i] * r[i] * wnorm) for i in range(dim))
            if rn2_0 is None:
                rn2_0 = rn2
            if rn2 <= FLOOR2 * rn2_0:          # converged to rounding: further iterations divide 0 by 0
                break
            if cfg.fixed_iters_v <= 0:
                if rn2 < cfg.vtol ** 2:
                    break
            w = self.helm_apply(p, h2)
            pw = sum(np.sum(p[i] * w[i] * s.vmult) for i in range(dim))
            alpha = rz / pw
            for i in range(dim):
                x[i] += alpha * p[i]
                r[i] -= alpha * w[i]
                z[i] = s.mask[i] * minv * r[i]
            rz_new = sum(np.sum(r[i] * z[i] * s.vmult) for i in range(dim))
            beta = rz_new / rz
            rz = rz_new
            for i in range(dim):
                p[i] = z[i] + beta * p[i]
            it += 1
        self.stats["v_iters"] += it
        return x

    def pcg_helm_single_reduction(self, b, h2):
        """The twin of run_pcg's single-reduction branch (csrc/lns.hip, cg_post_logic mode 4): Chronopoulos & Gear's PCG, in which (w, u),
        (r, u) and |r|^2 -- u = M^-1 r, w = A u -- come from ONE reduction per iteration and p, s = A p are recurrences.  Same iterates as
        pcg_helm in exact arithmetic; the convergence test comes one operator application late.  Returns (x, iterations)."""
        s, cfg = self.sem, self.cfg
        dim = s.dim
        minv = self.hdiag_inv(h2)
        wnorm = s.binvm1 * s.vmult / s.volvm1
        x = [np.zeros(s.shape1) for _ in range(dim)]
        r = [bi.copy() for bi in b]
        u = [s.mask[i] * minv * r[i] for i in range(dim)]
        p = [None] * dim
        sv = [None] * dim
        maxit = cfg.fixed_iters_v if cfg.fixed_iters_v > 0 else cfg.maxit_v
        rn2_0 = sum(np.sum(r[i] * r[i] * wnorm) for i in range(dim))
        if (cfg.fixed_iters_v <= 0 and rn2_0 < cfg.vtol ** 2) or maxit <= 0 or rn2_0 <= 0.0:
            return x, 0
        it, gamma_old, alpha = 0, None, None
        while True:
            w = self.helm_apply(u, h2)
            delta = sum(np.sum(u[i] * w[i] * s.vmult) for i in range(dim))          # the three sums of ONE reduction
            gamma = sum(np.sum(r[i] * u[i] * s.vmult) for i in range(dim))
            rn2 = sum(np.sum(r[i] * r[i] * wnorm) for i in range(dim))
            if it > 0 and ((cfg.fixed_iters_v <= 0 and rn2 < cfg.vtol ** 2) or it >= maxit or rn2 <= FLOOR2 * rn2_0):
                break
            beta = 0.0 if it == 0 else gamma / gamma_old
            alpha = gamma / delta if it == 0 else gamma / (delta - beta * gamma / alpha)
            gamma_old = gamma
            it += 1
            for i in range(dim):
                p[i] = u[i].copy() if it == 1 else u[i] + beta * p[i]
                sv[i] = w[i].copy() if it == 1 else w[i] + beta * sv[i]
                x[i] += alpha * p[i]
                r[i] -= alpha * sv[i]
                r[i] = np.where(s.mask[i] == 0.0, 0.0, r[i])
                u[i] = s.mask[i] * minv * r[i]
        return x, it

    def pcg_heat(self, b, h1, h2):
        """Jacobi-PCG for the scalar Helmholtz problem tmask QQ^T (h1 A + h2 B) x = b (same stopping rule as the velocity)."""
        s, cfg = self.sem, self.cfg
        minv = s.tmask / s.gs(s.helm_diag_local(h1, h2))
        wnorm = s.binvm1 * s.vmult / s.volvm1

        def A(p):
            return s.tmask * s.gs(s.axhelm_local(p, h1, h2))

        x = np.zeros(s.shape1)
        r = b.copy()
        z = minv * r
        p = z.copy()
        rz = np.sum(r * z * s.vmult)
        it, rn2_0 = 0, None
        maxit = cfg.fixed_iters_v if cfg.fixed_iters_v > 0 else cfg.maxit_v
        while it < maxit:
            rn2 = np.sum(r * r * wnorm)
            if rn2_0 is None:
                rn2_0 = rn2
            if rn2 <= FLOOR2 * rn2_0:
                break
            if cfg.fixed_iters_v <= 0 and rn2 < cfg.vtol ** 2:
                break
            w = A(p)
            alpha = rz / np.sum(p * w * s.vmult)
            x += alpha * p
            r -= alpha * w
            z = minv * r
            rz_new = np.sum(r * z * s.vmult)
            p = z + (rz_new / rz) * p
            rz = rz_new
            it += 1
        self.stats["t_iters"] = self.stats.get("t_iters", 0) + it
        return x

    def pcg_E(self, b, scale):
        """Solve E x = b; converged when scale*||r||_p < ptol (remaining divergence).

        Without outflow the pressure is defined up to a constant.  On deformed elements the GL(lx2)
        quadrature does not reproduce D^T 1 = 0 exactly, so E is only NEARLY singular; iterating on it
        amplifies the near-null mode (observed: the time stepper blows up).  The solve therefore runs
        on P E P with P = I - 1 1^T / n restricted to the mean-free subspace (Nek5000 `ortho` applied
        consistently to operator, preconditioner and right-hand side)."""
        s, cfg = self.sem, self.cfg
        minv = self.ediag_inv
        proj = not s.has_outflow
        npts = b.size

        def P(a):
            return a - np.sum(a) / npts if proj else a

        x = np.zeros(s.shape2)
        r = P(b.copy())
        z = minv * r
        p = P(z)
        rz = float(np.sum(r * z))
        it = 0
        maxit = cfg.fixed_iters_p if cfg.fixed_iters_p > 0 else cfg.maxit_p
        rn2_0 = None
        while it < maxit:
            rn2 = float(np.sum(r * r / s.bm2)) / s.volvm2
            if rn2_0 is None:
                rn2_0 = rn2
            if rn2 <= FLOOR2 * rn2_0:
                break
            if cfg.fixed_iters_p <= 0:
                if rn2 < (cfg.ptol / scale) ** 2:
                    break
            w = s.cdabdtp(p)
            pw = float(np.sum(p * w))
            alpha = rz / pw
            x += alpha * p
            r -= alpha * P(w)
            z = minv * r
            rz_new = float(np.sum(r * z))
            beta = rz_new / rz
            rz = rz_new
            p = P(z) + beta * p
            it += 1
        self.stats["p_iters"] += it
        return x

    # ---------------- one time step (restated nek_advance, perturbation mode) ----------------
    def _reset_state(self, vec: NekDVector, adjoint):
        s = self.sem
        self.u = [a.copy() for a in vec.v]
        self.p = vec.pr.copy()
        self.ulag = [[np.zeros(s.shape1) for _ in range(s.dim)] for _ in range(2)]
        self.flag = [[np.zeros(s.shape1) for _ in range(s.dim)] for _ in range(2)]
        self.istep = 0
        self.adjoint = adjoint
        if self.cfg.ifheat:
            self.t = vec.theta[0].copy()
            self.tlag = [np.zeros(s.shape1) for _ in range(2)]
            self.ftlag = [np.zeros(s.shape1) for _ in range(2)]

    def advance(self):
        s, cfg = self.sem, self.cfg
        dim, dt = s.dim, self.dt
        self.istep += 1
        # gauge: without outflow the pressure is defined up to a constant; keep it mean-free so that the
        # constant mode (whose discrete gradient is not exactly zero on deformed elements) can never act
        # on the velocity -- otherwise it shows up as a spurious eigenvalue mu = 1 of the propagator
        self.p = s.ortho(self.p)
        k = min(self.istep, cfg.torder)
        b0, bd = BDF[k]
        ab = EXT[k]
        if cfg.ifheat:
            # scalar first, then the fluid with the buoyancy of the NEW temperature in its explicit term (the order of
            # Nek5000's nek_advance: heat, then fluid, whose userf reads the updated scalar)
            rc = cfg.rhocp
            if getattr(self, "nonlinear", False):
                Nt = s.conv_weak(self.u, self.t)                  # full equation: (u.grad) theta
            elif self.adjoint:
                # adjoint of the coupled operator in the reference's inner product (velocity + temperature, weight bm1, no
                # rhocp factor: real_vectors.f90:217-224): with A = [[L_u, b], [-grad Theta . , L_theta / rhocp]] the adjoint is
                #   u+_t     = L_u^+ u+ - theta+ grad Theta
                #   rhocp theta+_t = rhocp (U . grad) theta+ + conductivity lap theta+ + rhocp b . u+
                # (exponential_propagator_temp.f90:62-107 integrates Nek5000's adjoint equations; restated like the rest)
                Nt = -s.conv_weak(self.U, self.t) - s.bm1 * sum(cfg.buoy[i] * self.u[i] for i in range(dim))
            else:
                Nt = s.conv_weak(self.U, self.t) + s.conv_weak(self.u, self.Theta)
            Ft = -rc * Nt
            hist_ft = [Ft] + self.ftlag
            hist_t = [self.t] + self.tlag
            rhs_t = sum(ab[j] * hist_ft[j] for j in range(k)) + (rc * s.bm1 / dt) * sum(bd[j] * hist_t[j] for j in range(k))
            self.ftlag = [Ft, self.ftlag[0]]
            self.tlag = [self.t.copy(), self.tlag[0]]
            h1t, h2t = cfg.conductivity, rc * b0 / dt
            res_t = s.tmask * s.gs(rhs_t - s.axhelm_local(self.t, h1t, h2t))
            self.t = self.t + self.pcg_heat(res_t, h1t, h2t)
        force = getattr(self, "force", None)
        if getattr(self, "nonlinear", False):
            # full Navier-Stokes step (nonlinear_map, /root/reference/src/systems/fixed_point.f90:4-38):
            # (u.grad)u = 1/2 [(U.grad)u + (u.grad)U] at U = u
            N = [0.5 * a for a in s.lns_conv_weak(self.u, self.u, adjoint=False)]
        else:
            N = s.lns_conv_weak(self.U, self.u, adjoint=self.adjoint)
        F = [-a for a in N]
        if force is not None:
            # time-harmonic body force Re[f exp(i s omega t)] at the level the step starts from (resolvent.f90:97-103)
            f_re, f_im, omega, sign = force
            ph = sign * omega * (self.istep - 1) * dt
            for i in range(dim):
                F[i] = F[i] + s.bm1 * (np.cos(ph) * f_re[i] - (np.sin(ph) * f_im[i] if f_im is not None else 0.0))
        if cfg.ifheat and self.adjoint and not getattr(self, "nonlinear", False):
            # - theta+ grad Theta, weak and dealiased like the convective terms, with the NEW theta+
            tg = s.scalar_times_grad_weak(self.t, self.Theta)
            for i in range(dim):
                F[i] = F[i] - tg[i]
        elif cfg.ifheat:
            for i in range(dim):
                if cfg.buoy[i] != 0.0:
                    F[i] = F[i] + s.bm1 * cfg.buoy[i] * self.t
        hist_f = [F] + self.flag
        hist_u = [self.u] + self.ulag
        rhs = []
        for i in range(dim):
            acc = sum(ab[j] * hist_f[j][i] for j in range(k))
            acc = acc + (s.bm1 / dt) * sum(bd[j] * hist_u[j][i] for j in range(k))
            rhs.append(acc)
        # shift histories
        self.flag = [F, self.flag[0]]
        self.ulag = [[a.copy() for a in self.u], self.ulag[0]]
        # tentative velocity, residual form with lagged pressure
        h2 = b0 / dt
        gp = s.opgradt(self.p)
        res = [s.mask[i] * s.gs(rhs[i] + gp[i] - s.axhelm_local(self.u[i], self.nu, h2)) for i in range(dim)]
        du = self.pcg_helm(res, h2)
        uh = [self.u[i] + du[i] for i in range(dim)]
        # pressure correction
        rp = -(b0 / dt) * s.opdiv(uh)
        rp = s.ortho(rp)
        dp = self.pcg_E(rp, dt / b0)
        self.p = self.p + dp
        w = s.opbinv(s.opgradt(dp))
        self.u = [uh[i] + (dt / b0) * w[i] for i in range(dim)]
        self.stats["steps"] += 1

    def _load(self, vec):
        self.u = [a.copy() for a in vec.v]
        self.p = vec.pr.copy()
        if self.cfg.ifheat:
            self.t = vec.theta[0].copy()

    def _store(self, vec):
        for a, b in zip(vec.v, self.u):
            a[...] = b
        vec.pr[...] = self.p
        if self.cfg.ifheat:
            vec.theta[0][...] = self.t

    # ---------------- reference: exponential_propagator.f90:15-60 / :62-107 ----------------
    def matvec(self, vec_in: NekDVector, adjoint=False) -> NekDVector:
        nrst = 0 if self.cfg.no_history else self.cfg.torder - 1
        vec_out = NekDVector(self.sem, vec_in.nscal, vec_in.lorder)   # intent(out): default-initialised
        self._reset_state(vec_in, adjoint)
        if getattr(self, "proj_lab", None) is not None:
            self.u = self.proj(self.u)                               # exptA_proj_matvec, exponential_propagator_proj.f90:51
            self.p = self.proj_pressure(self.p)
        for istep in range(1, self.nsteps + 1):
            self.advance()
            if istep <= nrst and vec_in.has_rst_fields():          # get_rst, :129-142
                tmp = NekDVector(self.sem, vec_in.nscal, vec_in.lorder)
                vec_in.get_rst(tmp, istep)
                self._load(tmp)
                if getattr(self, "proj_lab", None) is not None:      # replayed states are projected too (csrc/lns.hip do_matvec)
                    self.u = self.proj(self.u)
                    self.p = self.proj_pressure(self.p)
        if getattr(self, "proj_lab", None) is not None:
            self.u = self.proj(self.u)                               # :66
            self.p = self.proj_pressure(self.p)
            self.ulag = [self.proj(a) for a in self.ulag]            # ... and the lagged states of the multistep scheme
        self._store(vec_out)
        # compute_rst, :109-127
        for irst in range(1, nrst + 1):
            self.advance()
            tmp = NekDVector(self.sem, vec_in.nscal, vec_in.lorder)
            self._store(tmp)
            vec_out.save_rst(tmp, irst)
        return vec_out

    def rmatvec(self, vec_in):
        return self.matvec(vec_in, adjoint=True)

    # ---------------- reference: exponential_propagator_proj.f90:18-28, :135-173 ----------------
    def set_projection(self, alpha, idir, line_label, line_label2=None, x2=None):
        """Wavenumber projection of the projected propagator: lines of points along direction idir (1-based); with
        line_label2 / x2 the pressure is projected as well (lines and coordinate along idir of the pressure points)."""
        s = self.sem
        self.proj_lab = np.asarray(line_label).reshape(-1)
        self.proj_cv = np.cos(alpha * s.X[idir - 1])
        self.proj_sv = np.sin(alpha * s.X[idir - 1])
        self.proj_den = np.bincount(self.proj_lab, weights=s.bm1.ravel())
        self.proj_lab2 = None
        if line_label2 is not None:
            self.proj_lab2 = np.asarray(line_label2).reshape(-1)
            self.proj_cv2 = np.cos(alpha * np.asarray(x2).reshape(s.shape2))
            self.proj_sv2 = np.sin(alpha * np.asarray(x2).reshape(s.shape2))
            self.proj_den2 = np.bincount(self.proj_lab2, weights=s.bm2.ravel())

    def proj_pressure(self, p):
        if getattr(self, "proj_lab2", None) is None:
            return p
        s = self.sem
        c = np.bincount(self.proj_lab2, weights=(2.0 * p * self.proj_cv2 * s.bm2).ravel()) / self.proj_den2
        d = np.bincount(self.proj_lab2, weights=(2.0 * p * self.proj_sv2 * s.bm2).ravel()) / self.proj_den2
        return self.proj_cv2 * c[self.proj_lab2].reshape(s.shape2) + self.proj_sv2 * d[self.proj_lab2].reshape(s.shape2)

    def proj(self, u):
        """u <- cv <2 u cv> + sv <2 u sv>, <.> the bm1-weighted average along the line (proj_alpha)."""
        s = self.sem
        out = []
        for a in u:
            c = np.bincount(self.proj_lab, weights=(2.0 * a * self.proj_cv * s.bm1).ravel()) / self.proj_den
            d = np.bincount(self.proj_lab, weights=(2.0 * a * self.proj_sv * s.bm1).ravel()) / self.proj_den
            out.append(self.proj_cv * c[self.proj_lab].reshape(s.shape1) + self.proj_sv * d[self.proj_lab].reshape(s.shape1))
        return out

    # ---------------- reference: src/linops/resolvent.f90:80-111, :133-166 ----------------
    def set_tau(self, tau):
        self.cfg.tau = tau
        self.set_baseflow(self.U)

    def integrate_forced(self, ic, f_re, f_im, omega, adjoint=False):
        """State after nsteps from `ic` (None: rest) under the body force Re[(f_re + i f_im) exp(i s omega t)]."""
        start = ic if ic is not None else NekDVector(self.sem)
        self._reset_state(start, adjoint)
        if ic is None:
            self.p = np.zeros(self.sem.shape2)
        self.force = ([np.asarray(a) for a in f_re], None if f_im is None else [np.asarray(a) for a in f_im], omega,
                      -1.0 if adjoint else 1.0)
        try:
            for _ in range(self.nsteps):
                self.advance()
        finally:
            self.force = None
        out = NekDVector(self.sem)
        self._store(out)
        return out

    # ---------------- reference: src/systems/fixed_point.f90:4-38 ----------------
    def set_baseflow(self, baseflow):
        """`self%X` of jac_exptA_matvec (fixed_point.f90:52): new frozen base flow, dt / nsteps from its CFL number."""
        self.U = [self.sem.f1(a).copy() for a in baseflow]
        if self.cfg.dt > 0:
            self.nsteps = int(math.ceil(self.cfg.tau / self.cfg.dt - 1e-12))
            self.dt = self.cfg.tau / self.nsteps
        else:
            c1 = self.sem.compute_cfl(self.U, 1.0)
            self.dt, self.nsteps = dt_rule(self.cfg.tau, c1, self.cfg.cfl_limit)
        self._hdiag = {}

    def nonlinear_map(self, vec_in: NekDVector) -> NekDVector:
        """F(X) = Phi_tau(X) - X with the nonlinear integrator; the time step follows the CFL number of X."""
        self.set_baseflow(vec_in.v)
        if self.cfg.ifheat:
            self.Theta = vec_in.theta[0].copy()
        vec_out = NekDVector(self.sem, vec_in.nscal, vec_in.lorder)
        self._reset_state(vec_in, False)
        self.nonlinear = True
        try:
            for _ in range(self.nsteps):
                self.advance()
        finally:
            self.nonlinear = False
        self._store(vec_out)
        vec_out.axpby(-1.0, vec_in, 1.0)
        return vec_out
