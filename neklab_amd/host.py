"""Host-side mirror of the reference's operator / plugin interface for the hot path, on top of the
C ABI (include/neklab_gpu.h).  Names, argument meaning and error behaviour follow the reference:

* `nek_dvector`   <- type nek_dvector           /root/reference/src/vectors/neklab_vectors.f90:26-50
* `exptA_linop`   <- type exptA_linop           /root/reference/src/linops/neklab_linops.f90:35-44
* `linear_stability_analysis_fixed_point` <-    /root/reference/src/neklab_analysis.f90:38-105
* `nek2vec` / `vec2nek` field movers       <-    /root/reference/src/neklab_utils.f90:84-134

In the reference these are Fortran 2008 types driven by LightKrylov; the Fortran shim with the same
bindings is neklab_amd/fortran/neklab_gpu.f90.  This Python mirror exists so that parity tests can be
written the way the reference's own driver reads.  All arithmetic happens in HIP kernels; nothing here
computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import NlgError, check, dptr, vp

VX, VY, VZ, PR, THETA = 0, 1, 2, 3, 4


class Context:
    """Device + stream (+ RCCL communicator).  One per process / GPU."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = vp()
        check(self.lib.nlg_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.rank, self.nranks = 0, 1

    def comm_init(self, rank: int, nranks: int, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        check(self.lib.nlg_ctx_comm_init(self.h, rank, nranks, C.cast(buf, vp)))
        self.rank, self.nranks = rank, nranks

    def comm_init_shm(self, rank: int, nranks: int, name: str, slot_bytes: int = 64 << 20):
        """Validation transport (several test ranks on one GPU), see include/neklab_gpu.h."""
        check(self.lib.nlg_ctx_comm_init_shm(self.h, rank, nranks, name.encode(), slot_bytes))
        self.rank, self.nranks = rank, nranks

    @staticmethod
    def unique_id() -> bytes:
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        check(lib.nlg_comm_unique_id(C.cast(buf, vp)))
        return buf.raw

    def sync(self):
        check(self.lib.nlg_ctx_sync(self.h))

    def close(self):
        if self.h:
            self.lib.nlg_ctx_destroy(self.h)
            self.h = None


class Mesh:
    """The Nek5000 state the reference reads from SIZE/TOTAL commons, uploaded once."""

    def __init__(self, ctx: Context, mesh, lxd: int = 0):
        self.ctx, self.lib = ctx, ctx.lib
        self.host = mesh
        d = _lib.MeshDesc()
        d.dim, d.n, d.lxd, d.nelv = mesh.dim, mesh.n, int(lxd), mesh.E
        keep = []

        def f64(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return dptr(a)

        d.xm1, d.ym1 = f64(mesh.x), f64(mesh.y)
        d.zm1 = f64(mesh.z) if mesh.dim == 3 else None
        g = np.ascontiguousarray(mesh.glo_num, dtype=np.int64)
        keep.append(g)
        d.glo_num = g.ctypes.data_as(_lib.c_int64_p)
        if mesh.elem_gid is not None:
            eg = np.ascontiguousarray(mesh.elem_gid, dtype=np.int64)
            keep.append(eg)
            d.lglel = eg.ctypes.data_as(_lib.c_int64_p)
        d.v1mask, d.v2mask = f64(mesh.mask[0]), f64(mesh.mask[1])
        d.v3mask = f64(mesh.mask[2]) if mesh.dim == 3 else None
        d.tmask = f64(mesh.tmask)
        d.has_outflow = int(mesh.has_outflow)
        h = vp()
        check(self.lib.nlg_mesh_create(ctx.h, C.byref(d), C.byref(h)))
        self.h = h
        lvn, lpn = C.c_int64(), C.c_int64()
        check(self.lib.nlg_mesh_sizes(h, C.byref(lvn), C.byref(lpn), None, None))
        self.lvn, self.lpn = lvn.value, lpn.value
        self.dim, self.n = mesh.dim, mesh.n
        self.E = mesh.E

    def get(self, name: str, on_mesh: int = 1) -> np.ndarray:
        npts = {1: self.lvn, 2: self.lpn}.get(on_mesh, on_mesh)
        out = np.empty(npts)
        check(self.lib.nlg_mesh_get(self.h, name.encode(), dptr(out), npts))
        return out

    def close(self):
        if self.h:
            self.lib.nlg_mesh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class nek_dvector:
    """reference: type nek_dvector (neklab_vectors.f90:26-50).  Fields live in HBM."""

    def __init__(self, mesh: Mesh, nscal: int = 0, lorder: int = 3, _handle=None, _owns=True):
        self.mesh, self.lib = mesh, mesh.lib
        self.nscal, self.lorder = nscal, lorder
        self._owns = _owns
        if _handle is None:
            h = vp()
            check(self.lib.nlg_vec_create(mesh.h, nscal, lorder, C.byref(h)))
            self.h = h
        else:
            self.h = _handle

    # ---- the six deferred procedures of abstract_vector_rdp (neklab_vectors.f90:39-44)
    def zero(self):
        check(self.lib.nlg_vec_zero(self.h))

    def rand(self, ifnorm: bool = False, seed: int = 0):
        check(self.lib.nlg_vec_rand(self.h, int(bool(ifnorm)), int(seed)))

    def scal(self, alpha: float):
        check(self.lib.nlg_vec_scal(self.h, float(alpha)))

    def axpby(self, alpha: float, vec: "nek_dvector", beta: float):
        """self = alpha*vec + beta*self  (argument order of nek_daxpby with pass(self))."""
        if not isinstance(vec, nek_dvector):
            raise TypeError("type_error: vec must be nek_dvector (reference: real_vectors.f90:202-204)")
        check(self.lib.nlg_vec_axpby(float(alpha), vec.h, float(beta), self.h))

    def dot(self, vec: "nek_dvector") -> float:
        if not isinstance(vec, nek_dvector):
            raise TypeError("type_error: vec must be nek_dvector (reference: real_vectors.f90:229-231)")
        out = C.c_double()
        check(self.lib.nlg_vec_dot(self.h, vec.h, C.byref(out)))
        return out.value

    def get_size(self) -> int:
        out = C.c_int64()
        check(self.lib.nlg_vec_size(self.h, C.byref(out)))
        return out.value

    # ---- inherited helpers LightKrylov provides on top of the deferred set
    def norm(self) -> float:
        out = C.c_double()
        check(self.lib.nlg_vec_norm(self.h, C.byref(out)))
        return out.value

    def sub(self, vec):
        self.axpby(-1.0, vec, 1.0)

    def add(self, vec):
        self.axpby(1.0, vec, 1.0)

    # ---- neklab-specific restart history (neklab_vectors.f90:46-49)
    def save_rst(self, vec_rst: "nek_dvector", irst: int):
        check(self.lib.nlg_vec_save_rst(self.h, vec_rst.h, int(irst)))

    def get_rst(self, vec_rst: "nek_dvector", irst: int):
        check(self.lib.nlg_vec_get_rst(self.h, vec_rst.h, int(irst)))

    def has_rst_fields(self) -> bool:
        out = C.c_int()
        check(self.lib.nlg_vec_has_rst_fields(self.h, C.byref(out)))
        return bool(out.value)

    def clear_rst_fields(self):
        check(self.lib.nlg_vec_clear_rst_fields(self.h))

    @property
    def nrst(self) -> int:
        out = C.c_int()
        check(self.lib.nlg_vec_nrst(self.h, C.byref(out)))
        return out.value

    # ---- Fortran assignment semantics / host transfer
    def copy(self) -> "nek_dvector":
        h = vp()
        check(self.lib.nlg_vec_clone(self.h, C.byref(h)))
        return nek_dvector(self.mesh, self.nscal, self.lorder, _handle=h)

    def assign(self, other: "nek_dvector"):
        check(self.lib.nlg_vec_copy(self.h, other.h))

    def set_field(self, field: int, data, irst: int = 0):
        a = np.ascontiguousarray(data, dtype=np.float64).reshape(-1)
        check(self.lib.nlg_vec_set_field(self.h, int(field), int(irst), dptr(a), a.size))

    def get_field(self, field: int, irst: int = 0) -> np.ndarray:
        n = self.mesh.lpn if field == PR else self.mesh.lvn
        out = np.empty(n)
        check(self.lib.nlg_vec_get_field(self.h, int(field), int(irst), dptr(out), n))
        return out

    def close(self):
        if self.h and self._owns:
            self.lib.nlg_vec_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def nek2vec(vec: nek_dvector, vx, vy, vz=None, pr=None, t=None):
    """reference: neklab_utils.f90:84-108 (nopcopy :279-301): host fields -> vector."""
    vec.set_field(VX, vx)
    vec.set_field(VY, vy)
    if vec.mesh.dim == 3 and vz is not None:
        vec.set_field(VZ, vz)
    if pr is not None:
        vec.set_field(PR, pr)
    if t is not None:
        for m, tm in enumerate(t[: vec.nscal]):
            vec.set_field(THETA + m, tm)


def vec2nek(vec: nek_dvector):
    """reference: neklab_utils.f90:110-134: vector -> host fields (vx, vy, vz|None, pr, [theta])."""
    vz = vec.get_field(VZ) if vec.mesh.dim == 3 else None
    return (vec.get_field(VX), vec.get_field(VY), vz, vec.get_field(PR),
            [vec.get_field(THETA + m) for m in range(vec.nscal)])


class KrylovBasis:
    """Contiguous array of nek_dvector (what LightKrylov allocates as X(kdim+1)) with block kernels."""

    def __init__(self, mesh: Mesh, nvec: int, nscal: int = 0, lorder: int = 3):
        self.mesh, self.lib, self.nvec = mesh, mesh.lib, nvec
        h = vp()
        check(self.lib.nlg_basis_create(mesh.h, nscal, lorder, nvec, C.byref(h)))
        self.h = h
        self.nscal, self.lorder = nscal, lorder

    def __getitem__(self, i: int) -> nek_dvector:
        h = vp()
        check(self.lib.nlg_basis_vec(self.h, int(i), C.byref(h)))
        return nek_dvector(self.mesh, self.nscal, self.lorder, _handle=h, _owns=False)

    def block_dot(self, k: int, w: nek_dvector) -> np.ndarray:
        out = np.empty(k)
        check(self.lib.nlg_basis_block_dot(self.h, k, w.h, dptr(out)))
        return out

    def block_axpy(self, k: int, h, w: nek_dvector):
        a = np.ascontiguousarray(h, dtype=np.float64)
        check(self.lib.nlg_basis_block_axpy(self.h, k, dptr(a), w.h))

    def cgs2(self, k: int, w: nek_dvector):
        h = np.empty(max(k, 1))
        beta = C.c_double()
        check(self.lib.nlg_basis_cgs2(self.h, k, w.h, dptr(h), C.byref(beta)))
        return h[:k], beta.value

    def block_cgs2(self, k: int, s: int) -> np.ndarray:
        """Columns k .. k+s-1 orthogonalised against 0 .. k-1 and among themselves (nlg_basis_block_cgs2); returns the
        (k+s, s) coefficient matrix: projection coefficients on top, the upper-triangular R below."""
        coef = np.zeros((k + s, s), order="F")
        check(self.lib.nlg_basis_block_cgs2(self.h, int(k), int(s), dptr(coef)))
        return coef

    def last_block_rank(self) -> int:
        """columns the last block_cgs2 kept (< s: numerically dependent columns were deflated to zero vectors)"""
        r = C.c_int()
        check(self.lib.nlg_basis_last_block_rank(self.h, C.byref(r)))
        return r.value

    def combine(self, k: int, c, out: nek_dvector):
        a = np.ascontiguousarray(c, dtype=np.float64)
        check(self.lib.nlg_basis_combine(self.h, k, dptr(a), out.h))

    def close(self):
        if self.h:
            self.lib.nlg_basis_destroy(self.h)
            self.h = None


class exptA_linop:
    """reference: type exptA_linop (neklab_linops.f90:35-44); constructor idiom `exptA_linop(tau, bf)`
    (examples/cylinder/stability/direct/1cyl.usr:20)."""

    def __init__(self, tau: float, baseflow: nek_dvector, **cfg):
        self.mesh, self.lib = baseflow.mesh, baseflow.lib
        c = _lib.ExptAConfig()
        check(self.lib.nlg_exptA_config_default(C.byref(c)))
        c.tau = float(tau)
        for k, v in cfg.items():
            if not hasattr(c, k):
                raise TypeError("unknown exptA option %r" % k)
            if k == "buoy":
                v = (C.c_double * 3)(*[float(a) for a in tuple(v) + (0.0,) * (3 - len(tuple(v)))])
            setattr(c, k, v)
        self.cfg = c
        self.baseflow = baseflow
        h = vp()
        check(self.lib.nlg_linop_create(self.mesh.h, C.byref(c), baseflow.h, C.byref(h)))
        self.h = h

    @property
    def tau(self) -> float:
        return self.info()["tau"]

    @tau.setter
    def tau(self, v: float):
        check(self.lib.nlg_linop_set_tau(self.h, float(v)))

    def init(self):
        check(self.lib.nlg_linop_init(self.h))

    def matvec(self, vec_in: nek_dvector, vec_out: nek_dvector):
        if not isinstance(vec_in, nek_dvector) or not isinstance(vec_out, nek_dvector):
            raise TypeError("type_error: nek_dvector expected (reference: exponential_propagator.f90:53-58)")
        check(self.lib.nlg_linop_matvec(self.h, vec_in.h, vec_out.h))

    def rmatvec(self, vec_in: nek_dvector, vec_out: nek_dvector):
        if not isinstance(vec_in, nek_dvector) or not isinstance(vec_out, nek_dvector):
            raise TypeError("type_error: nek_dvector expected (reference: exponential_propagator.f90:100-105)")
        check(self.lib.nlg_linop_rmatvec(self.h, vec_in.h, vec_out.h))

    def matvec_block(self, vecs_in: list, vecs_out: list, transpose: bool = False):
        """len(vecs_in) <= 4 vectors advanced together (nlg_linop_matvec_block)."""
        s = len(vecs_in)
        assert len(vecs_out) == s
        ai = (vp * s)(*[x.h for x in vecs_in])
        ao = (vp * s)(*[x.h for x in vecs_out])
        check(self.lib.nlg_linop_matvec_block(self.h, s, ai, ao, int(bool(transpose))))

    def info(self) -> dict:
        tau, dt, cfl, ns = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        check(self.lib.nlg_linop_get_info(self.h, C.byref(tau), C.byref(dt), C.byref(ns), C.byref(cfl)))
        return {"tau": tau.value, "dt": dt.value, "nsteps": ns.value, "cfl": cfl.value}

    def stats(self) -> dict:
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.nlg_linop_get_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"steps": a.value, "v_iters": b.value, "p_iters": c.value, "matvecs": d.value}

    def close(self):
        if self.h:
            self.lib.nlg_linop_destroy(self.h)
            self.h = None


# ------------------------------------------------------------------------------------------------------------------
# Resolvent operator by time stepping.  Reference: resolvent_linop (src/linops/neklab_linops.f90:198-205,
# src/linops/resolvent.f90) acting on nek_zvector = (re, im) pairs of nek_dvector (src/vectors/neklab_vectors.f90).
class nek_zvector:
    """Complex state as a pair of real vectors (the reference's nek_zvector holds `re` and `im` of type nek_dvector)."""

    def __init__(self, mesh: "Mesh"):
        self.mesh = mesh
        self.re, self.im = nek_dvector(mesh), nek_dvector(mesh)

    def zero(self):
        self.re.zero()
        self.im.zero()

    def axpby(self, alpha: complex, vec: "nek_zvector", beta: complex):
        """self = alpha * vec + beta * self with complex scalars."""
        a, b = complex(alpha), complex(beta)
        re, im = self.re.copy(), self.im.copy()
        self.re.scal(b.real)
        self.re.axpby(-b.imag, im, 1.0)
        self.re.axpby(a.real, vec.re, 1.0)
        self.re.axpby(-a.imag, vec.im, 1.0)
        self.im.scal(b.real)
        self.im.axpby(b.imag, re, 1.0)
        self.im.axpby(a.real, vec.im, 1.0)
        self.im.axpby(a.imag, vec.re, 1.0)

    def dot(self, vec: "nek_zvector") -> complex:
        """<self, vec> = sum conj(self) vec in the mass-weighted velocity inner product."""
        return complex(self.re.dot(vec.re) + self.im.dot(vec.im), self.re.dot(vec.im) - self.im.dot(vec.re))

    def norm(self) -> float:
        return float(np.sqrt(self.re.dot(self.re) + self.im.dot(self.im)))


def integrate_forced(exptA: exptA_linop, ic, f_re: nek_dvector, f_im, omega: float, adjoint: bool, out: nek_dvector):
    check(exptA.lib.nlg_linop_integrate_forced(exptA.h, ic.h if ic is not None else None, f_re.h,
                                               f_im.h if f_im is not None else None, float(omega), int(bool(adjoint)), out.h))


class resolvent_linop:
    """R(omega) f: the time-periodic response to the forcing Re[f exp(i omega t)] about `baseflow`, by time stepping
    (resolvent_matvec, src/linops/resolvent.f90:17-44): b = one period from rest under the forcing (evaluate_rhs),
    Re x from (I - exp(T L)) x = b by GMRES(64), rtol 1e-6 (solve_resolvent_real_part, :113-131), Im part = the state a
    quarter period later (evaluate_imaginary_part).  `rmatvec` integrates the adjoint equations with exp(-i omega t)."""

    def __init__(self, omega: float, baseflow: nek_dvector, **cfg):
        self.omega, self.baseflow, self.cfg = float(omega), baseflow, dict(cfg)
        self.mesh = baseflow.mesh
        self.gmres_matvecs = 0

    def _apply(self, vin: nek_zvector, vout: nek_zvector, adjoint: bool):
        tau = 1.0 if self.omega == 0.0 else 2.0 * np.pi / abs(self.omega)
        A = exptA_linop(tau, self.baseflow, **self.cfg)
        A.init()
        b = nek_dvector(self.mesh)
        integrate_forced(A, None, vin.re, vin.im, self.omega, adjoint, b)
        # (I - exp(T L)) x = b  <=>  (exp(T L) - I) x = -b
        rhs = b.copy()
        rhs.scal(-1.0)
        x = nek_dvector(self.mesh)
        res, nmv = gmres(A, rhs, x, atol=max(1.0e-6 * b.norm(), 1.0e-12), kdim=64, transpose=adjoint)
        self.gmres_matvecs += nmv
        x.clear_rst_fields()
        vout.re.assign(x)
        A.tau = tau / 4.0                      # exptA%tau = tau/4; call exptA%init(), resolvent.f90:35
        integrate_forced(A, x, vin.re, vin.im, self.omega, adjoint, vout.im)
        self.last_operator = A
        return res

    def matvec(self, vin: nek_zvector, vout: nek_zvector):
        return self._apply(vin, vout, False)

    def rmatvec(self, vin: nek_zvector, vout: nek_zvector):
        return self._apply(vin, vout, True)


def line_labels(mesh: "Mesh", idir: int, decimals: int = 9) -> np.ndarray:
    """One label per line of velocity points along direction idir (1-based): points with the same remaining coordinates
    (rounded) share it.  What Nek5000's gtpp_gs_setup derives from (nelx, nely, nelz) for extruded box meshes
    (init_exptA_proj, src/linops/exponential_propagator_proj.f90:18-22); needs a mesh that is not deformed along idir."""
    hm = mesh.host
    coords = [hm.x, hm.y] + ([hm.z] if hm.dim == 3 else [])
    others = [np.round(np.asarray(c).ravel(), decimals) for d, c in enumerate(coords) if d != idir - 1]
    key = np.stack(others, axis=1)
    _, lab = np.unique(key, axis=0, return_inverse=True)
    return np.ascontiguousarray(lab.ravel(), dtype=np.int64)


def _gll_to_gl_matrix(n: int) -> np.ndarray:
    """(n-2 x n) Lagrange interpolation from the GLL velocity points to the Gauss-Legendre pressure points."""
    from .mesh import gll_points
    z1 = gll_points(n)
    z2 = np.polynomial.legendre.leggauss(n - 2)[0]
    M = np.ones((n - 2, n))
    for k in range(n):
        for l in range(n):
            if l != k:
                M[:, k] *= (z2 - z1[l]) / (z1[k] - z1[l])
    return M


def pressure_mesh_coords(mesh: "Mesh"):
    """Coordinates of the pressure (GL) points: the isoparametric map evaluated there."""
    hm = mesh.host
    n, dim, E = hm.n, hm.dim, hm.E
    M = _gll_to_gl_matrix(n)
    out = []
    for c in [hm.x, hm.y] + ([hm.z] if dim == 3 else []):
        a = np.asarray(c).reshape((E,) + (n,) * dim)
        for ax in range(1, dim + 1):
            a = np.moveaxis(np.tensordot(M, a, axes=([1], [ax])), 0, ax)
        out.append(a.reshape(-1))
    return out


def line_labels_pressure(mesh: "Mesh", idir: int, decimals: int = 9):
    """(labels, coordinate along idir) of the pressure points, see line_labels."""
    X2 = pressure_mesh_coords(mesh)
    key = np.stack([np.round(c, decimals) for d, c in enumerate(X2) if d != idir - 1], axis=1)
    _, lab = np.unique(key, axis=0, return_inverse=True)
    return np.ascontiguousarray(lab.ravel(), dtype=np.int64), np.ascontiguousarray(X2[idir - 1], dtype=np.float64)


class exptA_proj_linop(exptA_linop):
    """reference: exptA_proj_linop(tau=, baseflow=, alpha=) (src/linops/neklab_linops.f90:130-152; constructed with
    keywords at examples/poiseuille/stability/direct_alpha_1/poiseuille.usr:24): the propagator restricted to the
    streamwise wavenumber alpha by projecting the initial condition and the final state."""

    def __init__(self, tau: float, baseflow: nek_dvector, alpha: float, idir: int = 1, project_pressure: bool = True, **cfg):
        super().__init__(tau, baseflow, **cfg)
        self.alpha, self.idir, self.project_pressure = float(alpha), int(idir), bool(project_pressure)

    def init(self):
        super().init()
        lab = line_labels(self.mesh, self.idir)
        if self.project_pressure:
            lab2, x2 = line_labels_pressure(self.mesh, self.idir)
            check(self.lib.nlg_linop_set_projection(self.h, self.alpha, self.idir, lab.ctypes.data_as(_lib.c_int64_p),
                                                    lab2.ctypes.data_as(_lib.c_int64_p), x2.ctypes.data_as(_lib.c_double_p)))
        else:
            check(self.lib.nlg_linop_set_projection(self.h, self.alpha, self.idir, lab.ctypes.data_as(_lib.c_int64_p), None, None))

    def proj(self, vec: nek_dvector):
        check(self.lib.nlg_linop_project(self.h, vec.h))


# ------------------------------------------------------------------------------------------------------------------
# Newton-Krylov fixed-point solver.  Reference: nek_system / nek_jacobian
# (src/systems/neklab_systems.f90, fixed_point.f90:4-96) driven by LightKrylov's `newton` + `gmres_rdp`
# (src/neklab_analysis.f90:158-205).  LightKrylov is absent from the reference tree, so Newton and GMRES are restated
# from the published algorithms (Saad & Schultz 1986: restarted GMRES with Givens rotations; inexact Newton with the
# reference's own tolerance schedulers) -- parity at LightKrylov's internals is unpinned, as for `eigs`.
class nek_system:
    """`eval` = nonlinear_map: F(X) = Phi_tau(X) - X (fixed_point.f90:4-38); `jacobian` = exp(tau J(X)) - I about the
    frozen state X (fixed_point.f90:40-96).  Two operator objects, as the reference switches Nek5000 between its
    nonlinear (CFL limit 0.4) and linearised (0.5) set-ups."""

    def __init__(self, tau: float, X0: nek_dvector, re: float, torder: int = 3, **cfg):
        self.mesh, self.lib, self.tau = X0.mesh, X0.lib, float(tau)
        self.nl = exptA_linop(tau, X0, re=re, torder=torder, cfl_limit=0.4, **cfg)
        self.jac = exptA_linop(tau, X0, re=re, torder=torder, cfl_limit=0.5, **cfg)
        self.nl.init()
        self.jac.init()
        self.n_eval = 0

    def set_tolerance(self, tol: float):
        """nek_constant_tol / nek_dynamic_tol set param(21) = param(22) = tol; nonlinear_map then solves to tol * 0.1
        (fixed_point.f90:16-17), the Jacobian to tol * 0.5 (:58-59)."""
        check(self.lib.nlg_linop_set_tolerances(self.nl.h, 0.1 * tol, 0.1 * tol))
        check(self.lib.nlg_linop_set_tolerances(self.jac.h, 0.5 * tol, 0.5 * tol))

    def eval(self, X: nek_dvector, out: nek_dvector):
        check(self.lib.nlg_linop_nonlinear_map(self.nl.h, X.h, out.h))
        self.n_eval += 1

    def set_jacobian_state(self, X: nek_dvector):
        check(self.lib.nlg_linop_set_baseflow(self.jac.h, X.h))


def nek_dynamic_tol(tol_old: float, target: float, rnorm: float):
    """neklab_systems.f90:273-335: tol = max(0.1 rnorm, target), snapped to the target when within a decade of it,
    capped at 1e-4; the target itself is clipped to [10 atol_dp, 1e-4]."""
    maxtol, mintol = 1.0e-4, 10.0 * 10.0 ** -12
    target = min(max(target, mintol), maxtol)
    tol = max(0.1 * rnorm, target)
    if tol < 10.0 * target:
        tol = target
    return min(tol, maxtol)


def nek_constant_tol(tol_old: float, target: float, rnorm: float):
    """neklab_systems.f90:229-261."""
    return max(target, 10.0 * 10.0 ** -12)


def gmres(exptA: exptA_linop, b: nek_dvector, x: nek_dvector, atol: float, kdim: int = 30, maxiter: int = 10,
          shift: float = -1.0, basis: "KrylovBasis | None" = None, replay_history: bool = False, transpose: bool = False,
          history: "list | None" = None):
    """Restarted GMRES(kdim) for (A + shift I) x = b, zero initial guess, stop at |residual| <= atol.  The Krylov space of
    A + shift I is that of A, so the Arnoldi relation comes from the device Arnoldi step of A (`nlg_arnoldi_step`:
    matvec + block CGS2) with `shift` added to the diagonal of H.  Returns (residual norm, number of matvecs).

    history (a list): receives the residual norm at the start and after every inner step (what LightKrylov logs as
    "GMRES(k) init step" / "inner step").
    replay_history = False strips the restart history from every new Krylov vector, so that each matvec starts
    impulsively like the nonlinear map whose Jacobian it stands for.  The reference's jac_exptA_matvec replays the
    history (fixed_point.f90:73); measured on a lid-driven cavity (scripts/newton_cavity_oracle.py, tau = 0.4): Newton
    needs 17 iterations with the replay (the Krylov operator is then the start-up-free continuation map, not the
    derivative of the map being solved) and 3 without.  The converged fixed point is the same."""
    mesh = b.mesh
    B = basis if basis is not None else KrylovBasis(mesh, kdim + 1, b.nscal, b.lorder)
    x.zero()
    r = b.copy()
    nmv = 0
    res = r.norm()
    if history is not None:
        history.append(res)
    for _ in range(maxiter):
        beta = res
        if beta <= atol:
            break
        v0 = B[0]
        v0.assign(r)
        v0.scal(1.0 / beta)
        H = np.zeros((kdim + 2, kdim + 1), order="F")
        R = np.zeros((kdim + 1, kdim))
        cs, sn = np.zeros(kdim), np.zeros(kdim)
        g = np.zeros(kdim + 1)
        g[0] = beta
        k = 0
        while k < kdim:
            arnoldi_step(exptA, B, k, H, transpose)
            nmv += 1
            if not replay_history:
                B[k + 1].clear_rst_fields()
            h = H[: k + 2, k].copy()
            h[k] += shift
            for i in range(k):                                   # previous rotations
                t = cs[i] * h[i] + sn[i] * h[i + 1]
                h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1]
                h[i] = t
            d = np.hypot(h[k], h[k + 1])
            cs[k], sn[k] = (1.0, 0.0) if d == 0.0 else (h[k] / d, h[k + 1] / d)
            h[k], h[k + 1] = d, 0.0
            R[: k + 1, k] = h[: k + 1]
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            k += 1
            res = abs(g[k])
            if history is not None:
                history.append(res)
            if res <= atol:
                break
        y = np.linalg.solve(np.triu(R[:k, :k]), g[:k])
        dx = nek_dvector(mesh, b.nscal, b.lorder)
        B.combine(k, y, dx)
        x.axpby(1.0, dx, 1.0)
        if res <= atol:
            break
        # true residual for the restart: r = b - (A + shift I) x
        Ax = nek_dvector(mesh, b.nscal, b.lorder)
        (exptA.rmatvec if transpose else exptA.matvec)(x, Ax)
        nmv += 1
        r.assign(b)
        r.axpby(-1.0, Ax, 1.0)
        r.axpby(-shift, x, 1.0)
        res = r.norm()
    return res, nmv


def newton_fixed_point_iteration(sys: nek_system, bf: nek_dvector, tol: float, tol_mode: int = 1, maxiter: int = 40,
                                 kdim: int = 30, outdir: str | None = None, session: str = "neklab", log=None,
                                 replay_history: bool = False):
    """reference: newton_fixed_point_iteration (src/neklab_analysis.f90:158-205): Newton on F(X) = Phi_tau(X) - X with
    GMRES on the Jacobian exp(tau J) - I, tolerance scheduler nek_constant_tol (tol_mode 1) or nek_dynamic_tol, 40
    iterations at most, no bisection; on convergence the fixed point is written as `nwt<session>0.f00001`.
    `bf` is updated in place.  Returns dict(converged, iterations, residuals, gmres_matvecs, evals)."""
    sched = nek_constant_tol if tol_mode == 1 else nek_dynamic_tol
    mesh = bf.mesh
    r = nek_dvector(mesh, bf.nscal, bf.lorder)
    dx = nek_dvector(mesh, bf.nscal, bf.lorder)
    B = KrylovBasis(mesh, kdim + 1, bf.nscal, bf.lorder)     # (with the scalar of a temperature-coupled system)
    final = sched(0.0, tol, 0.0)               # the tightest level the scheduler will ever set
    cur, rnorm = 0.0, 1.0
    residuals, nmv_total, converged = [], 0, False
    gmres_hist = []
    for it in range(maxiter + 1):
        new = sched(cur, tol, rnorm)           # scheduler first, as LightKrylov's newton calls it at the top of an iteration
        if new != cur:
            cur = new
            sys.set_tolerance(cur)
        sys.eval(bf, r)
        rnorm = r.norm()
        residuals.append(rnorm)
        if log is not None:
            log("newton %2d  |F(X)| = %.6e  solver tol %.3e" % (it, rnorm, cur))
        if rnorm < tol and cur <= final:       # a residual below the target only counts when computed at the final level
            converged = True
            break
        if it == maxiter:
            break
        sys.set_jacobian_state(bf)
        r.scal(-1.0)
        gh = []
        res, nmv = gmres(sys.jac, r, dx, atol=sched(cur, tol, rnorm), kdim=kdim, basis=B, replay_history=replay_history, history=gh)
        gmres_hist.append(gh)
        nmv_total += nmv
        bf.axpby(1.0, dx, 1.0)
    if converged and outdir is not None:
        outpost_dnek(bf, "nwt", session, outdir)
    return {"converged": converged, "iterations": len(residuals) - 1, "residuals": residuals, "gmres_matvecs": nmv_total,
            "evals": sys.n_eval, "gmres_residuals": gmres_hist}


def arnoldi_step(exptA: exptA_linop, basis: KrylovBasis, k: int, H: np.ndarray, transpose: bool = False):
    """H is Fortran-ordered (kdim+1, kdim)."""
    assert H.flags.f_contiguous
    check(exptA.lib.nlg_arnoldi_step(exptA.h, basis.h, int(k), dptr(H), H.shape[0], int(bool(transpose))))


def block_arnoldi_step(exptA: exptA_linop, basis: KrylovBasis, k: int, s: int, H: np.ndarray, transpose: bool = False):
    """Columns k .. k+s-1 -> k+s .. k+2s-1; H (Fortran-ordered) receives H[0:k+2s, k:k+s]."""
    assert H.flags.f_contiguous
    check(exptA.lib.nlg_block_arnoldi_step(exptA.h, basis.h, int(k), int(s), dptr(H), H.shape[0], int(bool(transpose))))


def eigs(exptA: exptA_linop, X: list, kdim: int = 0, tol: float = 0.0, x0: nek_dvector | None = None,
         transpose: bool = False, write_intermediate: bool = True, logfile: str | None = None, seed: int = 0,
         max_restarts: int = 50, block_size: int = 0, warm_start: bool = False):
    """reference call: eigs(exptA, eigvecs, eigvals, residuals, info, x0=, kdim=, transpose=,
    write_intermediate=) at neklab_analysis.f90:80-81.  Returns (eigvals complex[nev], residuals, info)."""
    lib = exptA.lib
    nev = len(X)
    o = _lib.EigsOpts()
    check(lib.nlg_eigs_opts_default(C.byref(o)))
    o.kdim, o.transpose, o.write_intermediate, o.tol, o.seed = int(kdim), int(bool(transpose)), int(bool(write_intermediate)), float(tol), int(seed)
    o.max_restarts = int(max_restarts)
    o.block_size, o.warm_start = int(block_size), int(bool(warm_start))
    if logfile is not None:
        o.logfile = logfile.encode()
    re, im, res = np.zeros(nev), np.zeros(nev), np.zeros(nev)
    info = C.c_int()
    arr = (vp * nev)(*[x.h for x in X])
    check(lib.nlg_eigs(exptA.h, arr, nev, dptr(re), dptr(im), dptr(res), C.byref(info), x0.h if x0 is not None else None,
                       C.byref(o)))
    return re + 1j * im, res, info.value


def svds(exptA: exptA_linop, U: list, V: list, kdim: int = 0, tol: float = 0.0, u0: nek_dvector | None = None,
         write_intermediate: bool = True, logfile: str | None = None, seed: int = 0):
    """reference call: svds(exptA, U, S, V, residuals, info, kdim=, write_intermediate=) at neklab_analysis.f90:136.
    Returns (S[nsv], residuals, info)."""
    lib = exptA.lib
    nsv = len(U)
    assert len(V) == nsv
    o = _lib.EigsOpts()
    check(lib.nlg_eigs_opts_default(C.byref(o)))
    o.kdim, o.write_intermediate, o.tol, o.seed = int(kdim), int(bool(write_intermediate)), float(tol), int(seed)
    if logfile is not None:
        o.logfile = logfile.encode()
    S, res = np.zeros(nsv), np.zeros(nsv)
    info = C.c_int()
    au = (vp * nsv)(*[x.h for x in U])
    av = (vp * nsv)(*[x.h for x in V])
    check(lib.nlg_svds(exptA.h, au, av, nsv, dptr(S), dptr(res), C.byref(info), u0.h if u0 is not None else None, C.byref(o)))
    return S, res, info.value


def transient_growth_analysis_fixed_point(exptA: exptA_linop, nsv: int, kdim: int, tol: float = 0.0, outdir: str = ".",
                                          seed: int = 0, outpost: bool = False, session: str = "neklab"):
    """reference: neklab_analysis.f90:107-156.  Returns (S, residuals, U (optimal responses), V (optimal
    perturbations), info) and writes singular_spectrum.dat (:139-143)."""
    mesh = exptA.mesh
    U = [nek_dvector(mesh, 0, 3) for _ in range(nsv)]
    V = [nek_dvector(mesh, 0, 3) for _ in range(nsv)]
    S, residuals, info = svds(exptA, U, V, kdim=kdim, tol=tol, write_intermediate=True,
                              logfile=os.path.join(outdir, "svds_output.txt"), seed=seed)
    with open(os.path.join(outdir, "singular_spectrum.dat"), "w") as f:
        f.write(" ".join("%.16e" % s for s in S) + "\n")
    if outpost:                                                      # :146-147
        outpost_dnek(V, "prt", session, outdir)
        outpost_dnek(U, "rsp", session, outdir)
    return S, residuals, U, V, info


def _gl_to_gll_matrix(n: int) -> np.ndarray:
    """(n x n-2) Lagrange interpolation from the Gauss-Legendre pressure points to the GLL velocity points."""
    from .mesh import gll_points
    z1 = gll_points(n)
    z2 = np.polynomial.legendre.leggauss(n - 2)[0]
    M = np.ones((n, n - 2))
    for k in range(n - 2):
        for l in range(n - 2):
            if l != k:
                M[:, k] *= (z1 - z2[l]) / (z2[k] - z2[l])
    return M


def pressure_from_mesh1(mesh: "Mesh", p1) -> np.ndarray:
    """A pressure given on the velocity mesh (what a field file holds) on the pressure mesh: tensor interpolation GLL -> GL inside
    every element, as Nek5000's restart does for a Pn-Pn-2 run (`map_pm1_to_pr`).  Returns the flat array for set_field(PR, .)."""
    n, dim, E = mesh.host.n, mesh.host.dim, mesh.host.E
    M = _gll_to_gl_matrix(n)
    p = np.asarray(p1, dtype=np.float64).reshape((E,) + (n,) * dim)
    for ax in range(1, dim + 1):
        p = np.moveaxis(np.tensordot(M, p, axes=([1], [ax])), 0, ax)
    return np.ascontiguousarray(p.reshape(-1))


def pressure_to_mesh1(vec: nek_dvector) -> np.ndarray:
    """Pressure of `vec` on the velocity mesh, as Nek5000's outpost writes it for a Pn-Pn-2 run (`mappr`: tensor
    interpolation GL -> GLL inside every element, then the direct-stiffness average across elements)."""
    mesh = vec.mesh
    n, dim, E = mesh.host.n, mesh.host.dim, mesh.host.E
    M = _gl_to_gll_matrix(n)
    p = vec.get_field(PR).reshape((E,) + (n - 2,) * dim)
    for ax in range(1, dim + 1):
        p = np.moveaxis(np.tensordot(M, p, axes=([1], [ax])), 0, ax)
    tmp = nek_dvector(mesh)
    tmp.set_field(VX, p.reshape(-1))
    check(mesh.lib.nlg_op_dssum(mesh.h, tmp.h))
    return tmp.get_field(VX) * mesh.get("vmult")


def outpost_dnek(vecs, prefix: str, session: str = "neklab", outdir: str = ".", first_index: int = 1, time: float = 0.0):
    """reference: outpost_dnek (src/neklab_utils.f90:305-333, called at neklab_analysis.f90:93,146,147): every vector
    becomes one Nek5000 field file `<prefix><session>0.f%05d` with the GLL coordinates in the first file only (Nek5000's
    default `ifxyo` behaviour), velocity and the pressure mapped to the velocity mesh.  Single-rank writer."""
    from . import nekio
    if isinstance(vecs, nek_dvector):
        vecs = [vecs]
    if len(prefix) != 3:
        raise ValueError("outpost_dnek: the prefix has three characters in Nek5000 (got %r)" % prefix)
    paths = []
    for i, v in enumerate(vecs):
        hm = v.mesh.host
        n, dim, E = hm.n, hm.dim, hm.E
        coords = None
        if i == 0:
            coords = [hm.x, hm.y] + ([hm.z] if dim == 3 else [])
            coords = [np.asarray(c).reshape(E, -1) for c in coords]
        vel = [v.get_field(c).reshape(E, -1) for c in range(dim)]
        path = os.path.join(outdir, "%s%s0.f%05d" % (prefix, session, first_index + i))
        nekio.write_fld(path, n, dim, coords=coords, vel=vel, p=pressure_to_mesh1(v).reshape(E, -1), time=time,
                        istep=first_index + i)
        paths.append(path)
    return paths


def save_eigenspectrum(eigvals, residuals, filename: str):
    """reference call site: neklab_analysis.f90:90 (LightKrylov save_eigenspectrum): (n,3) array
    [Re, Im, residual] in .npy format, the layout examples/*/plot_eigenvalues.py reads."""
    data = np.column_stack([np.real(eigvals), np.imag(eigvals), residuals])
    np.save(filename, data)


def linear_stability_analysis_fixed_point(exptA: exptA_linop, kdim: int, nev: int, adjoint: bool = False,
                                          X0: nek_dvector | None = None, tol: float = 0.0, outdir: str = ".",
                                          seed: int = 0, outpost: bool = False, session: str = "neklab",
                                          block_size: int = 0, warm_start: bool = False):
    """reference: neklab_analysis.f90:38-105.  Returns (eigvals continuous-time, residuals, eigvecs)."""
    mesh = exptA.mesh
    eigvecs = [nek_dvector(mesh, 0, 3) for _ in range(nev)]   # lorder = 3 as in every reference SIZE file
    for v in eigvecs:
        v.zero()                                                     # zero_basis, :77
    prefix = "adj" if adjoint else "dir"
    mu, residuals, info = eigs(exptA, eigvecs, kdim=kdim, x0=X0, transpose=adjoint, write_intermediate=True,
                               logfile=os.path.join(outdir, "eigs_output.txt"), tol=tol, seed=seed,
                               block_size=block_size, warm_start=warm_start)
    eigvals = np.log(mu.astype(complex)) / exptA.info()["tau"]       # :84
    save_eigenspectrum(eigvals, residuals, os.path.join(outdir, prefix + "_eigenspectrum.npy"))   # :90
    if outpost:
        outpost_dnek(eigvecs, prefix, session, outdir)               # :93
    return eigvals, residuals, eigvecs, mu, info
