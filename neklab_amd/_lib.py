"""ctypes binding of libneklab_gpu.so -- the C ABI declared in include/neklab_gpu.h.

The product path has NO fallback: if the library is missing or no MI355X is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libneklab_gpu.so")

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_int_p = C.POINTER(C.c_int)
vp = C.c_void_p


class NlgError(RuntimeError):
    pass


class MeshDesc(C.Structure):
    _fields_ = [
        ("dim", C.c_int), ("n", C.c_int), ("lxd", C.c_int), ("nelv", C.c_int64),
        ("xm1", c_double_p), ("ym1", c_double_p), ("zm1", c_double_p),
        ("glo_num", c_int64_p), ("lglel", c_int64_p),
        ("v1mask", c_double_p), ("v2mask", c_double_p), ("v3mask", c_double_p), ("tmask", c_double_p),
        ("has_outflow", C.c_int),
    ]


class ExptAConfig(C.Structure):
    _fields_ = [
        ("tau", C.c_double), ("re", C.c_double), ("cfl_limit", C.c_double), ("vtol", C.c_double),
        ("ptol", C.c_double), ("dt", C.c_double),
        ("torder", C.c_int), ("maxit_v", C.c_int), ("maxit_p", C.c_int),
        ("fixed_iters_v", C.c_int), ("fixed_iters_p", C.c_int), ("pprecond", C.c_int), ("pproj", C.c_int),
        ("ifheat", C.c_int), ("conductivity", C.c_double), ("rhocp", C.c_double), ("buoy", C.c_double * 3),
        ("no_history", C.c_int),
    ]


class EigsOpts(C.Structure):
    _fields_ = [
        ("kdim", C.c_int), ("transpose", C.c_int), ("max_restarts", C.c_int), ("write_intermediate", C.c_int),
        ("tol", C.c_double), ("logfile", C.c_char_p), ("seed", C.c_uint64), ("block_size", C.c_int), ("warm_start", C.c_int),
    ]


# every symbol include/neklab_gpu.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "nlg_last_error": (C.c_char_p, []),
    "nlg_version": (C.c_int, []),
    "nlg_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "nlg_ctx_destroy": (C.c_int, [vp]),
    "nlg_ctx_sync": (C.c_int, [vp]),
    "nlg_comm_unique_id": (C.c_int, [vp]),
    "nlg_ctx_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "nlg_ctx_rank": (C.c_int, [vp, c_int_p, c_int_p]),
    "nlg_ctx_comm_init_shm": (C.c_int, [vp, C.c_int, C.c_int, C.c_char_p, C.c_int64]),
    "nlg_halo_plan": (C.c_int64, [C.c_int, C.c_int, c_int64_p, c_int64_p, c_int64_p, c_int64_p, C.c_int64]),
    "nlg_halo_boundary_labels": (C.c_int64, [C.c_int, C.c_int, C.c_int64, c_int64_p, c_int64_p, C.c_int64]),
    "nlg_halo_lists": (C.c_int64, [C.c_int, C.c_int, C.c_int64, c_int64_p, C.c_int, C.c_int, c_int64_p, c_int64_p, c_int64_p,
                                   C.POINTER(C.c_int32), C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), C.c_int64, c_int64_p]),
    "nlg_prof_enable": (C.c_int, [vp, C.c_int]),
    "nlg_prof_sample": (C.c_int, [vp, C.c_int]),
    "nlg_prof_reset": (C.c_int, [vp]),
    "nlg_prof_get": (C.c_int, [vp, C.c_char_p, c_int64_p, c_double_p]),
    "nlg_counters": (C.c_int, [c_int64_p, c_int64_p]),
    "nlg_basis_last_block_rank": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "nlg_vec_generation": (C.c_int, [vp, c_int64_p]),
    "nlg_vec_release": (C.c_int, [vp]),
    "nlg_vec_adopt": (C.c_int, [vp, C.c_int64, C.POINTER(C.c_int)]),
    "nlg_vec_pin": (C.c_int, [vp, C.c_int64]),
    "nlg_vec_unpin": (C.c_int, [vp, C.c_int64]),
    "nlg_vec_size_checked": (C.c_int64, [vp, C.c_int64]),
    "nlg_vec_has_rst_checked": (C.c_int, [vp, C.c_int64]),
    "nlg_vec_pool_limit": (C.c_int, [C.c_int64]),
    "nlg_vec_pool_trim": (C.c_int, [c_int64_p]),
    "nlg_mesh_create": (C.c_int, [vp, C.POINTER(MeshDesc), C.POINTER(vp)]),
    "nlg_mesh_destroy": (C.c_int, [vp]),
    "nlg_mesh_sizes": (C.c_int, [vp, c_int64_p, c_int64_p, c_int_p, c_int_p]),
    "nlg_mesh_get": (C.c_int, [vp, C.c_char_p, c_double_p, C.c_int64]),
    "nlg_vec_create": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "nlg_vec_destroy": (C.c_int, [vp]),
    "nlg_vec_clone": (C.c_int, [vp, C.POINTER(vp)]),
    "nlg_vec_copy": (C.c_int, [vp, vp]),
    "nlg_vec_zero": (C.c_int, [vp]),
    "nlg_vec_rand": (C.c_int, [vp, C.c_int, C.c_uint64]),
    "nlg_vec_rand_noise": (C.c_int, [vp, C.c_uint64]),
    "nlg_vec_size_value": (C.c_int64, [vp]),
    "nlg_vec_has_rst_value": (C.c_int, [vp]),
    "nlg_basis_block_cgs2": (C.c_int, [vp, C.c_int, C.c_int, c_double_p]),
    "nlg_linop_matvec_block": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int]),
    "nlg_block_arnoldi_step": (C.c_int, [vp, vp, C.c_int, C.c_int, c_double_p, C.c_int, C.c_int]),
    "nlg_vec_outpost": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_double, C.c_int]),
    "nlg_vec_rand_finish": (C.c_int, [vp, C.c_int]),
    "nlg_vec_scal": (C.c_int, [vp, C.c_double]),
    "nlg_vec_axpby": (C.c_int, [C.c_double, vp, C.c_double, vp]),
    "nlg_vec_dot": (C.c_int, [vp, vp, c_double_p]),
    "nlg_vec_norm": (C.c_int, [vp, c_double_p]),
    "nlg_vec_size": (C.c_int, [vp, c_int64_p]),
    "nlg_vec_save_rst": (C.c_int, [vp, vp, C.c_int]),
    "nlg_vec_get_rst": (C.c_int, [vp, vp, C.c_int]),
    "nlg_vec_has_rst_fields": (C.c_int, [vp, c_int_p]),
    "nlg_vec_clear_rst_fields": (C.c_int, [vp]),
    "nlg_vec_nrst": (C.c_int, [vp, c_int_p]),
    "nlg_vec_set_field": (C.c_int, [vp, C.c_int, C.c_int, c_double_p, C.c_int64]),
    "nlg_vec_get_field": (C.c_int, [vp, C.c_int, C.c_int, c_double_p, C.c_int64]),
    "nlg_set_axpby_rst_consistent": (C.c_int, [C.c_int]),
    "nlg_basis_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "nlg_basis_destroy": (C.c_int, [vp]),
    "nlg_basis_vec": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "nlg_basis_block_dot": (C.c_int, [vp, C.c_int, vp, c_double_p]),
    "nlg_basis_block_axpy": (C.c_int, [vp, C.c_int, c_double_p, vp]),
    "nlg_basis_cgs2": (C.c_int, [vp, C.c_int, vp, c_double_p, c_double_p]),
    "nlg_basis_combine": (C.c_int, [vp, C.c_int, c_double_p, vp]),
    "nlg_exptA_config_default": (C.c_int, [C.POINTER(ExptAConfig)]),
    "nlg_linop_create": (C.c_int, [vp, C.POINTER(ExptAConfig), vp, C.POINTER(vp)]),
    "nlg_linop_destroy": (C.c_int, [vp]),
    "nlg_linop_init": (C.c_int, [vp]),
    "nlg_linop_nonlinear_map": (C.c_int, [vp, vp, vp]),
    "nlg_linop_set_baseflow": (C.c_int, [vp, vp]),
    "nlg_linop_set_tolerances": (C.c_int, [vp, C.c_double, C.c_double]),
    "nlg_linop_set_projection": (C.c_int, [vp, C.c_double, C.c_int, c_int64_p, c_int64_p, c_double_p]),
    "nlg_linop_project": (C.c_int, [vp, vp]),
    "nlg_linop_integrate_forced": (C.c_int, [vp, vp, vp, vp, C.c_double, C.c_int, vp]),
    "nlg_linop_matvec": (C.c_int, [vp, vp, vp]),
    "nlg_linop_rmatvec": (C.c_int, [vp, vp, vp]),
    "nlg_linop_set_tau": (C.c_int, [vp, C.c_double]),
    "nlg_linop_get_info": (C.c_int, [vp, c_double_p, c_double_p, c_int_p, c_double_p]),
    "nlg_linop_get_stats": (C.c_int, [vp, c_int64_p, c_int64_p, c_int64_p, c_int64_p]),
    "nlg_op_helmholtz": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, C.c_int]),
    "nlg_op_dssum": (C.c_int, [vp, vp]),
    "nlg_op_cdabdtp": (C.c_int, [vp, vp, vp]),
    "nlg_op_pprec": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
    "nlg_op_opdiv": (C.c_int, [vp, vp, vp]),
    "nlg_op_opgradt": (C.c_int, [vp, vp, vp]),
    "nlg_op_conv": (C.c_int, [vp, vp, vp, vp, C.c_int]),
    "nlg_op_cfl": (C.c_int, [vp, vp, C.c_double, c_double_p]),
    "nlg_arnoldi_step": (C.c_int, [vp, vp, C.c_int, c_double_p, C.c_int, C.c_int]),
    "nlg_eigs_opts_default": (C.c_int, [C.POINTER(EigsOpts)]),
    "nlg_eigs": (C.c_int, [vp, C.POINTER(vp), C.c_int, c_double_p, c_double_p, c_double_p, c_int_p, vp,
                           C.POINTER(EigsOpts)]),
    "nlg_svds": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, c_double_p, c_double_p, c_int_p, vp, C.POINTER(EigsOpts)]),
    "nlg_symtridiag_eig": (C.c_int, [C.c_int, c_double_p, c_double_p, c_double_p]),
    "nlg_dense_eig": (C.c_int, [C.c_int, c_double_p, C.c_int, c_double_p, c_double_p, c_double_p, C.c_int]),
}

_lib = None


def load():
    """Load libneklab_gpu.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NlgError("%s not found: build it with `python -m neklab_amd.build` (hipcc, gfx950). "
                       "The product has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().nlg_last_error()
        raise NlgError((msg or b"").decode("utf-8", "replace") or ("libneklab_gpu error %d" % rc))


def dptr(a):
    return a.ctypes.data_as(c_double_p)
