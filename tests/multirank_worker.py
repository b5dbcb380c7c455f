"""One rank of tests/test_gpu_multirank.py: `world` processes share cuda:0 through the shared-memory validation
transport (nlg_ctx_comm_init_shm) and run the distributed hot path on their element slab of a global box:
vector-space reductions, the propagator (halo exchange in natural and face-grouped layout, split reductions of the
PCG solvers, rank-local preconditioner levels), its adjoint, and a short Arnoldi factorisation.
usage: multirank_worker.py rank world segment outdir case"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neklab_amd import host  # noqa: E402
from neklab_amd.mesh import box_mesh  # noqa: E402

CASES = {
    # name: (elements per rank, lx1, periodic, pprecond)
    "box3d": ((2, 2, 2), 6, (False, False, False), 0),
    "per3d": ((2, 2, 2), 6, (True, False, True), 0),      # periodic across the rank boundary: two shared interfaces
    "jac3d": ((2, 2, 1), 5, (False, False, False), 1),
    "box2d": ((3, 2), 7, (False, False), 0),
    # with NLG_COARSE_EXACT_MAX=50 in the environment: aggregated (not exact) coarse level, as on production meshes
    "agg3d": ((4, 4, 3), 5, (False, False, False), 0),
    # the block propagator across ranks: three lanes in every launch, ONE halo exchange / all-reduce carrying all of them
    "blk3d": ((2, 2, 2), 6, (False, True, False), 0),
}


def run_cylinder(rank, world, segment, outdir, case):
    """The reference's cylinder mesh (unstructured, periodic in y, outflow), contiguous element blocks per rank."""
    import dataclasses
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from refdata import load_cylinder
    from neklab_amd.mesh import partition_elements
    hm, ux, uy, _, re, lxd, _ = load_cylinder(with_bcs=True)
    mine = partition_elements(hm.x.shape[0], world)[rank]
    loc = dataclasses.replace(hm, nel=(len(mine), 1), x=hm.x[mine], y=hm.y[mine], glo_num=hm.glo_num[mine],
                              mask=[m[mine] for m in hm.mask], tmask=hm.tmask[mine], elem_gid=hm.elem_gid[mine])
    ctx = host.Context(0)
    if world > 1:
        ctx.comm_init_shm(rank, world, segment)
    gm = host.Mesh(ctx, loc, lxd=lxd)
    bf = host.nek_dvector(gm)
    bf.set_field(host.VX, ux[mine])
    bf.set_field(host.VY, uy[mine])
    v = host.nek_dvector(gm)
    v.rand(True, seed=3)
    A = host.exptA_linop(0.05, bf, re=re, torder=3, vtol=1e-13, ptol=1e-13, maxit_v=400, maxit_p=4000)
    A.init()
    out, outT = host.nek_dvector(gm), host.nek_dvector(gm)
    A.matvec(v, out)
    A.rmatvec(v, outT)
    m = 4
    B = host.KrylovBasis(gm, m + 1)
    B[0].assign(v)
    H = np.zeros((m + 1, m), order="F")
    for k in range(m):
        host.arnoldi_step(A, B, k, H)
    fields = {"scal": np.array([v.norm(), out.norm(), outT.norm(), out.dot(v), float(A.info()["nsteps"])]), "H": H,
              "stats": np.array([A.stats()["p_iters"], A.stats()["v_iters"]], dtype=float)}
    for i in range(2):
        fields["out%d" % i] = out.get_field(i)
        fields["outT%d" % i] = outT.get_field(i)
        fields["v%d" % i] = v.get_field(i)
    fields["outp"] = out.get_field(3)
    np.savez(os.path.join(outdir, "%s_w%d_r%d.npz" % (case, world, rank)), **fields)
    ctx.sync()


def run_heat(rank, world, segment, outdir, case):
    """Boussinesq coupling (cfg.ifheat) and the nonlinear map on a slab partition: the scalar solve, the buoyancy and the
    scalar convection use the same halo / reductions as the fluid."""
    _, _, mult = case.partition("@")
    nel, n = (2, 2, 2 * int(mult or 1)), 6
    ctx = host.Context(0)
    if world > 1:
        ctx.comm_init_shm(rank, world, segment)
    gnel = (nel[0], nel[1], nel[2] * world)
    hm = box_mesh(gnel, n, lengths=(2.0, 1.0, 1.0 * gnel[2] / 2), periodic=(True, False, True), deform=0.03,
                  last_range=(rank * nel[2], (rank + 1) * nel[2]))
    gm = host.Mesh(ctx, hm)
    gb = host.nek_dvector(gm, 1)
    gb.set_field(0, hm.mask[0] * (4 * hm.y * (1 - hm.y)))
    gb.set_field(host.THETA, 1.0 - hm.y + 0.1 * np.sin(np.pi * hm.x) * np.sin(np.pi * hm.y))
    kw = dict(re=5.0, torder=3, vtol=1e-13, ptol=1e-13, maxit_v=600, maxit_p=4000, dt=0.01, ifheat=1, conductivity=0.3,
              rhocp=1.5, buoy=(0.0, 50.0, 0.0))
    A = host.exptA_linop(0.05, gb, **kw)
    A.init()
    v, out, nl = host.nek_dvector(gm, 1), host.nek_dvector(gm, 1), host.nek_dvector(gm, 1)
    v.rand(True, seed=5)
    A.matvec(v, out)
    host.check(A.lib.nlg_linop_nonlinear_map(A.h, v.h, nl.h))
    fields = {"scal": np.array([v.norm(), out.norm(), nl.norm(), out.dot(nl)]), "H": np.zeros((2, 1)),
              "stats": np.array([A.stats()["p_iters"], A.stats()["v_iters"]], dtype=float)}
    for i in range(3):
        fields["out%d" % i] = out.get_field(i)
        fields["outT%d" % i] = nl.get_field(i)
        fields["v%d" % i] = v.get_field(i)
    fields["outp"] = out.get_field(host.THETA)
    fields["outq"] = nl.get_field(host.THETA)
    np.savez(os.path.join(outdir, "%s_w%d_r%d.npz" % (case, world, rank)), **fields)
    ctx.sync()


def run_proj(rank, world, segment, outdir, case):
    """Wavenumber-projected propagator with the homogeneous direction ACROSS the rank boundaries: the lines of the planar
    averages (exponential_propagator_proj.f90:146-169) have parts on every rank; labels are global line names."""
    _, _, mult = case.partition("@")
    nel, n = (2, 2, 2 * int(mult or 1)), 6
    ctx = host.Context(0)
    if world > 1:
        ctx.comm_init_shm(rank, world, segment)
    gnel = (nel[0], nel[1], nel[2] * world)
    Lz = 2.0 * np.pi
    hm = box_mesh(gnel, n, lengths=(2.0, 2.0, Lz), periodic=(True, False, True), deform=0.0,
                  last_range=(rank * nel[2], (rank + 1) * nel[2]))
    gm = host.Mesh(ctx, hm)
    bf = host.nek_dvector(gm)
    bf.set_field(0, hm.mask[0] * (hm.y * (2.0 - hm.y)))
    A = host.exptA_linop(0.03, bf, re=40.0, dt=0.01, vtol=1e-13, ptol=1e-13, maxit_p=2000)
    A.init()

    def name(*coords):      # a global name for the line through a point: its coordinates across the line, hashed
        k = [np.round(np.asarray(c).ravel() / 1e-7).astype(np.int64) for c in coords]
        return np.ascontiguousarray(k[0] * 40000003 + k[1], dtype=np.int64)

    X2 = host.pressure_mesh_coords(gm)
    lab, lab2, z2 = name(hm.x, hm.y), name(X2[0], X2[1]), np.ascontiguousarray(X2[2], dtype=np.float64)
    from neklab_amd import _lib
    host.check(gm.lib.nlg_linop_set_projection(A.h, 1.0, 3, lab.ctypes.data_as(_lib.c_int64_p), lab2.ctypes.data_as(_lib.c_int64_p),
                                                z2.ctypes.data_as(_lib.c_double_p)))
    v, pv, out = host.nek_dvector(gm), host.nek_dvector(gm), host.nek_dvector(gm)
    v.rand(True, seed=21)
    v.set_field(host.PR, np.cos(X2[2]) * X2[1])
    pv.assign(v)
    host.check(gm.lib.nlg_linop_project(A.h, pv.h))
    A.matvec(v, out)
    fields = {"scal": np.array([v.norm(), pv.norm(), out.norm(), out.dot(pv)]), "H": np.zeros((2, 1)),
              "stats": np.array([A.stats()["p_iters"], A.stats()["v_iters"]], dtype=float)}
    for i in range(3):
        fields["out%d" % i] = out.get_field(i)
        fields["outT%d" % i] = pv.get_field(i)
        fields["v%d" % i] = v.get_field(i)
    fields["outp"] = out.get_field(host.PR)
    fields["outq"] = pv.get_field(host.PR)
    np.savez(os.path.join(outdir, "%s_w%d_r%d.npz" % (case, world, rank)), **fields)
    ctx.sync()


def run(rank, world, segment, outdir, case):
    if case.startswith("proj"):
        return run_proj(rank, world, segment, outdir, case)
    if case.startswith("heat"):
        return run_heat(rank, world, segment, outdir, case)
    if case.startswith("cyl"):
        return run_cylinder(rank, world, segment, outdir, case)
    base, _, mult = case.partition("@")          # "box3d@2" with world 1: the global mesh of the 2-rank run
    base = base.replace("+ovl", "")              # "+ovl": the same case with NLG_HALO_OVERLAP=1 in the environment (set by the test)
    base = base.replace("+sr", "")               # "+sr": NLG_PCG_SINGLE_RED=1, the one-reduction PCG of the velocity / scalar solves
    nel, n, periodic, pprecond = CASES[base]
    nel = tuple(nel[:-1]) + (nel[-1] * int(mult or 1),)
    dim = len(nel)
    ctx = host.Context(0)
    if world > 1:
        ctx.comm_init_shm(rank, world, segment)
    gnel = tuple(nel[:-1]) + (nel[-1] * world,)
    hm = box_mesh(gnel, n, periodic=periodic, deform=0.04, last_range=(rank * nel[-1], (rank + 1) * nel[-1]))
    gm = host.Mesh(ctx, hm)
    X = [hm.x, hm.y] + ([hm.z] if dim == 3 else [])
    L = hm.lengths
    ph = [2 * np.pi * X[d] / L[d] for d in range(dim)]
    bf = host.nek_dvector(gm)
    U = [np.sin(ph[1]) * np.cos(ph[-1]), 0.5 * np.sin(ph[0])] + ([0.3 * np.cos(ph[0]) * np.sin(ph[1])] if dim == 3 else [])
    for i in range(dim):
        bf.set_field(i, U[i] * hm.mask[i])
    v, w = host.nek_dvector(gm), host.nek_dvector(gm)
    v.rand(True, seed=11)
    w.rand(False, seed=12)
    scal = [v.dot(w), w.norm(), v.norm()]
    A = host.exptA_linop(0.03, bf, re=40.0, dt=0.01, vtol=1e-13, ptol=1e-13, maxit_p=2000, pprecond=pprecond)
    A.init()
    out, outT = host.nek_dvector(gm), host.nek_dvector(gm)
    if base == "blk3d":
        # three vectors advanced together (nlg_linop_matvec_block); `out` / `outT` = the first and the last lane
        w.scal(1.0 / w.norm())
        mid, outm = host.nek_dvector(gm), host.nek_dvector(gm)
        mid.rand(True, seed=13)
        A.matvec_block([v, mid, w], [out, outm, outT])
        scal += [outm.norm(), outm.dot(v)]
    else:
        A.matvec(v, out)
        A.rmatvec(v, outT)
    scal += [out.norm(), outT.norm(), out.dot(w)]
    m = 6
    B = host.KrylovBasis(gm, m + 1)
    B[0].assign(v)
    H = np.zeros((m + 1, m), order="F")
    for k in range(m):
        host.arnoldi_step(A, B, k, H)
    fields = {"scal": np.array(scal), "H": H, "stats": np.array([A.stats()["p_iters"], A.stats()["v_iters"]], dtype=float)}
    for i in range(dim):
        fields["out%d" % i] = out.get_field(i)
        fields["outT%d" % i] = outT.get_field(i)
        fields["v%d" % i] = v.get_field(i)
    fields["outp"] = out.get_field(3)
    np.savez(os.path.join(outdir, "%s_w%d_r%d.npz" % (case, world, rank)), **fields)
    ctx.sync()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5])
    print("WORKER_OK")
