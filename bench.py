#!/usr/bin/env python3
"""Benchmark of the hot path: exptA matvec + Arnoldi orthogonalisation on the MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.  Started without a
launcher (WORLD_SIZE unset) and with N > 1 it starts the N rank processes itself (`python -m torch.distributed.run`,
before anything in this process touches the GPU) and relays rank 0's line.

Scaling (`--scaling`): "strong" (default, BASELINE.json configs[2] "1 GPU vs 8 GPU element-partitioned") -- ONE global
problem of E = 10 000 elements split into N sub-boxes by recursive coordinate bisection; "weak" -- every GPU holds an
E-element block of an N-times larger box.  In both, `value` is matvecs per second of the GLOBAL operator.

A "step" is one Arnoldi iteration at full basis size on BASELINE.json's headline configuration
(configs[2]: 3-D, E = 10 000 = 25x20x20 spectral elements, N = 7 i.e. lx1 = 8, Krylov dimension m = 64):
one `exptA` matvec (nsteps + torder-1 restated nek_advance steps, each with a dealiased convective
term, one joint Helmholtz Jacobi-PCG solve and one consistent-Poisson Jacobi-PCG solve to the
reference tolerances 1e-9 / 1e-7 of examples/cylinder/stability/direct/1cyl.par:22,27) followed by the
CGS2 orthogonalisation of the result against the m basis vectors, normalisation included.
`value` = matvecs per second of the whole job; inputs are resident in HBM before the timed region.

The JSON also carries `roofline` for the kernel class with the largest share of the timed region
(HIP events on the launch stream, algorithmic bytes from DESIGN.md §5) and `cpu_baseline` (the CPU
restatement in oracle/, timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: the global --nel box is split into N sub-boxes (recursive coordinate bisection); weak: every rank holds "
                         "a --nel block of an N times larger box")
    ap.add_argument("--transport", choices=("rccl", "shm"), default="rccl",
                    help="shm: REHEARSAL of the N>1 path with all ranks on GPU 0 through the library's shared-memory "
                         "validation transport (RCCL refuses two ranks per device); its numbers are not bench results")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nel", type=str, default="25,20,20", help="elements per direction (E = product)")
    ap.add_argument("--lx1", type=int, default=8)
    ap.add_argument("--kdim", type=int, default=64)
    ap.add_argument("--nsteps", type=int, default=2, help="time steps per matvec before the torder-1 history steps")
    ap.add_argument("--re", type=float, default=100.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--pprecond", type=int, default=0, help="pressure preconditioner (0 default, 2 = no overlap, 1 = Jacobi)")
    ap.add_argument("--no-units", action="store_true", help="skip the U1+U2 / U3 unit timings after the timed region (profiling runs)")
    ap.add_argument("--pproj", type=int, default=1, help="pressure residual projection (1 default, 0 = off)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--ifheat", action="store_true", help="Boussinesq coupling with one scalar (BASELINE config 4 shape)")
    ap.add_argument("--no-history", action="store_true",
                    help="vectors without restart-history copies (cfg.no_history, lorder = 1): a third of the basis memory; "
                         "the memory plan for BASELINE configs 4 / 5 (DESIGN.md)")
    ap.add_argument("--block", type=int, default=1,
                    help="> 1: a step is one BLOCK Arnoldi step with this many vectors (<= 4) advanced together "
                         "(nlg_block_arnoldi_step); value counts every vector's matvec")
    return ap.parse_args()


def algorithmic_bytes(cls, E, n, dim, k, ncomp, lvs, lps, nshared, main_len, lorder=3):
    """Algorithmic HBM bytes of ONE launch of a kernel class, fp64."""
    n2 = n - 2
    np1, np2 = n ** dim, n2 ** dim
    if cls == "opgradt":      # p + dim^2 metric terms in, dim velocity-mesh fields out
        return 8.0 * E * (np2 * (1 + dim * dim) + dim * np1)
    if cls == "opdiv":        # dim fields + dim fused weights in, dim^2 metric terms, p out; 3-D: the weights are one array and a mask
        # byte per point (k_opdiv3n, NLG_OPDIV_MASKB=0: dim arrays as before)
        if dim == 3 and os.environ.get("NLG_OPDIV_MASKB", "1") != "0":
            return 8.0 * E * (dim * np1 + 1.125 * np1 + np2 * (dim * dim + 1))
        return 8.0 * E * (2 * dim * np1 + np2 * (dim * dim + 1))
    if cls == "axhelm":       # NF = dim fields in/out, 6 (3) metric factors + mass; 3-D: the fused PCG direction
        ng = 6 if dim == 3 else 3   # update p <- z + beta p adds z in and p out
        return 8.0 * E * np1 * ((4 if dim == 3 else 2) * dim + ng + 1)
    if cls == "gs":           # value in + out per shared local dof and field, 4-byte index once
        return nshared * (16.0 * dim + 4.0)
    if cls == "cg_update":    # k_cg_update<dim>: r in/out, w, pc in, z out per field (5 arrays) + the two weight arrays; x in/out and
        # p in as well (8 arrays) when the deferred solution update is switched off (NLG_PCG_DEFER_X=0) or the one-reduction PCG runs
        deferred = os.environ.get("NLG_PCG_DEFER_X", "16") != "0" and os.environ.get("NLG_PCG_SINGLE_RED", "0") == "0"
        # 3-D: the preconditioner is one array 1 / diag plus a mask byte per point instead of dim masked arrays (NLG_PC_MASKB=0: as before)
        compact = dim == 3 and os.environ.get("NLG_PC_MASKB", "1") != "0" and os.environ.get("NLG_PCG_SINGLE_RED", "0") == "0"
        per_field = (5 if deferred else 8) - (1 if compact else 0)
        return 8.0 * lvs * (per_field * dim + 2 + (1.125 if compact else 0.0))
    if cls == "block_dot":    # k basis vectors + w + bm1 over the inner-product dofs
        return 8.0 * (k * ncomp + ncomp + 1) * lvs
    fused = k >= 24           # CGS2: first subtraction + second projection in one sweep over the last min(k, 64) vectors
    if cls == "axpy_dot":     # kf basis vectors + w in/out + bm1 over the inner-product dofs
        return 8.0 * ((min(k, 64) + 2) * ncomp + 1) * lvs if fused else None
    if cls == "block_axpy" and k > 64:
        return None           # three launches of different sizes per CGS2: not a single-kernel class any more
    if cls == "block_axpy":   # k basis vectors + w in/out.  CGS2 launches it twice: first pass over the main fields only
        # (fused sweep: only the pressure part is left to it), second pass over main + the lorder-1 history blocks
        # (consistent restart history) -> mean per launch
        first = lps if fused else main_len
        return 8.0 * (k + 2) * (first + main_len * lorder) / 2.0
    return None


def spawn_ranks(args):
    """No launcher around us and --gpus N > 1: start the N ranks as fresh children (one process per GPU) and relay
    their output; this parent never touches the GPU (nothing GPU-related has been imported at this point)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        # Control plane only (rendezvous, unique-id broadcast, barrier, max-over-ranks): gloo on CPU tensors.
        # The data-path collectives are RCCL calls inside libneklab_gpu.so (one communicator per process);
        # torch's bundled HIP runtime is deliberately never initialised next to the system one the library links.
        import torch.distributed as dist_mod
        dist_mod.init_process_group(backend="gloo")
        dist = dist_mod
    from neklab_amd import host
    from neklab_amd.mesh import box_mesh, partition_elements

    nel = tuple(int(x) for x in args.nel.split(","))
    n, dim, m = args.lx1, len(nel), args.kdim
    E = int(np.prod(nel))
    ctx = host.Context(local_rank if args.transport == "rccl" else 0)
    if world > 1 and args.transport == "shm":
        seg = ["/nlg_bench_%d" % os.getpid() if rank == 0 else None]
        dist.broadcast_object_list(seg, src=0)
        ctx.comm_init_shm(rank, world, seg[0], 256 << 20)
    elif world > 1:
        uid = [host.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(rank, world, uid[0])
    lib = ctx.lib

    # ---- synthetic inputs: deformed box, wall-masked, C0 noise on a smooth shear flow
    t0 = time.time()
    # Every rank generates only its own sub-box of elements with the labels and element ids of the global mesh.  Shared faces
    # are exchanged by the library's gather-scatter halo (RCCL send/recv); every reduction is a RCCL all-reduce.
    # strong: the global box is --nel, partitioned by recursive coordinate bisection; weak: the global box is `world` copies
    # of --nel stacked in the last direction.
    if args.scaling == "strong":
        # recursive coordinate bisection of the ONE global box: 25 x 20 x 20 on 8 ranks = 2 x 2 x 2 sub-boxes of
        # 12|13 x 10 x 10 elements (max / mean 1.04, at most 3 face neighbours + edges / corners) instead of slabs of z layers
        # (3,3,3,3,2,2,2,2 of 20: max / mean 1.2 and 2 x 500 element faces of halo per rank)
        from neklab_amd.mesh import rcb_boxes
        gnel = nel
        try:
            boxes = rcb_boxes(gnel, world)
        except ValueError as exc:
            raise SystemExit("bench.py: %s" % exc)
        my_box = boxes[rank]
        part_sizes = [int(np.prod([b - a for a, b in bx])) for bx in boxes]
    else:
        gnel = tuple(nel[:-1]) + (nel[-1] * world,)
        my_box = tuple((0, e) for e in nel[:-1]) + ((rank * nel[-1], (rank + 1) * nel[-1]),)
        part_sizes = [E] * world
    E_global = int(np.prod(gnel))
    hm = box_mesh(gnel, n, deform=0.05, ranges=my_box)
    E = hm.E                                   # local element count from here on
    gm = host.Mesh(ctx, hm)
    nscal = 1 if args.ifheat else 0
    lorder = 1 if args.no_history else 3
    bf = host.nek_dvector(gm, nscal)
    L = hm.lengths
    X = [hm.x, hm.y] + ([hm.z] if dim == 3 else [])
    ph = [2 * np.pi * X[d] / L[d] for d in range(dim)]
    if dim == 3:
        U = [np.sin(ph[1]) * np.cos(ph[2]), 0.5 * np.sin(ph[2]) * np.cos(ph[0]), 0.5 * np.sin(ph[0]) * np.cos(ph[1])]
    else:
        U = [np.sin(ph[1]), 0.5 * np.sin(ph[0])]
    for i in range(dim):
        bf.set_field(i, U[i] * hm.mask[i])
    if args.ifheat:
        bf.set_field(host.THETA, 1.0 - X[1] / L[1])      # conduction profile between the walls y = 0, Ly
    noise = host.nek_dvector(gm, nscal)
    noise.rand(False, seed=0)
    bf.axpby(0.05, noise, 1.0)
    # tau such that the CFL rule (cfl_limit 0.5) gives exactly args.nsteps steps
    cfl1 = C.c_double()
    host.check(lib.nlg_op_cfl(gm.h, bf.h, 1.0, C.byref(cfl1)))
    dt0 = 0.5 / cfl1.value
    tau = dt0 * (args.nsteps - 0.5)
    heat = dict(ifheat=1, conductivity=1.0 / args.re, rhocp=1.0, buoy=(0.0, 1.0, 0.0)) if args.ifheat else {}
    A = host.exptA_linop(tau, bf, re=args.re, torder=3, vtol=1e-9, ptol=1e-7, maxit_v=200, maxit_p=4000,
                         pprecond=args.pprecond, pproj=args.pproj, no_history=int(args.no_history), **heat)
    A.init()
    info = A.info()
    assert info["nsteps"] == args.nsteps, info

    # ---- Krylov basis of m orthonormal vectors + one work column, resident in HBM
    B = host.KrylovBasis(gm, m + 1, nscal, lorder)
    for j in range(m):
        v = B[j]
        v.rand(False, seed=100 + j)
        B.cgs2(j, v)
    ctx.sync()
    setup_s = time.time() - t0
    H = np.zeros((m + 2, m + 1), order="F")
    names = ["axhelm", "gs", "opgradt", "opdiv", "colmul", "block_dot", "block_axpy", "cg_vec", "conv", "vec_ops", "pprec", "axpy_dot", "cg_update"]

    sblk = max(1, min(args.block, 4))

    def step():
        if sblk > 1:      # columns m+1-2s .. m-s  ->  m+1-s .. m
            host.block_arnoldi_step(A, B, m + 1 - 2 * sblk, sblk, H)
        else:
            host.arnoldi_step(A, B, m - 1, H)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # ---- warmup (all kernel classes timed to find the dominant one)
    host.check(lib.nlg_prof_reset(ctx.h))
    host.check(lib.nlg_prof_enable(ctx.h, -1))
    st0 = A.stats()
    nwarm = max(args.warmup, 1)
    for w in range(nwarm):
        if w == 1:      # the first step is cold (first-chunk predictions, caches): classes are timed over the warm ones
            ctx.sync()
            host.check(lib.nlg_prof_reset(ctx.h))
        step()
    ctx.sync()
    nprof = max(nwarm - 1, 1)
    SUBCLASS = ("cg_update",)      # timed inside another class (cg_vec): never added to a sum over classes
    prof = {}
    for nm in names:
        cnt, ms = C.c_int64(), C.c_double()
        host.check(lib.nlg_prof_get(ctx.h, nm.encode(), C.byref(cnt), C.byref(ms)))
        prof[nm] = (cnt.value, ms.value)
    # the roofline is quoted for a single kernel: classes that bundle several kernels of different sizes (cg_vec,
    # pprec, conv, vec_ops) stay in share_of_step but are not candidates.  `roofline` = the single-kernel class with the largest share
    # of the step (no other rule); `roofline_gs` = the gather-scatter, the kernel north_star sets its 40 % target on, always reported.
    # cg_update is a sub-class of cg_vec (k_cg_update alone): it takes part in the selection, not in the share sums.
    single = [k for k in prof if algorithmic_bytes(k, 1, n, dim, m, dim, 1, 1, 1, 1) is not None]
    by_share = max(single, key=lambda k: prof[k][1])
    dominant = by_share
    timed = [dominant] + (["gs"] if dominant != "gs" and "gs" in single else [])
    host.check(lib.nlg_prof_enable(ctx.h, sum(1 << names.index(k) for k in timed)))
    host.check(lib.nlg_prof_reset(ctx.h))
    # every 8th launch of the dominant class is timed in the timed region: a pair of events costs ~12 us of stream time around
    # a 50-us kernel (measured in the rocprofv3 trace: 5.9 us idle before and after every timed launch, 1.0 ms per step)
    PROF_STRIDE = 8
    host.check(lib.nlg_prof_sample(ctx.h, PROF_STRIDE))
    st1 = A.stats()
    nlaunch0, ncoll0 = C.c_int64(), C.c_int64()
    host.check(lib.nlg_counters(C.byref(nlaunch0), C.byref(ncoll0)))

    # ---- timed region: exactly K steps
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st2 = A.stats()
    nlaunch1, ncoll1 = C.c_int64(), C.c_int64()
    host.check(lib.nlg_counters(C.byref(nlaunch1), C.byref(ncoll1)))
    launches_per_step = (nlaunch1.value - nlaunch0.value) / max(args.steps, 1)
    coll_per_step = (ncoll1.value - ncoll0.value) / max(args.steps, 1)
    tim = {}
    for kcls in timed:
        cnt, ms = C.c_int64(), C.c_double()
        host.check(lib.nlg_prof_get(ctx.h, kcls.encode(), C.byref(cnt), C.byref(ms)))
        tim[kcls] = (cnt.value, ms.value / max(cnt.value, 1))
    host.check(lib.nlg_prof_enable(ctx.h, 0))
    host.check(lib.nlg_prof_sample(ctx.h, 1))
    # shared local dofs: copies of labels that occur more than once
    _, inv, counts = np.unique(hm.glo_num.ravel(), return_inverse=True, return_counts=True)
    nshared = int(np.sum(counts[inv] > 1))
    lvs = -(-gm.lvn // 32) * 32
    lps = -(-gm.lpn // 32) * 32
    LANE_BATCHED = ("axhelm", "gs", "opgradt", "opdiv", "cg_update")   # one launch carries all lanes of a block step (gridDim.y)

    def abytes_of(kcls):
        ab = algorithmic_bytes(kcls, E, n, dim, m, dim + nscal, lvs, lps, nshared, (dim + nscal) * lvs + lps, lorder)
        return ab * sblk if (ab is not None and kcls in LANE_BATCHED) else ab

    def roof(kcls):
        cnt_, avg_ = tim[kcls]
        ab = abytes_of(kcls)
        r = {"kernel": kcls, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
             "launches": cnt_, "timed_every": PROF_STRIDE, "avg_ms": avg_, "algorithmic_bytes_per_launch": ab, "lanes_per_launch": sblk if kcls in LANE_BATCHED else 1}
        if ab is not None and avg_ > 0:
            r["achieved"] = ab / (avg_ * 1e-3) / 1e9
            r["frac"] = r["achieved"] / HBM_PEAK_GBS
        return r

    roofline = roof(dominant)
    avg_ms = tim[dominant][1]
    tot_ms = max(sum(x[1] for kk_, x in prof.items() if kk_ not in SUBCLASS), 1e-30)
    roofline.update({
        "share_of_step": {k: round(v[1] / tot_ms, 4) for k, v in prof.items()},
        # absolute: event-timed milliseconds per step and launches per step of every class (warm-up steps); cg_update is part of cg_vec
        "class_ms_per_step": {k: round(v[1] / nprof, 3) for k, v in prof.items()},
        "class_launches_per_step": {k: round(v[0] / nprof, 1) for k, v in prof.items()}})
    roofline["dominant_by_share"] = by_share
    # algorithmic bytes / event-timed duration / peak of every single-kernel class over the warm-up steps (every launch timed there:
    # each figure carries ~5 us of event overhead per launch, i.e. is a lower bound)
    cf = {}
    for kcls in single:
        ab = abytes_of(kcls)
        if ab is not None and prof[kcls][0] > 0 and prof[kcls][1] > 0:
            cf[kcls] = round(ab * prof[kcls][0] / (prof[kcls][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    roofline["class_frac_warmup"] = cf
    # HBM bytes per launch from the PMC counters: they cannot be sampled from inside this process, so the figure is the
    # one produced by scripts/pmc_traffic.py from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of THIS command on
    # THIS configuration (newest profiles/r*_pmc/traffic_per_launch.json whose config matches); null for any other config.
    import glob
    roofline_gs = roof("gs") if "gs" in tim else None
    for rl in (roofline, roofline_gs):
        if rl is None or sblk > 1:      # the PMC passes were made on the single-vector command
            continue
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc", "traffic_per_launch.json")), reverse=True):
            try:
                tr = json.load(open(tf))
                if tr["config"] == {"E": E, "lx1": n, "dim": dim} and rl["kernel"] in tr:
                    rl["traffic"] = tr[rl["kernel"]]["traffic_bytes"]
                    rl["traffic_source"] = os.path.relpath(tf, ROOT)
                    break
            except (OSError, ValueError, KeyError):
                pass
    if roofline_gs is None:
        roofline_gs = {k: v for k, v in roofline.items() if k not in ("share_of_step", "class_ms_per_step", "class_launches_per_step",
                                                                       "dominant_by_share", "class_frac_warmup")}

    steps_per_mv = (st2["steps"] - st1["steps"]) / max(args.steps, 1)
    p_iters = (st2["p_iters"] - st1["p_iters"]) / max(st2["steps"] - st1["steps"], 1)
    v_iters = (st2["v_iters"] - st1["v_iters"]) / max(st2["steps"] - st1["steps"], 1)

    # ---- the two roofline-accountable units of SURVEY.md 8(d), timed outside the timed region (collective, all ranks):
    # U1+U2 = element-local Helmholtz operator + gather-scatter per scalar field; U3 = CGS2 + norm + scale at k = m
    u12_per_s, u3_ms, u3_vs_k = None, None, None
    if not args.no_units:
        va, vb = host.nek_dvector(gm, nscal), host.nek_dvector(gm, nscal)
        va.rand(False, seed=7)
        nrep = 20
        host.check(lib.nlg_op_helmholtz(gm.h, va.h, vb.h, 1.0 / args.re, 1.0, 1))
        ctx.sync()
        t_u = time.perf_counter()
        for _ in range(nrep):
            host.check(lib.nlg_op_helmholtz(gm.h, va.h, vb.h, 1.0 / args.re, 1.0, 1))
        ctx.sync()
        u12_per_s = dim * nrep / (time.perf_counter() - t_u)
        wv = B[m]
        wv.rand(False, seed=8)
        ctx.sync()
        t_u = time.perf_counter()
        nrep3 = 3
        for _ in range(nrep3):
            B.cgs2(m, wv)
        ctx.sync()
        u3_ms = 1e3 * (time.perf_counter() - t_u) / nrep3
        u3_vs_k = {}
        for kk in sorted({1, 2, 4, 8, 16, 32, 48, m}):
            if kk > m:
                continue
            wv.rand(False, seed=9)
            ctx.sync()
            t_u = time.perf_counter()
            B.cgs2(kk, wv)
            ctx.sync()
            u3_vs_k[str(kk)] = round(1e3 * (time.perf_counter() - t_u), 3)
        del va, vb
    # block Arnoldi (BASELINE.json config 5 names it): s = 4 vectors per step at the same basis size -- s matvecs, then
    # ONE block CGS2 that reads the basis once per pass for all four (nlg_basis_block_cgs2)
    blk = None
    if not args.no_units and m + 1 >= 12:
        sb = 4
        kb = m + 1 - 2 * sb
        Hb = np.zeros((m + 2, m + 1), order="F")
        host.block_arnoldi_step(A, B, kb, sb, Hb)          # warm-up (the last 2 s basis columns are sacrificed)
        B.block_cgs2(kb, sb)                                 # re-orthonormalise the input block
        ctx.sync()
        t_u = time.perf_counter()
        host.block_arnoldi_step(A, B, kb, sb, Hb)
        ctx.sync()
        t_b = time.perf_counter() - t_u
        for v in range(sb):
            B[kb + sb + v].rand(False, seed=300 + v)
        ctx.sync()
        t_u = time.perf_counter()
        B.block_cgs2(kb + sb, sb)
        ctx.sync()
        t_o = time.perf_counter() - t_u
        blk = {"block_size": sb, "k": kb + sb, "matvecs_per_s": round(sb / t_b, 3), "ms_per_block_step": round(1e3 * t_b, 3),
               "block_cgs2_ms": round(1e3 * t_o, 3), "block_cgs2_ms_per_vector": round(1e3 * t_o / sb, 3)}


    # ---- CPU baseline (rank 0, N = 1 only): oracle restatement on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            from oracle import cpu_baseline
            cpu = cpu_baseline.run(hm, U_fields=[bf.get_field(i) for i in range(dim)], re=args.re, dt=info["dt"],
                                   kdim=m, v_iters=max(int(round(v_iters)), 1), p_iters=max(int(round(p_iters)), 1),
                                   steps_per_matvec=steps_per_mv, budget_s=args.cpu_seconds)
        except Exception as exc:   # the baseline is a report, never a reason to lose the GPU number
            cpu = {"value": None, "unit": "matvecs/s", "cores": None, "kind": "port", "sample": "failed: %r" % (exc,)}

    if rank == 0:
        out = {
            "metric": "linop matvecs/sec + Arnoldi iter time, E=10k N=7, 1/2/4/8 GPU",
            # matvecs per second of the GLOBAL operator (strong: the same E-element problem at every N)
            "value": sblk * args.steps / elapsed,
            "unit": "matvecs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if args.transport == "rccl" else "synthetic; REHEARSAL on one GPU (shm transport), not a result",
            "config": {"workload": "3-D deformed box E=%d (%s) lx1=%d (N=%d), Krylov dim m=%d, exptA: Re=%g bdf3/ext3 "
                                   "nsteps=%d(%s), tol 1e-9/1e-7, one Arnoldi iteration per step at k=m"
                                   % (E_global, "x".join(map(str, gnel)), n, n - 1, m, args.re, args.nsteps,
                                      "+0 history steps, BDF start-up every matvec" if args.no_history else "+2 history steps"),
                       "vectors_per_step": sblk, "restart_history": not args.no_history, "ifheat": bool(args.ifheat),
                       "basis_bytes": (m + 1) * ((dim + nscal) * lvs + lps) * lorder * 8,
                       "elements_per_gpu": E_global / world, "partition": "rcb" if args.scaling == "strong" else "stacked",
                       "partition_sizes": part_sizes, "time_steps_per_matvec": steps_per_mv / sblk,
                       # comparable between the history-carrying and the --no-history protocol (2 + 2 against 2 time steps per matvec)
                       "time_steps_per_s": round(steps_per_mv * args.steps / elapsed, 2),
                       "pressure_iters_per_time_step": p_iters, "helmholtz_iters_per_time_step": v_iters,
                       # kernel launches and collective sites (all-reduce, all-gather, gather-scatter / Schwarz halo exchanges; counted on
                       # one rank too) per step, and per vector of a block step
                       "launches_per_step": round(launches_per_step, 1), "collectives_per_step": round(coll_per_step, 1),
                       "launches_per_vector": round(launches_per_step / sblk, 1), "collectives_per_vector": round(coll_per_step / sblk, 1),
                       "dt": info["dt"], "tau": info["tau"], "setup_s": round(setup_s, 2),
                       "global_elements": E_global,
                       "element_matvecs_per_s": E_global * sblk * args.steps / elapsed,
                       "value_definition": "exptA matvecs (+ CGS2 at k = m) of the global %d-element operator per second"
                                           % E_global,
                       "operator_applies_per_s_per_field_per_gpu": None if u12_per_s is None else round(u12_per_s, 1),
                       # the figures that carry over to the reference's tau = 1 cases (one matvec there = 100 + 2 time steps): the cost
                       # of ONE time step of the propagator inside the timed region (orthogonalisation at k = m, history blocks
                       # included, taken off: the block kernels' share of the step from the class timing)
                       "ms_per_time_step": round((1e3 * elapsed / args.steps)
                                                 * (1.0 - sum(prof[k][1] for k in ("block_dot", "block_axpy", "axpy_dot")) / max(sum(v[1] for kk_, v in prof.items() if kk_ not in SUBCLASS), 1e-30))
                                                 / max(steps_per_mv / sblk, 1e-30), 3),
                       "arnoldi_orthogonalisation_ms_at_k=m": None if u3_ms is None else round(u3_ms, 3),
                       "arnoldi_orthogonalisation_ms_vs_k": u3_vs_k,
                       "block_arnoldi": blk,
                       "parallelism": "1 process per GPU, recursive-coordinate-bisection element blocks, RCCL all-reduce for every reduction, "
                                      "RCCL send/recv halo for the gather-scatter"},
            "roofline": roofline,
            "roofline_gs": roofline_gs,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
