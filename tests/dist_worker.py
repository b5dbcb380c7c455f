"""world_size-2 worker (gloo, CPU): the multi-rank protocol of the path, with the oracle as local compute.

Checks, against the serial oracle on the full mesh:
  * nlg_halo_plan (the host planning step the GPU library uses, csrc/halo.hip) + the pack / exchange /
    ordered-unpack protocol reproduce the global gather-scatter,
  * multiplicity and assembled mass derived through it,
  * element-partitioned inner products = local glsc3 + allreduce (reference: glsc3, real_vectors.f90:217-224),
  * the partition-independent start vector (counter-based RNG keyed on global element ids),
  * one CGS2 orthogonalisation step on partitioned vectors.
Launched by tests/test_cpu_dist.py through torch.distributed.run.
"""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neklab_amd import _lib  # noqa: E402
from neklab_amd.mesh import box_mesh  # noqa: E402
from oracle.sem import SEM  # noqa: E402
from oracle.vectors import NekDVector  # noqa: E402


def halo_gs(sem, hm, u, rank, world, lib):
    """QQ^T across ranks following csrc/halo.hip."""
    n, dim = hm.n, hm.dim
    loc = sem.gs(u).reshape(-1).copy()                      # local gather-scatter first
    glo = hm.glo_num.reshape(-1)
    # element-boundary labels
    p = np.arange(n ** dim)
    ijk = [p % n, (p // n) % n] + ([p // (n * n)] if dim == 3 else [])
    onb = np.zeros(n ** dim, dtype=bool)
    for a in ijk:
        onb |= (a == 0) | (a == n - 1)
    bidx = np.nonzero(np.tile(onb, hm.E))[0]
    ulab = np.unique(glo[bidx])
    gathered = [None] * world
    dist.all_gather_object(gathered, ulab)
    counts = np.array([len(g) for g in gathered], dtype=np.int64)
    concat = np.concatenate(gathered).astype(np.int64)
    ncnt = np.zeros(world, dtype=np.int64)
    shared = np.zeros(len(ulab) * max(world - 1, 1) + 1, dtype=np.int64)
    tot = lib.nlg_halo_plan(rank, world, counts.ctypes.data_as(_lib.c_int64_p), concat.ctypes.data_as(_lib.c_int64_p),
                            ncnt.ctypes.data_as(_lib.c_int64_p), shared.ctypes.data_as(_lib.c_int64_p), shared.size)
    assert tot >= 0
    first = {}
    for i in bidx:
        first.setdefault(int(glo[i]), int(i))
    send = np.array([loc[first[int(l)]] for l in shared[:tot]])
    # exchange: every rank publishes its send buffer; receiver picks the segment addressed to it
    allsend, allcnt, allshared = [None] * world, [None] * world, [None] * world
    dist.all_gather_object(allsend, send)
    dist.all_gather_object(allcnt, ncnt)
    dist.all_gather_object(allshared, shared[:tot])
    out = loc.copy()
    copies = {}
    for i in bidx:
        copies.setdefault(int(glo[i]), []).append(int(i))
    add = {}
    for q in range(world):                                   # ascending neighbour order
        if q == rank or ncnt[q] == 0:
            continue
        off_q = int(np.sum(allcnt[q][:rank]))                # where q put the segment for `rank`
        seg = allsend[q][off_q: off_q + int(allcnt[q][rank])]
        labs = allshared[q][off_q: off_q + int(allcnt[q][rank])]
        mine_off = int(np.sum(ncnt[:q]))
        assert np.array_equal(labs, shared[mine_off: mine_off + int(ncnt[q])])   # same order on both sides
        for l, v in zip(labs, seg):
            add[int(l)] = add.get(int(l), 0.0) + v
    for l, v in add.items():
        for i in copies[l]:
            out[i] += v
    return out.reshape(np.shape(u)), ncnt


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.load()
    nel, n = (3, 2, 4), 5
    kw = dict(periodic=(False, False, True), deform=0.05)
    full = box_mesh(nel, n, **kw)
    fsem = SEM(full)
    per = nel[2] // world
    k0, k1 = rank * per, (rank + 1) * per if rank < world - 1 else nel[2]
    hm = box_mesh(nel, n, last_range=(k0, k1), **kw)
    sem = SEM(hm)
    sl = slice(int(hm.elem_gid[0]), int(hm.elem_gid[-1]) + 1)
    rng = np.random.default_rng(0)
    ufull = rng.standard_normal(fsem.shape1)
    u = ufull[sl]
    # 1. gather-scatter with halo == serial gather-scatter (periodic wrap makes both slab faces shared)
    g, ncnt = halo_gs(sem, hm, u, rank, world, lib)
    assert np.max(np.abs(g - fsem.gs(ufull)[sl])) < 1e-13, "halo gs mismatch"
    assert ncnt[rank] == 0 and np.sum(ncnt) > 0
    # 2. multiplicity / assembled mass through the same path
    mult, _ = halo_gs(sem, hm, np.ones(sem.shape1), rank, world, lib)
    assert np.array_equal(mult, fsem.mult[sl])
    bsum, _ = halo_gs(sem, hm, sem.bm1, rank, world, lib)
    assert np.max(np.abs(1.0 / bsum - fsem.binvm1[sl])) < 1e-13 * np.max(fsem.binvm1)
    # 3. inner product: local glsc3 + allreduce
    import torch
    vfull = rng.standard_normal(fsem.shape1)
    t = torch.tensor([sem.glsc3(u, vfull[sl])], dtype=torch.float64)
    dist.all_reduce(t)
    assert abs(float(t) - fsem.glsc3(ufull, vfull)) < 1e-12 * abs(fsem.glsc3(ufull, vfull))
    # 4. partition-independent noise
    raw = NekDVector(sem).raw_noise(7, hm.elem_gid, 2)
    assert np.array_equal(raw, NekDVector(fsem).raw_noise(7, full.elem_gid, 2)[sl])
    # 5. CGS2 step on partitioned vectors == serial
    k = 4
    V = [rng.standard_normal(fsem.shape1) for _ in range(k)]
    w = rng.standard_normal(fsem.shape1)

    def gdot(a, b, s):
        tt = torch.tensor([s.glsc3(a, b)], dtype=torch.float64)
        dist.all_reduce(tt)
        return float(tt)

    wl = w[sl].copy()
    ws = w.copy()
    for _ in range(2):
        hl = [gdot(V[j][sl], wl, sem) for j in range(k)]
        hs = [fsem.glsc3(V[j], ws) for j in range(k)]
        for j in range(k):
            wl -= hl[j] * V[j][sl]
            ws -= hs[j] * V[j]
    assert np.max(np.abs(wl - ws[sl])) < 1e-12 * np.max(np.abs(ws))
    dist.barrier()
    if rank == 0:
        print("DIST_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
