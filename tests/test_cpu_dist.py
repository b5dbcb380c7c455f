"""world_size-2 gloo test of the multi-rank path (CPU)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_protocol():
    from neklab_amd import build
    build.build_library()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("dim,nel,nranks", [(3, (3, 2, 4), 2), (3, (2, 2, 6), 3), (2, (4, 6), 4), (3, (2, 3, 2), 4)])
def test_halo_index_lists_with_emulated_ranks(dim, nel, nranks):
    """The index lists the GPU gather-scatter halo uses (csrc/halo.hip: boundary labels, plan, send / receive / copy
    lists -- pure host code exported as nlg_halo_boundary_labels / nlg_halo_lists) checked with emulated ranks: local
    gather-scatter + pack / exchange / ordered unpack in numpy must reproduce the serial QQ^T on every rank, also
    with ranks that are not neighbours and with periodic wrap-around."""
    import ctypes as C
    from neklab_amd import _lib
    from neklab_amd.mesh import box_mesh, partition_elements
    lib = _lib.load()
    n = 5
    hm = box_mesh(nel, n, periodic=(False,) * (dim - 1) + (True,), deform=0.03)
    np1 = n ** dim
    glo = hm.glo_num.reshape(hm.E, np1)
    rng = np.random.default_rng(0)
    f = rng.standard_normal((hm.E, np1))
    # serial reference
    _, inv = np.unique(glo.ravel(), return_inverse=True)
    ref = np.bincount(inv, weights=f.ravel())[inv].reshape(hm.E, np1)
    parts = partition_elements(hm.E, nranks)
    i64p = _lib.c_int64_p
    i32p = C.POINTER(C.c_int32)
    labs, gl = [], []
    for r in range(nranks):
        g = np.ascontiguousarray(glo[parts[r]].ravel(), dtype=np.int64)
        gl.append(g)
        out = np.empty(g.size, dtype=np.int64)
        cnt = lib.nlg_halo_boundary_labels(n, dim, len(parts[r]), g.ctypes.data_as(i64p), out.ctypes.data_as(i64p), out.size)
        assert cnt > 0
        assert np.all(np.diff(out[:cnt]) > 0)
        labs.append(out[:cnt].copy())
    counts = np.array([len(a) for a in labs], dtype=np.int64)
    concat = np.ascontiguousarray(np.concatenate(labs))
    lists = []
    for r in range(nranks):
        E_r = len(parts[r])
        cap = gl[r].size * nranks
        ncnt = np.zeros(nranks, dtype=np.int64)
        send_idx, rpos, cidx = (np.zeros(cap, dtype=np.int32) for _ in range(3))
        roff, coff = np.zeros(cap + 1, dtype=np.int32), np.zeros(cap + 1, dtype=np.int32)
        nlab = C.c_int64()
        tot = lib.nlg_halo_lists(n, dim, E_r, gl[r].ctypes.data_as(i64p), r, nranks, counts.ctypes.data_as(i64p),
                                 concat.ctypes.data_as(i64p), ncnt.ctypes.data_as(i64p), send_idx.ctypes.data_as(i32p), cap,
                                 roff.ctypes.data_as(i32p), rpos.ctypes.data_as(i32p), coff.ctypes.data_as(i32p),
                                 cidx.ctypes.data_as(i32p), cap, C.byref(nlab))
        assert tot >= 0 and ncnt[r] == 0 and ncnt.sum() == tot
        lists.append(dict(tot=tot, ncnt=ncnt, send_idx=send_idx[:tot], nlab=nlab.value, roff=roff[:nlab.value + 1],
                          rpos=rpos, coff=coff[:nlab.value + 1], cidx=cidx))
    # the plans are symmetric
    for r in range(nranks):
        for q in range(nranks):
            assert lists[r]["ncnt"][q] == lists[q]["ncnt"][r]
    # local gather-scatter, pack
    loc, send = [], []
    for r in range(nranks):
        fr = f[parts[r]].ravel().copy()
        _, inv_r = np.unique(gl[r], return_inverse=True)
        fr = np.bincount(inv_r, weights=fr)[inv_r]
        loc.append(fr)
        send.append(fr[lists[r]["send_idx"]])
    # exchange: the segment rank r keeps for neighbour q arrives in q's segment for r (same label order on both sides)
    def seg(r, q):
        off = int(lists[r]["ncnt"][:q].sum())
        return slice(off, off + int(lists[r]["ncnt"][q]))
    for r in range(nranks):
        L = lists[r]
        recv = np.zeros(L["tot"])
        for q in range(nranks):
            if L["ncnt"][q]:
                recv[seg(r, q)] = send[q][seg(q, r)]
        out = loc[r].copy()
        for l in range(L["nlab"]):
            s = recv[L["rpos"][L["roff"][l]:L["roff"][l + 1]]].sum()
            out[L["cidx"][L["coff"][l]:L["coff"][l + 1]]] += s
        assert np.max(np.abs(out - ref[parts[r]].ravel())) < 1e-13 * np.abs(ref).max(), (r, nranks)
