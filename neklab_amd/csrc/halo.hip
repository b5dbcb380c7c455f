// Multi-GPU part of the gather-scatter: exchange of the dofs shared between element partitions.
//
// Reference behaviour being replaced: Nek5000's gslib `gs_op` behind `opdssum`/`dsavg`
// (/root/reference/src/vectors/real_vectors.f90:100-104) and behind every `dssum` inside `nek_advance`
// (SURVEY.md §2b); elements are block-distributed over ranks.
//
// Protocol (placement-independent, deterministic):
//   set-up : every rank gathers the sorted unique element-boundary labels of all ranks (ncclAllGather),
//            intersects them with its own -> per neighbour a list of shared labels in ascending label order
//            (the same order on both sides), see nlg_halo_plan (pure host code, unit-tested on CPU).
//   gs_op  : local gather-scatter of the groups that hold a dof another rank shares -> pack one representative value per
//            (neighbour, shared label) -> grouped ncclSend/ncclRecv with every neighbour  ||  local gather-scatter of all
//            other groups -> unpack: one thread per shared label sums the received contributions in ascending neighbour
//            order and adds the sum to all local copies.  The two local parts touch disjoint dofs, so the result does not
//            depend on whether the exchange runs beside the second part (side stream, NLG_HALO_OVERLAP=1) or between the
//            two (default, and always with the shared-memory validation transport).
#include <algorithm>
#include <numeric>

#include "internal.h"

using namespace nlg;

namespace {

constexpr int NT = 256;

struct F3 {
    double *p[3];
};

template <int NF>
__global__ __launch_bounds__(NT) void k_halo_pack(const int *__restrict__ send_idx, int64_t ntot, F3 f,
                                                  double *__restrict__ buf, int64_t ld) {
    // blockIdx.y = lane of a block step: fields ld doubles behind lane 0's, buffer rows lane * NF + c (one exchange for all lanes)
    const int64_t e = blockIdx.x * (int64_t)NT + threadIdx.x;
    if (e >= ntot) return;
    const int i = send_idx[e];
    const int64_t lo = (int64_t)blockIdx.y * ld;
#pragma unroll
    for (int c = 0; c < NF; ++c) buf[((int64_t)blockIdx.y * NF + c) * ntot + e] = f.p[c][lo + i];
}

template <int NF>
__global__ __launch_bounds__(NT) void k_halo_unpack(int64_t nlab, const int *__restrict__ roff,
                                                    const int *__restrict__ rpos, const int *__restrict__ coff,
                                                    const int *__restrict__ cidx, int64_t ntot,
                                                    const double *__restrict__ buf, F3 f, int64_t ld) {
    const int64_t l = blockIdx.x * (int64_t)NT + threadIdx.x;
    if (l >= nlab) return;
    buf += (int64_t)blockIdx.y * NF * ntot;
#pragma unroll
    for (int c = 0; c < NF; ++c) f.p[c] += (int64_t)blockIdx.y * ld;
    double s[NF];
#pragma unroll
    for (int c = 0; c < NF; ++c) s[c] = 0.0;
    for (int q = roff[l]; q < roff[l + 1]; ++q) {
        const int e = rpos[q];
#pragma unroll
        for (int c = 0; c < NF; ++c) s[c] += buf[c * ntot + e];
    }
    for (int q = coff[l]; q < coff[l + 1]; ++q) {
        const int i = cidx[q];
#pragma unroll
        for (int c = 0; c < NF; ++c) f.p[c][i] += s[c];
    }
}

}  // namespace

// Pure host: which labels does `rank` share with every other rank?  labels_concat holds, rank after rank, the
// ascending unique labels of each rank (counts[q] of them).  On return neigh_counts[q] = number of labels shared
// with rank q (0 for q == rank) and shared_out holds those labels, neighbour after neighbour in ascending rank
// order, each list ascending.  Returns the total number of shared entries, or -1 if capacity is too small.
extern "C" int64_t nlg_halo_plan(int rank, int nranks, const int64_t *counts, const int64_t *labels_concat,
                                 int64_t *neigh_counts, int64_t *shared_out, int64_t capacity) {
    std::vector<int64_t> off(nranks + 1, 0);
    for (int q = 0; q < nranks; ++q) off[q + 1] = off[q] + counts[q];
    const int64_t *mine = labels_concat + off[rank];
    const int64_t nm = counts[rank];
    int64_t tot = 0;
    for (int q = 0; q < nranks; ++q) {
        neigh_counts[q] = 0;
        if (q == rank) continue;
        const int64_t *oth = labels_concat + off[q];
        const int64_t no = counts[q];
        int64_t i = 0, j = 0, cnt = 0;
        while (i < nm && j < no) {
            if (mine[i] < oth[j])
                ++i;
            else if (mine[i] > oth[j])
                ++j;
            else {
                if (tot + cnt >= capacity) return -1;
                shared_out[tot + cnt] = mine[i];
                ++cnt;
                ++i;
                ++j;
            }
        }
        neigh_counts[q] = cnt;
        tot += cnt;
    }
    return tot;
}

// Pure host: the sorted unique labels of the element-boundary dofs of one rank (what it contributes to the all-gather of
// the set-up).  Returns their number, or -1 if `capacity` is too small.
namespace {
struct BoundaryLabels {
    std::vector<int> bidx;       // local boundary dofs sorted by (label, index)
    std::vector<int64_t> ulab;   // unique labels, ascending
    std::vector<int> ubeg;       // begin of each label's copies in bidx (+ end sentinel)
};
BoundaryLabels boundary_labels(int n, int dim, int64_t E, const int64_t *glo) {
    BoundaryLabels b;
    const int np1 = dim == 3 ? n * n * n : n * n;
    b.bidx.reserve((size_t)(E * np1) / 2);
    for (int64_t e = 0; e < E; ++e)
        for (int p = 0; p < np1; ++p) {
            const int i = p % n, j = (p / n) % n, k = p / (n * n);
            const bool onb = i == 0 || i == n - 1 || j == 0 || j == n - 1 || (dim == 3 && (k == 0 || k == n - 1));
            if (onb) b.bidx.push_back((int)(e * np1 + p));
        }
    std::sort(b.bidx.begin(), b.bidx.end(), [glo](int a, int c) { return glo[a] < glo[c] || (glo[a] == glo[c] && a < c); });
    for (size_t q = 0; q < b.bidx.size(); ++q)
        if (q == 0 || glo[b.bidx[q]] != glo[b.bidx[q - 1]]) {
            b.ulab.push_back(glo[b.bidx[q]]);
            b.ubeg.push_back((int)q);
        }
    b.ubeg.push_back((int)b.bidx.size());
    return b;
}
struct HaloLists {
    std::vector<int64_t> ncnt;   // per rank: number of shared labels
    int64_t tot = 0, nlab = 0;
    std::vector<int> send_idx, roff{0}, rpos, coff{0}, cidx;
};
// index lists of one rank from its boundary labels and the labels of all ranks; false if the plan failed
bool halo_lists(const BoundaryLabels &b, int rank, int nranks, const int64_t *counts, const int64_t *labels_concat, HaloLists &L) {
    L.ncnt.assign(nranks, 0);
    std::vector<int64_t> shared(b.ulab.size() * (size_t)std::max(nranks - 1, 1) + 1);
    L.tot = nlg_halo_plan(rank, nranks, counts, labels_concat, L.ncnt.data(), shared.data(), (int64_t)shared.size());
    if (L.tot < 0) return false;
    L.send_idx.resize((size_t)L.tot);
    auto find_lab = [&](int64_t lab) { return (int)(std::lower_bound(b.ulab.begin(), b.ulab.end(), lab) - b.ulab.begin()); };
    std::vector<std::vector<int>> rpos_of(b.ulab.size());   // per unique label: entry positions, ascending neighbour
    for (int64_t e = 0; e < L.tot; ++e) {
        const int u = find_lab(shared[e]);
        L.send_idx[e] = b.bidx[b.ubeg[u]];
        rpos_of[u].push_back((int)e);
    }
    for (size_t u = 0; u < b.ulab.size(); ++u) {
        if (rpos_of[u].empty()) continue;
        ++L.nlab;
        L.rpos.insert(L.rpos.end(), rpos_of[u].begin(), rpos_of[u].end());
        L.roff.push_back((int)L.rpos.size());
        for (int q = b.ubeg[u]; q < b.ubeg[u + 1]; ++q) L.cidx.push_back(b.bidx[q]);
        L.coff.push_back((int)L.cidx.size());
    }
    return true;
}
}  // namespace

extern "C" int64_t nlg_halo_boundary_labels(int n, int dim, int64_t E, const int64_t *glo, int64_t *labels_out, int64_t capacity) {
    const BoundaryLabels b = boundary_labels(n, dim, E, glo);
    if ((int64_t)b.ulab.size() > capacity) return -1;
    std::copy(b.ulab.begin(), b.ulab.end(), labels_out);
    return (int64_t)b.ulab.size();
}

extern "C" int64_t nlg_halo_lists(int n, int dim, int64_t E, const int64_t *glo, int rank, int nranks, const int64_t *counts,
                                  const int64_t *labels_concat, int64_t *neigh_counts, int32_t *send_idx, int64_t cap_send,
                                  int32_t *roff, int32_t *rpos, int32_t *coff, int32_t *cidx, int64_t cap_copies, int64_t *nlab_out) {
    const BoundaryLabels b = boundary_labels(n, dim, E, glo);
    HaloLists L;
    if (!halo_lists(b, rank, nranks, counts, labels_concat, L)) return -1;
    if (L.tot > cap_send || (int64_t)L.cidx.size() > cap_copies) return -1;
    std::copy(L.ncnt.begin(), L.ncnt.end(), neigh_counts);
    std::copy(L.send_idx.begin(), L.send_idx.end(), send_idx);
    std::copy(L.roff.begin(), L.roff.end(), roff);
    std::copy(L.rpos.begin(), L.rpos.end(), rpos);
    std::copy(L.coff.begin(), L.coff.end(), coff);
    std::copy(L.cidx.begin(), L.cidx.end(), cidx);
    *nlab_out = L.nlab;
    return L.tot;
}

namespace nlg {

int halo_setup(nlg_mesh *m, const int64_t *glo) {
    nlg_ctx *ctx = m->ctx;
    nlg_halo &h = m->halo;
    h.active = false;
    if (!ctx->distributed()) return 0;
    hipStream_t st = ctx->stream;
    const int nr = ctx->nranks, me = ctx->rank;
    const int n = m->n, dim = m->dim, np1 = m->np1;
    // ---- local element-boundary labels -> sorted unique, with the list of local copies of each (pure host code,
    // shared with the CPU test of the list construction)
    const BoundaryLabels bl = boundary_labels(n, dim, m->E, glo);
    const std::vector<int64_t> &ulab = bl.ulab;
    // ---- gather counts and labels of all ranks (device buffers, RCCL)
    int64_t *d_cnt = nullptr;
    NLG_HIP(hipMalloc(&d_cnt, sizeof(int64_t) * (nr + 1)));
    int64_t mycnt = (int64_t)ulab.size();
    NLG_HIP(hipMemcpy(d_cnt + nr, &mycnt, sizeof(int64_t), hipMemcpyHostToDevice));
    if (ctx->shm)
        NLG_TRY(shm_allgather_i64(ctx, d_cnt + nr, d_cnt, 1));
    else
        NLG_NCCL(ncclAllGather(d_cnt + nr, d_cnt, 1, ncclInt64, ctx->comm, st));
    std::vector<int64_t> counts(nr);
    NLG_HIP(hipMemcpyAsync(counts.data(), d_cnt, sizeof(int64_t) * nr, hipMemcpyDeviceToHost, st));
    NLG_HIP(hipStreamSynchronize(st));
    const int64_t maxc = *std::max_element(counts.begin(), counts.end());
    int64_t *d_lab = nullptr;
    NLG_HIP(hipMalloc(&d_lab, sizeof(int64_t) * (size_t)maxc * (nr + 1)));
    NLG_HIP(hipMemsetAsync(d_lab, 0xff, sizeof(int64_t) * (size_t)maxc * (nr + 1), st));
    NLG_HIP(hipMemcpyAsync(d_lab + (size_t)maxc * nr, ulab.data(), sizeof(int64_t) * ulab.size(), hipMemcpyHostToDevice, st));
    if (ctx->shm)
        NLG_TRY(shm_allgather_i64(ctx, d_lab + (size_t)maxc * nr, d_lab, maxc));
    else
        NLG_NCCL(ncclAllGather(d_lab + (size_t)maxc * nr, d_lab, maxc, ncclInt64, ctx->comm, st));
    std::vector<int64_t> padded((size_t)maxc * nr), concat;
    NLG_HIP(hipMemcpyAsync(padded.data(), d_lab, sizeof(int64_t) * padded.size(), hipMemcpyDeviceToHost, st));
    NLG_HIP(hipStreamSynchronize(st));
    hipFree(d_cnt);
    hipFree(d_lab);
    for (int q = 0; q < nr; ++q) concat.insert(concat.end(), padded.begin() + (size_t)q * maxc, padded.begin() + (size_t)q * maxc + counts[q]);
    // ---- plan + index lists
    HaloLists L;
    NLG_CHECK(halo_lists(bl, me, nr, counts.data(), concat.data(), L), "halo_setup: plan capacity exceeded");
    const std::vector<int64_t> &ncnt = L.ncnt;
    const int64_t tot = L.tot;
    h.neigh.clear();
    h.noff.clear();
    h.ncnt.clear();
    int64_t off = 0;
    for (int q = 0; q < nr; ++q)
        if (ncnt[q] > 0) {
            h.neigh.push_back(q);
            h.noff.push_back(off);
            h.ncnt.push_back(ncnt[q]);
            off += ncnt[q];
        }
    h.ntot = tot;
    if (tot == 0) return 0;
    const std::vector<int> &send_idx = L.send_idx, &roff = L.roff, &rpos = L.rpos, &coff = L.coff, &cidx = L.cidx;
    const int64_t nlab = L.nlab;
    h.nlab = nlab;
    auto up = [&](const std::vector<int> &v, int **d) -> int {
        NLG_HIP(hipMalloc(d, sizeof(int) * std::max<size_t>(v.size(), 1)));
        NLG_HIP(hipMemcpy(*d, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
        return 0;
    };
    NLG_TRY(up(send_idx, &h.d_send_idx));
    h.h_cidx = cidx;
    NLG_TRY(up(roff, &h.d_roff));
    NLG_TRY(up(rpos, &h.d_rpos));
    NLG_TRY(up(coff, &h.d_coff));
    NLG_TRY(up(cidx, &h.d_cidx));
    if (!m->h_slot.empty()) {
        // the same lists for fields kept in the face-grouped element layout (3-D pressure operator)
        auto to_fg = [&](std::vector<int> v) {
            for (int &i : v) i = (i / np1) * np1 + m->h_slot[i % np1];
            return v;
        };
        NLG_TRY(up(to_fg(send_idx), &h.d_send_idx_fg));
        NLG_TRY(up(to_fg(cidx), &h.d_cidx_fg));
    }
    if (!m->h_slot_xp.empty()) {
        auto to_xp = [&](std::vector<int> v) {
            for (int &i : v) i = (i / np1) * np1 + m->h_slot_xp[i % np1];
            return v;
        };
        NLG_TRY(up(to_xp(send_idx), &h.d_send_idx_xp));
        NLG_TRY(up(to_xp(cidx), &h.d_cidx_xp));
    }
    NLG_HIP(hipMalloc(&h.d_send, sizeof(double) * (size_t)tot * 3 * kMaxLanes));   // up to three fields of every lane of a block step
    NLG_HIP(hipMalloc(&h.d_recv, sizeof(double) * (size_t)tot * 3 * kMaxLanes));
    NLG_HIP(hipEventCreateWithFlags(&h.ev_packed, hipEventDisableTiming));
    NLG_HIP(hipEventCreateWithFlags(&h.ev_recv, hipEventDisableTiming));
    const char *ov = getenv("NLG_HALO_OVERLAP");
    h.overlap = ov && atoi(ov) != 0;   // (also under the shm validation transport: the event choreography is the same)
    h.active = true;
    NLG_TRY(gs_split(m));
    return 0;
}

// ---- the local groups split by "holds a dof that another rank shares" ----------------------------------------------
static int upload_tab(std::vector<std::vector<int>> &gl, int np1, bool partner_order, nlg_gs_tab &t) {
    auto cls = [](size_t sz) { return sz == 2 ? 0 : (sz == 4 ? 1 : 2); };
    std::sort(gl.begin(), gl.end(), [&](const std::vector<int> &a, const std::vector<int> &b) {
        const int ca = cls(a.size()), cb = cls(b.size());
        if (ca != cb) return ca < cb;
        if (ca == 0 && partner_order) {   // as in nlg_mesh_create: pairs by (element, partner element, index)
            const int ea = a[0] / np1, eb = b[0] / np1, pa = a[1] / np1, pb = b[1] / np1;
            if (ea != eb) return ea < eb;
            if (pa != pb) return pa < pb;
        }
        return a[0] < b[0];
    });
    std::vector<int> off{0}, idx;
    t.ngroups = (int64_t)gl.size();
    t.npairs = t.nquads = 0;
    for (auto &v : gl) {
        t.npairs += v.size() == 2;
        t.nquads += v.size() == 4;
        idx.insert(idx.end(), v.begin(), v.end());
        off.push_back((int)idx.size());
    }
    NLG_HIP(hipMalloc(&t.d_off, sizeof(int) * off.size()));
    NLG_HIP(hipMalloc(&t.d_idx, sizeof(int) * std::max<size_t>(idx.size(), 4)));
    NLG_HIP(hipMemcpy(t.d_off, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice));
    if (!idx.empty()) NLG_HIP(hipMemcpy(t.d_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    return 0;
}

int gs_split(nlg_mesh *m) {
    nlg_gs &g = m->gs;
    g.split = false;
    if (!m->halo.active || g.h_groups.empty()) {
        g.h_groups.clear();
        g.h_groups.shrink_to_fit();
        return 0;
    }
    const int np1 = m->np1;
    std::vector<char> shared((size_t)m->lvn, 0);
    for (int i : m->halo.h_cidx) shared[i] = 1;
    for (int layout = 0; layout < 3; ++layout) {
        const std::vector<int> *slot = layout == LAYOUT_FG ? &m->h_slot : (layout == LAYOUT_XP ? &m->h_slot_xp : nullptr);
        if (slot && slot->empty()) continue;
        std::vector<std::vector<int>> part[2];
        for (const auto &grp : g.h_groups) {
            bool hal = false;
            for (int i : grp) hal = hal || shared[i];
            std::vector<int> v(grp);
            if (slot) {
                for (int &i : v) i = (i / np1) * np1 + (*slot)[i % np1];
                std::sort(v.begin(), v.end());
            }
            part[hal ? 0 : 1].push_back(std::move(v));
        }
        NLG_TRY(upload_tab(part[0], np1, layout == LAYOUT_XP, g.tab_halo[layout]));
        NLG_TRY(upload_tab(part[1], np1, layout == LAYOUT_XP, g.tab_rest[layout]));
    }
    g.split = true;
    g.h_groups.clear();
    g.h_groups.shrink_to_fit();
    return 0;
}

int halo_begin(nlg_mesh *m, double *const *fields, int nf, int layout, int nl, int64_t ld) {
    nlg_halo &h = m->halo;
    if (!h.active) return 0;
    NLG_CHECK(nl >= 1 && nl <= kMaxLanes, "halo_exchange: %d lanes", nl);
    NLG_CHECK(layout != LAYOUT_FG || h.d_send_idx_fg, "halo_exchange: no face-grouped index lists");
    NLG_CHECK(layout != LAYOUT_XP || h.d_send_idx_xp, "halo_exchange: no slab-permuted index lists");
    const int *send_idx = layout == LAYOUT_FG ? h.d_send_idx_fg : (layout == LAYOUT_XP ? h.d_send_idx_xp : h.d_send_idx);
    nlg_ctx *ctx = m->ctx;
    hipStream_t st = ctx->stream;
    F3 f = {{fields[0], nf > 1 ? fields[1] : nullptr, nf > 2 ? fields[2] : nullptr}};
    const dim3 g1((unsigned)((h.ntot + NT - 1) / NT), (unsigned)nl);
    if (nf == 1)
        NLG_LAUNCH(k_halo_pack<1>, g1, dim3(NT), 0, st, send_idx, h.ntot, f, h.d_send, ld);
    else if (nf == 2)
        NLG_LAUNCH(k_halo_pack<2>, g1, dim3(NT), 0, st, send_idx, h.ntot, f, h.d_send, ld);
    else
        NLG_LAUNCH(k_halo_pack<3>, g1, dim3(NT), 0, st, send_idx, h.ntot, f, h.d_send, ld);
    NLG_HIP(hipGetLastError());
    nf *= nl;   // rows of the packed buffers from here on: one send / receive group carries every lane
    if (ctx->shm) {
        if (h.overlap) {
            // validation transport with NLG_HALO_OVERLAP=1: the staging copy and the host-side exchange are ordered on the side stream
            // by the same two events the RCCL path uses, so that the fork / join choreography itself runs on one GPU
            NLG_HIP(hipEventRecord(h.ev_packed, st));
            NLG_HIP(hipStreamWaitEvent(ctx->stream2, h.ev_packed, 0));
            NLG_TRY(shm_exchange(ctx, h, nf, ctx->stream2));
            NLG_HIP(hipEventRecord(h.ev_recv, ctx->stream2));
            return 0;
        }
        return shm_exchange(ctx, h, nf, st);
    }
    // the send / receive group goes to the side stream when the overlap is switched on: it starts when the pack kernel has
    // finished and halo_finish makes the launch stream wait for it, so the communicator never sees two operations at once
    hipStream_t sx = h.overlap ? ctx->stream2 : st;
    if (h.overlap) {
        NLG_HIP(hipEventRecord(h.ev_packed, st));
        NLG_HIP(hipStreamWaitEvent(sx, h.ev_packed, 0));
    }
    NLG_NCCL(ncclGroupStart());
    for (size_t q = 0; q < h.neigh.size(); ++q)
        for (int c = 0; c < nf; ++c) {
            NLG_NCCL(ncclSend(h.d_send + (size_t)c * h.ntot + h.noff[q], (size_t)h.ncnt[q], ncclDouble, h.neigh[q], ctx->comm, sx));
            NLG_NCCL(ncclRecv(h.d_recv + (size_t)c * h.ntot + h.noff[q], (size_t)h.ncnt[q], ncclDouble, h.neigh[q], ctx->comm, sx));
        }
    NLG_NCCL(ncclGroupEnd());
    if (h.overlap) NLG_HIP(hipEventRecord(h.ev_recv, sx));
    return 0;
}

int halo_finish(nlg_mesh *m, double *const *fields, int nf, int layout, int nl, int64_t ld) {
    nlg_halo &h = m->halo;
    if (!h.active) return 0;
    const int *cidx = layout == LAYOUT_FG ? h.d_cidx_fg : (layout == LAYOUT_XP ? h.d_cidx_xp : h.d_cidx);
    hipStream_t st = m->ctx->stream;
    if (h.overlap) NLG_HIP(hipStreamWaitEvent(st, h.ev_recv, 0));
    F3 f = {{fields[0], nf > 1 ? fields[1] : nullptr, nf > 2 ? fields[2] : nullptr}};
    const dim3 g2((unsigned)((h.nlab + NT - 1) / NT), (unsigned)nl);
    if (nf == 1)
        NLG_LAUNCH(k_halo_unpack<1>, g2, dim3(NT), 0, st, h.nlab, h.d_roff, h.d_rpos, h.d_coff, cidx, h.ntot, h.d_recv, f, ld);
    else if (nf == 2)
        NLG_LAUNCH(k_halo_unpack<2>, g2, dim3(NT), 0, st, h.nlab, h.d_roff, h.d_rpos, h.d_coff, cidx, h.ntot, h.d_recv, f, ld);
    else
        NLG_LAUNCH(k_halo_unpack<3>, g2, dim3(NT), 0, st, h.nlab, h.d_roff, h.d_rpos, h.d_coff, cidx, h.ntot, h.d_recv, f, ld);
    NLG_HIP(hipGetLastError());
    return 0;
}

int halo_exchange(nlg_mesh *m, double *const *fields, int nf, int layout, int nl, int64_t ld) {
    NLG_TRY(halo_begin(m, fields, nf, layout, nl, ld));
    return halo_finish(m, fields, nf, layout, nl, ld);
}

void halo_free(nlg_mesh *m) {
    nlg_halo &h = m->halo;
    int *ip[] = {h.d_send_idx, h.d_roff, h.d_rpos, h.d_coff, h.d_cidx, h.d_send_idx_fg, h.d_cidx_fg, h.d_send_idx_xp, h.d_cidx_xp};
    for (int *p : ip)
        if (p) hipFree(p);
    if (h.d_send) hipFree(h.d_send);
    if (h.d_recv) hipFree(h.d_recv);
    if (h.ev_packed) hipEventDestroy(h.ev_packed);
    if (h.ev_recv) hipEventDestroy(h.ev_recv);
    for (int l = 0; l < 3; ++l)
        for (nlg_gs_tab *t : {&m->gs.tab_halo[l], &m->gs.tab_rest[l]}) {
            if (t->d_off) hipFree(t->d_off);
            if (t->d_idx) hipFree(t->d_idx);
            *t = nlg_gs_tab();
        }
    m->gs.split = false;
    h = nlg_halo();
}

}  // namespace nlg
