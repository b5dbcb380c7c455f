// Host-staged shared-memory transport: the VALIDATION twin of the RCCL communicator.
//
// RCCL refuses two ranks on one device, and the development box has one MI355X.  This transport lets several
// processes share that one GPU and run the whole distributed path (halo index lists, face-grouped halo, split
// reductions, rank-local preconditioner levels) with every collective staged through a POSIX shared-memory
// segment: device -> own slot -> barrier -> read the peers' slots -> device.  It implements exactly the four
// operations the library asks of RCCL (all-reduce sum / max of doubles, all-gather of int64, pairwise exchange of
// the halo segments), so that everything except the RCCL calls themselves is exercised by tests/test_gpu_multirank.py.
// It is selected only by nlg_ctx_comm_init_shm; a production host never calls that.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>

#include "internal.h"

namespace nlg {

struct shm_header {
    std::atomic<int> attached;
    std::atomic<int> arrived;
    std::atomic<int> sense;
    int pad[13];
};

struct nlg_shm {
    int rank = 0, nranks = 1;
    size_t slot = 0, total = 0;
    char *base = nullptr;
    int local_sense = 0;
    std::string name;
    std::vector<char> stage;
    shm_header *hdr() const { return reinterpret_cast<shm_header *>(base); }
    char *slot_of(int q) const { return base + sizeof(shm_header) + (size_t)q * slot; }
};

static constexpr double kShmTimeoutS = 120.0;   // a dead peer must end in an error, never in a hang

static int shm_wait(const std::atomic<int> &a, int want) {
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (a.load(std::memory_order_acquire) != want) {
        if ((++spins & 1023) == 0) {
            sched_yield();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            NLG_CHECK(dt < kShmTimeoutS, "shm transport: peer did not arrive within %.0f s", kShmTimeoutS);
        }
    }
    return 0;
}

static int shm_barrier(nlg_shm *s) {
    s->local_sense ^= 1;
    shm_header *h = s->hdr();
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == s->nranks) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->sense.store(s->local_sense, std::memory_order_release);
        return 0;
    }
    return shm_wait(h->sense, s->local_sense);
}

void shm_close(nlg_ctx *ctx) {
    nlg_shm *s = ctx->shm;
    if (!s) return;
    if (s->base) munmap(s->base, s->total);
    delete s;
    ctx->shm = nullptr;
}

int shm_allreduce(nlg_ctx *ctx, double *d_buf, int count, bool is_max) {
    nlg_shm *s = ctx->shm;
    const size_t bytes = sizeof(double) * (size_t)count;
    NLG_CHECK(bytes <= s->slot, "shm transport: all-reduce of %d doubles exceeds the slot", count);
    NLG_HIP(hipMemcpyAsync(s->slot_of(s->rank), d_buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NLG_HIP(hipStreamSynchronize(ctx->stream));
    NLG_TRY(shm_barrier(s));
    s->stage.resize(bytes);
    double *out = reinterpret_cast<double *>(s->stage.data());
    for (int i = 0; i < count; ++i) {   // rank order: every rank computes the same bits
        double acc = reinterpret_cast<const double *>(s->slot_of(0))[i];
        for (int q = 1; q < s->nranks; ++q) {
            const double v = reinterpret_cast<const double *>(s->slot_of(q))[i];
            acc = is_max ? (v > acc ? v : acc) : acc + v;
        }
        out[i] = acc;
    }
    NLG_TRY(shm_barrier(s));   // nobody may overwrite a slot before all have read it
    NLG_HIP(hipMemcpyAsync(d_buf, out, bytes, hipMemcpyHostToDevice, ctx->stream));
    NLG_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int shm_allgather_i64(nlg_ctx *ctx, const int64_t *d_in, int64_t *d_out, int64_t count) {
    nlg_shm *s = ctx->shm;
    const size_t bytes = sizeof(int64_t) * (size_t)count;
    NLG_CHECK(bytes <= s->slot, "shm transport: all-gather of %lld words exceeds the slot", (long long)count);
    NLG_HIP(hipMemcpyAsync(s->slot_of(s->rank), d_in, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NLG_HIP(hipStreamSynchronize(ctx->stream));
    NLG_TRY(shm_barrier(s));
    s->stage.resize(bytes * s->nranks);
    for (int q = 0; q < s->nranks; ++q) memcpy(s->stage.data() + bytes * q, s->slot_of(q), bytes);
    NLG_TRY(shm_barrier(s));
    NLG_HIP(hipMemcpyAsync(d_out, s->stage.data(), bytes * s->nranks, hipMemcpyHostToDevice, ctx->stream));
    NLG_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// Slot layout of an exchange: [nneigh, ntot, nf, (neigh, noff, ncnt) * nneigh] as int64, then nf * ntot doubles.
int shm_exchange(nlg_ctx *ctx, const nlg_halo &h, int nf, hipStream_t st) {
    nlg_shm *s = ctx->shm;
    const size_t nn = h.neigh.size();
    const size_t head = sizeof(int64_t) * (3 + 3 * nn);
    const size_t bytes = sizeof(double) * (size_t)nf * (size_t)h.ntot;
    NLG_CHECK(head + bytes <= s->slot, "shm transport: halo of %lld doubles exceeds the slot", (long long)(nf * h.ntot));
    int64_t *mine = reinterpret_cast<int64_t *>(s->slot_of(s->rank));
    mine[0] = (int64_t)nn;
    mine[1] = h.ntot;
    mine[2] = nf;
    for (size_t q = 0; q < nn; ++q) {
        mine[3 + 3 * q] = h.neigh[q];
        mine[4 + 3 * q] = h.noff[q];
        mine[5 + 3 * q] = h.ncnt[q];
    }
    NLG_HIP(hipMemcpyAsync(reinterpret_cast<char *>(mine) + head, h.d_send, bytes, hipMemcpyDeviceToHost, st));
    NLG_HIP(hipStreamSynchronize(st));
    NLG_TRY(shm_barrier(s));
    s->stage.resize(bytes);
    double *recv = reinterpret_cast<double *>(s->stage.data());
    for (size_t q = 0; q < nn; ++q) {
        const int64_t *peer = reinterpret_cast<const int64_t *>(s->slot_of(h.neigh[q]));
        const int64_t pn = peer[0], ptot = peer[1];
        NLG_CHECK(peer[2] == nf, "shm transport: rank %d exchanges %lld fields, rank %d %d", h.neigh[q], (long long)peer[2], s->rank, nf);
        const double *pdata = reinterpret_cast<const double *>(reinterpret_cast<const char *>(peer) + sizeof(int64_t) * (3 + 3 * pn));
        int64_t poff = -1;
        for (int64_t e = 0; e < pn; ++e)
            if (peer[3 + 3 * e] == s->rank) {
                NLG_CHECK(peer[5 + 3 * e] == h.ncnt[q], "shm transport: segment sizes of ranks %d and %d differ", s->rank, h.neigh[q]);
                poff = peer[4 + 3 * e];
            }
        NLG_CHECK(poff >= 0, "shm transport: rank %d does not list rank %d as a neighbour", h.neigh[q], s->rank);
        for (int c = 0; c < nf; ++c)
            memcpy(recv + (size_t)c * h.ntot + h.noff[q], pdata + (size_t)c * ptot + poff, sizeof(double) * (size_t)h.ncnt[q]);
    }
    NLG_TRY(shm_barrier(s));
    NLG_HIP(hipMemcpyAsync(h.d_recv, recv, bytes, hipMemcpyHostToDevice, st));
    NLG_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // namespace nlg

using namespace nlg;

extern "C" int nlg_ctx_comm_init_shm(nlg_ctx *ctx, int rank, int nranks, const char *name, int64_t slot_bytes) {
    NLG_CHECK(ctx && name, "nlg_ctx_comm_init_shm: NULL argument");
    NLG_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "nlg_ctx_comm_init_shm: bad rank %d / %d", rank, nranks);
    NLG_CHECK(!ctx->comm && !ctx->shm, "nlg_ctx_comm_init_shm: the context already has a communicator");
    NLG_CHECK(slot_bytes >= 4096, "nlg_ctx_comm_init_shm: slot of %lld bytes is too small", (long long)slot_bytes);
    static_assert(sizeof(shm_header) == 64, "header is one cache line");
    nlg_shm *s = new nlg_shm;
    s->rank = rank;
    s->nranks = nranks;
    s->slot = ((size_t)slot_bytes + 63) / 64 * 64;
    s->total = sizeof(shm_header) + s->slot * (size_t)nranks;
    s->name = name;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)s->total) != 0) {
        if (fd >= 0) close(fd);
        delete s;
        set_error("nlg_ctx_comm_init_shm: cannot create segment %s", name);
        return 1;
    }
    void *p = mmap(nullptr, s->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete s;
        set_error("nlg_ctx_comm_init_shm: mmap of %s failed", name);
        return 1;
    }
    s->base = static_cast<char *>(p);
    ctx->shm = s;
    ctx->rank = rank;
    ctx->nranks = nranks;
    // the segment starts zero-filled; once every rank is attached the name can go
    s->hdr()->attached.fetch_add(1, std::memory_order_acq_rel);
    if (shm_wait(s->hdr()->attached, nranks) != 0) {
        shm_unlink(name);
        shm_close(ctx);
        return 1;
    }
    NLG_TRY(shm_barrier(s));
    if (rank == 0) shm_unlink(name);
    return 0;
}
