// Context, error reporting and the RCCL communicator of libneklab_gpu.
#include <cstdarg>
#include <cstdlib>

#include "internal.h"

namespace nlg {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int64_t g_launches = 0, g_collectives = 0;

int allreduce_sum(nlg_ctx *ctx, double *d_buf, int count) {
    ++g_collectives;   // counted whether or not a communicator exists: a one-rank run reports what several ranks would issue
    if (ctx->shm) return shm_allreduce(ctx, d_buf, count, false);
    if (ctx->comm) NLG_NCCL(ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, ctx->comm, ctx->stream));
    return 0;
}

int allreduce_max(nlg_ctx *ctx, double *d_buf, int count) {
    ++g_collectives;
    if (ctx->shm) return shm_allreduce(ctx, d_buf, count, true);
    if (ctx->comm) NLG_NCCL(ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclMax, ctx->comm, ctx->stream));
    return 0;
}

int allgather_f64(nlg_ctx *ctx, const double *d_in, double *d_out, int64_t count) {
    ++g_collectives;
    if (ctx->shm)
        return shm_allgather_i64(ctx, reinterpret_cast<const int64_t *>(d_in), reinterpret_cast<int64_t *>(d_out), count);
    if (ctx->comm) {
        NLG_NCCL(ncclAllGather(d_in, d_out, (size_t)count, ncclDouble, ctx->comm, ctx->stream));
        return 0;
    }
    NLG_HIP(hipMemcpyAsync(d_out, d_in, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int scalars_to_host(nlg_ctx *ctx, int first, int count, double *out) {
    NLG_HIP(hipMemcpyAsync(ctx->h_scalars + first, ctx->d_scalars + first, sizeof(double) * count,
                           hipMemcpyDeviceToHost, ctx->stream));
    NLG_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < count; ++i) out[i] = ctx->h_scalars[first + i];
    return 0;
}

int reduce_ws_reserve(nlg_ctx *ctx, int nvec) {
    if (nvec <= ctx->max_red_vec) return 0;
    // the first reservation already covers a Krylov dimension of 256 (2 MB): later bases then never reallocate the
    // workspace under kernels or captured graphs that hold its address (hipFree synchronises the device in any case)
    if (nvec < 264) nvec = 264;
    if (ctx->d_partial) NLG_HIP(hipFree(ctx->d_partial));
    ctx->d_partial = nullptr;
    NLG_HIP(hipMalloc(&ctx->d_partial, sizeof(double) * (size_t)nvec * kMaxBlocksReduce));
    ctx->max_red_vec = nvec;
    return 0;
}

static const char *kProfNames[P_COUNT] = {"axhelm", "gs", "opgradt", "opdiv", "colmul", "block_dot", "block_axpy",
                                          "cg_vec", "conv", "vec_ops", "pprec", "axpy_dot", "cg_update"};

void prof_begin(nlg_ctx *ctx, int id) {
    nlg_prof_slot &s = ctx->prof[id];
    if (s.used + 2 > (int)s.ev.size()) {
        if (s.ev.size() >= 16384) {
            prof_flush(ctx);
        } else {
            const size_t old = s.ev.size();
            s.ev.resize(old + 1024);
            for (size_t i = old; i < s.ev.size(); ++i) hipEventCreate(&s.ev[i]);
        }
    }
    hipEventRecord(s.ev[s.used], ctx->stream);
}

void prof_end(nlg_ctx *ctx, int id) {
    nlg_prof_slot &s = ctx->prof[id];
    hipEventRecord(s.ev[s.used + 1], ctx->stream);
    s.used += 2;
}

int prof_flush(nlg_ctx *ctx) {
    hipStreamSynchronize(ctx->stream);
    for (int id = 0; id < P_COUNT; ++id) {
        nlg_prof_slot &s = ctx->prof[id];
        for (int i = 0; i + 1 < s.used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]) == hipSuccess) {
                s.total_ms += ms;
                s.count += 1;
            }
        }
        s.used = 0;
    }
    return 0;
}

}  // namespace nlg

using namespace nlg;

extern "C" int nlg_prof_enable(nlg_ctx *ctx, int on) {
    NLG_CHECK(ctx, "nlg_prof_enable: NULL ctx");
    prof_flush(ctx);
    ctx->prof_on = on;   // bit mask over kernel classes (1 << class index); -1 = all
    return 0;
}

extern "C" int nlg_prof_sample(nlg_ctx *ctx, int stride) {
    NLG_CHECK(ctx && stride >= 1, "nlg_prof_sample: NULL ctx or stride < 1");
    prof_flush(ctx);
    ctx->prof_stride = stride;
    for (int id = 0; id < P_COUNT; ++id) ctx->prof_seq[id] = 0;
    return 0;
}

extern "C" int nlg_prof_reset(nlg_ctx *ctx) {
    NLG_CHECK(ctx, "nlg_prof_reset: NULL ctx");
    prof_flush(ctx);
    for (int id = 0; id < P_COUNT; ++id) {
        ctx->prof[id].total_ms = 0.0;
        ctx->prof[id].count = 0;
    }
    return 0;
}

extern "C" int nlg_prof_get(nlg_ctx *ctx, const char *name, int64_t *count, double *total_ms) {
    NLG_CHECK(ctx && name, "nlg_prof_get: NULL argument");
    prof_flush(ctx);
    for (int id = 0; id < P_COUNT; ++id)
        if (strcmp(name, kProfNames[id]) == 0) {
            if (count) *count = ctx->prof[id].count;
            if (total_ms) *total_ms = ctx->prof[id].total_ms;
            return 0;
        }
    set_error("nlg_prof_get: unknown kernel class '%s'", name);
    return 1;
}

extern "C" int nlg_counters(int64_t *launches, int64_t *collectives) {
    if (launches) *launches = nlg::g_launches;
    if (collectives) *collectives = nlg::g_collectives;
    return 0;
}

extern "C" {

const char *nlg_last_error(void) { return nlg::g_err; }

int nlg_version(void) { return 100; }

int nlg_ctx_create(int device, nlg_ctx **out) {
    NLG_CHECK(out != nullptr, "nlg_ctx_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("nlg_ctx_create: no HIP device available (%s); this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return 2;
    }
    NLG_CHECK(device >= 0 && device < ndev, "nlg_ctx_create: device %d out of range [0,%d)", device, ndev);
    NLG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    NLG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("nlg_ctx_create: device %d is %s, this library is built for gfx950 (MI355X) only", device,
                  prop.gcnArchName);
        return 2;
    }
    nlg_ctx *ctx = new nlg_ctx();
    ctx->device = device;
    NLG_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    NLG_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    NLG_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    NLG_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    ctx->n_scalars = 4096;
    NLG_HIP(hipMalloc(&ctx->d_scalars, sizeof(double) * ctx->n_scalars));
    NLG_HIP(hipMemsetAsync(ctx->d_scalars, 0, sizeof(double) * ctx->n_scalars, ctx->stream));
    NLG_HIP(hipHostMalloc(&ctx->h_scalars, sizeof(double) * ctx->n_scalars, hipHostMallocDefault));
    NLG_TRY(reduce_ws_reserve(ctx, 16));
    *out = ctx;
    return 0;
}

int nlg_ctx_destroy(nlg_ctx *ctx) {
    if (!ctx) return 0;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->comm) ncclCommDestroy(ctx->comm);
    shm_close(ctx);
    if (ctx->d_partial) hipFree(ctx->d_partial);
    if (ctx->d_scalars) hipFree(ctx->d_scalars);
    if (ctx->h_scalars) hipHostFree(ctx->h_scalars);
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int nlg_ctx_sync(nlg_ctx *ctx) {
    NLG_CHECK(ctx, "nlg_ctx_sync: NULL ctx");
    NLG_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int nlg_comm_unique_id(void *out128) {
    NLG_CHECK(out128, "nlg_comm_unique_id: NULL");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    NLG_NCCL(ncclGetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return 0;
}

int nlg_ctx_comm_init(nlg_ctx *ctx, int rank, int nranks, const void *unique_id128) {
    NLG_CHECK(ctx && unique_id128, "nlg_ctx_comm_init: NULL argument");
    NLG_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "nlg_ctx_comm_init: bad rank %d / %d", rank, nranks);
    NLG_CHECK(!ctx->comm && !ctx->shm, "nlg_ctx_comm_init: the context already has a communicator");
    NLG_HIP(hipSetDevice(ctx->device));
    ctx->rank = rank;
    ctx->nranks = nranks;
    // a single rank needs no communicator; NLG_FORCE_COMM=1 creates one anyway so that the RCCL code paths
    // (all-reduce, all-gather of labels) can be exercised on a one-GPU box
    if (nranks == 1 && !getenv("NLG_FORCE_COMM")) return 0;
    ncclUniqueId id;
    memcpy(&id, unique_id128, sizeof(id));
    NLG_NCCL(ncclCommInitRank(&ctx->comm, nranks, id, rank));
    return 0;
}

int nlg_ctx_rank(const nlg_ctx *ctx, int *rank, int *nranks) {
    NLG_CHECK(ctx, "nlg_ctx_rank: NULL ctx");
    if (rank) *rank = ctx->rank;
    if (nranks) *nranks = ctx->nranks;
    return 0;
}

}  // extern "C"
