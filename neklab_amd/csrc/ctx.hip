// Context, error reporting and the RCCL communicator of libneklab_gpu.
#include <cstdarg>

#include "internal.h"

namespace nlg {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int allreduce_sum(nlg_ctx *ctx, double *d_buf, int count) {
    if (ctx->nranks > 1) NLG_NCCL(ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, ctx->comm, ctx->stream));
    return 0;
}

int allreduce_max(nlg_ctx *ctx, double *d_buf, int count) {
    if (ctx->nranks > 1) NLG_NCCL(ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclMax, ctx->comm, ctx->stream));
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
    if (ctx->d_partial) NLG_HIP(hipFree(ctx->d_partial));
    ctx->d_partial = nullptr;
    NLG_HIP(hipMalloc(&ctx->d_partial, sizeof(double) * (size_t)nvec * kMaxBlocksReduce));
    ctx->max_red_vec = nvec;
    return 0;
}

}  // namespace nlg

using namespace nlg;

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
    if (ctx->d_partial) hipFree(ctx->d_partial);
    if (ctx->d_scalars) hipFree(ctx->d_scalars);
    if (ctx->h_scalars) hipHostFree(ctx->h_scalars);
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
    NLG_HIP(hipSetDevice(ctx->device));
    ctx->rank = rank;
    ctx->nranks = nranks;
    if (nranks == 1) return 0;
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
