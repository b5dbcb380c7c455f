// Internal declarations of libneklab_gpu (gfx950 only). Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "neklab_gpu.h"

namespace nlg {

void set_error(const char *fmt, ...);

#define NLG_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            nlg::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

#define NLG_NCCL(call)                                                                            \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            nlg::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

#define NLG_CHECK(cond, ...)             \
    do {                                 \
        if (!(cond)) {                   \
            nlg::set_error(__VA_ARGS__); \
            return 1;                    \
        }                                \
    } while (0)

#define NLG_TRY(call)           \
    do {                        \
        int rc_ = (call);       \
        if (rc_) return rc_;    \
    } while (0)

// every kernel launch of the library goes through this macro: the count is what bench.py reports as launches per step
extern int64_t g_launches, g_collectives;
#define NLG_LAUNCH(...)                     \
    do {                                    \
        ++nlg::g_launches;                  \
        hipLaunchKernelGGL(__VA_ARGS__);    \
    } while (0)

constexpr int kMaxLanes = 4;            // vectors advanced together by the block stepper
constexpr int kMaxBlocksReduce = 1024;  // fixed first-stage grid => run-to-run deterministic sums
constexpr int kAlign = 32;              // field starts aligned to 32 doubles (256 B)

inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

}  // namespace nlg

// kernel classes that can be timed with HIP events on the launch stream (bench.py roofline leg)
enum { P_AXHELM = 0, P_GS, P_OPGRADT, P_OPDIV, P_COLMUL, P_BLOCKDOT, P_BLOCKAXPY, P_CGVEC, P_CONV, P_VECOPS, P_PPREC, P_AXPYDOT, P_CGUPDATE, P_COUNT };

struct nlg_prof_slot {
    std::vector<hipEvent_t> ev;   // pairs (begin, end)
    int used = 0;
    double total_ms = 0.0;
    int64_t count = 0;
};

namespace nlg {
struct nlg_shm;   // host-staged validation transport (shm_transport.hip)
}

struct nlg_ctx {
    int prof_on = 0;
    int prof_stride = 1;                      // time every prof_stride-th launch of an enabled class (nlg_prof_sample)
    int64_t prof_seq[16] = {};
    nlg_prof_slot prof[P_COUNT];
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // side stream: coarse-grid branch of the pressure preconditioner
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    ncclComm_t comm = nullptr;
    nlg::nlg_shm *shm = nullptr;              // set only by nlg_ctx_comm_init_shm (several test ranks on one GPU)
    bool distributed() const { return comm != nullptr || shm != nullptr; }
    int rank = 0, nranks = 1;
    // reduction workspace
    double *d_partial = nullptr;   // [kMaxVecReduce * kMaxBlocksReduce]
    double *d_scalars = nullptr;   // small device scalars (results of reductions)
    double *h_scalars = nullptr;   // pinned host mirror
    int n_scalars = 0;
    int max_red_vec = 0;
};

// Small dense operators of the discretisation, host copies (device copies live in nlg_mesh::d_ops)
struct nlg_ops1d {
    int n = 0, n2 = 0, nd = 0;
    std::vector<double> z1, w1, z2, w2, zd, wd;
    std::vector<double> D;     // n  x n   d/dr on GLL
    std::vector<double> I12;   // n2 x n   GLL -> GL(n2)
    std::vector<double> D12;   // n2 x n
    std::vector<double> Jd;    // nd x n   GLL -> GL(nd)
    std::vector<double> DJd;   // nd x n
    std::vector<double> rdr;   // n  inverse reference spacing (CFL)
};

// Face-grouped slot of point (a, j, k) inside an element: the 8 corners, the 12 edges (interior points of an edge
// contiguous), the 6 face interiors (contiguous (N-2)^2 blocks), then the element interior.  The copies of a shared
// face or edge are then contiguous runs in every element that shares it, so the gather-scatter moves whole runs
// instead of one 8-byte word per 64-byte line.  FG = false gives the natural ix-fastest index.
__host__ __device__ inline int fg_slot(int N, int a, int j, int k) {
    const int M = N - 2;
    const int ba = (a == 0 || a == N - 1), bj = (j == 0 || j == N - 1), bk = (k == 0 || k == N - 1);
    const int sa = a == N - 1, sj = j == N - 1, sk = k == N - 1;
    const int nb = ba + bj + bk;
    if (nb == 3) return sa + 2 * sj + 4 * sk;
    if (nb == 2) {
        if (!ba) return 8 + (0 + sj + 2 * sk) * M + (a - 1);
        if (!bj) return 8 + (4 + sa + 2 * sk) * M + (j - 1);
        return 8 + (8 + sa + 2 * sj) * M + (k - 1);
    }
    const int fbase = 8 + 12 * M;
    if (nb == 1) {
        if (ba) return fbase + (0 + sa) * M * M + (j - 1) + M * (k - 1);
        if (bj) return fbase + (2 + sj) * M * M + (a - 1) + M * (k - 1);
        return fbase + (4 + sk) * M * M + (a - 1) + M * (j - 1);
    }
    return fbase + 6 * M * M + (a - 1) + M * ((j - 1) + M * (k - 1));
}

// "Slab-permuted" slot of point (a, j, k) (historically "xp", x-planes first): every k-slab of an element stays one dense run
// of N N doubles, but inside the slab the points are permuted so that the element-boundary points are contiguous:
//   [ a = 0 row (j = 0..N-1) | a = N-1 row | j = 0 row (a = 1..N-2) | j = N-1 row | interior ]
// and, at N = 8, two interior points pad each j-row so that it sits inside one 64-byte sector.  A kernel whose lanes are
// (a, j) columns reads a slab as ONE coalesced run exactly as in the natural layout (the permutation only decides which
// lane gets which word), while the copies of a shared x-face -- one 8-byte word per 64-byte row in the natural layout --
// become sector-aligned runs of N, y-faces runs of N-2 inside one sector, z-faces whole slabs.  Used for the vectors of
// the velocity PCG (3-D), whose gather-scatter moved 3.2 times its algorithmic bytes in the natural layout.
// Measured on the way here (10^4 elements, lx1 = 8, ms per step): natural gs 7.09 / operator 6.07; x-faces as two separate
// N x N planes: gs 6.2 / operator 6.7 (half-line rows); both x-rows of a slab in one line, rest natural: 6.07 / 6.32.
inline void sp_slab_table(int N, int *tab /* [N*N], index a + N*j */) {
    std::vector<int> order;   // slab positions (a + N*j) in storage order
    std::vector<char> used((size_t)N * N, 0);
    auto put = [&](int a, int j) {
        order.push_back(a + N * j);
        used[a + N * j] = 1;
    };
    for (int j = 0; j < N; ++j) put(0, j);
    for (int j = 0; j < N; ++j) put(N - 1, j);
    auto filler = [&](int count) {
        for (int j = 1; j < N - 1 && count > 0; ++j)
            for (int a = 1; a < N - 1 && count > 0; ++a)
                if (!used[a + N * j]) {
                    put(a, j);
                    --count;
                }
    };
    for (int a = 1; a < N - 1; ++a) put(a, 0);
    if (N == 8) filler(2);
    for (int a = 1; a < N - 1; ++a) put(a, N - 1);
    if (N == 8) filler(2);
    filler(N * N);
    for (int q = 0; q < N * N; ++q) tab[order[q]] = q;
}
inline int xp_slot(int N, int a, int j, int k) {
    static thread_local int cachedN = 0;
    static thread_local std::vector<int> tab;
    if (cachedN != N) {
        tab.assign((size_t)N * N, 0);
        sp_slab_table(N, tab.data());
        cachedN = N;
    }
    // NLG_XP_LAYOUT=1 (lx1 = 8, experiment): a 128-byte line must not hold rows of two different FACES -- the two faces are summed by
    // different blocks of the gather-scatter, on different XCDs, and each fetches the whole line (measured traffic 2.03 x algorithmic).
    // Interior slabs 1..6 are interleaved in pairs, row by row: line r of a pair block = [row r of slab k | row r of slab k + 1], so the
    // a = 0 rows of two consecutive slabs (same face) share a line; slabs 0 and 7 (z faces) hold their 36 face points, then the four
    // edges, then the corners.  The operator kernels address through the element's slot table, so any permutation will do for them.
    static const int mode = getenv("NLG_XP_LAYOUT") ? atoi(getenv("NLG_XP_LAYOUT")) : 0;
    if (mode == 2) return fg_slot(N, a, j, k);   // (experiment: the face-grouped layout of the pressure operator for the velocity PCG too)
    if (mode == 1 && N == 8) {
        const int NS = N * N;
        if (k >= 1 && k <= N - 2) {
            const int t = tab[a + N * j], pair = (k - 1) / 2, half = (k - 1) % 2;
            return NS + pair * 2 * NS + (t / 8) * 16 + half * 8 + (t % 8);
        }
        const int base = k == 0 ? 0 : NS + (N - 2) * NS;
        const bool ia = a > 0 && a < N - 1, ij = j > 0 && j < N - 1;
        if (ia && ij) return base + (a - 1) + (N - 2) * (j - 1);
        if (!ia && !ij) return base + 60 + (a == 0 ? 0 : 1) + 2 * (j == 0 ? 0 : 1);
        const int edge = !ia ? (a == 0 ? 0 : 1) : (j == 0 ? 2 : 3);
        return base + 36 + edge * 6 + (!ia ? j - 1 : a - 1);
    }
    return N * N * k + tab[a + N * j];
}

struct nlg_gs_tab {   // one table of groups as k_gs reads it: pairs first, then quads, then the rest (CSR)
    int64_t ngroups = 0, npairs = 0, nquads = 0;
    int *d_off = nullptr, *d_idx = nullptr;
};
struct nlg_gs {
    // groups of local dofs that share a global label (only groups of size >= 2 are stored)
    int64_t ngroups = 0;
    int64_t nshared = 0;
    int64_t npairs = 0;         // the first npairs groups have exactly two copies (same count in both layouts)
    int64_t nquads = 0;         // the next nquads groups have exactly four
    int *d_offsets = nullptr;   // [ngroups + 1]
    int *d_indices = nullptr;   // [nshared] local dof index
    // the same groups in the face-grouped element layout used for the intermediate fields of the consistent
    // Poisson operator (3-D): element-boundary points first, face by face, so that the copies of a shared face
    // are contiguous runs instead of stride-n points
    int *d_offsets_fg = nullptr;
    int *d_indices_fg = nullptr;
    // ... and in the x-planes-first layout of the velocity PCG (3-D)
    int *d_offsets_xp = nullptr;
    int *d_indices_xp = nullptr;
    // several ranks: the groups split into those holding a dof that another rank shares (summed first, so that the halo
    // exchange can start) and all others (summed while the exchange is under way); one pair of tables per layout
    bool split = false;
    nlg_gs_tab tab_halo[3], tab_rest[3];
    std::vector<std::vector<int>> h_groups;   // the groups in natural indices; kept only until halo_setup has split them
};
enum { LAYOUT_NAT = 0, LAYOUT_FG = 1, LAYOUT_XP = 2 };

struct nlg_halo {
    bool active = false;
    std::vector<int> neigh;          // neighbour ranks (ascending)
    std::vector<int64_t> noff, ncnt; // offset / count of each neighbour's shared labels in the packed buffers
    int64_t ntot = 0, nlab = 0;
    int *d_send_idx = nullptr;       // [ntot] representative local dof per (neighbour, label)
    int *d_roff = nullptr, *d_rpos = nullptr;   // per distinct shared label: positions in the recv buffer
    int *d_coff = nullptr, *d_cidx = nullptr;   // per distinct shared label: all local copies
    int *d_send_idx_fg = nullptr, *d_cidx_fg = nullptr;   // d_send_idx / d_cidx for the face-grouped element layout
    int *d_send_idx_xp = nullptr, *d_cidx_xp = nullptr;   // ... and for the x-planes-first layout
    double *d_send = nullptr, *d_recv = nullptr;
    std::vector<int> h_cidx;         // host copy of d_cidx: every local dof that another rank shares
    bool overlap = false;            // send / receive on the side stream while the interior groups are summed (NLG_HALO_OVERLAP)
    hipEvent_t ev_packed = nullptr, ev_recv = nullptr;
};

// two-level preconditioner of the pressure operator (pprec.hip)
struct nlg_pprec {
    bool ready = false;
    int na = 0;                                  // aggregates
    int nvert = 0, ncorner = 0;                  // coarse dofs: element vertices (trilinear coarse space), 2^dim per element
    int *d_vg = nullptr;                         // [E][ncorner] vertex id of every element corner
    int *d_v2e_p = nullptr, *d_v2e_i = nullptr;  // vertex -> incident (element*ncorner + corner) entries, CSR
    double *d_tq = nullptr;                      // [E][ncorner] element-local restriction
    double hat1[16] = {};                        // (1 + z2)/2 at the GL points: the 1-D hat function of the upper corner
    double *d_S = nullptr, *d_invden = nullptr;  // FDM: [E][3][n2*n2] eigenvector matrices, [E][n2^dim] 1/(sum of eigenvalues)
    double *d_dinv = nullptr;                    // 1 / diag(A_c)
    // overlapping variant (3-D, lx1 <= 8): extended 1-D eigen-decompositions [E][3][n*n], eigenvalues [E][3][n], the
    // velocity-shaped exchange array (face-grouped layout) and the zero-denominator threshold
    bool overlap = false;
    int *d_exttab = nullptr;                     // [n^3] packed per-point constants of the extended grid (k_fdm_ext)
    int *d_wslot = nullptr;                      // [n2^3][3] face slot of W next to a pressure point per direction, -1 = none
    double *d_Sx = nullptr, *d_lamx = nullptr, *d_W = nullptr, *d_wq = nullptr;   // d_wq: count^-1/2 weights [E][n2^3]
    double thrx = 0.0;
    int *d_agg = nullptr, *d_ap = nullptr, *d_am = nullptr;
    double *d_Ainv = nullptr;                    // dense inverse on the aggregates: this rank's rows, [na][ncols]
    float *d_Ainv32 = nullptr;                   // several ranks: the rows in single precision instead of d_Ainv
    int na_max = 0, ncols = 0;                   // several ranks: ncols = nranks * na_max columns (global aggregate level)
    double *d_rag = nullptr;                     // [ncols] aggregate residuals of all ranks (all-gather of d_ra)
    double *d_rc = nullptr, *d_x = nullptr, *d_ra = nullptr, *d_xa = nullptr;
    // lanes of a block step: `lanes_cap` copies of W, tq, rc / x, ra, xa at these strides (pprec_reserve_lanes)
    int lanes_cap = 1;
    int64_t lW = 0, lt = 0, lv = 0, la = 0, la_x = 0;
    bool coarse_pending = false;                 // pprec_coarse has left its chain (gather, restriction, dense solve) to the merged launches of pprec_fine
};

struct nlg_mesh {
    nlg_pprec pprec;
    nlg_halo halo;
    nlg_ctx *ctx = nullptr;
    int dim = 3, n = 8, n2 = 6, nd = 12;
    int64_t E = 0;
    int np1 = 0, np2 = 0, npd = 0;      // points per element on the three meshes
    int64_t lvn = 0, lpn = 0, lvs = 0, lps = 0, lfn = 0;  // lvs/lps: padded field strides
    int has_outflow = 0;
    nlg_ops1d ops;
    // device copies of the 1-D operators, row-major; transposes stored too
    double *d_D = nullptr, *d_Dt = nullptr, *d_I12 = nullptr, *d_I12t = nullptr, *d_D12 = nullptr, *d_D12t = nullptr;
    double *d_Jd = nullptr, *d_Jdt = nullptr, *d_DJd = nullptr, *d_DJdt = nullptr, *d_rdr = nullptr;
    double *d_w1 = nullptr, *d_w2 = nullptr, *d_wd = nullptr;
    // geometry (velocity mesh), each lvn
    double *d_x[3] = {nullptr, nullptr, nullptr};
    double *d_rst[9] = {};     // rst[j*dim+i] = J * dr_j/dx_i
    double *d_jac = nullptr, *d_bm1 = nullptr, *d_binvm1 = nullptr, *d_vmult = nullptr;
    double *d_G[6] = {};       // 3-D: 11,12,13,22,23,33 ; 2-D: 11,12,22
    double *d_mask[3] = {nullptr, nullptr, nullptr};
    double *d_tmask = nullptr;
    double *d_mbinv[3] = {nullptr, nullptr, nullptr};   // mask_i * binvm1 (fused opbinv weight)
    double *d_mbinv_fg[3] = {nullptr, nullptr, nullptr};   // the same in the face-grouped layout
    double *d_binv_fg = nullptr;                           // binvm1 alone, face-grouped, and ...
    unsigned char *d_maskb_fg = nullptr;                   // ... one byte per point, bit i = mask_i: the three weights of k_opdiv3n in 8.5 bytes per point
    // pressure mesh
    double *d_rst2w[9] = {};   // each lpn
    double *d_bm2 = nullptr, *d_bm2inv = nullptr;
    // fine mesh
    double *d_rstdw[9] = {};   // each lfn
    int64_t *d_lglel = nullptr;
    std::vector<int64_t> h_lglel;
    std::vector<int> h_slot;   // natural point -> face-grouped slot inside an element (3-D)
    std::vector<int> h_slot_xp;   // natural point -> x-planes-first slot (3-D)
    int *d_slot_xp = nullptr;
    int *d_slot_fg = nullptr;   // device copy of h_slot (lx1 > 8 pressure kernels read it instead of computing the slot)
    double *d_vmult_xp = nullptr;
    double *d_wlanes = nullptr;   // [kMaxLanes][3][lvs] intermediate fields of the pressure operator of a block step (lazy)
    nlg_gs gs;
    double volvm1 = 0, volvm2 = 0;
    int64_t lpn_global = 0;   // global pressure dof count (ortho)
    // scratch fields
    std::vector<double *> scratch1, scratchd, scratch2;   // velocity-mesh / fine-mesh / pressure-mesh scratch
    std::map<std::string, const double *> named;          // for nlg_mesh_get
    std::map<std::string, int64_t> named_len;
};

struct nlg_vec {
    nlg_mesh *mesh = nullptr;
    int nscal = 0, lorder = 3;
    int ncomp = 0;             // dim + nscal : fields in the inner product
    int64_t main_len = 0;      // ncomp*lvs + lps
    int64_t total_len = 0;     // main_len * lorder
    double *d = nullptr;
    bool owns = true;
    int nrst = 0;
    // field pointers
    double *vel(int i, int irst = 0) const { return d + irst * main_len + (int64_t)i * mesh->lvs; }
    double *theta(int m, int irst = 0) const { return d + irst * main_len + (int64_t)(mesh->dim + m) * mesh->lvs; }
    double *pr(int irst = 0) const { return d + irst * main_len + (int64_t)ncomp * mesh->lvs; }
};

struct nlg_basis {
    nlg_mesh *mesh = nullptr;
    int nvec = 0, nscal = 0, lorder = 3;
    int64_t stride = 0;        // doubles between consecutive vectors
    double *d = nullptr;
    std::vector<nlg_vec *> views;
    double *d_h = nullptr;     // [2 * nvec + 8] device coefficients
    double *d_hb = nullptr;    // block orthogonalisation: [2 * nvec * 4 + 64] coefficients of up to 4 vectors (lazy)
    int last_block_rank = 0;   // columns kept by the last nlg_basis_block_cgs2 (< s: dependent columns were deflated)
};

namespace nlg {

// ---- ctx.hip: optional per-kernel-class event timing ----
void prof_begin(nlg_ctx *ctx, int id);
void prof_end(nlg_ctx *ctx, int id);
int prof_flush(nlg_ctx *ctx);
// is this launch of class `id` to be timed?  Enabled classes are sampled: every prof_stride-th launch gets its pair of events
// (an event pair around a 50-us kernel costs ~12 us of stream time: timing every launch of the dominant class took 2 % off the
// benchmark it was measuring)
inline bool prof_want(nlg_ctx *c, int id) {
    if (!(c->prof_on & (1 << id))) return false;
    return (c->prof_seq[id]++ % c->prof_stride) == 0;
}
struct ProfScope {
    nlg_ctx *c;
    int id;
    bool on;
    ProfScope(nlg_ctx *ctx, int i) : c(ctx), id(i), on(prof_want(ctx, i)) {
        if (on) prof_begin(c, id);
    }
    ~ProfScope() {
        if (on) prof_end(c, id);
    }
};

// ---- vec.hip ----
int reduce_ws_reserve(nlg_ctx *ctx, int nvec);
// weighted dot over the inner-product part, result left on device at ctx->d_scalars[slot]; allreduced
int dev_dot(const nlg_vec *a, const nlg_vec *b, int slot);
int scalars_to_host(nlg_ctx *ctx, int first, int count, double *out);   // syncs the stream
int allreduce_sum(nlg_ctx *ctx, double *d_buf, int count);
int allreduce_max(nlg_ctx *ctx, double *d_buf, int count);
int allgather_f64(nlg_ctx *ctx, const double *d_in, double *d_out, int64_t count);   // count doubles per rank
int shm_allreduce(nlg_ctx *ctx, double *d_buf, int count, bool is_max);
int shm_allgather_i64(nlg_ctx *ctx, const int64_t *d_in, int64_t *d_out, int64_t count);
int shm_exchange(nlg_ctx *ctx, const nlg_halo &h, int nf, hipStream_t st);   // st: the stream the staging copies are ordered on
void shm_close(nlg_ctx *ctx);

int basis_block_dot_dev(const nlg_basis *b, int k, const nlg_vec *w, double *d_out, double *d_acc);
int basis_block_axpy_dev(const nlg_basis *b, int k, const double *d_h, nlg_vec *w, double sign, const double *d_hh = nullptr,
                         bool main_only = false);
int basis_cgs2_dev(nlg_basis *b, int k, nlg_vec *w);

// ---- pprec.hip ----
int pprec_setup(nlg_mesh *m, const nlg_mesh_desc *d);
// PCG update fused into the first kernel of the preconditioner (null alpha: none): x += alpha p, r -= alpha (w - wmean)
// written back, rr_part[block] = sum r^2 nw -- the residual is in registers there anyway
struct nlg_pcg_upd {
    const double *alpha = nullptr, *wmean = nullptr;
    double *x = nullptr;
    const double *p = nullptr, *w = nullptr, *nw = nullptr;
    double *rr_part = nullptr;   // [(E + 3) / 4]
};
int pprec_coarse(nlg_mesh *m, hipStream_t st, const double *flag, const double *r, const double **xc, bool overlap = false,
                 const nlg_pcg_upd *upd = nullptr, int nl = 1, int64_t ld = 0);
int pprec_fine(nlg_mesh *m, hipStream_t st, const double *flag, const double *r, const double *xc, double *z,
               double *rz_part = nullptr, bool overlap = false, int nl = 1, int64_t ld = 0);
int pprec_reserve_lanes(nlg_mesh *m, int nl);
void pprec_free(nlg_mesh *m);

// ---- lns.hip
bool linop_can_block(const nlg_linop *op);   // the multi-vector stepper covers this operator (round 4: also with the Boussinesq coupling and the wavenumber projection)

// ---- halo.hip ----
int halo_setup(nlg_mesh *m, const int64_t *glo_num);
int halo_exchange(nlg_mesh *m, double *const *fields, int nf, int layout = 0, int nl = 1, int64_t ld = 0);   // LAYOUT_*; = halo_begin + halo_finish
int gs_split(nlg_mesh *m);   // builds gs.tab_halo / tab_rest from gs.h_groups and the halo lists
int halo_begin(nlg_mesh *m, double *const *fields, int nf, int layout, int nl = 1, int64_t ld = 0);    // pack + start of the exchange
int halo_finish(nlg_mesh *m, double *const *fields, int nf, int layout, int nl = 1, int64_t ld = 0);   // end of the exchange + unpack
void halo_free(nlg_mesh *m);

// ---- sem.hip (device-pointer level operators; all on ctx->stream) ----
// nl / ld / ldg (everywhere below): nl lanes of a block step in ONE launch (gridDim.y = nl): lane v's fields sit v * ld doubles
// behind the given (lane-0) pointers, its gate v * ldg doubles behind `gate`
int sem_gs(nlg_mesh *m, double *const *fields, int nf, const double *gate = nullptr, int layout = 0, int nl = 1, int64_t ld = 0, int64_t ldg = 0);   // in place QQ^T; gate: device flag, non-zero = skip; layout: LAYOUT_NAT or LAYOUT_XP
int sem_to_xp(nlg_mesh *m, double *const *src, double *const *dst, int nf, int nl = 1, int64_t ld = 0, double *const *wts = nullptr);     // natural -> x-planes-first (out of place); wts: dst = wts * src
int sem_from_xp(nlg_mesh *m, double *const *src, double *const *dst, int nf, int nl = 1, int64_t ld = 0);
int sem_gs_pairs_fg(nlg_mesh *m, double *w, const double *gate = nullptr, int nl = 1, int64_t ld = 0, int64_t ldg = 0);
int sem_gs_pairs(nlg_mesh *m, double *w, const double *gate = nullptr, int nl = 1, int64_t ld = 0, int64_t ldg = 0);   // the same in the natural layout (2-D Schwarz exchange)   // rank-local QQ^T over the two-copy groups (face interiors) of one field in the face-grouped layout
int sem_axhelm(nlg_mesh *m, double *const *u, double *const *w, int nf, double h1, double h2, double *pw_part = nullptr,
               double *const *zf = nullptr, const double *beta_p = nullptr, const double *done_p = nullptr, bool xp = false, int nl = 1, int64_t ld = 0, int64_t uoff = 0);   // uoff: the updated direction is stored uoff doubles behind u (direction history of the PCG); zf: fused u <- zf + beta u; xp: u, zf, w in the x-planes-first layout (3-D, lx1 <= 8)
int sem_axhelm_lanes(nlg_mesh *m, int nl, double *const *const *u, double *const *const *w, double h1, double h2, double *const *pw,
                     double *const *const *zf, const double *const *beta, const double *const *done, bool xp);
int sem_opdiv_blocks(const nlg_mesh *m);
int sem_axhelm_blocks(nlg_mesh *m, int nf);   // 3-D: number of per-block sums of u . w_local written to pw_part
int sem_helm_diag(nlg_mesh *m, double *out, double h1, double h2);   // local diag (not assembled)
struct nlg_pupd;
bool sem_opgradt_has_fg(const nlg_mesh *m);
int sem_opgradt(nlg_mesh *m, const double *p, double *const *w, bool face_grouped = false, const double *gate = nullptr, const nlg_pupd *upd = nullptr);
int sem_opdiv(nlg_mesh *m, double *const *u, double *out, double scale, double *const *wts = nullptr, bool face_grouped = false,
              const double *pdot = nullptr, double *pw_part = nullptr, const double *gate = nullptr);
int sem_opbinv(nlg_mesh *m, double *const *w, int nl = 1, int64_t ld = 0);                       // w_i <- mask_i binv QQ^T w_i
// direction update of a PCG performed by the operator while it loads p:  p <- (z - zmean[0]) + beta[0] p   (device scalars)
struct nlg_pupd {
    const double *z = nullptr, *beta = nullptr, *zmean = nullptr;
    double *p = nullptr;
};
bool sem_opgradt_fuses_pupdate(const nlg_mesh *m);
bool sem_small_mesh(const nlg_mesh *m);   // local element count below the threshold of the strong-scaling kernel variants (NLG_SMALL_E)
int sem_cdabdtp(nlg_mesh *m, const double *p, double *out, double *pw_part = nullptr, const double *gate = nullptr, const nlg_pupd *upd = nullptr);
int sem_cdabdtp_lanes(nlg_mesh *m, int nl, const double *const *p, double *const *out, double *const *pw_part, const double *const *gate,
                      const nlg_pupd *upd = nullptr);
int sem_opgradt_lanes(nlg_mesh *m, int nl, const double *const *p, double *const *const *w, bool face_grouped, const double *const *gate,
                      const nlg_pupd *upd = nullptr);
int sem_opdiv_lanes(nlg_mesh *m, int nl, double *const *const *u, double *const *out, double scale, double *const *wts, bool face_grouped,
                    const double *const *pdot, double *const *pw_part, const double *const *gate);
int sem_ediag(nlg_mesh *m, double *out);
int sem_tensor(nlg_mesh *m, const double *in, double *out, int nin, int nout, const double *Mx, const double *My,
               const double *Mz, const double *wt);
int sem_conv_setup(nlg_mesh *m, double *const *U, double **Ur, double **GU);
int sem_conv_apply(nlg_mesh *m, double *const *Ur, double *const *GU, double *const *u, double *const *out, int adjoint);
int sem_conv_apply_lanes(nlg_mesh *m, double *const *Ur, double *const *GU, int nl, double *const *const *ulanes, double *const *const *olanes, int adjoint);
int sem_conv_apply_generic(nlg_mesh *m, double *const *Ur, double *const *GU, double *const *u, double *const *out, int adjoint);
int sem_conv_scalar_setup(nlg_mesh *m, const double *Theta, double **GT);
int sem_conv_scalar_apply(nlg_mesh *m, double *const *Ur, double *const *GT, double *const *u, const double *theta, double *out, int adjoint = 0);
int sem_scalar_grad_apply(nlg_mesh *m, double *const *GT, const double *theta, double *const *out, double sgn);
int sem_cfl(nlg_mesh *m, double *const *U, double dt, double *cfl_host);
int sem_ortho(nlg_mesh *m, double *p, int nl = 1, int64_t ld = 0);
double *sem_scratch1(nlg_mesh *m, int i);
double *sem_scratchd(nlg_mesh *m, int i);
double *sem_scratch2(nlg_mesh *m, int i);

}  // namespace nlg
