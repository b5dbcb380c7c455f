// nek_dvector vector space and Krylov-basis block kernels (gfx950).
//
// Layout of one vector in HBM (doubles):
//   [ v_0 | v_1 | (v_2) | theta_0.. | pr ]  main block, `main_len = ncomp*lvs + lps`
//   followed by lorder-1 history blocks of the same shape (reference: neklab_vectors.f90:30-35).
// `lvs`/`lps` are the field lengths rounded up to 32 doubles; the padding is zero in every vector and
// in the weight `bm1`, so every kernel can sweep whole padded ranges with 16-byte accesses.
//
// Roofline: every kernel here is HBM-bound (<= 0.25 flop/B).  Algorithmic bytes per call:
//   scal 16 B/dof, axpby 24 B/dof, dot 24 B/dof (a, b, bm1), block_dot 8*(k+2) B per inner-product
//   dof, block_axpy 8*(k+2) B per dof.
#include <algorithm>

#include "internal.h"

using namespace nlg;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the 256 threads of a block; valid in thread 0. `sm` needs 4 doubles.
__device__ __forceinline__ double block_sum(double v, double *sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += sm[i];
    }
    return r;
}

__global__ __launch_bounds__(NT) void k_fill(double2 *x, double v, int64_t n2) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) x[i] = make_double2(v, v);
}

__global__ __launch_bounds__(NT) void k_copy(double2 *y, const double2 *x, int64_t n2) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) y[i] = x[i];
}

// reference cmult: x(i) = x(i)*alpha  (real_vectors.f90:131)
__global__ __launch_bounds__(NT) void k_scal(double2 *x, double a, int64_t n2) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) {
        double2 v = x[i];
        v.x = __dmul_rn(v.x, a);
        v.y = __dmul_rn(v.y, a);
        x[i] = v;
    }
}

// x *= f(s[0]) with the scalar on device: mode 0: 1/sqrt(s), mode 1: 1/s
__global__ __launch_bounds__(NT) void k_scal_dev(double2 *x, const double *s, int mode, int64_t n2) {
    const double sv = s[0];
    const double a = (mode == 0) ? (sv > 0.0 ? 1.0 / sqrt(sv) : 0.0) : 1.0 / sv;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) {
        double2 v = x[i];
        v.x = __dmul_rn(v.x, a);
        v.y = __dmul_rn(v.y, a);
        x[i] = v;
    }
}

// reference nek_daxpby (real_vectors.f90:162-206): scal(beta) then add2s2(self, vec, alpha) on the main
// block, and for each valid history slot of self: slot += alpha * (vec main | vec slot).
// Rounding mirrors the two sweeps: fl(fl(beta*y) + fl(alpha*x)).
__global__ __launch_bounds__(NT) void k_axpby(double2 *y, const double2 *x, double a, double b, int64_t n2,
                                              int nrst, int64_t blk2, int consistent) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) {
        const double2 xv = x[i];
        const double ax0 = __dmul_rn(a, xv.x), ax1 = __dmul_rn(a, xv.y);
        double2 yv = y[i];
        yv.x = __dadd_rn(__dmul_rn(yv.x, b), ax0);
        yv.y = __dadd_rn(__dmul_rn(yv.y, b), ax1);
        y[i] = yv;
        for (int r = 1; r <= nrst; ++r) {
            double2 rv = y[i + r * blk2];
            double s0 = ax0, s1 = ax1;
            if (consistent) {
                const double2 xr = x[i + r * blk2];
                s0 = __dmul_rn(a, xr.x);
                s1 = __dmul_rn(a, xr.y);
            }
            rv.x = __dadd_rn(__dmul_rn(rv.x, b), s0);
            rv.y = __dadd_rn(__dmul_rn(rv.y, b), s1);
            y[i + r * blk2] = rv;
        }
    }
}

// first stage of glsc3-type sums over the ncomp inner-product fields:
// partial[c*nblk + blk] = sum_{i in chunk} a_c[i] b_c[i] bm1[i]
__global__ __launch_bounds__(NT) void k_dot_partial(const double *a, const double *b, const double *bm1, int64_t lvs,
                                                    int nblk, double *partial) {
    __shared__ double sm[4];
    const int c = blockIdx.y;
    const int64_t n2 = lvs >> 1;
    const int64_t per = (n2 + nblk - 1) / nblk;
    const int64_t beg = blockIdx.x * per, end = (beg + per < n2) ? beg + per : n2;
    const double2 *a2 = reinterpret_cast<const double2 *>(a + c * lvs);
    const double2 *b2 = reinterpret_cast<const double2 *>(b + c * lvs);
    const double2 *m2 = reinterpret_cast<const double2 *>(bm1);
    double acc = 0.0;
    for (int64_t i = beg + threadIdx.x; i < end; i += NT) {
        const double2 av = a2[i], bv = b2[i], mv = m2[i];
        acc += av.x * bv.x * mv.x + av.y * bv.y * mv.y;
    }
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[c * nblk + blockIdx.x] = s;
}

// second stage, fixed order => deterministic: out[row] (+)= sum_b partial[row*nper + b]
__global__ __launch_bounds__(NT) void k_reduce_rows(const double *partial, int nper, double *out, int accumulate,
                                                    double *out2) {
    __shared__ double sm[4];
    const double *p = partial + (int64_t)blockIdx.x * nper;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nper; i += NT) acc += p[i];
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        out[blockIdx.x] = s;
        if (accumulate) out2[blockIdx.x] += s;
    }
}

// h_j = V_j^T (bm1 o w) for a tile of KB basis vectors; V is read exactly once.
template <int KB>
__global__ __launch_bounds__(NT) void k_block_dot(const double *V, int64_t vstride, int k, const double *w,
                                                  const double *bm1, int64_t lvs, int nblk, int nper,
                                                  double *partial) {
    __shared__ double sm[4 * KB];
    const int c = blockIdx.y;
    const int j0 = blockIdx.z * KB;
    const int kc = (k - j0 < KB) ? (k - j0) : KB;
    const int64_t n2 = lvs >> 1;
    const int64_t per = (n2 + nblk - 1) / nblk;
    const int64_t beg = blockIdx.x * per, end = (beg + per < n2) ? beg + per : n2;
    const double2 *w2 = reinterpret_cast<const double2 *>(w + c * lvs);
    const double2 *m2 = reinterpret_cast<const double2 *>(bm1);
    const double2 *v2[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        const int jj = (j < kc) ? j : 0;
        v2[j] = reinterpret_cast<const double2 *>(V + (int64_t)(j0 + jj) * vstride + c * lvs);
    }
    double acc[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) acc[j] = 0.0;
    for (int64_t i = beg + threadIdx.x; i < end; i += NT) {
        const double2 wv = w2[i], mv = m2[i];
        const double x0 = wv.x * mv.x, x1 = wv.y * mv.y;
        double2 vv[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) vv[j] = v2[j][i];
#pragma unroll
        for (int j = 0; j < KB; ++j) acc[j] += vv[j].x * x0 + vv[j].y * x1;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        const double s = wave_sum(acc[j]);
        if (lane == 0) sm[wid * KB + j] = s;
    }
    __syncthreads();
    if (threadIdx.x < kc) {
        const int j = threadIdx.x;
        const double s = sm[j] + sm[KB + j] + sm[2 * KB + j] + sm[3 * KB + j];
        partial[(int64_t)(j0 + j) * nper + c * nblk + blockIdx.x] = s;
    }
}

// CGS2, first subtraction and second projection in ONE sweep over the basis (k <= KMAX):
//   w <- w - V h   on the velocity (+ scalar) part of the main block,   partial[j] = V_j^T (bm1 o w_new).
// The second projection needs the finished w at a point and the k basis values at the same point — which the thread
// that has just computed w there still holds in registers.  One point per lane, the k values and the k running sums
// in registers (2 x KMAX doubles: one wave per SIMD, 64 independent loads in flight per wave), one block per CU.
// Saves one of the four reads of the basis that CGS2 otherwise makes.
template <int KMAX>
__global__ __launch_bounds__(NT) void k_block_axpy_dot(const double *__restrict__ V, int64_t vstride, int k,
                                                       const double *__restrict__ h, double *__restrict__ w,
                                                       const double *__restrict__ bm1, int64_t lvs, int64_t nv,
                                                       double *__restrict__ partial) {
    __shared__ double sh[KMAX];
    __shared__ double sm[4][KMAX];
    for (int j = threadIdx.x; j < KMAX; j += NT) sh[j] = j < k ? h[j] : 0.0;
    __syncthreads();
    double acc[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) acc[j] = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < nv; i += (int64_t)gridDim.x * NT) {
        double v[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) v[j] = j < k ? V[(int64_t)j * vstride + i] : 0.0;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) s += sh[j] * v[j];
        const double wn = w[i] - s;
        w[i] = wn;
        const double x = wn * bm1[i % lvs];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) acc[j] += v[j] * x;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        const double t = wave_sum(acc[j]);
        if (lane == 0) sm[wid][j] = t;
    }
    __syncthreads();
    if (threadIdx.x < k) {
        const int j = threadIdx.x;
        partial[(int64_t)j * gridDim.x + blockIdx.x] = (sm[0][j] + sm[1][j]) + (sm[2][j] + sm[3][j]);
    }
}

// The same sweep for 64 < k <= 128 (round 4; BASELINE configs 4 and 5 name m = 128 and 256): k values and k running sums no longer fit
// one lane's registers, so the SECOND half of the basis values of a point is parked in LDS (one column of 64 doubles per thread: 128 KB of
// dynamic LDS, one block per CU as before) between the subtraction, which needs all k of them, and the projection, which needs them again
// together with the finished w: the basis is still read ONCE per pass.  Before, the 64 vectors in front of the fused tile cost a separate
// subtraction and a separate projection (k = 128: 448 vector reads per CGS2 instead of 384).
template <int KH>
__global__ __launch_bounds__(NT) void k_block_axpy_dot2(const double *__restrict__ V, int64_t vstride, int k,
                                                        const double *__restrict__ h, double *__restrict__ w,
                                                        const double *__restrict__ bm1, int64_t lvs, int64_t nv,
                                                        double *__restrict__ partial) {
    extern __shared__ double dyn2[];
    double *sh = dyn2;                    // 2 KH coefficients (zero beyond k)
    double *stash = dyn2 + 2 * KH;        // [KH][NT]: the second half of a point's basis values, column of this thread
    __shared__ double sm[4][2 * KH];
    for (int j = threadIdx.x; j < 2 * KH; j += NT) sh[j] = j < k ? h[j] : 0.0;
    __syncthreads();
    double acc0[KH], acc1[KH];
#pragma unroll
    for (int j = 0; j < KH; ++j) acc0[j] = 0.0, acc1[j] = 0.0;
    double *mine = stash + threadIdx.x;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < nv; i += (int64_t)gridDim.x * NT) {
        double v[KH];
#pragma unroll
        for (int j = 0; j < KH; ++j) v[j] = KH + j < k ? V[(int64_t)(KH + j) * vstride + i] : 0.0;
        double s1 = 0.0;
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            s1 += sh[KH + j] * v[j];
            mine[j * NT] = v[j];
        }
#pragma unroll
        for (int j = 0; j < KH; ++j) v[j] = V[(int64_t)j * vstride + i];
        double s0 = 0.0;
#pragma unroll
        for (int j = 0; j < KH; ++j) s0 += sh[j] * v[j];
        const double wn = w[i] - (s0 + s1);
        w[i] = wn;
        const double x = wn * bm1[i % lvs];
#pragma unroll
        for (int j = 0; j < KH; ++j) {
            acc0[j] += v[j] * x;
            acc1[j] += mine[j * NT] * x;
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < KH; ++j) {
        const double t0 = wave_sum(acc0[j]), t1 = wave_sum(acc1[j]);
        if (lane == 0) sm[wid][j] = t0, sm[wid][KH + j] = t1;
    }
    __syncthreads();
    if (threadIdx.x < k) {
        const int j = threadIdx.x;
        partial[(int64_t)j * gridDim.x + blockIdx.x] = (sm[0][j] + sm[1][j]) + (sm[2][j] + sm[3][j]);
    }
}

// w -= sum_j h_j V_j on the main block; history slots of w receive the same correction
// (reference axpby quirk) or the combination of the basis history (consistent mode).
// `hh` (may be null): a second coefficient set used for the entries at or beyond blk2 (the history blocks of a sweep
// that covers main + history in one range): CGS2 applies the second-pass coefficients to the main block and the SUM
// of both passes to the history, which no inner product ever reads -> the history is swept once, not twice.
__global__ __launch_bounds__(NT) void k_block_axpy(const double *V, int64_t vstride, int k, const double *h, double *w,
                                                   int64_t n2, int nrst, int64_t blk2, int consistent, double sign,
                                                   const double *hh) {
    extern __shared__ double sh0[];
    double *sh1 = sh0 + k;
    for (int j = threadIdx.x; j < k; j += NT) {
        sh0[j] = h[j];
        sh1[j] = hh ? hh[j] : h[j];
    }
    __syncthreads();
    double2 *w2 = reinterpret_cast<double2 *>(w);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) {
        const double *sh = (hh && i >= blk2) ? sh1 : sh0;
        double s0 = 0.0, s1 = 0.0;
        const double2 *v = reinterpret_cast<const double2 *>(V) + i;
        const int64_t vs2 = vstride >> 1;
        int j = 0;
        for (; j + 8 <= k; j += 8) {
            double2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = v[(int64_t)(j + u) * vs2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0 += sh[j + u] * t[u].x;
                s1 += sh[j + u] * t[u].y;
            }
        }
        for (; j < k; ++j) {
            const double2 t = v[(int64_t)j * vs2];
            s0 += sh[j] * t.x;
            s1 += sh[j] * t.y;
        }
        double2 wv = w2[i];
        wv.x += sign * s0;
        wv.y += sign * s1;
        w2[i] = wv;
        for (int r = 1; r <= nrst; ++r) {
            double c0 = s0, c1 = s1;
            if (consistent) {
                c0 = c1 = 0.0;
                for (int jj = 0; jj < k; ++jj) {
                    const double2 t = v[(int64_t)jj * vs2 + r * blk2];
                    c0 += sh[jj] * t.x;
                    c1 += sh[jj] * t.y;
                }
            }
            double2 rv = w2[i + r * blk2];
            rv.x += sign * c0;
            rv.y += sign * c1;
            w2[i + r * blk2] = rv;
        }
    }
}

// ---- block (multi-vector) orthogonalisation: S new vectors W = [w_0 .. w_{S-1}] (consecutive basis columns, wstride apart)
// against k basis vectors in ONE sweep over the basis per pass -- the basis is read once per S vectors instead of once per
// vector.  Partial sums: row (j * S + v) of the reduction workspace holds V_j^T (bm1 o w_v).
template <int KB, int S>
__global__ __launch_bounds__(NT) void k_block_dot_s(const double *V, int64_t vstride, int k, const double *W, int64_t wstride,
                                                    const double *bm1, int64_t lvs, int nblk, int nper, double *partial) {
    __shared__ double sm[4 * KB * S];
    const int c = blockIdx.y;
    const int j0 = blockIdx.z * KB;
    const int kc = (k - j0 < KB) ? (k - j0) : KB;
    const int64_t n2 = lvs >> 1;
    const int64_t per = (n2 + nblk - 1) / nblk;
    const int64_t beg = blockIdx.x * per, end = (beg + per < n2) ? beg + per : n2;
    const double2 *m2 = reinterpret_cast<const double2 *>(bm1);
    const double2 *v2[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
        const int jj = (j < kc) ? j : 0;
        v2[j] = reinterpret_cast<const double2 *>(V + (int64_t)(j0 + jj) * vstride + c * lvs);
    }
    double acc[KB][S];
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int v = 0; v < S; ++v) acc[j][v] = 0.0;
    for (int64_t i = beg + threadIdx.x; i < end; i += NT) {
        const double2 mv = m2[i];
        double x0[S], x1[S];
#pragma unroll
        for (int v = 0; v < S; ++v) {
            const double2 wv = reinterpret_cast<const double2 *>(W + (int64_t)v * wstride + c * lvs)[i];
            x0[v] = wv.x * mv.x;
            x1[v] = wv.y * mv.y;
        }
        double2 vv[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) vv[j] = v2[j][i];
#pragma unroll
        for (int j = 0; j < KB; ++j)
#pragma unroll
            for (int v = 0; v < S; ++v) acc[j][v] += vv[j].x * x0[v] + vv[j].y * x1[v];
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int v = 0; v < S; ++v) {
            const double t = wave_sum(acc[j][v]);
            if (lane == 0) sm[wid * KB * S + j * S + v] = t;
        }
    __syncthreads();
    if (threadIdx.x < kc * S) {
        const int q = threadIdx.x;   // j * S + v
        const double t = (sm[q] + sm[KB * S + q]) + (sm[2 * KB * S + q] + sm[3 * KB * S + q]);
        partial[((int64_t)j0 * S + q) * nper + c * nblk + blockIdx.x] = t;
    }
}

// w_v += sign * sum_j h[j * S + v] V_j for v < S over [0, n2) double2 entries; entries at or beyond blk2 (the history
// blocks) use the coefficient set hh instead (CGS2: second-pass coefficients on the main block, the sum of both passes on
// the history, which no inner product reads).  The k basis values of a point are loaded once for all S vectors.
template <int S>
__global__ __launch_bounds__(NT) void k_block_axpy_s(const double *V, int64_t vstride, int k, const double *h, const double *hh,
                                                     double *W, int64_t wstride, int64_t n2, int64_t blk2, double sign) {
    extern __shared__ double shs[];
    double *sh0 = shs, *sh1 = shs + (size_t)k * S;
    for (int j = threadIdx.x; j < k * S; j += NT) {
        sh0[j] = h[j];
        sh1[j] = hh ? hh[j] : h[j];
    }
    __syncthreads();
    const int64_t vs2 = vstride >> 1, ws2 = wstride >> 1;
    double2 *w2 = reinterpret_cast<double2 *>(W);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) {
        const double *sh = (hh && i >= blk2) ? sh1 : sh0;
        const double2 *v = reinterpret_cast<const double2 *>(V) + i;
        double s0[S], s1[S];
#pragma unroll
        for (int q = 0; q < S; ++q) s0[q] = s1[q] = 0.0;
        int j = 0;
        for (; j + 8 <= k; j += 8) {
            double2 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = v[(int64_t)(j + u) * vs2];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < S; ++q) {
                    s0[q] += sh[(j + u) * S + q] * t[u].x;
                    s1[q] += sh[(j + u) * S + q] * t[u].y;
                }
        }
        for (; j < k; ++j) {
            const double2 t = v[(int64_t)j * vs2];
#pragma unroll
            for (int q = 0; q < S; ++q) {
                s0[q] += sh[j * S + q] * t.x;
                s1[q] += sh[j * S + q] * t.y;
            }
        }
#pragma unroll
        for (int q = 0; q < S; ++q) {
            double2 wv = w2[i + q * ws2];
            wv.x += sign * s0[q];
            wv.y += sign * s1[q];
            w2[i + q * ws2] = wv;
        }
    }
}

// W <- W T with a small S x S matrix T (row-major, on the device) over [0, n2) double2 entries of every vector: the
// triangular solve of a Cholesky QR, applied to all fields and history blocks like `scal`
template <int S>
__global__ __launch_bounds__(NT) void k_block_rmul(double *W, int64_t wstride, const double *T, int64_t n2) {
    double t[S][S];
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int b = 0; b < S; ++b) t[a][b] = T[a * S + b];
    const int64_t ws2 = wstride >> 1;
    double2 *w2 = reinterpret_cast<double2 *>(W);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) {
        double2 x[S], y[S];
#pragma unroll
        for (int a = 0; a < S; ++a) x[a] = w2[i + a * ws2];
#pragma unroll
        for (int b = 0; b < S; ++b) {
            y[b].x = y[b].y = 0.0;
#pragma unroll
            for (int a = 0; a < S; ++a) {
                y[b].x += x[a].x * t[a][b];
                y[b].y += x[a].y * t[a][b];
            }
        }
#pragma unroll
        for (int b = 0; b < S; ++b) w2[i + b * ws2] = y[b];
    }
}

__global__ void k_vadd(double *a, const double *b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

inline int grid_for(int64_t n2) {
    int64_t g = (n2 + NT - 1) / NT;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

inline int dot_nblk(const nlg_vec *v) {
    int64_t nb = v->mesh->lvs / 4096;
    int cap = kMaxBlocksReduce / v->ncomp;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

int check_same(const nlg_vec *a, const nlg_vec *b, const char *who) {
    NLG_CHECK(a && b, "%s: NULL vector", who);
    NLG_CHECK(a->mesh == b->mesh && a->nscal == b->nscal && a->lorder == b->lorder,
              "%s: vectors live on different meshes / layouts (reference: type_error, real_vectors.f90:202-204)", who);
    return 0;
}

}  // namespace

namespace nlg {

int dev_dot(const nlg_vec *a, const nlg_vec *b, int slot) {
    nlg_ctx *ctx = a->mesh->ctx;
    const int nblk = dot_nblk(a);
    NLG_LAUNCH(k_dot_partial, dim3(nblk, a->ncomp), dim3(NT), 0, ctx->stream, a->d, b->d, a->mesh->d_bm1,
                       a->mesh->lvs, nblk, ctx->d_partial);
    NLG_LAUNCH(k_reduce_rows, dim3(1), dim3(NT), 0, ctx->stream, ctx->d_partial, nblk * a->ncomp,
                       ctx->d_scalars + slot, 0, (double *)nullptr);
    NLG_HIP(hipGetLastError());
    NLG_TRY(allreduce_sum(ctx, ctx->d_scalars + slot, 1));
    return 0;
}

}  // namespace nlg

static int g_axpby_consistent = 1;   // default reproduces the reference's published eigenvalue, see include/neklab_gpu.h

extern "C" {

int nlg_set_axpby_rst_consistent(int flag) {
    g_axpby_consistent = flag ? 1 : 0;
    return 0;
}

// ---- handle lifetimes for by-value host languages (the Fortran shim) ---------------------------------------------------
// The reference's vectors are plain arrays: intrinsic assignment, sourced allocation, array constructors and reallocation on
// assignment copy them bit by bit (neklab_analysis.f90:250 `allocate(Lu(r), source=..)`).  A shim object holding a handle is
// then duplicated behind its type's back, and its finaliser may run on the ORIGINAL while a copy is still going to be used
// (`X = [X, v]`: the temporary holds the copies, the old X is finalised, the new X receives the bits).  So the owner's
// finaliser does not free: it RELEASES the handle (state "released", buffer kept); a copy that turns up later ADOPTS it --
// released: the copy becomes the owner, nothing is copied or leaked; still owned elsewhere: the caller clones.  Every handle
// carries a generation number, so that a copy whose handle has meanwhile been freed and whose address has been reused is
// recognised instead of reading someone else's data.  Released buffers are freed oldest first once they hold more than
// g_pool_limit bytes (nlg_vec_pool_limit) or when an allocation fails, and all of them by nlg_vec_pool_trim.
namespace {
struct VecInfo {
    uint64_t gen = 0;
    bool released = false;      // the owner's finaliser ran: in the pool, evictable, any copy may adopt it
    bool pinned = false;        // released, then READ by a copy that cannot take ownership (intent(in) use): out of the pool, still adoptable
    uint64_t released_at = 0;
};
std::map<const nlg_vec *, VecInfo> g_vecs;      // every live or released handle
uint64_t g_next_gen = 1, g_release_tick = 0;
int64_t g_pool_bytes = 0, g_pool_limit = (int64_t)32 << 30;

void pool_free_oldest() {
    const nlg_vec *old = nullptr;
    uint64_t at = ~0ull;
    for (auto &kv : g_vecs)
        if (kv.second.released && kv.second.released_at < at) at = kv.second.released_at, old = kv.first;
    if (!old) return;
    nlg_vec *v = const_cast<nlg_vec *>(old);
    g_pool_bytes -= (int64_t)sizeof(double) * v->total_len;
    g_vecs.erase(old);
    if (v->owns && v->d) hipFree(v->d);
    delete v;
}
}  // namespace

int nlg_vec_create(nlg_mesh *mesh, int nscal, int lorder, nlg_vec **out) {
    NLG_CHECK(mesh && out, "nlg_vec_create: NULL argument");
    NLG_CHECK(nscal >= 0 && nscal <= 8, "nlg_vec_create: nscal %d out of range", nscal);
    NLG_CHECK(lorder >= 1 && lorder <= 4, "nlg_vec_create: lorder %d out of range", lorder);
    nlg_vec *v = new nlg_vec();
    v->mesh = mesh;
    v->nscal = nscal;
    v->lorder = lorder;
    v->ncomp = mesh->dim + nscal;
    v->main_len = (int64_t)v->ncomp * mesh->lvs + mesh->lps;
    v->total_len = v->main_len * lorder;
    NLG_HIP(hipSetDevice(mesh->ctx->device));
    hipError_t e = hipMalloc(&v->d, sizeof(double) * (size_t)v->total_len);
    while (e != hipSuccess && g_pool_bytes > 0) {   // give released buffers back before failing
        (void)hipGetLastError();
        pool_free_oldest();
        e = hipMalloc(&v->d, sizeof(double) * (size_t)v->total_len);
    }
    if (e != hipSuccess) {
        delete v;
        set_error("nlg_vec_create: hipMalloc(%zu bytes) failed: %s", sizeof(double) * (size_t)v->total_len,
                  hipGetErrorString(e));
        return 1;
    }
    NLG_HIP(hipMemsetAsync(v->d, 0, sizeof(double) * (size_t)v->total_len, mesh->ctx->stream));
    g_vecs[v].gen = g_next_gen++;
    *out = v;
    return 0;
}

int nlg_vec_destroy(nlg_vec *v) {
    if (!v) return 0;
    auto it = g_vecs.find(v);
    if (it != g_vecs.end()) {
        if (it->second.released) g_pool_bytes -= (int64_t)sizeof(double) * v->total_len;
        g_vecs.erase(it);
    }
    if (v->owns && v->d) hipFree(v->d);   // hipFree synchronises; never touch the parent mesh here
    delete v;
    return 0;
}

int nlg_vec_generation(const nlg_vec *v, int64_t *gen) {
    NLG_CHECK(v && gen, "nlg_vec_generation: NULL argument");
    auto it = g_vecs.find(v);
    NLG_CHECK(it != g_vecs.end(), "nlg_vec_generation: unknown handle (freed, or not created by nlg_vec_create)");
    *gen = (int64_t)it->second.gen;
    return 0;
}

int nlg_vec_release(nlg_vec *v) {
    if (!v) return 0;
    auto it = g_vecs.find(v);
    NLG_CHECK(it != g_vecs.end(), "nlg_vec_release: unknown handle");
    if (it->second.released) return 0;
    it->second.released = true;
    it->second.released_at = ++g_release_tick;
    g_pool_bytes += (int64_t)sizeof(double) * v->total_len;
    while (g_pool_bytes > g_pool_limit) pool_free_oldest();
    return 0;
}

int nlg_vec_adopt(nlg_vec *v, int64_t gen, int *status) {
    NLG_CHECK(v && status, "nlg_vec_adopt: NULL argument");
    auto it = g_vecs.find(v);
    NLG_CHECK(it != g_vecs.end() && (int64_t)it->second.gen == gen,
              "nlg_vec_adopt: the handle of this copy has been freed (a bitwise copy of a vector was used after more than "
              "nlg_vec_pool_limit bytes of released vectors piled up, or after nlg_vec_pool_trim)");
    if (it->second.released || it->second.pinned) {
        if (it->second.released) g_pool_bytes -= (int64_t)sizeof(double) * v->total_len;
        it->second.released = it->second.pinned = false;
        *status = 1;   // the caller owns the handle now
    } else {
        *status = 0;   // owned by a live object: the caller clones
    }
    return 0;
}

int nlg_vec_pin(nlg_vec *v, int64_t gen) {
    NLG_CHECK(v, "nlg_vec_pin: NULL argument");
    auto it = g_vecs.find(v);
    NLG_CHECK(it != g_vecs.end() && (int64_t)it->second.gen == gen,
              "nlg_vec_pin: the handle of this copy has been freed (a bitwise copy of a vector was used after more than "
              "nlg_vec_pool_limit bytes of released vectors piled up, or after nlg_vec_pool_trim)");
    if (it->second.released) {   // read through a released handle: keep it out of the eviction pool from now on
        it->second.released = false;
        it->second.pinned = true;
        g_pool_bytes -= (int64_t)sizeof(double) * v->total_len;
    }
    return 0;
}

int nlg_vec_unpin(nlg_vec *v, int64_t gen) {
    if (!v) return 0;
    auto it = g_vecs.find(v);
    if (it == g_vecs.end() || (int64_t)it->second.gen != gen || !it->second.pinned) return 0;   // not this copy's business
    it->second.pinned = false;
    it->second.released = true;
    it->second.released_at = ++g_release_tick;
    g_pool_bytes += (int64_t)sizeof(double) * v->total_len;
    while (g_pool_bytes > g_pool_limit) pool_free_oldest();
    return 0;
}

int64_t nlg_vec_size_checked(const nlg_vec *v, int64_t gen) {
    auto it = g_vecs.find(v);
    if (!v || it == g_vecs.end() || (int64_t)it->second.gen != gen) return -1;
    return (int64_t)v->ncomp * v->mesh->lvn + v->mesh->lpn;
}

int nlg_vec_has_rst_checked(const nlg_vec *v, int64_t gen) {
    auto it = g_vecs.find(v);
    if (!v || it == g_vecs.end() || (int64_t)it->second.gen != gen) return -1;
    return v->nrst > 0 ? 1 : 0;
}

int nlg_vec_pool_limit(int64_t bytes) {
    NLG_CHECK(bytes >= 0, "nlg_vec_pool_limit: negative limit");
    g_pool_limit = bytes;
    while (g_pool_bytes > g_pool_limit) pool_free_oldest();
    return 0;
}

int nlg_vec_pool_trim(int64_t *freed_bytes) {
    const int64_t before = g_pool_bytes;
    while (g_pool_bytes > 0) pool_free_oldest();
    if (freed_bytes) *freed_bytes = before;
    return 0;
}

int nlg_vec_copy(nlg_vec *dst, const nlg_vec *src) {
    NLG_TRY(check_same(dst, src, "nlg_vec_copy"));
    if (dst->d != src->d)
        NLG_HIP(hipMemcpyAsync(dst->d, src->d, sizeof(double) * (size_t)src->total_len, hipMemcpyDeviceToDevice,
                               src->mesh->ctx->stream));
    dst->nrst = src->nrst;
    return 0;
}

int nlg_vec_clone(const nlg_vec *src, nlg_vec **out) {
    NLG_CHECK(src && out, "nlg_vec_clone: NULL argument");
    NLG_TRY(nlg_vec_create(src->mesh, src->nscal, src->lorder, out));
    return nlg_vec_copy(*out, src);
}

int nlg_vec_zero(nlg_vec *self) {
    NLG_CHECK(self, "nlg_vec_zero: NULL vector");
    NLG_HIP(hipMemsetAsync(self->d, 0, sizeof(double) * (size_t)self->total_len, self->mesh->ctx->stream));
    self->nrst = 0;
    return 0;
}

int nlg_vec_scal(nlg_vec *self, double alpha) {
    NLG_CHECK(self, "nlg_vec_scal: NULL vector");
    const int64_t n2 = self->main_len * (1 + self->nrst) / 2;
    NLG_LAUNCH(k_scal, dim3(grid_for(n2)), dim3(NT), 0, self->mesh->ctx->stream,
                       reinterpret_cast<double2 *>(self->d), alpha, n2);
    NLG_HIP(hipGetLastError());
    return 0;
}

int nlg_vec_axpby(double alpha, const nlg_vec *vec, double beta, nlg_vec *self) {
    NLG_TRY(check_same(self, vec, "nlg_vec_axpby"));
    const int64_t n2 = self->main_len / 2;
    NLG_LAUNCH(k_axpby, dim3(grid_for(n2)), dim3(NT), 0, self->mesh->ctx->stream,
                       reinterpret_cast<double2 *>(self->d), reinterpret_cast<const double2 *>(vec->d), alpha, beta, n2,
                       self->nrst, n2, g_axpby_consistent);
    NLG_HIP(hipGetLastError());
    return 0;
}

int nlg_vec_dot(const nlg_vec *self, const nlg_vec *vec, double *out) {
    NLG_TRY(check_same(self, vec, "nlg_vec_dot"));
    NLG_CHECK(out, "nlg_vec_dot: out is NULL");
    NLG_TRY(dev_dot(self, vec, 0));
    return scalars_to_host(self->mesh->ctx, 0, 1, out);
}

int nlg_vec_norm(const nlg_vec *self, double *out) {
    double d = 0.0;
    NLG_TRY(nlg_vec_dot(self, self, &d));
    *out = sqrt(d);
    return 0;
}

int nlg_vec_size(const nlg_vec *self, int64_t *out) {
    NLG_CHECK(self && out, "nlg_vec_size: NULL argument");
    *out = (int64_t)self->ncomp * self->mesh->lvn + self->mesh->lpn;
    return 0;
}

int nlg_vec_save_rst(nlg_vec *self, const nlg_vec *vec_rst, int irst) {
    NLG_TRY(check_same(self, vec_rst, "nlg_vec_save_rst"));
    // reference: "Cannot save rst fields <torder> for a simulation of temporal order <torder>"
    NLG_CHECK(irst >= 1 && irst < self->lorder, "nlg_vec_save_rst: cannot save rst fields %d for temporal order %d", irst,
              self->lorder);
    NLG_HIP(hipMemcpyAsync(self->d + (int64_t)irst * self->main_len, vec_rst->d, sizeof(double) * (size_t)self->main_len,
                           hipMemcpyDeviceToDevice, self->mesh->ctx->stream));
    if (irst > self->nrst) self->nrst = irst;
    return 0;
}

int nlg_vec_get_rst(const nlg_vec *self, nlg_vec *vec_rst, int irst) {
    NLG_TRY(check_same(self, vec_rst, "nlg_vec_get_rst"));
    NLG_CHECK(irst >= 1 && irst < self->lorder, "nlg_vec_get_rst: invalid input for irst: %d", irst);
    NLG_HIP(hipMemcpyAsync(vec_rst->d, self->d + (int64_t)irst * self->main_len, sizeof(double) * (size_t)self->main_len,
                           hipMemcpyDeviceToDevice, self->mesh->ctx->stream));
    return 0;
}

int nlg_vec_has_rst_fields(const nlg_vec *self, int *out) {
    NLG_CHECK(self && out, "nlg_vec_has_rst_fields: NULL argument");
    *out = self->nrst > 0;
    return 0;
}

int nlg_vec_clear_rst_fields(nlg_vec *self) {
    NLG_CHECK(self, "nlg_vec_clear_rst_fields: NULL vector");
    self->nrst = 0;
    return 0;
}

int nlg_vec_nrst(const nlg_vec *self, int *out) {
    NLG_CHECK(self && out, "nlg_vec_nrst: NULL argument");
    *out = self->nrst;
    return 0;
}

static int field_ptr(const nlg_vec *v, int field, int irst, double **p, int64_t *len) {
    NLG_CHECK(v, "field access: NULL vector");
    NLG_CHECK(irst >= 0 && irst < v->lorder, "field access: irst %d out of range", irst);
    const nlg_mesh *m = v->mesh;
    if (field >= NLG_VX && field <= NLG_VZ) {
        NLG_CHECK(field < m->dim, "field access: component %d on a %d-D mesh", field, m->dim);
        *p = v->vel(field, irst);
        *len = m->lvn;
    } else if (field == NLG_PR) {
        *p = v->pr(irst);
        *len = m->lpn;
    } else {
        const int s = field - NLG_THETA;
        NLG_CHECK(s >= 0 && s < v->nscal, "field access: scalar %d not active (nscal=%d)", s, v->nscal);
        *p = v->theta(s, irst);
        *len = m->lvn;
    }
    return 0;
}

int nlg_vec_set_field(nlg_vec *self, int field, int irst, const double *host, int64_t count) {
    double *p;
    int64_t len;
    NLG_TRY(field_ptr(self, field, irst, &p, &len));
    NLG_CHECK(host && count == len, "nlg_vec_set_field: count %lld != field length %lld", (long long)count, (long long)len);
    NLG_HIP(hipMemcpyAsync(p, host, sizeof(double) * (size_t)len, hipMemcpyHostToDevice, self->mesh->ctx->stream));
    NLG_HIP(hipStreamSynchronize(self->mesh->ctx->stream));
    return 0;
}

int nlg_vec_get_field(const nlg_vec *self, int field, int irst, double *host, int64_t count) {
    double *p;
    int64_t len;
    NLG_TRY(field_ptr(self, field, irst, &p, &len));
    NLG_CHECK(host && count == len, "nlg_vec_get_field: count %lld != field length %lld", (long long)count, (long long)len);
    NLG_HIP(hipMemcpyAsync(host, p, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, self->mesh->ctx->stream));
    NLG_HIP(hipStreamSynchronize(self->mesh->ctx->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Krylov basis
// ------------------------------------------------------------------------------------------------
int nlg_basis_create(nlg_mesh *mesh, int nscal, int lorder, int nvec, nlg_basis **out) {
    NLG_CHECK(mesh && out, "nlg_basis_create: NULL argument");
    NLG_CHECK(nvec >= 1 && nvec <= 4096, "nlg_basis_create: nvec %d out of range", nvec);
    nlg_basis *b = new nlg_basis();
    b->mesh = mesh;
    b->nvec = nvec;
    b->nscal = nscal;
    b->lorder = lorder;
    const int ncomp = mesh->dim + nscal;
    const int64_t main_len = (int64_t)ncomp * mesh->lvs + mesh->lps;
    b->stride = main_len * lorder;
    NLG_HIP(hipSetDevice(mesh->ctx->device));
    const size_t bytes = sizeof(double) * (size_t)b->stride * nvec;
    hipError_t e = hipMalloc(&b->d, bytes);
    if (e != hipSuccess) {
        delete b;
        set_error("nlg_basis_create: hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return 1;
    }
    NLG_HIP(hipMemsetAsync(b->d, 0, bytes, mesh->ctx->stream));
    NLG_HIP(hipMalloc(&b->d_h, sizeof(double) * (size_t)(3 * nvec + 16)));
    NLG_HIP(hipMemsetAsync(b->d_h, 0, sizeof(double) * (size_t)(3 * nvec + 16), mesh->ctx->stream));
    NLG_TRY(reduce_ws_reserve(mesh->ctx, nvec + 8));
    b->views.resize(nvec);
    for (int i = 0; i < nvec; ++i) {
        nlg_vec *v = new nlg_vec();
        v->mesh = mesh;
        v->nscal = nscal;
        v->lorder = lorder;
        v->ncomp = ncomp;
        v->main_len = main_len;
        v->total_len = b->stride;
        v->d = b->d + (int64_t)i * b->stride;
        v->owns = false;
        b->views[i] = v;
    }
    *out = b;
    return 0;
}

int nlg_basis_destroy(nlg_basis *b) {
    if (!b) return 0;
    for (auto *v : b->views) delete v;
    if (b->d) hipFree(b->d);
    if (b->d_h) hipFree(b->d_h);
    if (b->d_hb) hipFree(b->d_hb);
    delete b;
    return 0;
}

int nlg_basis_vec(nlg_basis *b, int i, nlg_vec **out) {
    NLG_CHECK(b && out, "nlg_basis_vec: NULL argument");
    NLG_CHECK(i >= 0 && i < b->nvec, "nlg_basis_vec: index %d out of range [0,%d)", i, b->nvec);
    *out = b->views[i];
    return 0;
}

}  // extern "C"

namespace nlg {

// device-resident block projection: d_out[0:k] = V^T B w (allreduced); optionally d_acc += d_out
int basis_block_dot_dev(const nlg_basis *b, int k, const nlg_vec *w, double *d_out, double *d_acc) {
    nlg_ctx *ctx = b->mesh->ctx;
    const nlg_vec *v0 = b->views[0];
    constexpr int KB = 8;
    const int nblk = dot_nblk(v0);
    const int nper = nblk * v0->ncomp;
    NLG_TRY(reduce_ws_reserve(ctx, k));
    ProfScope ps(ctx, P_BLOCKDOT);
    NLG_LAUNCH(k_block_dot<KB>, dim3(nblk, v0->ncomp, (k + KB - 1) / KB), dim3(NT), 0, ctx->stream, b->d,
                       b->stride, k, w->d, b->mesh->d_bm1, b->mesh->lvs, nblk, nper, ctx->d_partial);
    if (ctx->distributed()) {
        NLG_LAUNCH(k_reduce_rows, dim3(k), dim3(NT), 0, ctx->stream, ctx->d_partial, nper, d_out, 0,
                           (double *)nullptr);
        NLG_TRY(allreduce_sum(ctx, d_out, k));
        if (d_acc) NLG_LAUNCH(k_vadd, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, d_acc, d_out, k);
    } else {
        ++g_collectives;   // (counted on one rank too, nlg_counters)
        NLG_LAUNCH(k_reduce_rows, dim3(k), dim3(NT), 0, ctx->stream, ctx->d_partial, nper, d_out,
                           d_acc ? 1 : 0, d_acc);
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

int basis_block_axpy_dev(const nlg_basis *b, int k, const double *d_h, nlg_vec *w, double sign, const double *d_hh,
                         bool main_only) {
    nlg_ctx *ctx = b->mesh->ctx;
    const int64_t n2 = w->main_len / 2;
    ProfScope ps(ctx, P_BLOCKAXPY);
    if (g_axpby_consistent) {
        // consistent history: main block and the nrst valid history blocks are one contiguous range in which every
        // entry receives the same linear combination -> one sweep through the unrolled path
        const int64_t n2all = main_only ? n2 : n2 * (1 + w->nrst);
        NLG_LAUNCH(k_block_axpy, dim3(grid_for(n2all)), dim3(NT), sizeof(double) * 2 * k, ctx->stream, b->d, b->stride,
                           k, d_h, w->d, n2all, 0, n2, 0, sign, d_hh);
    } else {
        NLG_LAUNCH(k_block_axpy, dim3(grid_for(n2)), dim3(NT), sizeof(double) * 2 * k, ctx->stream, b->d, b->stride, k,
                           d_h, w->d, n2, w->nrst, n2, 0, sign, (const double *)nullptr);
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

// CGS2 + norm + scale, everything stream-ordered on device. Results: d_h[0:k] coefficients,
// d_h[k] = ||w||^2 before normalisation.
int basis_cgs2_dev(nlg_basis *b, int k, nlg_vec *w) {
    nlg_ctx *ctx = b->mesh->ctx;
    double *h = b->d_h, *h2 = b->d_h + b->nvec + 8;
    if (k > 0) {
        NLG_TRY(basis_block_dot_dev(b, k, w, h, nullptr));
        if (g_axpby_consistent) {
            // first pass on the main block only; the second pass applies h2 to the main block and h + h2 (accumulated
            // in h by the second block_dot) to the history blocks: the result is the one of two full sweeps up to
            // rounding, at two thirds of the traffic
            // (measured, CGS2 + norm + scale at 10^4 elements: k = 64: 6.29 -> 4.94 ms, 32: 3.28 -> 2.79, 16: 1.78 -> 1.88,
            // 8: 1.04 -> 1.46 — the fully unrolled KMAX = 64 kernel does all 128 FMAs whatever k — hence the lower bound)
            static const int fuse_max = getenv("NLG_CGS2_FUSE_MAX") ? atoi(getenv("NLG_CGS2_FUSE_MAX")) : 128;
            if (k >= 24 && fuse_max >= 24) {
                // first subtraction and second projection in one sweep over the LAST kf <= 64 basis vectors
                // (k_block_axpy_dot); the k0 = k - kf vectors before them are subtracted first and projected after
                const nlg_vec *v0 = b->views[0];
                // k <= 64: one register tile (k_block_axpy_dot<64>); up to 128: two tiles, the second parked in LDS (k_block_axpy_dot2<64>)
                const int kf = std::min(std::min(k, fuse_max), 128) > 64 ? std::min(std::min(k, fuse_max), 128) : std::min(std::min(k, fuse_max), 64);
                const int k0 = k - kf;
                const int64_t nv = (int64_t)v0->ncomp * b->mesh->lvs;   // velocity (+ scalar) part; the pressure follows
                const int G = 256;                                     // one block per CU (one wave per SIMD)
                NLG_TRY(reduce_ws_reserve(ctx, k));
                if (k0 > 0) NLG_TRY(basis_block_axpy_dev(b, k0, h, w, -1.0, nullptr, true));
                const double *Vf = b->d + (int64_t)k0 * b->stride;
                {
                    ProfScope ps(ctx, P_AXPYDOT);
                    if (kf > 64) {
                        constexpr size_t lds2 = sizeof(double) * (2 * 64 + 64 * NT);
                        static bool attr_set = false;
                        if (!attr_set) {
                            NLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_block_axpy_dot2<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
                            attr_set = true;
                        }
                        NLG_LAUNCH(k_block_axpy_dot2<64>, dim3(G), dim3(NT), lds2, ctx->stream, Vf, b->stride, kf,
                                           (const double *)(h + k0), w->d, (const double *)b->mesh->d_bm1, b->mesh->lvs, nv, ctx->d_partial);
                    } else {
                        NLG_LAUNCH(k_block_axpy_dot<64>, dim3(G), dim3(NT), 0, ctx->stream, Vf, b->stride, kf,
                                           (const double *)(h + k0), w->d, (const double *)b->mesh->d_bm1, b->mesh->lvs, nv, ctx->d_partial);
                    }
                }
                {
                    ProfScope ps(ctx, P_BLOCKAXPY);
                    const int64_t np2 = (w->main_len - nv) / 2;        // pressure part of the main block: subtraction only
                    if (np2 > 0)
                        NLG_LAUNCH(k_block_axpy, dim3(grid_for(np2)), dim3(NT), sizeof(double) * 2 * kf, ctx->stream,
                                           Vf + nv, b->stride, kf, (const double *)(h + k0), w->d + nv, np2, 0, np2, 0, -1.0,
                                           (const double *)nullptr);
                }
                if (ctx->distributed()) {
                    NLG_LAUNCH(k_reduce_rows, dim3(kf), dim3(NT), 0, ctx->stream, ctx->d_partial, G, h2 + k0, 0, (double *)nullptr);
                    NLG_TRY(allreduce_sum(ctx, h2 + k0, kf));
                    NLG_LAUNCH(k_vadd, dim3((kf + 255) / 256), dim3(256), 0, ctx->stream, h + k0, h2 + k0, kf);
                } else {
                    ++g_collectives;
                    NLG_LAUNCH(k_reduce_rows, dim3(kf), dim3(NT), 0, ctx->stream, ctx->d_partial, G, h2 + k0, 1, h + k0);
                }
                if (k0 > 0) NLG_TRY(basis_block_dot_dev(b, k0, w, h2, h));
            } else {
                NLG_TRY(basis_block_axpy_dev(b, k, h, w, -1.0, nullptr, true));
                NLG_TRY(basis_block_dot_dev(b, k, w, h2, h));
            }
            NLG_TRY(basis_block_axpy_dev(b, k, h2, w, -1.0, h, false));
        } else {
            NLG_TRY(basis_block_axpy_dev(b, k, h, w, -1.0));
            NLG_TRY(basis_block_dot_dev(b, k, w, h2, h));
            NLG_TRY(basis_block_axpy_dev(b, k, h2, w, -1.0));
        }
    }
    // norm
    const int nblk = dot_nblk(w);
    NLG_LAUNCH(k_dot_partial, dim3(nblk, w->ncomp), dim3(NT), 0, ctx->stream, w->d, w->d, b->mesh->d_bm1,
                       b->mesh->lvs, nblk, ctx->d_partial);
    NLG_LAUNCH(k_reduce_rows, dim3(1), dim3(NT), 0, ctx->stream, ctx->d_partial, nblk * w->ncomp, h + k, 0,
                       (double *)nullptr);
    NLG_TRY(allreduce_sum(ctx, h + k, 1));
    const int64_t n2 = w->main_len * (1 + w->nrst) / 2;
    NLG_LAUNCH(k_scal_dev, dim3(grid_for(n2)), dim3(NT), 0, ctx->stream, reinterpret_cast<double2 *>(w->d),
                       h + k, 0, n2);
    NLG_HIP(hipGetLastError());
    return 0;
}

}  // namespace nlg

extern "C" {

int nlg_basis_block_dot(const nlg_basis *b, int k, const nlg_vec *w, double *h) {
    NLG_CHECK(b && w && h, "nlg_basis_block_dot: NULL argument");
    NLG_CHECK(k >= 1 && k <= b->nvec, "nlg_basis_block_dot: k=%d out of range [1,%d]", k, b->nvec);
    NLG_TRY(check_same(b->views[0], w, "nlg_basis_block_dot"));
    NLG_TRY(nlg::basis_block_dot_dev(b, k, w, b->d_h, nullptr));
    NLG_HIP(hipMemcpyAsync(h, b->d_h, sizeof(double) * k, hipMemcpyDeviceToHost, b->mesh->ctx->stream));
    NLG_HIP(hipStreamSynchronize(b->mesh->ctx->stream));
    return 0;
}

int nlg_basis_block_axpy(const nlg_basis *b, int k, const double *h, nlg_vec *w) {
    NLG_CHECK(b && w && h, "nlg_basis_block_axpy: NULL argument");
    NLG_CHECK(k >= 1 && k <= b->nvec, "nlg_basis_block_axpy: k=%d out of range [1,%d]", k, b->nvec);
    NLG_TRY(check_same(b->views[0], w, "nlg_basis_block_axpy"));
    NLG_HIP(hipMemcpyAsync(b->d_h, h, sizeof(double) * k, hipMemcpyHostToDevice, b->mesh->ctx->stream));
    NLG_TRY(nlg::basis_block_axpy_dev(b, k, b->d_h, w, -1.0));
    NLG_HIP(hipStreamSynchronize(b->mesh->ctx->stream));
    return 0;
}

int nlg_basis_cgs2(const nlg_basis *bc, int k, nlg_vec *w, double *h, double *beta) {
    nlg_basis *b = const_cast<nlg_basis *>(bc);
    NLG_CHECK(b && w && h && beta, "nlg_basis_cgs2: NULL argument");
    NLG_CHECK(k >= 0 && k <= b->nvec, "nlg_basis_cgs2: k=%d out of range [0,%d]", k, b->nvec);
    NLG_TRY(check_same(b->views[0], w, "nlg_basis_cgs2"));
    NLG_TRY(nlg::basis_cgs2_dev(b, k, w));
    std::vector<double> tmp(k + 1);
    NLG_HIP(hipMemcpyAsync(tmp.data(), b->d_h, sizeof(double) * (k + 1), hipMemcpyDeviceToHost, b->mesh->ctx->stream));
    NLG_HIP(hipStreamSynchronize(b->mesh->ctx->stream));
    for (int j = 0; j < k; ++j) h[j] = tmp[j];
    *beta = sqrt(tmp[k]);
    return 0;
}

// Block classical Gram-Schmidt with re-orthogonalisation for the s consecutive columns k .. k+s-1 against the columns
// 0 .. k-1, followed by a Cholesky QR (twice) among the s columns themselves; see include/neklab_gpu.h.
int nlg_basis_block_cgs2(nlg_basis *b, int k, int s, double *coef) {
    NLG_CHECK(b && coef, "nlg_basis_block_cgs2: NULL argument");
    NLG_CHECK(s >= 1 && s <= 4, "nlg_basis_block_cgs2: block size %d unsupported (1..4)", s);
    NLG_CHECK(k >= 0 && k + s <= b->nvec, "nlg_basis_block_cgs2: columns %d..%d outside the basis (nvec=%d)", k, k + s - 1, b->nvec);
    NLG_CHECK(g_axpby_consistent, "nlg_basis_block_cgs2: the block path implements the consistent restart-history update only");
    nlg_ctx *ctx = b->mesh->ctx;
    hipStream_t st = ctx->stream;
    const int ld = k + s;
    if (s == 1) {   // the single-vector path, same output convention
        std::vector<double> h(std::max(k, 1));
        double beta = 0.0;
        NLG_TRY(nlg_basis_cgs2(b, k, b->views[k], h.data(), &beta));
        for (int j = 0; j < k; ++j) coef[j] = h[j];
        coef[k] = beta;
        return 0;
    }
    if (!b->d_hb) NLG_HIP(hipMalloc(&b->d_hb, sizeof(double) * ((size_t)2 * b->nvec * 4 + 64)));
    nlg_vec *w0 = b->views[k];
    int nrst = 0;
    for (int v = 0; v < s; ++v) nrst = std::max(nrst, b->views[k + v]->nrst);
    for (int v = 0; v < s; ++v) {
        // the sweeps below run over the main block and `nrst` history blocks of EVERY column: a column that carries fewer (a
        // freshly drawn one next to an x0 with a restart history) gets zeros there, not whatever its memory held
        nlg_vec *c = b->views[k + v];
        if (c->nrst < nrst)
            NLG_HIP(hipMemsetAsync(c->d + (int64_t)(1 + c->nrst) * c->main_len, 0, sizeof(double) * (size_t)((nrst - c->nrst) * c->main_len), st));
        c->nrst = nrst;
    }
    b->last_block_rank = s;
    double *W = w0->d;
    double *H1 = b->d_hb, *H2 = b->d_hb + (size_t)b->nvec * 4, *dG = b->d_hb + (size_t)2 * b->nvec * 4, *dT = dG + 16;
    const int nblk = dot_nblk(w0), nper = nblk * w0->ncomp;
    const int64_t n2 = w0->main_len / 2, n2all = n2 * (1 + nrst);
    constexpr int KB = 8;
    NLG_TRY(reduce_ws_reserve(ctx, std::max(k, s) * s));
    auto dots = [&](const double *V, int kk, double *out, double *acc) -> int {
        ProfScope ps(ctx, P_BLOCKDOT);
        const dim3 g(nblk, w0->ncomp, (kk + KB - 1) / KB);
#define BD(S_) NLG_LAUNCH((k_block_dot_s<KB, S_>), g, dim3(NT), 0, st, V, b->stride, kk, (const double *)W, b->stride, \
                                  (const double *)b->mesh->d_bm1, b->mesh->lvs, nblk, nper, ctx->d_partial)
        if (s == 2) BD(2);
        else if (s == 3) BD(3);
        else BD(4);
#undef BD
        if (ctx->distributed()) {
            NLG_LAUNCH(k_reduce_rows, dim3(kk * s), dim3(NT), 0, st, ctx->d_partial, nper, out, 0, (double *)nullptr);
            NLG_TRY(allreduce_sum(ctx, out, kk * s));
            if (acc) NLG_LAUNCH(k_vadd, dim3((kk * s + 255) / 256), dim3(256), 0, st, acc, out, kk * s);
        } else {
            ++g_collectives;
            NLG_LAUNCH(k_reduce_rows, dim3(kk * s), dim3(NT), 0, st, ctx->d_partial, nper, out, acc ? 1 : 0, acc);
        }
        return 0;
    };
    auto axpy = [&](const double *h, const double *hh, int64_t n, int64_t blk2) -> int {
        ProfScope ps(ctx, P_BLOCKAXPY);
        const size_t lds = sizeof(double) * 2 * (size_t)k * s;
#define BA(S_) NLG_LAUNCH((k_block_axpy_s<S_>), dim3(grid_for(n)), dim3(NT), lds, st, (const double *)b->d, b->stride, k, h, hh, W, \
                                  b->stride, n, blk2, -1.0)
        if (s == 2) BA(2);
        else if (s == 3) BA(3);
        else BA(4);
#undef BA
        return 0;
    };
    if (k > 0) {
        NLG_TRY(dots(b->d, k, H1, nullptr));
        NLG_TRY(axpy(H1, nullptr, n2, n2));          // first pass: main block only
        NLG_TRY(dots(b->d, k, H2, H1));              // H1 <- H1 + H2
        NLG_TRY(axpy(H2, H1, n2all, n2));            // second pass: H2 on the main block, the sum on the history blocks
    }
    // Cholesky QR among the s new columns, twice.  A (numerically) dependent column -- a block Krylov space that has reached an
    // invariant subspace, the "lucky breakdown" of converged eigenvalues, or a drawn column close to x0 -- is not an error: what is
    // left of it after the projections is below the rounding level of what it was, so it is DEFLATED: its coefficients on the
    // columns before it go into R as for any column, its diagonal entry of R is 0 and the column itself becomes the zero vector
    // (harmless in every later projection).  b->last_block_rank = number of columns kept; nlg_eigs stops expanding the space when
    // it falls below s, as it does at beta = 0 in the single-vector iteration.
    // Two tests, both relative to what the column was when the step that can lose it began: (1) against the BASIS -- the squared
    // norm a column brought to this call is what is left of it plus the squares of its coefficients on the (orthonormal) basis,
    // |w|^2 = |w - V h|^2 + |h|^2; CGS2 leaves a remainder of about eps |w| of a column that lies in span(V), so a remainder below
    // 1e-12 |w| (1e-24 on the squares) is rounding noise, whatever the other columns of the block look like; (2) among the columns
    // of the block -- the Cholesky pivot against the diagonal entry at entry to the CURRENT round (the second round sees
    // normalised columns: a threshold carried over from the first would deflate any healthy column that came in with a norm above 1e7).
    double R[4][4] = {};
    for (int a = 0; a < s; ++a) R[a][a] = 1.0;
    bool dep[4] = {false, false, false, false};
    double hsq[4] = {0.0, 0.0, 0.0, 0.0};   // |h_v|^2: squared norm of the part of column v that the projections removed
    std::vector<double> hs((size_t)std::max(k, 1) * s);
    if (k > 0) {
        NLG_HIP(hipMemcpyAsync(hs.data(), H1, sizeof(double) * (size_t)k * s, hipMemcpyDeviceToHost, st));
        NLG_HIP(hipStreamSynchronize(st));
        for (int v = 0; v < s; ++v)
            for (int j = 0; j < k; ++j) hsq[v] += hs[(size_t)j * s + v] * hs[(size_t)j * s + v];
    }
    for (int round = 0; round < 2; ++round) {
        NLG_TRY(dots(W, s, dG, nullptr));
        double G[16];
        NLG_HIP(hipMemcpyAsync(G, dG, sizeof(double) * s * s, hipMemcpyDeviceToHost, st));
        NLG_HIP(hipStreamSynchronize(st));
        double L[4][4] = {};
        for (int i = 0; i < s; ++i) {
            NLG_CHECK(std::isfinite(G[i * s + i]), "nlg_basis_block_cgs2: column %d is not finite", k + i);
            const double gin = G[i * s + i];   // squared norm of column i at entry to this round
            if (round == 0 && !(gin > 1e-24 * (gin + hsq[i]))) dep[i] = true;   // nothing left of it beside span(V)
            for (int j = 0; j <= i; ++j) {
                double a = G[i * s + j];
                for (int q = 0; q < j; ++q) a -= L[i][q] * L[j][q];
                if (i == j) {
                    // the pivot is the squared norm of column i after the columns before it have been projected out: a block with
                    // condition number above ~1e7 loses it to rounding (CholQR works on the SQUARED condition number)
                    if (dep[i] || !(a > 1e-14 * gin) || !(gin > 0.0)) {
                        dep[i] = true;
                        L[i][i] = 0.0;
                    } else {
                        L[i][i] = std::sqrt(a);
                    }
                } else {
                    L[i][j] = dep[j] ? 0.0 : a / L[j][j];
                }
            }
        }
        // Rr = L^T (upper); T = Rr^-1 (upper, back substitution column by column); W <- W T; R <- Rr R.  A deflated column takes a
        // unit pivot in the inverse and a zero column in T afterwards: W T then holds the zero vector in its place.
        double T[4][4] = {};
        for (int c = 0; c < s; ++c) {
            T[c][c] = dep[c] ? 1.0 : 1.0 / L[c][c];
            for (int r = c - 1; r >= 0; --r) {
                double a = 0.0;
                for (int q = r + 1; q <= c; ++q) a += L[q][r] * T[q][c];
                T[r][c] = dep[r] ? -a : -a / L[r][r];
            }
        }
        for (int c = 0; c < s; ++c)
            if (dep[c])
                for (int r = 0; r < s; ++r) T[r][c] = 0.0;
        double Tf[16], Rn[4][4] = {};
        for (int a = 0; a < s; ++a)
            for (int c = 0; c < s; ++c) {
                Tf[a * s + c] = T[a][c];
                for (int q = 0; q < s; ++q) Rn[a][c] += L[q][a] * R[q][c];
            }
        memcpy(R, Rn, sizeof(R));
        NLG_HIP(hipMemcpyAsync(dT, Tf, sizeof(double) * s * s, hipMemcpyHostToDevice, st));
        {
            ProfScope ps(ctx, P_BLOCKAXPY);
#define BR(S_) NLG_LAUNCH((k_block_rmul<S_>), dim3(grid_for(n2all)), dim3(NT), 0, st, W, b->stride, (const double *)dT, n2all)
            if (s == 2) BR(2);
            else if (s == 3) BR(3);
            else BR(4);
#undef BR
        }
        NLG_HIP(hipStreamSynchronize(st));   // Tf lives on this stack frame
    }
    for (int v = 0; v < s; ++v)
        if (dep[v]) --b->last_block_rank;
    for (int v = 0; v < s; ++v) {
        for (int j = 0; j < k; ++j) coef[(size_t)v * ld + j] = hs[(size_t)j * s + v];
        for (int a = 0; a < s; ++a) coef[(size_t)v * ld + k + a] = R[a][v];
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

int nlg_basis_last_block_rank(const nlg_basis *b, int *rank) {
    NLG_CHECK(b && rank, "nlg_basis_last_block_rank: NULL argument");
    *rank = b->last_block_rank;
    return 0;
}

int nlg_basis_combine(const nlg_basis *b, int k, const double *c, nlg_vec *out) {
    NLG_CHECK(b && c && out, "nlg_basis_combine: NULL argument");
    NLG_CHECK(k >= 1 && k <= b->nvec, "nlg_basis_combine: k=%d out of range [1,%d]", k, b->nvec);
    NLG_TRY(check_same(b->views[0], out, "nlg_basis_combine"));
    NLG_TRY(nlg_vec_zero(out));
    if (g_axpby_consistent) {
        // consistent history: the combination carries the combined history slots of the basis vectors
        int nr = 0;
        for (int j = 0; j < k; ++j) nr = std::max(nr, b->views[j]->nrst);
        out->nrst = nr;
    }
    NLG_HIP(hipMemcpyAsync(b->d_h, c, sizeof(double) * k, hipMemcpyHostToDevice, b->mesh->ctx->stream));
    NLG_TRY(nlg::basis_block_axpy_dev(b, k, b->d_h, out, +1.0));
    NLG_HIP(hipStreamSynchronize(b->mesh->ctx->stream));
    return 0;
}

}  // extern "C"
