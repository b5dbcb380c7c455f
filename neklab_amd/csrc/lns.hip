// exptA_linop: exponential propagator of the linearised Navier-Stokes operator (gfx950).
//
// In-tree protocol followed (file:line under /root/reference):
//   exptA_matvec / exptA_rmatvec   src/linops/exponential_propagator.f90:15-60, :62-107
//   exptA_compute_rst / get_rst    src/linops/exponential_propagator.f90:109-142
//   init_exptA                     src/linops/exponential_propagator.f90:4-13
//   dt / nsteps rule               src/neklab_nek_setup.f90:195-200
// The time integrator behind `nek_advance` is Nek5000 code that is not in the reference tree; it is
// restated here from the published Pn-Pn-2 BDFk/EXTk splitting exactly as in oracle/lns.py.
//
// All solver scalars (alpha, beta, residual norms, the convergence flag) live in device memory; the
// host only reads the flag once per chunk of launched iterations, and every kernel of an iteration
// returns immediately once the flag is set, so the result is identical to stopping at convergence.
#include <cmath>
#include <functional>
#include <map>
#include <string>

#include "internal.h"

using namespace nlg;

namespace {

constexpr int NT = 256;
constexpr int NB = 512;   // fixed first-stage reduction grid

struct F3 {
    double *p[3];
};
struct CF3 {
    const double *p[3];
};
// Lane batching (block stepper): the per-lane work buffers of all lanes live in ONE slab at a constant stride `ld` doubles, so a
// kernel launched with gridDim.y = lanes reaches lane v's copy of every per-lane argument at `lane-0 pointer + v * ld`
// (blockIdx.y is wave-uniform: the offset is scalar arithmetic, no lane loop in the kernel body).  Single-vector launches pass
// gridDim.y = 1.  Arrays that belong to the mesh or to the base flow (weights, preconditioners, masks) are not offset.
__device__ __forceinline__ int64_t lane_lo(int64_t ld) { return (int64_t)blockIdx.y * ld; }
__device__ __forceinline__ F3 lane_f3(F3 a, int64_t lo) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (a.p[c]) a.p[c] += lo;
    return a;
}
__device__ __forceinline__ CF3 lane_f3(CF3 a, int64_t lo) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (a.p[c]) a.p[c] += lo;
    return a;
}

// solver scalar slots (doubles) inside a per-solver device block
enum { S_RZ = 0, S_PW = 1, S_RZN = 2, S_RN2 = 3, S_DONE = 4, S_ITERS = 5, S_ALPHA = 6, S_BETA = 7, S_T0 = 8, S_T1 = 9, S_T2 = 10,
       S_WMEAN = 11, S_ZMEAN = 12, S_RN20 = 13, S_AH = 16, S_N = 48 };
constexpr int kAlphaRing = 32;   // S_AH .. S_AH + 31: the step lengths of the last 32 iterations (iteration i at i mod 32), for the deferred solution update
constexpr double kFloor2 = 1e-28;   // stop when |r|^2 has dropped by 1e-28: further iterations only divide 0 by 0

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// two simultaneous block sums; results valid in thread 0
__device__ __forceinline__ void block_sum2(double &a, double &b, double *sm) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        sm[wid] = a;
        sm[4 + wid] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = sm[0] + sm[1] + sm[2] + sm[3];
        b = sm[4] + sm[5] + sm[6] + sm[7];
    }
}

// three simultaneous block sums; results valid in thread 0
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *sm) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        sm[wid] = a;
        sm[4 + wid] = b;
        sm[8 + wid] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = sm[0] + sm[1] + sm[2] + sm[3];
        b = sm[4] + sm[5] + sm[6] + sm[7];
        c = sm[8] + sm[9] + sm[10] + sm[11];
    }
}

// Projected PCG (P = I - 1 1^T / n on the mean-free subspace, see oracle/lns.py pcg_E): the means of
// w = A p and of z = M^-1 r ride along as extra partial sums of kernels that exist anyway.
// x = 0, r = b (in place), z = pc*r ; partial sums of (r,z)_ipw, (r,r)_nw and sum(z)
template <int NF>
__global__ __launch_bounds__(NT) void k_cg_init(int64_t n, F3 x, F3 r, F3 z, CF3 pc, const double *ipw,
                                                const double *nw, double *partial, int64_t ld, const unsigned char *__restrict__ mb, int defer_x) {
    __shared__ double sm[12];
    const int64_t lo = lane_lo(ld);
    x = lane_f3(x, lo), r = lane_f3(r, lo), z = lane_f3(z, lo), partial += lo;
    double a = 0.0, b = 0.0, c3 = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double wi = ipw ? ipw[i] : 1.0, wn = nw[i];
        // mb: the preconditioner of component c is pc.p[0] (1 / diag) where bit c of the point's mask byte is set and 0 elsewhere
        const double inv = mb ? pc.p[0][i] : 0.0;
        const unsigned mk = mb ? mb[i] : 0u;
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            const double rv = r.p[c][i];
            if (!defer_x) x.p[c][i] = 0.0;   // (deferred solution update: x is first WRITTEN by k_x_flush / never read before that)
            b += rv * rv * wn;
            if (mb || pc.p[c]) {      // pointwise (Jacobi) preconditioner; otherwise z comes from an operator
                const double zv = (mb ? (((mk >> c) & 1u) ? inv : 0.0) : pc.p[c][i]) * rv;
                z.p[c][i] = zv;
                a += rv * zv * wi;
                c3 += zv;
            }
        }
    }
    block_sum3(a, b, c3, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = a;
        partial[NB + blockIdx.x] = b;
        partial[2 * NB + blockIdx.x] = c3;
    }
}

template <int NF>
__global__ __launch_bounds__(NT) void k_cg_pw(const double *s, int64_t n, CF3 p, CF3 w, const double *ipw, double *partial, int64_t ld) {
    __shared__ double sm[8];
    const int64_t lo = lane_lo(ld);
    s += lo, p = lane_f3(p, lo), w = lane_f3(w, lo), partial += lo;
    if (s[S_DONE] != 0.0) return;
    double a = 0.0, b = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double wi = ipw ? ipw[i] : 1.0;
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            const double wv = w.p[c][i];
            a += p.p[c][i] * wv * wi;
            b += wv;
        }
    }
    block_sum2(a, b, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = a;
        partial[NB + blockIdx.x] = b;
    }
}

// x += alpha p (unless deferred) ; r -= alpha (w - wmean) ; z = pc r ; partial (r,z)_ipw, (r,r)_nw, sum(z)
// Two points per lane (16-byte accesses; every field length is a multiple of 32): eight streams per component are
// what this kernel is, so the width of an access is its efficiency.
template <int NF>
__global__ __launch_bounds__(NT) void k_cg_update(const double *s, int64_t n, F3 x, F3 r, F3 z, CF3 p, CF3 w, CF3 pc,
                                                  const double *ipw, const double *nw, double *partial, int64_t ld, int defer_x,
                                                  const unsigned char *__restrict__ mb) {
    __shared__ double sm[12];
    const int64_t lo = lane_lo(ld);
    s += lo, x = lane_f3(x, lo), r = lane_f3(r, lo), z = lane_f3(z, lo), p = lane_f3(p, lo), w = lane_f3(w, lo), partial += lo;
    if (s[S_DONE] != 0.0) return;
    const double alpha = s[S_ALPHA], wmean = s[S_WMEAN];
    const bool dox = defer_x == 0;   // deferred: x is assembled from the direction history afterwards (k_x_flush / k_add_hist), p is not read here
    double a = 0.0, b = 0.0, c3 = 0.0;
    const int64_t n2 = n >> 1;
    const double2 *ipw2 = reinterpret_cast<const double2 *>(ipw), *nw2 = reinterpret_cast<const double2 *>(nw);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n2; i += (int64_t)gridDim.x * NT) {
        const double2 wi = ipw ? ipw2[i] : make_double2(1.0, 1.0), wn = nw2[i];
        // mb: pc.p[0] = 1 / diag for all components, the masks are the bits of one byte per point (3 arrays -> 1 + 1/8)
        const double2 inv = mb ? reinterpret_cast<const double2 *>(pc.p[0])[i] : make_double2(0.0, 0.0);
        const unsigned mk = mb ? reinterpret_cast<const unsigned short *>(mb)[i] : 0u;
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            double2 *x2 = reinterpret_cast<double2 *>(x.p[c]), *r2 = reinterpret_cast<double2 *>(r.p[c]);
            const double2 wv = reinterpret_cast<const double2 *>(w.p[c])[i];
            double2 rv = r2[i];
            if (dox) {
                const double2 pv = reinterpret_cast<const double2 *>(p.p[c])[i];
                double2 xv = x2[i];
                xv.x += alpha * pv.x;
                xv.y += alpha * pv.y;
                x2[i] = xv;
            }
            rv.x -= alpha * (wv.x - wmean);
            rv.y -= alpha * (wv.y - wmean);
            double2 pcv = make_double2(1.0, 1.0);
            if (mb)
                pcv = make_double2(((mk >> c) & 1u) ? inv.x : 0.0, ((mk >> (8 + c)) & 1u) ? inv.y : 0.0);
            else if (pc.p[c])
                pcv = reinterpret_cast<const double2 *>(pc.p[c])[i];
            if (pcv.x == 0.0) rv.x = 0.0;   // Dirichlet dof (the Helmholtz preconditioner carries the mask): w is not masked
            if (pcv.y == 0.0) rv.y = 0.0;
            r2[i] = rv;
            b += rv.x * rv.x * wn.x + rv.y * rv.y * wn.y;
            if (mb || pc.p[c]) {
                double2 zv;
                zv.x = pcv.x * rv.x;
                zv.y = pcv.y * rv.y;
                reinterpret_cast<double2 *>(z.p[c])[i] = zv;
                a += rv.x * zv.x * wi.x + rv.y * zv.y * wi.y;
                c3 += zv.x + zv.y;
            }
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // odd length: the last point
        const int64_t i = n - 1;
        const double wi = ipw ? ipw[i] : 1.0, wn = nw[i];
        for (int c = 0; c < NF; ++c) {
            if (dox) x.p[c][i] += alpha * p.p[c][i];
            double rv = r.p[c][i] - alpha * (w.p[c][i] - wmean);
            const double pcv = mb ? (((mb[i] >> c) & 1u) ? pc.p[0][i] : 0.0) : (pc.p[c] ? pc.p[c][i] : 1.0);
            if (pcv == 0.0) rv = 0.0;
            r.p[c][i] = rv;
            b += rv * rv * wn;
            if (mb || pc.p[c]) {
                const double zv = pcv * rv;
                z.p[c][i] = zv;
                a += rv * zv * wi;
                c3 += zv;
            }
        }
    }
    block_sum3(a, b, c3, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = a;
        partial[NB + blockIdx.x] = b;
        partial[2 * NB + blockIdx.x] = c3;
    }
}

// single-reduction PCG: p <- u + beta p ; s <- w + beta s ; x += alpha p ; r -= alpha s ; u = pc r  (u lives in `z`); partial (r, u)_ipw, (r, r)_nw.
// The first update of a solve (ITERS == 1 after the logic) takes p = u, s = w whatever the buffers held.
template <int NF>
__global__ __launch_bounds__(NT) void k_cg_update_sr(const double *s, int64_t n, F3 x, F3 r, F3 z, F3 p, F3 sd, CF3 w, CF3 pc,
                                                     const double *ipw, const double *nw, double *partial, int64_t ld) {
    __shared__ double sm[12];
    const int64_t lo = lane_lo(ld);
    s += lo, x = lane_f3(x, lo), r = lane_f3(r, lo), z = lane_f3(z, lo), p = lane_f3(p, lo), sd = lane_f3(sd, lo), w = lane_f3(w, lo), partial += lo;
    if (s[S_DONE] != 0.0) return;
    const double alpha = s[S_ALPHA], beta = s[S_BETA];
    const bool first = s[S_ITERS] == 1.0;
    double a = 0.0, b = 0.0, c3 = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double wi = ipw ? ipw[i] : 1.0, wn = nw[i];
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            const double uv = z.p[c][i], wv = w.p[c][i];
            const double pv = first ? uv : uv + beta * p.p[c][i];
            const double sv = first ? wv : wv + beta * sd.p[c][i];
            p.p[c][i] = pv;
            sd.p[c][i] = sv;
            x.p[c][i] += alpha * pv;
            double rv = r.p[c][i] - alpha * sv;
            const double pcv = pc.p[c][i];
            if (pcv == 0.0) rv = 0.0;   // Dirichlet dof (the preconditioner carries the mask): w is not masked
            r.p[c][i] = rv;
            const double zv = pcv * rv;
            z.p[c][i] = zv;
            a += rv * zv * wi;
            b += rv * rv * wn;
        }
    }
    block_sum3(a, b, c3, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = a;
        partial[NB + blockIdx.x] = b;
    }
}

// partial sums of (r,z)_ipw and sum(z) when z was produced by a preconditioning operator
// `xc`/`npe` (xc may be null): the coarse-grid part of z, one value per element of npe points, kept separate so
// that the coarse branch can run concurrently with the element-wise solves: z_total = z + xc[i / npe]
// Deferred solution update of the velocity PCG.  x = sum_i alpha_i p_i never changes the iteration, so k_cg_update does not touch it:
// the fused direction update of the operator kernel stores direction i into slot i mod PH of a ring of PH direction buffers (the
// write it makes anyway, at another address), the scalar logic keeps alpha_i, and x is assembled afterwards in the order of the
// iteration -- the same additions on the same operands, hence the same bits -- by k_add_hist (the velocity update that consumes x)
// and, for the rare solve with more than PH iterations, by k_x_flush every PH iterations.  Saves the two x streams and the p
// stream of every iteration of k_cg_update (26 -> 17 streams at three components) for one sweep over the directions per solve.
struct PHist {
    const double *p0[3];   // slot 0, one pointer per component; slot q is q * stride doubles behind
    int64_t stride;
    int ph;
};
// x += sum over the PH iterations `last` - PH + 1 .. `last`; launched right after the k_cg_update of iteration `last` (before the scalar
// logic that counts it), so "not converged yet" = the solve did run them all
template <int NF>
__global__ __launch_bounds__(NT) void k_x_flush(const double *__restrict__ s, int64_t n, F3 x, PHist H, int last, int64_t ld) {
    const int64_t lo = lane_lo(ld);
    s += lo, x = lane_f3(x, lo);
    if (s[S_DONE] != 0.0) return;   // converged inside this window: k_add_hist takes what there is of it
    const int w0 = last - H.ph + 1;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            const double *__restrict__ pc = H.p0[c] + lo + i;
            double t = w0 > 0 ? x.p[c][i] : 0.0;   // the first window starts the sum (k_cg_init leaves x alone)
#pragma unroll 4
            for (int q = 0; q < H.ph; ++q) t += s[S_AH + ((w0 + q) & (kAlphaRing - 1))] * pc[q * H.stride];   // (alpha: wave-uniform scalar loads)
            x.p[c][i] = t;
        }
    }
}
// y = a + (x + sum of alpha_i p_i over the iterations since the last full window); x, p in the layout of the solve (slot: slab-permuted -> natural)
template <int NF>
__global__ __launch_bounds__(NT) void k_add_hist(const double *__restrict__ s, int64_t n, int np, const int *__restrict__ slot, F3 y, CF3 a, CF3 x, PHist H,
                                                 int64_t ld) {
    const int64_t lo = lane_lo(ld);
    s += lo, y = lane_f3(y, lo), a = lane_f3(a, lo), x = lane_f3(x, lo);
    const int iters = (int)s[S_ITERS];
    const int w0 = (iters / H.ph) * H.ph, cnt = iters - w0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        int64_t q = i;
        if (slot) {
            const int64_t e = i / np;
            q = e * np + slot[(int)(i - e * np)];
        }
        // the components side by side, four directions per trip: 4 NF independent loads in flight per lane (the sums stay per component, in
        // iteration order)
        double t[NF];
#pragma unroll
        for (int c = 0; c < NF; ++c) t[c] = w0 > 0 ? x.p[c][q] : 0.0;   // x holds the full windows, if there were any
#pragma unroll 4
        for (int k = 0; k < cnt; ++k) {
            const double al = s[S_AH + ((w0 + k) & (kAlphaRing - 1))];
#pragma unroll
            for (int c = 0; c < NF; ++c) t[c] += al * H.p0[c][lo + q + k * H.stride];
        }
#pragma unroll
        for (int c = 0; c < NF; ++c) y.p[c][i] = a.p[c] ? a.p[c][i] + t[c] : t[c];   // (a == null: the solution itself, the pressure solve)
    }
}

template <int NF>
__global__ __launch_bounds__(NT) void k_cg_rz(const double *s, int gate, int64_t n, CF3 r, CF3 z, const double *ipw,
                                              const double *xc, int npe, double *partial, int64_t ld) {
    __shared__ double sm[8];
    const int64_t lo = lane_lo(ld);
    s += lo, r = lane_f3(r, lo), z = lane_f3(z, lo), partial += lo;
    if (gate && s[S_DONE] != 0.0) return;
    double a = 0.0, b = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double wi = ipw ? ipw[i] : 1.0;
        const double xv = xc ? xc[i / npe] : 0.0;
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            const double zv = z.p[c][i] + xv;
            a += r.p[c][i] * zv * wi;
            b += zv;
        }
    }
    block_sum2(a, b, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = a;
        partial[2 * NB + blockIdx.x] = b;
    }
}

template <int NF>
__global__ __launch_bounds__(NT) void k_cg_pupdate(const double *s, int64_t n, F3 p, CF3 z, const double *xc, int npe, int64_t ld) {
    const int64_t lo = lane_lo(ld);
    s += lo, p = lane_f3(p, lo), z = lane_f3(z, lo);
    if (s[S_DONE] != 0.0) return;
    const double beta = s[S_BETA], zmean = s[S_ZMEAN];
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double xv = xc ? xc[i / npe] : 0.0;
#pragma unroll
        for (int c = 0; c < NF; ++c) p.p[c][i] = (z.p[c][i] + xv - zmean) + beta * p.p[c][i];
    }
}

// where the first-stage partial sums of up to three reductions live (they may come from different kernels)
struct Red {
    const double *p[3];
    int n[3];
};

// scalar logic, one thread. mode 0: after init (T0 = rz, T1 = rn2, T2 = sum z) ; 1: after pw (T0 = pw, T1 = sum w) ;
// 2: after update (T0 = rz, T1 = rn2, T2 = sum z).  inv_n = 1/n for the projected solve, 0 otherwise.
__device__ __forceinline__ void cg_post_logic(double *s, int mode, double tol2, int use_tol, int maxit, double inv_n) {
    if (mode != 0 && s[S_DONE] != 0.0) return;
    if (mode == 0) {
        s[S_RZ] = s[S_T0];
        s[S_RN2] = s[S_T1];
        s[S_ZMEAN] = s[S_T2] * inv_n;
        s[S_WMEAN] = 0.0;
        s[S_BETA] = 0.0;
        s[S_ITERS] = 0.0;
        s[S_DONE] = 0.0;     // the first p-update must run; the convergence test happens in mode 3
    } else if (mode == 3) {
        s[S_RN20] = s[S_RN2];
        s[S_DONE] = ((use_tol && s[S_RN2] < tol2) || maxit <= 0 || s[S_RN2] <= 0.0) ? 1.0 : 0.0;
    } else if (mode == 1) {
        s[S_PW] = s[S_T0];
        s[S_WMEAN] = s[S_T1] * inv_n;
        s[S_ALPHA] = s[S_RZ] / s[S_T0];
        s[S_AH + ((int)s[S_ITERS] & (kAlphaRing - 1))] = s[S_ALPHA];
    } else if (mode == 4) {
        // single-reduction PCG (Chronopoulos & Gear 1989): T0 = (w, u) with w = A u, u = M^-1 r;  T1 = (r, u);  T2 = |r|^2 -- all of the
        // CURRENT residual, in one reduction.  The convergence test standard PCG makes before applying the operator comes one operator
        // application late here (the price of the merged reduction): one wasted application per solve.
        if (s[S_ITERS] > 0.0 && ((use_tol && s[S_T2] < tol2) || s[S_ITERS] >= (double)maxit || s[S_T2] <= kFloor2 * s[S_RN20])) {
            s[S_RN2] = s[S_T2];
            s[S_DONE] = 1.0;
            return;
        }
        const bool first = s[S_ITERS] == 0.0;
        const double beta = first ? 0.0 : s[S_T1] / s[S_RZ];
        s[S_ALPHA] = first ? s[S_T1] / s[S_T0] : s[S_T1] / (s[S_T0] - beta * s[S_T1] / s[S_ALPHA]);
        s[S_BETA] = beta;
        s[S_RZ] = s[S_T1];
        s[S_RN2] = s[S_T2];
        s[S_PW] = s[S_T0];
        s[S_ITERS] += 1.0;
    } else {
        s[S_BETA] = s[S_T0] / s[S_RZ];
        s[S_RZ] = s[S_T0];
        s[S_RN2] = s[S_T1];
        s[S_ZMEAN] = s[S_T2] * inv_n;
        s[S_ITERS] += 1.0;
        if ((use_tol && s[S_T1] < tol2) || s[S_ITERS] >= (double)maxit || s[S_T1] <= kFloor2 * s[S_RN20]) s[S_DONE] = 1.0;
    }
}

// `red` (several ranks): the all-reduced sums of all lanes, [lane][3]; copied into the lane's S_T slots first
__global__ void k_cg_post(double *s, int mode, double tol2, int use_tol, int maxit, double inv_n, int64_t ld, const double *red, int nsums) {
    s += lane_lo(ld);
    if (red)
        for (int q = 0; q < nsums; ++q) s[S_T0 + q] = red[3 * blockIdx.y + q];
    cg_post_logic(s, mode, tol2, use_tol, maxit, inv_n);
}

// Second-stage reduction and (single rank: no all-reduce in between) the scalar logic in one launch.  One block of
// 1024 threads; the (up to three) sums are accumulated together so that their loads overlap, and the per-thread
// loop is unrolled four-fold for the same reason: the kernel is pure load latency.
constexpr int NTF = 1024;
__global__ __launch_bounds__(NTF) void k_cg_final_post(double *s, Red rd, int nsums, int gate, int mode, double tol2,
                                                       int use_tol, int maxit, double inv_n, int post, int64_t ld, double *red) {
    __shared__ double sm[3][NTF / 64];
    const int64_t lo = lane_lo(ld);
    s += lo;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (rd.p[q]) rd.p[q] += lo;
    if (gate && s[S_DONE] != 0.0) {
        // several ranks: a finished lane still takes part in the all-reduce of the lanes that are not; it contributes zeros
        if (red && threadIdx.x < 3) red[3 * blockIdx.y + threadIdx.x] = 0.0;
        return;
    }
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q >= nsums) break;
        const double *__restrict__ p = rd.p[q];
        const int n = rd.n[q];
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int i = threadIdx.x;
        for (; i + 3 * NTF < n; i += 4 * NTF) {
            a0 += p[i];
            a1 += p[i + NTF];
            a2 += p[i + 2 * NTF];
            a3 += p[i + 3 * NTF];
        }
        for (; i < n; i += NTF) a0 += p[i];
        acc[q] = (a0 + a1) + (a2 + a3);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double a = acc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if (lane == 0) sm[q][wid] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < nsums; ++q) {
            double a = 0.0;
            for (int w = 0; w < NTF / 64; ++w) a += sm[q][w];
            s[S_T0 + q] = a;
            if (red) red[3 * blockIdx.y + q] = a;
        }
        if (post) cg_post_logic(s, mode, tol2, use_tol, maxit, inv_n);   // several ranks: the all-reduce comes first
    }
}

// ---- residual projection of the pressure solve (Fischer 1998; Nek5000 `residualProj = yes`, 1cyl.par:23) ----
// X = up to PROJ_L previous solution increments, A-orthonormal (A = P E P), B = A X.  All on the device, no host sync.
constexpr int PROJ_L = 8;
// partial[v * NB + blk] = sum over the block's share of X_v . y   (v < nvec <= PROJ_L)
__global__ __launch_bounds__(NT) void k_proj_dots(int64_t n, const double *__restrict__ X, int64_t stride, int nvec,
                                                  const double *__restrict__ y, double *__restrict__ partial, int64_t ld) {
    __shared__ double sm[PROJ_L][NT / 64];
    const int64_t lo = lane_lo(ld);
    X += lo, y += lo, partial += lo;
    double a[PROJ_L];
#pragma unroll
    for (int v = 0; v < PROJ_L; ++v) a[v] = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double yv = y[i];
#pragma unroll
        for (int v = 0; v < PROJ_L; ++v)
            if (v < nvec) a[v] += X[v * stride + i] * yv;
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < PROJ_L; ++v) {
        double t = a[v];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
        if (lane == 0) sm[v][wid] = t;
    }
    __syncthreads();
    if (threadIdx.x < PROJ_L) {
        double t = 0.0;
        for (int w = 0; w < NT / 64; ++w) t += sm[threadIdx.x][w];
        partial[threadIdx.x * NB + blockIdx.x] = t;
    }
}
// out[v] = sum_blk partial[v * NB + blk]
__global__ __launch_bounds__(NT) void k_proj_reduce(const double *__restrict__ partial, int nblk, int nvec, double *__restrict__ out, int64_t ld) {
    __shared__ double sm[NT / 64];
    partial += lane_lo(ld), out += lane_lo(ld);
    for (int v = 0; v < nvec; ++v) {
        double t = 0.0;
        for (int i = threadIdx.x; i < nblk; i += NT) t += partial[v * NB + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0;
            for (int w = 0; w < NT / 64; ++w) a += sm[w];
            out[v] = a;
        }
        __syncthreads();
    }
}
// y += sgn * sum_v c[v] M_v
__global__ __launch_bounds__(NT) void k_proj_comb(int64_t n, double *__restrict__ y, const double *__restrict__ M, int64_t stride,
                                                  int nvec, const double *__restrict__ c, double sgn, int64_t ld) {
    double cv[PROJ_L];
    y += lane_lo(ld), M += lane_lo(ld), c += lane_lo(ld);
#pragma unroll
    for (int v = 0; v < PROJ_L; ++v) cv[v] = v < nvec ? sgn * c[v] : 0.0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        double a = y[i];
#pragma unroll
        for (int v = 0; v < PROJ_L; ++v)
            if (v < nvec) a += cv[v] * M[v * stride + i];
        y[i] = a;
    }
}
// new basis pair: X_new = v / sqrt(nrm2), B_new = w / sqrt(nrm2); a non-positive nrm2 stores zeros (a harmless member)
__global__ __launch_bounds__(NT) void k_proj_store(int64_t n, const double *__restrict__ v, const double *__restrict__ w,
                                                   const double *__restrict__ nrm2, double *__restrict__ Xn, double *__restrict__ Bn, int64_t ld) {
    const int64_t lo = lane_lo(ld);
    v += lo, w += lo, nrm2 += lo, Xn += lo, Bn += lo;
    const double q = nrm2[0];
    const double sc = q > 0.0 ? 1.0 / sqrt(q) : 0.0;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        Xn[i] = sc * v[i];
        Bn[i] = sc * w[i];
    }
}

// ---- wavenumber projection of the projected propagator (proj_alpha, src/linops/exponential_propagator_proj.f90:135-173):
// every velocity component keeps only its cos(alpha x) / sin(alpha x) content along the homogeneous direction,
//   u <- cv <2 u cv> + sv <2 u sv>,   <f> = sum_line bm1 f / sum_line bm1 over the points of a line along that direction
// (Nek5000's planar_avg over the gtpp gather-scatter handle).  One wave per line.
template <int NF>
__global__ __launch_bounds__(NT) void k_proj_alpha(int64_t nlines, const int *__restrict__ off, const int *__restrict__ idx,
                                                   const double *__restrict__ bm1, const double *__restrict__ cv,
                                                   const double *__restrict__ sv, const double *__restrict__ inv_den, F3 u) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (g >= nlines) return;
    const int b = off[g], e = off[g + 1];
    double ac[NF], as[NF];
#pragma unroll
    for (int c = 0; c < NF; ++c) ac[c] = as[c] = 0.0;
    for (int q = b + lane; q < e; q += 64) {
        const int i = idx[q];
        const double w = 2.0 * bm1[i], c0 = cv[i], s0 = sv[i];
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            const double v = u.p[c][i];
            ac[c] += w * v * c0;
            as[c] += w * v * s0;
        }
    }
#pragma unroll
    for (int c = 0; c < NF; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ac[c] += __shfl_xor(ac[c], o, 64);
            as[c] += __shfl_xor(as[c], o, 64);
        }
    }
    const double id = inv_den[g];
    for (int q = b + lane; q < e; q += 64) {
        const int i = idx[q];
#pragma unroll
        for (int c = 0; c < NF; ++c) u.p[c][i] = cv[i] * (ac[c] * id) + sv[i] * (as[c] * id);
    }
}
// Several ranks: a line along the homogeneous direction may cross rank boundaries.  The weighted sums of a rank's part of a
// line go to the line's slot in a global array (one slot per distinct line label over all ranks), the array is all-reduced,
// and a second kernel applies the projection with the global sums.
template <int NF>
__global__ __launch_bounds__(NT) void k_proj_sums(int64_t nlines, const int *__restrict__ off, const int *__restrict__ idx,
                                                  const int *__restrict__ gslot, const double *__restrict__ bm1,
                                                  const double *__restrict__ cv, const double *__restrict__ sv, CF3 u,
                                                  double *__restrict__ glob) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (g >= nlines) return;
    const int b = off[g], e = off[g + 1];
    double ac[NF], as[NF];
#pragma unroll
    for (int c = 0; c < NF; ++c) ac[c] = as[c] = 0.0;
    for (int q = b + lane; q < e; q += 64) {
        const int i = idx[q];
        const double w = 2.0 * bm1[i], c0 = cv[i], s0 = sv[i];
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            const double v = u.p[c][i];
            ac[c] += w * v * c0;
            as[c] += w * v * s0;
        }
    }
#pragma unroll
    for (int c = 0; c < NF; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ac[c] += __shfl_xor(ac[c], o, 64);
            as[c] += __shfl_xor(as[c], o, 64);
        }
    }
    if (lane == 0) {
        double *dst = glob + (int64_t)gslot[g] * (2 * NF);
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            dst[2 * c] = ac[c];
            dst[2 * c + 1] = as[c];
        }
    }
}
template <int NF>
__global__ __launch_bounds__(NT) void k_proj_apply(int64_t nlines, const int *__restrict__ off, const int *__restrict__ idx,
                                                   const int *__restrict__ gslot, const double *__restrict__ cv,
                                                   const double *__restrict__ sv, const double *__restrict__ inv_den,
                                                   const double *__restrict__ glob, F3 u) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (g >= nlines) return;
    const int b = off[g], e = off[g + 1];
    const double *src = glob + (int64_t)gslot[g] * (2 * NF);
    const double id = inv_den[g];
    for (int q = b + lane; q < e; q += 64) {
        const int i = idx[q];
#pragma unroll
        for (int c = 0; c < NF; ++c) u.p[c][i] = cv[i] * (src[2 * c] * id) + sv[i] * (src[2 * c + 1] * id);
    }
}
// weights of a rank's part of every line into the global slots (set-up: the denominators)
__global__ __launch_bounds__(NT) void k_proj_wsum(int64_t nlines, const int *__restrict__ off, const int *__restrict__ idx,
                                                  const int *__restrict__ gslot, const double *__restrict__ w, double *__restrict__ glob) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (g >= nlines) return;
    double a = 0.0;
    for (int q = off[g] + lane; q < off[g + 1]; q += 64) a += w[idx[q]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) glob[gslot[g]] = a;
}
__global__ __launch_bounds__(NT) void k_proj_iden(int64_t nlines, const int *__restrict__ gslot, const double *__restrict__ glob, double *__restrict__ iden) {
    for (int64_t g = blockIdx.x * (int64_t)NT + threadIdx.x; g < nlines; g += (int64_t)gridDim.x * NT) iden[g] = 1.0 / glob[gslot[g]];
}

__global__ __launch_bounds__(NT) void k_cossin(int64_t n, const double *__restrict__ x, double alpha, double *__restrict__ cv,
                                               double *__restrict__ sv) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        cv[i] = cos(alpha * x[i]);
        sv[i] = sin(alpha * x[i]);
    }
}

// F -= bm1 (cr f_re + ci f_im): a body force in the explicit term (the stored F is +N, the right-hand side takes -EXT(F))
template <int NF>
__global__ __launch_bounds__(NT) void k_add_force(int64_t n, F3 F, const double *__restrict__ bm1, CF3 fre, CF3 fim, double cr, double ci) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double b = bm1[i];
#pragma unroll
        for (int c = 0; c < NF; ++c) F.p[c][i] -= b * (cr * fre.p[c][i] + (fim.p[c] ? ci * fim.p[c][i] : 0.0));
    }
}

// F_i -= bm1 * buoy_i * theta : Boussinesq buoyancy in the explicit term (the stored F is +N)
template <int NF>
__global__ __launch_bounds__(NT) void k_buoyancy(int64_t n, F3 F, const double *__restrict__ bm1, const double *__restrict__ theta,
                                                 double b0, double b1, double b2) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double t = bm1[i] * theta[i];
        F.p[0][i] -= b0 * t;
        if (NF > 1) F.p[1][i] -= b1 * t;
        if (NF > 2) F.p[2][i] -= b2 * t;
    }
}

// generic pointwise helpers
template <int NF>
__global__ __launch_bounds__(NT) void k_colmul_gated(const double *s, F3 w, CF3 wt, int64_t n, int64_t ld) {
    const int64_t lo = lane_lo(ld);
    if (s) s += lo;
    w = lane_f3(w, lo);
    if (s && s[S_DONE] != 0.0) return;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
#pragma unroll
        for (int c = 0; c < NF; ++c) w.p[c][i] *= wt.p[c][i];
    }
}

// rhs_i = sum_j ab_j F_j,i + (bm1/dt) sum_j bd_j u_j,i
struct Hist {
    const double *f[3][3];   // [level][component]
    const double *u[3][3];
    double ab[3], bd[3];
    int k;
};
// `gp` / `w` non-null: the residual form of the tentative-velocity problem in the same pass, rhs_i += gp_i - w_i (D^T p - H u)
// XP: the result is written masked and in the slab-permuted layout of the velocity solve (slot = element-local table), so that its
// gather-scatter runs on that layout's lists and the solve needs no permutation pass (the Dirichlet mask is the same for every copy
// of a dof, so masking before the gather-scatter gives the bits masking after it gave)
template <int NF, bool XP = false>
__global__ __launch_bounds__(NT) void k_rhs(int64_t n, Hist h, const double *bm1, double rdt, F3 rhs, int64_t ld, CF3 gp = CF3{{nullptr, nullptr, nullptr}},
                                            CF3 w = CF3{{nullptr, nullptr, nullptr}}, const int *__restrict__ slot = nullptr, int np = 1,
                                            CF3 mask = CF3{{nullptr, nullptr, nullptr}}) {
    // (the lane offset is added at the loads: writing it into `h` would move the by-value struct from the kernel-argument segment,
    //  where the runtime index j costs a scalar load, into scratch memory -- measured 123 -> 315 us)
    const int64_t lo = lane_lo(ld);
    rhs = lane_f3(rhs, lo);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const double b = bm1[i] * rdt;
        int64_t q = i;
        if (XP) {
            const int64_t e = i / np;
            q = e * np + slot[(int)(i - e * np)];
        }
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            double a = 0.0, u = 0.0;
            for (int j = 0; j < h.k; ++j) {
                a += h.ab[j] * h.f[j][c][lo + i];
                u += h.bd[j] * h.u[j][c][lo + i];
            }
            double v = a + b * u;
            if (gp.p[c]) v = (v + gp.p[c][lo + i]) - w.p[c][lo + i];
            rhs.p[c][q] = XP ? mask.p[c][i] * v : v;
        }
    }
}

// y_c = a_c + s1 * b_c + s2 * c_c   (any of b, c may be null)
template <int NF>
__global__ __launch_bounds__(NT) void k_lin3(int64_t n, F3 y, CF3 a, CF3 b, double s1, CF3 c, double s2, int64_t ld) {
    const int64_t lo = lane_lo(ld);
    y = lane_f3(y, lo), a = lane_f3(a, lo), b = lane_f3(b, lo), c = lane_f3(c, lo);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
#pragma unroll
        for (int q = 0; q < NF; ++q) {
            double v = a.p[q][i];
            if (b.p[q]) v += s1 * b.p[q][i];
            if (c.p[q]) v += s2 * c.p[q][i];
            y.p[q][i] = v;
        }
    }
}

// y_c = a_c + b_c with b stored in the slab-permuted element layout (the solution of the velocity PCG): the un-permutation rides in
// the update of the velocity instead of being a pass of its own
template <int NF>
__global__ __launch_bounds__(NT) void k_add_xp(int64_t n, int np, const int *__restrict__ slot, F3 y, CF3 a, CF3 b, int64_t ld) {
    const int64_t lo = lane_lo(ld);
    y = lane_f3(y, lo), a = lane_f3(a, lo), b = lane_f3(b, lo);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int64_t e = i / np;
        const int64_t q = e * np + slot[(int)(i - e * np)];
#pragma unroll
        for (int c = 0; c < NF; ++c) y.p[c][i] = a.p[c][i] + b.p[c][q];
    }
}

// y_c += s * wt_c * x_c  (velocity correction: the inverse mass / mask weights of opbinv ride in the update)
template <int NF, bool SLOT = false>
__global__ __launch_bounds__(NT) void k_axpy_w(int64_t n, F3 y, CF3 x, CF3 wt, double s, int64_t ld, const int *__restrict__ slot = nullptr, int np = 1) {
    // SLOT: x is stored in an element-local permutation (the face-grouped layout of the pressure operator's intermediates)
    const int64_t lo = lane_lo(ld);
    y = lane_f3(y, lo), x = lane_f3(x, lo);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        int64_t q = i;
        if (SLOT) {
            const int64_t e = i / np;
            q = e * np + slot[(int)(i - e * np)];
        }
#pragma unroll
        for (int c = 0; c < NF; ++c) y.p[c][i] += s * (wt.p[c][i] * x.p[c][q]);
    }
}

__global__ __launch_bounds__(NT) void k_scale1(int64_t n, double *y, const double *x, double s) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] = s * x[i];
}
__global__ __launch_bounds__(NT) void k_axpy1(int64_t n, double *y, const double *x, double s, int64_t ld) {
    y += lane_lo(ld), x += lane_lo(ld);
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] += s * x[i];
}
__global__ __launch_bounds__(NT) void k_recipmask(int64_t n, double *y, const double *d, const double *mask) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] = mask[i] / d[i];
}
__global__ __launch_bounds__(NT) void k_maskbits(int64_t n, double *y, const double *m0, const double *m1, const double *m2) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
        y[i] = (m0[i] != 0.0 ? 1.0 : 0.0) + (m1 && m1[i] != 0.0 ? 2.0 : 0.0) + (m2 && m2[i] != 0.0 ? 4.0 : 0.0);
}
__global__ __launch_bounds__(NT) void k_to_bytes(int64_t n, unsigned char *y, const double *x) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] = (unsigned char)x[i];
}
__global__ __launch_bounds__(NT) void k_recip1(int64_t n, double *y, const double *d) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] = 1.0 / d[i];
}
__global__ __launch_bounds__(NT) void k_mul3_acc(int64_t n, double *y, const double *a, const double *b, double s) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] += s * a[i] * b[i];
}
__global__ __launch_bounds__(NT) void k_mul3(int64_t n, double *y, const double *a, const double *b, double s) {
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) y[i] = s * a[i] * b[i];
}

inline int grid_for(int64_t n) {
    int64_t g = (n + NT - 1) / NT;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}
inline int red_grid(int64_t n) {
    int64_t g = (n + NT * 4 - 1) / (NT * 4);
    if (g > NB) g = NB;
    if (g < 1) g = 1;
    return (int)g;
}

const double BDF_B0[4] = {0.0, 1.0, 1.5, 11.0 / 6.0};
const double BDF_C[4][3] = {{0, 0, 0}, {1.0, 0, 0}, {2.0, -0.5, 0}, {3.0, -1.5, 1.0 / 3.0}};
const double EXT_C[4][3] = {{0, 0, 0}, {1.0, 0, 0}, {2.0, -1.0, 0}, {3.0, -3.0, 1.0}};

}  // namespace

struct nlg_linop {
    nlg_mesh *mesh = nullptr;
    nlg_exptA_config cfg;
    nlg_vec *baseflow = nullptr;
    bool inited = false;
    double dt = 0, cfl = 0;
    int nsteps = 0;
    // convective-term precomputation on the fine mesh
    double *Ur[3] = {}, *GU[9] = {};
    // state: three rotating velocity buffers and three rotating forcing buffers per component
    double *ubuf[3][3] = {};   // [slot][component]; slot 0 = current, 1 = lag1, 2 = lag2 (after rotation)
    double *fbuf[3][3] = {};
    double *p = nullptr;
    // work
    double *rhs[3] = {}, *x[3] = {}, *z[3] = {}, *pv[3] = {}, *w[3] = {}, *gp[3] = {};
    // single-reduction PCG (Chronopoulos-Gear; several ranks, NLG_PCG_SINGLE_RED): the search direction and its image as recurrences
    bool use_sr = false;
    int php = 0;               // the same for the pressure PCG (ring prh[php][lps]; the update rides in the preconditioner, the direction update in the gradient kernel)
    double *prh = nullptr;
    int ph = 0;                // depth of the direction ring of the velocity PCG (deferred solution update, k_add_hist); 0 = x updated every iteration
    double *phist = nullptr;   // [ph][dim][lvs]
    double *cgs[3] = {}, *tcgs = nullptr;
    bool rhs_in_xp = false;   // adv_a -> helm_problem: the right-hand side already is masked and in the slab-permuted layout
    double *pr_r = nullptr, *pr_x = nullptr, *pr_z = nullptr, *pr_p = nullptr, *pr_w = nullptr;
    double *pr_b = nullptr;    // the right-hand side the pressure PCG started from (after the projection): A x = b - r afterwards
    double *pcv[4][3] = {};    // mask_i / diag(H) per BDF order
    // velocity PCG in the x-planes-first layout (3-D, lx1 <= 8: internal.h xp_slot): the preconditioners and the residual
    // weight permuted once; 0 = natural layout (2-D, lx1 > 8, NLG_XP=0)
    int use_xp = -1;
    double *pcv_xp[4][3] = {}, *nwv_xp = nullptr;
    // the same preconditioners as ONE array 1 / diag(H) per BDF order plus one mask byte per point (bit c = mask of component c), in the
    // layout of the velocity solve: what k_cg_init / k_cg_update read (3 streams -> 1 + 1/8); pci_l[k] = {pci[k], pci[k], pci[k]}
    double *pci[4] = {}, *pci_l[4][3] = {};
    unsigned char *maskb_v = nullptr;
    double *pce = nullptr;     // 1 / diag(E)
    double *prX = nullptr, *prB = nullptr, *d_pc = nullptr;   // pressure residual projection: PROJ_L solution / image pairs, coefficients
    int nproj = 0;
    double *nwv = nullptr;     // binvm1 * vmult / volvm1
    double *nwp = nullptr;     // bm2inv / volvm2
    double *d_s = nullptr;     // solver scalars (two blocks of S_N)
    double *d_part = nullptr;  // first-stage sums written by opdiv ([2][E]) and by the FDM kernel ([2][E/4])
    double *d_cgpart = nullptr;   // [3][NB] first-stage sums of the PCG vector kernels
    double *d_red = nullptr;      // several ranks: [lanes][3] sums of all lanes for ONE all-reduce (owner only, not in the slab)
    double *h_s = nullptr;     // pinned
    // All per-lane work buffers (integrator state, PCG vectors, projection space, scalars, partial sums) are carved from ONE
    // allocation of `slab_cap` lanes at a constant stride `slab_ld` doubles: lane v's copy of any of them is the owner's
    // pointer + v * slab_ld, which is what lets one launch with gridDim.y = lanes serve the whole block (lane_lo()).
    double *slab = nullptr;
    int64_t slab_ld = 0;
    int slab_cap = 0;
    int64_t n_launch = 0, n_coll = 0;   // kernel launches / collectives issued by the time stepper (nlg_linop_get_counters)
    int istep = 0, adjoint = 0;
    // block stepper: lanes 1 .. 3 (created on first use by the operator that owns them); a lane shares the base-flow data
    // of its owner and must never free it
    nlg_linop *lanes[3] = {nullptr, nullptr, nullptr};
    bool is_lane = false;
    int adv_k = 1;             // state handed from one phase of a time step to the next (adv_a / adv_b / adv_c)
    double adv_b0 = 1.0, adv_h2 = 0.0;
    // wavenumber projection (exptA_proj_linop): lines along the homogeneous direction, cos / sin of alpha x, 1 / sum bm1
    int proj_nlines = 0, proj_nlines2 = 0;     // velocity-mesh lines; pressure-mesh lines (0 = pressure not projected)
    int *proj_off = nullptr, *proj_idx = nullptr, *proj_off2 = nullptr, *proj_idx2 = nullptr;
    // several ranks: slot of every local line in the global line list, number of global lines, all-reduce buffer
    int *proj_gslot = nullptr, *proj_gslot2 = nullptr;
    int64_t proj_nglob = 0, proj_nglob2 = 0;
    double *proj_glob = nullptr;
    double *proj_cv = nullptr, *proj_sv = nullptr, *proj_iden = nullptr, *proj_cv2 = nullptr, *proj_sv2 = nullptr, *proj_iden2 = nullptr;
    // time-harmonic body force Re(f exp(i s omega t)) of the resolvent integrations (null = none)
    const nlg_vec *force_re = nullptr, *force_im = nullptr;
    double force_omega = 0.0, force_sign = 1.0;
    // Boussinesq coupling (cfg.ifheat): temperature levels, its explicit terms, PCG work fields, Jacobi preconditioners per
    // BDF order, gradient of the base temperature on the fine mesh
    double *tbuf[3] = {}, *ftbuf[3] = {}, *trhs = nullptr, *tx = nullptr, *tz = nullptr, *tpv = nullptr, *tw = nullptr;
    double *pct[4] = {}, *GT[3] = {};
    int64_t st_titers = 0;
    int last_titers = 8;
    int nonlinear = 0;         // 1: full Navier-Stokes step, N(u) = (u.grad)u = half of the linearised term about U = u
    int64_t st_steps = 0, st_viters = 0, st_piters = 0, st_matvecs = 0;
    int last_piters = 16, last_viters = 8;
    std::vector<int> pit_hist, vit_hist;   // iteration counts of the previous matvec, by time-step index (the pattern repeats)
};

namespace {

int lalloc(nlg_linop *op, double **p, int64_t n) {
    NLG_HIP(hipMalloc(p, sizeof(double) * (size_t)n));
    NLG_HIP(hipMemsetAsync(*p, 0, sizeof(double) * (size_t)n, op->mesh->ctx->stream));
    return 0;
}

// ---- the lane slab ----------------------------------------------------------------------------------------------------
// every per-lane work buffer of the integrator, in slab order: f(pointer member, length in doubles).  The rotating history
// buffers come first and are contiguous, so that one strided memset clears the integrator state of all lanes.
template <typename F>
void lane_buffers(nlg_linop *op, F f) {
    nlg_mesh *m = op->mesh;
    const int dim = m->dim;
    for (int s = 0; s < 3; ++s)
        for (int c = 0; c < dim; ++c) f(&op->ubuf[s][c], m->lvs);
    for (int s = 0; s < 3; ++s)
        for (int c = 0; c < dim; ++c) f(&op->fbuf[s][c], m->lvs);
    if (op->cfg.ifheat) {
        for (int q = 0; q < 3; ++q) f(&op->tbuf[q], m->lvs);
        for (int q = 0; q < 3; ++q) f(&op->ftbuf[q], m->lvs);
    }
    for (int c = 0; c < dim; ++c) {
        f(&op->rhs[c], m->lvs);
        f(&op->x[c], m->lvs);
        f(&op->z[c], m->lvs);
        f(&op->pv[c], m->lvs);
        f(&op->w[c], m->lvs);
        f(&op->gp[c], m->lvs);
    }
    if (op->cfg.ifheat)
        for (double **v : {&op->trhs, &op->tx, &op->tz, &op->tpv, &op->tw}) f(v, m->lvs);
    if (op->ph > 0) f(&op->phist, (int64_t)op->ph * dim * m->lvs);
    if (op->php > 0) f(&op->prh, (int64_t)op->php * m->lps);
    if (op->use_sr) {
        for (int c = 0; c < dim; ++c) f(&op->cgs[c], m->lvs);
        if (op->cfg.ifheat) f(&op->tcgs, m->lvs);
    }
    for (double **q : {&op->p, &op->pr_r, &op->pr_x, &op->pr_z, &op->pr_p, &op->pr_w, &op->pr_b}) f(q, m->lps);
    if (op->cfg.pproj) {
        f(&op->prX, (int64_t)PROJ_L * m->lps);
        f(&op->prB, (int64_t)PROJ_L * m->lps);
        f(&op->d_pc, 4 * PROJ_L + PROJ_L * NB);
    }
    f(&op->d_s, 4 * S_N);
    f(&op->d_part, 3 * m->E + 16);
    f(&op->d_cgpart, 3 * NB);
}

int64_t lane_stride(nlg_linop *op) {
    int64_t off = 0;
    lane_buffers(op, [&](double **, int64_t len) { off += round_up(len, kAlign); });
    return off;
}

// point the members of `ln` (the owner itself for lane 0) at lane `v` of the owner's slab; the rotating buffers return to
// their canonical places, which keeps "lane v = lane 0 + v * slab_ld" true for every member whatever was run before
void lane_bind(nlg_linop *owner, nlg_linop *ln, int v) {
    int64_t off = 0;
    double *base = owner->slab + (int64_t)v * owner->slab_ld;
    const bool heat = ln->cfg.ifheat;
    ln->cfg.ifheat = owner->cfg.ifheat;   // (same buffer list as the owner's)
    ln->use_sr = owner->use_sr;
    ln->ph = owner->ph;
    ln->php = owner->php;
    lane_buffers(ln, [&](double **p, int64_t len) {
        *p = base + off;
        off += round_up(len, kAlign);
    });
    ln->cfg.ifheat = heat;
    ln->slab_ld = owner->slab_ld;
}

// make room for `cap` lanes (the work buffers hold no state between two matvecs, so growing = a new allocation)
int slab_ensure(nlg_linop *op, int cap) {
    if (op->slab && op->slab_cap >= cap) return 0;
    nlg_ctx *ctx = op->mesh->ctx;
    NLG_HIP(hipStreamSynchronize(ctx->stream));
    if (op->slab) NLG_HIP(hipFree(op->slab));   // (first: the old and the new slab need not exist together)
    op->slab = nullptr;
    op->slab_cap = 0;
    // the direction rings (deferred solution update) are bandwidth bought with memory: where the slab does not fit with them -- a block of
    // lanes next to a Krylov basis that fills the card -- they shrink and finally go (k_cg_update then updates x every iteration again)
    int64_t ld = 0;
    double *nslab = nullptr;
    for (;;) {
        ld = lane_stride(op);
        const bool rings = op->ph > 0 || op->php > 0;
        if (hipMalloc(&nslab, sizeof(double) * (size_t)(ld * cap)) == hipSuccess) {
            size_t free_b = 0, total_b = 0;
            // with rings, a tenth of the card must stay free for what is allocated later (lane scratch of the pressure operator and of the
            // preconditioner, solver work space): the rings are the one thing here that is optional
            if (!rings || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= total_b / 10) break;
            (void)hipFree(nslab);
        } else {
            (void)hipGetLastError();
            NLG_CHECK(rings, "exptA: out of device memory for the work buffers of %d lane(s) (%.1f GB)", cap, 8e-9 * (double)(ld * cap));
        }
        nslab = nullptr;
        if (op->php > 0)
            op->php = 0;
        else if (op->ph > 3)
            op->ph = std::max(3, op->ph / 2);
        else
            op->ph = 0;
    }
    NLG_HIP(hipMemsetAsync(nslab, 0, sizeof(double) * (size_t)(ld * cap), ctx->stream));
    op->slab = nslab;
    op->slab_ld = ld;
    op->slab_cap = cap;
    lane_bind(op, op, 0);
    for (int v = 1; v < kMaxLanes; ++v)
        if (op->lanes[v - 1]) lane_bind(op, op->lanes[v - 1], v);
    if (!op->d_red) NLG_TRY(lalloc(op, &op->d_red, 3 * kMaxLanes));
    return 0;
}

template <typename K, typename... A>
void launch_nf(int nf, K k1, K k2, K k3, dim3 g, hipStream_t s, A... a) {
    if (nf == 1)
        NLG_LAUNCH(k1, g, dim3(NT), 0, s, a...);
    else if (nf == 2)
        NLG_LAUNCH(k2, g, dim3(NT), 0, s, a...);
    else
        NLG_LAUNCH(k3, g, dim3(NT), 0, s, a...);
}

F3 f3(double *const *a, int nf) {
    F3 r = {{a[0], nf > 1 ? a[1] : nullptr, nf > 2 ? a[2] : nullptr}};
    return r;
}
CF3 cf3(double *const *a, int nf) {
    CF3 r = {{a[0], nf > 1 ? a[1] : nullptr, nf > 2 ? a[2] : nullptr}};
    return r;
}
inline dim3 lgrid(int g, int nl) { return dim3((unsigned)g, (unsigned)nl); }

// One PCG problem for nl lanes in lockstep: every pointer is lane 0's, lane v's copy sits v * ld doubles behind it (the lanes
// share a slab); tolerances, iteration limits and the operator are the same for all lanes, alpha / beta / residual norms / the
// convergence flag are per lane (device scalars), so each lane performs exactly the iteration it would perform on its own.
struct CGProblem {
    int nf;
    int64_t n;           // entries per field
    double *const *x, *const *r, *const *z, *const *p, *const *w;
    double *const *pc;
    const unsigned char *pc_mb = nullptr;   // non-null: pc[0] = 1 / diag for every component, bit c of byte i = mask of component c at point i
    const double *ipw, *nw;
    double tol2;         // squared tolerance on sum r^2 nw
    int use_tol, maxit;
    double *s;           // device scalars
    int chunk;           // iterations launched before the host first looks at the done flags (the prediction)
    int chunk_next = 2;  // ... and per look afterwards
    double inv_n;        // 1/n for the mean-free projected solve, 0 = no projection
    int nl = 1;          // lanes
    int64_t ld = 0;      // lane stride
    // non-pointwise M^-1 (nf = 1): writes the element-wise part to z and returns the coarse part per element in *xc
    std::function<int(const double *flag, const double *r, double *z, const double **xc)> precond;
    int npe = 1;         // points per element (for the per-element coarse part)
    // first-stage sums produced by the operator / preconditioner kernels themselves (null = separate kernels)
    const double *pw_part = nullptr;   // [2][pw_n]: sum p.w , sum w   (written by `apply`)
    int pw_n = 0;
    bool pw_sum = true;                // false: the sum of w is not provided (and not needed: inv_n == 0)
    bool fused_pupdate = false;        // `apply` itself performs p <- z + beta p (gated by the done flag) before w = A p
    const double *rz_part = nullptr;   // [2][rz_n]: sum r.z , sum z   (written by `precond`)
    int rz_n = 0;
    const double *rr_part = nullptr;   // [rr_n]: sum r^2 nw, written by `precond`, which then also performs the update
    int rr_n = 0;                      // x += alpha p, r -= alpha (w - wmean) of the iteration (no k_cg_update launch)
    // single-reduction variant (pointwise preconditioner only): `sd` holds s = A p by recurrence, `apply_plain` computes w = A z (the
    // preconditioned residual itself, no direction update inside) and its first-stage sums (z, w) into pw_part
    double *const *sd = nullptr;
    std::function<int()> apply_plain;
    // deferred solution update (k_x_flush / k_add_hist): `apply` stores direction i into slot i mod ph of the direction ring `hist`;
    // x is then NOT complete when run_pcg returns -- its consumer assembles it (adv_b)
    PHist hist = {{nullptr, nullptr, nullptr}, 0, 0};
};

// Device-scalar PCG for P.nl lanes.  `apply` computes w = A p for ALL lanes (stream-ordered, gated by each lane's s[S_DONE]).
// Every kernel of the iteration is ONE launch with gridDim.y = lanes; several ranks: ONE all-reduce per reduction carries the
// sums of all lanes.  iters_out[v]: iterations of lane v.
template <typename Apply>
int run_pcg(nlg_linop *op, const CGProblem &P, Apply apply, int *iters_out) {
    nlg_ctx *ctx = op->mesh->ctx;
    hipStream_t st = ctx->stream;
    const int nf = P.nf, nl = P.nl;
    const int64_t ld = P.ld;
    const int g = red_grid(P.n);
    double *partial = op->d_cgpart;
    double *s = P.s;
    F3 x = f3(P.x, nf), r = f3(P.r, nf), z = f3(P.z, nf), p = f3(P.p, nf);
    CF3 pc = cf3(P.pc, nf), cp = cf3(P.p, nf), cw = cf3(P.w, nf), cz = cf3(P.z, nf);
    CF3 cr = cf3(P.r, nf);
    const Red rd_std = {{partial, partial + NB, partial + 2 * NB}, {g, g, g}};
    auto reduce_post = [&](const Red &rd, int nsums, int gate, int mode) -> int {
        if (!ctx->distributed()) {
            ++g_collectives;   // (the all-reduce a partitioned run issues here; counted on one rank too, nlg_counters)
            NLG_LAUNCH(k_cg_final_post, lgrid(1, nl), dim3(NTF), 0, st, s, rd, nsums, gate, mode, P.tol2, P.use_tol, P.maxit,
                       P.inv_n, 1, ld, (double *)nullptr);
        } else {
            NLG_LAUNCH(k_cg_final_post, lgrid(1, nl), dim3(NTF), 0, st, s, rd, nsums, gate, mode, P.tol2, P.use_tol, P.maxit,
                       P.inv_n, 0, ld, op->d_red);
            NLG_TRY(allreduce_sum(ctx, op->d_red, 3 * nl));
            NLG_LAUNCH(k_cg_post, lgrid(1, nl), dim3(1), 0, st, s, mode, P.tol2, P.use_tol, P.maxit, P.inv_n, ld, (const double *)op->d_red, nsums);
        }
        return 0;
    };
    launch_nf(nf, k_cg_init<1>, k_cg_init<2>, k_cg_init<3>, lgrid(g, nl), st, P.n, x, r, z, pc, P.ipw, P.nw, partial, ld, P.pc_mb, (int)(P.hist.ph > 0));
    if (P.sd && P.apply_plain && !P.precond && P.pw_part) {
        // ---- single-reduction PCG (Chronopoulos-Gear): ONE reduction per iteration carries (w, u), (r, u) and |r|^2; several ranks: one
        // all-reduce instead of two.  u = M^-1 r lives in z, p and s = A p are recurrences.  Same iterates as the loop below in exact
        // arithmetic; the convergence test lags one operator application (cg_post_logic mode 4).
        NLG_TRY(reduce_post(rd_std, 3, 0, 0));
        NLG_LAUNCH(k_cg_post, lgrid(1, nl), dim3(1), 0, st, s, 3, P.tol2, P.use_tol, P.maxit, P.inv_n, ld, (const double *)nullptr, 0);
        const Red rd_sr = {{P.pw_part, partial, partial + NB}, {P.pw_n, g, g}};
        F3 sdv = f3(P.sd, nf);
        auto body_sr = [&]() -> int {
            NLG_TRY(P.apply_plain());
            NLG_TRY(reduce_post(rd_sr, 3, 1, 4));
            ProfScope pu(ctx, P_CGUPDATE);
            launch_nf(nf, k_cg_update_sr<1>, k_cg_update_sr<2>, k_cg_update_sr<3>, lgrid(g, nl), st, (const double *)s, P.n, x, r, z, p, sdv, cw,
                      pc, P.ipw, P.nw, partial, ld);
            return 0;
        };
        int launched = 0;
        while (true) {
            int todo = launched == 0 ? P.chunk + 1 : P.chunk_next;   // (+ 1: the iteration that notices the convergence)
            if (launched + todo > P.maxit + 1) todo = P.maxit + 1 - launched;
            for (int it = 0; it < todo; ++it) NLG_TRY(body_sr());
            launched += todo;
            NLG_HIP(hipGetLastError());
            for (int v = 0; v < nl; ++v)
                NLG_HIP(hipMemcpyAsync(op->h_s + (size_t)v * S_N, s + (int64_t)v * ld, sizeof(double) * S_N, hipMemcpyDeviceToHost, st));
            NLG_HIP(hipStreamSynchronize(st));
            bool all_done = true;
            for (int v = 0; v < nl; ++v) all_done = all_done && op->h_s[(size_t)v * S_N + S_DONE] != 0.0;
            if (all_done || launched >= P.maxit + 1) break;
        }
        for (int v = 0; v < nl; ++v) {
            const double *h = op->h_s + (size_t)v * S_N;
            if (!std::isfinite(h[S_RN2])) {
                set_error("PCG (single reduction) diverged (residual is not finite) after %d iterations (lane %d of %d)", (int)h[S_ITERS], v, nl);
                return 1;
            }
            iters_out[v] = (int)h[S_ITERS];
        }
        return 0;
    }
    const double *xc = nullptr;
    Red rd_rz = rd_std;
    if (P.precond) {
        NLG_TRY(P.precond(nullptr, P.r[0], P.z[0], &xc));
        if (P.rz_part) {
            rd_rz.p[0] = P.rz_part;
            rd_rz.n[0] = P.rz_n;
            rd_rz.p[2] = P.rz_part + P.rz_n;
            rd_rz.n[2] = P.rz_n;
        } else {
            launch_nf(nf, k_cg_rz<1>, k_cg_rz<2>, k_cg_rz<3>, lgrid(g, nl), st, (const double *)s, 0, P.n, cr, cz, P.ipw, xc, P.npe, partial, ld);
        }
    }
    Red rd_rz_loop = rd_rz;   // inside the loop the r^2 sums may come from the preconditioner's first kernel
    if (P.rr_part) {
        rd_rz_loop.p[1] = P.rr_part;
        rd_rz_loop.n[1] = P.rr_n;
    }
    Red rd_pw = rd_std;
    if (P.pw_part) {
        rd_pw.p[0] = P.pw_part;
        rd_pw.n[0] = P.pw_n;
        rd_pw.p[1] = P.pw_part + P.pw_n;
        rd_pw.n[1] = P.pw_sum ? P.pw_n : 0;
    }
    NLG_TRY(reduce_post(rd_rz, 3, 0, 0));
    if (!P.fused_pupdate)
        launch_nf(nf, k_cg_pupdate<1>, k_cg_pupdate<2>, k_cg_pupdate<3>, lgrid(g, nl), st, (const double *)s, P.n, p, cz, xc, P.npe, ld);   // p = z - zmean
    NLG_LAUNCH(k_cg_post, lgrid(1, nl), dim3(1), 0, st, s, 3, P.tol2, P.use_tol, P.maxit, P.inv_n, ld, (const double *)nullptr, 0);
    int nbody = 0;   // iterations launched so far = the iteration index the device is at while it has not converged
    auto body = [&]() -> int {
        NLG_TRY(apply(s));
        const bool prof_cg = prof_want(ctx, P_CGVEC);
        if (prof_cg) prof_begin(ctx, P_CGVEC);
        if (!P.pw_part)
            launch_nf(nf, k_cg_pw<1>, k_cg_pw<2>, k_cg_pw<3>, lgrid(g, nl), st, (const double *)s, P.n, cp, cw, P.ipw, partial, ld);
        NLG_TRY(reduce_post(rd_pw, 2, 1, 1));
        if (!P.rr_part) {
            ProfScope pu(ctx, P_CGUPDATE);   // the largest single kernel of a step by time: its own class inside cg_vec (bench.py quotes its roofline)
            launch_nf(nf, k_cg_update<1>, k_cg_update<2>, k_cg_update<3>, lgrid(g, nl), st, (const double *)s, P.n, x, r, z, cp, cw,
                      pc, P.ipw, P.nw, partial, ld, (int)(P.hist.ph > 0), P.pc_mb);
        }
        ++nbody;
        if (prof_cg) prof_end(ctx, P_CGVEC);
        if (P.precond) {
            NLG_TRY(P.precond(s + S_DONE, P.r[0], P.z[0], &xc));
            if (!P.rz_part)
                launch_nf(nf, k_cg_rz<1>, k_cg_rz<2>, k_cg_rz<3>, lgrid(g, nl), st, (const double *)s, 1, P.n, cr, cz, P.ipw, xc, P.npe, partial, ld);
        }
        if (P.hist.ph > 0 && nbody % P.hist.ph == 0)   // the direction ring is full: its PH terms go into x before slot 0 is overwritten
            launch_nf(nf, k_x_flush<1>, k_x_flush<2>, k_x_flush<3>, lgrid(grid_for(P.n), nl), st, (const double *)s, P.n, x, P.hist, nbody - 1, ld);
        NLG_TRY(reduce_post(rd_rz_loop, 3, 1, 2));
        if (!P.fused_pupdate)
            launch_nf(nf, k_cg_pupdate<1>, k_cg_pupdate<2>, k_cg_pupdate<3>, lgrid(g, nl), st, (const double *)s, P.n, p, cz, xc, P.npe, ld);
        return 0;
    };
    int launched = 0;
    while (true) {
        int todo = launched == 0 ? P.chunk : P.chunk_next;
        if (launched + todo > P.maxit) todo = P.maxit - launched;
        for (int it = 0; it < todo; ++it) NLG_TRY(body());
        launched += todo;
        NLG_HIP(hipGetLastError());
        for (int v = 0; v < nl; ++v)
            NLG_HIP(hipMemcpyAsync(op->h_s + (size_t)v * S_N, s + (int64_t)v * ld, sizeof(double) * S_N, hipMemcpyDeviceToHost, st));
        NLG_HIP(hipStreamSynchronize(st));
        bool all_done = true;
        for (int v = 0; v < nl; ++v) all_done = all_done && op->h_s[(size_t)v * S_N + S_DONE] != 0.0;
        if (all_done || launched >= P.maxit) break;
    }
    for (int v = 0; v < nl; ++v) {
        const double *h = op->h_s + (size_t)v * S_N;
        if (!std::isfinite(h[S_RN2])) {
            set_error("PCG diverged (residual is not finite) after %d iterations (lane %d of %d)", (int)h[S_ITERS], v, nl);
            return 1;
        }
        iters_out[v] = (int)h[S_ITERS];
    }
    return 0;
}

// The lanes of a time step: ops[0] is the operator itself, ops[1..] its lanes (all bound to one slab); nl = 1 is the
// single-vector path.  Every phase launches once for all lanes and keeps the host-side bookkeeping per lane.
struct Lanes {
    nlg_linop *const *ops;
    int nl;
    nlg_linop *op() const { return ops[0]; }
    int64_t ld() const { return nl > 1 ? ops[0]->slab_ld : 0; }
};

// The velocity solve in three pieces: problem set-up, one operator application, bookkeeping afterwards.
struct HelmSolve {
    CGProblem P;
    bool xp = false;
    double nu = 0.0, h2 = 0.0;
    double *pw_part = nullptr;
    mutable int it = 0;   // operator applications so far (= the slot of the direction ring the next one writes, modulo its depth)
};

int helm_problem(const Lanes &L, int order, double h2, HelmSolve &H) {
    nlg_linop *op = L.op();
    nlg_mesh *m = op->mesh;
    const int dim = m->dim;
    const auto &c = op->cfg;
    CGProblem &P = H.P;
    // x-planes-first layout for every vector of the iteration: the right-hand side is permuted on the way in (into gp,
    // free at this point), the solution on the way out (into rhs, which the caller reads as the increment)
    const bool xp = op->use_xp > 0;
    H.xp = xp;
    H.h2 = h2;
    const bool born_xp = xp && op->rhs_in_xp;   // adv_a wrote the right-hand side masked and slab-permuted already
    op->rhs_in_xp = false;
    if (xp && !born_xp) NLG_TRY(sem_to_xp(m, op->rhs, op->gp, dim, L.nl, L.ld(), m->d_mask));   // ... and masked on the way
    P.nf = dim;
    P.n = m->lvn;
    P.x = op->x;
    P.r = (xp && !born_xp) ? op->gp : op->rhs;
    P.z = op->z;
    P.p = op->pv;
    P.w = op->w;
    P.pc = xp ? op->pcv_xp[order] : op->pcv[order];
    if (op->maskb_v) {   // 1 / diag and the mask bytes instead of the three masked arrays (same values)
        P.pc = op->pci_l[order];
        P.pc_mb = op->maskb_v;
    }
    P.ipw = xp ? m->d_vmult_xp : m->d_vmult;
    P.nw = xp ? op->nwv_xp : op->nwv;
    P.tol2 = c.vtol * c.vtol;
    P.use_tol = c.fixed_iters_v > 0 ? 0 : 1;
    P.maxit = c.fixed_iters_v > 0 ? c.fixed_iters_v : c.maxit_v;
    P.s = op->d_s;
    P.inv_n = 0.0;
    P.nl = L.nl;
    P.ld = L.ld();
    // the iteration count barely changes from one time step to the next: launch exactly the previous count, then look at
    // the flags every second iteration; launches issued after convergence are gated on the device but still cost a launch
    P.chunk = 2;
    for (int v = 0; v < L.nl; ++v) {
        const nlg_linop *ln = L.ops[v];
        const int pred = (ln->istep < (int)ln->vit_hist.size() && ln->vit_hist[ln->istep] > 0) ? ln->vit_hist[ln->istep] : ln->last_viters;
        P.chunk = std::max(P.chunk, std::min(pred, 64));
    }
    H.nu = 1.0 / c.re;
    // w = QQ^T (nu A + h2 B) p.  The Dirichlet mask is not applied to w: p is masked (z = pc r with pc = mask/diag), so
    // (p, w) does not see the masked entries, and k_cg_update zeroes the residual where pc == 0.
    // (p, w) = sum over the local dofs of p . w_local (p is continuous), summed inside the operator kernel, which also
    // performs p <- z + beta p while it loads p.
    H.pw_part = op->d_part;
    P.pw_part = H.pw_part;
    P.pw_n = sem_axhelm_blocks(m, dim);
    P.pw_sum = false;
    P.fused_pupdate = true;
    if (op->ph > 0) {
        for (int q = 0; q < dim; ++q) P.hist.p0[q] = op->phist + (int64_t)q * m->lvs;
        P.hist.stride = (int64_t)dim * m->lvs;
        P.hist.ph = op->ph;
    }
    return 0;
}

int helm_apply(const Lanes &L, const HelmSolve &H) {
    nlg_linop *op = L.op();
    nlg_mesh *m = op->mesh;
    const int dim = m->dim;
    if (H.P.hist.ph > 0) {
        // direction ring: read direction it - 1, store direction it one slot further (the first application multiplies slot ph - 1 by
        // beta = 0: the ring is zeroed with the slab and holds finite values ever after)
        const int ph = H.P.hist.ph, so = H.it % ph, si = (H.it + ph - 1) % ph;
        ++H.it;
        double *pin[3] = {nullptr, nullptr, nullptr};
        for (int q = 0; q < dim; ++q) pin[q] = op->phist + si * H.P.hist.stride + (int64_t)q * m->lvs;
        NLG_TRY(sem_axhelm(m, pin, op->w, dim, H.nu, H.h2, H.pw_part, op->z, op->d_s + S_BETA, op->d_s + S_DONE, H.xp, L.nl, L.ld(),
                           (so - si) * H.P.hist.stride));
    } else {
        NLG_TRY(sem_axhelm(m, op->pv, op->w, dim, H.nu, H.h2, H.pw_part, op->z, op->d_s + S_BETA, op->d_s + S_DONE, H.xp, L.nl, L.ld()));
    }
    NLG_TRY(sem_gs(m, op->w, dim, op->d_s + S_DONE, H.xp ? LAYOUT_XP : LAYOUT_NAT, L.nl, L.ld(), L.ld()));
    return 0;
}

int helm_finish(const Lanes &L, const HelmSolve &H, const int *iters) {
    nlg_linop *op = L.op();
    (void)op;   // (slab-permuted solve: the increment stays in that layout, adv_b reads it through the slot table)
    for (int v = 0; v < L.nl; ++v) {
        nlg_linop *ln = L.ops[v];
        ln->st_viters += iters[v];
        ln->last_viters = iters[v];
        if ((int)ln->vit_hist.size() <= ln->istep) ln->vit_hist.resize(ln->istep + 1, 0);
        ln->vit_hist[ln->istep] = iters[v];
    }
    return 0;
}

int helm_solve(const Lanes &L, int order, double h2) {
    HelmSolve H;
    NLG_TRY(helm_problem(L, order, h2, H));
    if (L.op()->use_sr) {
        nlg_linop *op = L.op();
        nlg_mesh *m = op->mesh;
        H.P.sd = op->cgs;
        H.P.apply_plain = [&, op, m]() -> int {   // w = QQ^T (nu A + h2 B) z and the first-stage sums of (z, w): no direction update inside
            NLG_TRY(sem_axhelm(m, op->z, op->w, m->dim, H.nu, H.h2, H.pw_part, nullptr, nullptr, op->d_s + S_DONE, H.xp, L.nl, L.ld()));
            return sem_gs(m, op->w, m->dim, op->d_s + S_DONE, H.xp ? LAYOUT_XP : LAYOUT_NAT, L.nl, L.ld(), L.ld());
        };
    }
    auto apply = [&](double *) -> int { return helm_apply(L, H); };
    int iters[kMaxLanes] = {};
    NLG_TRY(run_pcg(L.op(), H.P, apply, iters));
    return helm_finish(L, H, iters);
}

// One scalar (temperature) step of the Boussinesq coupling, see oracle/lns.py advance (ifheat branch):
//   rhocp (b0 theta^{n+1} - sum bd_j theta^{n-j}) / dt = -rhocp EXT[(U.grad) theta + (u.grad) Theta] + conductivity lap theta^{n+1}
// in residual form, Jacobi-PCG; the new level ends in tbuf[0].
int heat_step(const Lanes &L, int k, double b0) {
    nlg_linop *op = L.op();
    nlg_mesh *m = op->mesh;
    hipStream_t st = m->ctx->stream;
    const auto &c = op->cfg;
    const int nl = L.nl;
    const int64_t ld = L.ld();
    const double dt = op->dt, rc = c.rhocp;
    // explicit term into the oldest buffer, then rotate.  The transport term is one launch per lane (base-flow data shared); everything
    // after it -- right-hand side, operator, gather-scatter, the whole PCG -- is ONE launch for all lanes (gridDim.y), as in the velocity solve
    const int adj = (op->adjoint && !op->nonlinear) ? 1 : 0;
    for (int v = 0; v < nl; ++v) {
        nlg_linop *ln = L.ops[v];
        NLG_TRY(sem_conv_scalar_apply(m, op->Ur, op->GT, ln->ubuf[0], ln->tbuf[0], ln->ftbuf[2], adj));
        if (adj) {
            // adjoint temperature equation: rhocp theta+_t = rhocp (U.grad) theta+ + conductivity lap theta+ + rhocp b . u+
            // (stored term N_t = -conv(U, theta+) - bm1 b . u+ ; oracle/lns.py advance, adjoint branch)
            for (int i = 0; i < m->dim; ++i)
                if (c.buoy[i] != 0.0)
                    NLG_LAUNCH(k_mul3_acc, dim3(grid_for(m->lvn)), dim3(NT), 0, st, m->lvn, ln->ftbuf[2], (const double *)m->d_bm1,
                                       (const double *)ln->ubuf[0][i], -c.buoy[i]);
        }
        double *t = ln->ftbuf[2];
        ln->ftbuf[2] = ln->ftbuf[1];
        ln->ftbuf[1] = ln->ftbuf[0];
        ln->ftbuf[0] = t;
    }
    Hist h;
    h.k = k;
    for (int j = 0; j < 3; ++j) {
        h.ab[j] = -(op->nonlinear ? 0.5 : 1.0) * rc * EXT_C[k][j];   // nonlinear: (u.grad)theta = 1/2 [(U.grad)theta + (u.grad)Theta] at U = u, Theta = theta
        h.bd[j] = BDF_C[k][j];
        for (int q = 0; q < 3; ++q) {
            h.f[j][q] = q == 0 ? op->ftbuf[j] : nullptr;
            h.u[j][q] = q == 0 ? op->tbuf[j] : nullptr;
        }
    }
    F3 rhs = {{op->trhs, nullptr, nullptr}};
    NLG_LAUNCH(k_rhs<1>, lgrid(grid_for(m->lvn), nl), dim3(NT), 0, st, m->lvn, h, (const double *)m->d_bm1, rc / dt, rhs, ld);
    const double h1 = c.conductivity, h2 = rc * b0 / dt;
    double *tin[1] = {op->tbuf[0]}, *tw[1] = {op->tw}, *trhs[1] = {op->trhs};
    NLG_TRY(sem_axhelm(m, tin, tw, 1, h1, h2, nullptr, nullptr, nullptr, nullptr, false, nl, ld));
    {
        CF3 a = {{op->trhs, nullptr, nullptr}}, b = {{op->tw, nullptr, nullptr}}, none = {{nullptr, nullptr, nullptr}};
        NLG_LAUNCH(k_lin3<1>, lgrid(grid_for(m->lvn), nl), dim3(NT), 0, st, m->lvn, rhs, a, b, -1.0, none, 0.0, ld);
    }
    NLG_TRY(sem_gs(m, trhs, 1, nullptr, LAYOUT_NAT, nl, ld, 0));
    {
        CF3 mk = {{m->d_tmask, nullptr, nullptr}};
        NLG_LAUNCH(k_colmul_gated<1>, lgrid(grid_for(m->lvn), nl), dim3(NT), 0, st, (const double *)nullptr, rhs, mk, m->lvn, ld);
    }
    // Jacobi-PCG, one field; the operator kernel sums (p, w) and updates p itself
    double *x[1] = {op->tx}, *z[1] = {op->tz}, *p[1] = {op->tpv}, *pc[1] = {op->pct[k]};
    CGProblem P;
    P.nf = 1;
    P.n = m->lvn;
    P.x = x;
    P.r = trhs;
    P.z = z;
    P.p = p;
    P.w = tw;
    P.pc = pc;
    P.ipw = m->d_vmult;
    P.nw = op->nwv;
    P.tol2 = c.vtol * c.vtol;
    P.use_tol = c.fixed_iters_v > 0 ? 0 : 1;
    P.maxit = c.fixed_iters_v > 0 ? c.fixed_iters_v : c.maxit_v;
    P.s = op->d_s;
    P.inv_n = 0.0;
    P.nl = nl;
    P.ld = ld;
    P.chunk = 2;
    for (int v = 0; v < nl; ++v) P.chunk = std::max(P.chunk, std::min(L.ops[v]->last_titers, 64));
    P.pw_part = op->d_part;
    P.pw_n = sem_axhelm_blocks(m, 1);
    P.pw_sum = false;
    P.fused_pupdate = true;
    // deferred solution update as in the velocity solve (helm_apply): the scalar's directions use the velocity solve's ring, one field wide
    // (the two solves never overlap: each ring is consumed right after its solve)
    const bool defer = op->ph > 0 && !op->use_sr;
    if (defer) {
        P.hist.p0[0] = op->phist;
        P.hist.stride = m->lvs;
        P.hist.ph = op->ph;
    }
    int napp = 0;
    auto apply = [&](double *) -> int {
        if (defer) {
            const int ph = op->ph, so = napp % ph, si = (napp + ph - 1) % ph;
            ++napp;
            double *pin[1] = {op->phist + (int64_t)si * m->lvs};
            NLG_TRY(sem_axhelm(m, pin, tw, 1, h1, h2, op->d_part, z, op->d_s + S_BETA, op->d_s + S_DONE, false, nl, ld, (so - si) * (int64_t)m->lvs));
        } else {
            NLG_TRY(sem_axhelm(m, p, tw, 1, h1, h2, op->d_part, z, op->d_s + S_BETA, op->d_s + S_DONE, false, nl, ld));
        }
        NLG_TRY(sem_gs(m, tw, 1, op->d_s + S_DONE, LAYOUT_NAT, nl, ld, ld));
        return 0;
    };
    double *sdt[1] = {op->tcgs};
    if (op->use_sr) {
        P.sd = sdt;
        P.apply_plain = [&]() -> int {
            NLG_TRY(sem_axhelm(m, z, tw, 1, h1, h2, op->d_part, nullptr, nullptr, op->d_s + S_DONE, false, nl, ld));
            return sem_gs(m, tw, 1, op->d_s + S_DONE, LAYOUT_NAT, nl, ld, ld);
        };
    }
    int iters[kMaxLanes] = {};
    NLG_TRY(run_pcg(op, P, apply, iters));
    for (int v = 0; v < nl; ++v) {
        L.ops[v]->st_titers += iters[v];
        L.ops[v]->last_titers = iters[v];
    }
    // theta^{n+1} = theta^n + x into the oldest level, then rotate: new -> current
    {
        F3 y = {{op->tbuf[2], nullptr, nullptr}};
        CF3 a = {{op->tbuf[0], nullptr, nullptr}}, b = {{op->tx, nullptr, nullptr}}, none = {{nullptr, nullptr, nullptr}};
        if (defer)
            NLG_LAUNCH(k_add_hist<1>, lgrid(grid_for(m->lvn), nl), dim3(NT), 0, st, (const double *)op->d_s, m->lvn, 1, (const int *)nullptr, y, a, b, P.hist, ld);
        else
            NLG_LAUNCH(k_lin3<1>, lgrid(grid_for(m->lvn), nl), dim3(NT), 0, st, m->lvn, y, a, b, 1.0, none, 0.0, ld);
        for (int v = 0; v < nl; ++v) {
            nlg_linop *ln = L.ops[v];
            double *t = ln->tbuf[2];
            ln->tbuf[2] = ln->tbuf[1];
            ln->tbuf[1] = ln->tbuf[0];
            ln->tbuf[0] = t;
        }
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

// The pressure solve in three pieces (see HelmSolve): set-up incl. the residual projection onto the previous increments,
// one application of E, and the update of the projection space afterwards.
struct PresSolve {
    CGProblem P;
    double *x[1], *r[1], *z[1], *p[1], *w[1], *pc[1], *nopc[1];
    double *pw_part = nullptr;
    bool proj = false;
    int nold = 0;
    mutable int it = 0;   // operator applications so far (slot of the direction ring, modulo its depth)
};

int pres_problem(const Lanes &L, double scale, PresSolve &Q) {
    nlg_linop *op = L.op();
    nlg_mesh *m = op->mesh;
    const auto &c = op->cfg;
    const int nl = L.nl;
    const int64_t ld = L.ld();
    Q.x[0] = op->pr_x, Q.r[0] = op->pr_r, Q.z[0] = op->pr_z, Q.p[0] = op->pr_p, Q.w[0] = op->pr_w, Q.pc[0] = op->pce, Q.nopc[0] = nullptr;
    CGProblem &P = Q.P;
    P.nf = 1;
    P.n = m->lpn;
    P.x = Q.x;
    P.r = Q.r;
    P.z = Q.z;
    P.p = Q.p;
    P.w = Q.w;
    P.pc = Q.pc;
    P.ipw = nullptr;
    P.nw = op->nwp;
    P.tol2 = (c.ptol / scale) * (c.ptol / scale);
    P.use_tol = c.fixed_iters_p > 0 ? 0 : 1;
    P.maxit = c.fixed_iters_p > 0 ? c.fixed_iters_p : c.maxit_p;
    P.s = op->d_s + S_N;
    P.inv_n = m->has_outflow ? 0.0 : 1.0 / (double)m->lpn_global;
    P.nl = nl;
    P.ld = ld;
    if (c.pprecond == 0 || c.pprecond == 2) {   // two-level Schwarz (pprec.hip): 0 = with overlap where available, 2 = without; 1 = Jacobi on diag(E), as in the oracle
        const bool overlap = c.pprecond == 0 && m->pprec.overlap;
        P.pc = Q.nopc;
        P.npe = m->np2;
        // r.z / z sums from the last kernel of the preconditioner: 3-D always, 2-D with the overlapping variant
        double *rzp = (m->dim == 3 || overlap) ? op->d_part + 2 * m->E : nullptr;
        // the PCG update of an iteration rides in the preconditioner's first kernel (which reads r anyway)
        nlg_pcg_upd upd;
        upd.alpha = op->d_s + S_N + S_ALPHA;
        upd.wmean = op->d_s + S_N + S_WMEAN;
        upd.x = op->pr_x;
        upd.p = op->pr_p;
        {
            // deferred solution update (as in the velocity solve, k_cg_update): the gradient kernel's fused direction update stores direction
            // i into slot i mod php of a ring, the update kernel of the preconditioner streams neither x nor p, pres_solve assembles x
            static const bool fuse = !(getenv("NLG_FUSE_PPUPDATE") && atoi(getenv("NLG_FUSE_PPUPDATE")) == 0);
            if (op->php > 0 && fuse && sem_opgradt_fuses_pupdate(m)) {
                upd.x = nullptr;
                upd.p = nullptr;
                P.hist.p0[0] = op->prh;
                P.hist.stride = m->lps;
                P.hist.ph = op->php;
            }
        }
        upd.w = op->pr_w;
        upd.nw = op->nwp;
        upd.rr_part = op->d_part + 2 * m->E + 2 * ((m->E + 3) / 4);
        P.rr_part = upd.rr_part;
        P.rr_n = (int)((m->E + 3) / 4);
        P.precond = [m, rzp, overlap, upd, nl, ld](const double *flag, const double *rr, double *zz, const double **xc) -> int {
            // one stream: a fork/join through events costs more than it hides (measured: 98 vs 84 us per apply)
            nlg_ctx *c = m->ctx;
            ProfScope ps(c, P_PPREC);
            const double *coarse = nullptr;
            NLG_TRY(pprec_coarse(m, c->stream, flag, rr, &coarse, overlap, flag ? &upd : nullptr, nl, ld));   // flag == null: the initial residual
            NLG_TRY(pprec_fine(m, c->stream, flag, rr, coarse, zz, rzp, overlap, nl, ld));   // z = local solves + prolonged coarse part
            *xc = nullptr;
            return 0;
        };
    }
    P.chunk = 2;
    for (int v = 0; v < nl; ++v) {
        const nlg_linop *ln = L.ops[v];
        const int pred = (ln->istep < (int)ln->pit_hist.size() && ln->pit_hist[ln->istep] > 0) ? ln->pit_hist[ln->istep] : ln->last_piters;
        P.chunk = std::max(P.chunk, std::min(pred, 96));
    }
    // fused first-stage sums (rank-local; the all-reduce follows the second stage): p.w and sum w from the divergence kernel,
    // r.z and sum z from the preconditioner's last kernel
    Q.pw_part = op->d_part;
    P.pw_part = Q.pw_part;
    P.pw_n = sem_opdiv_blocks(m);
    {
        static const bool fuse = !(getenv("NLG_FUSE_PPUPDATE") && atoi(getenv("NLG_FUSE_PPUPDATE")) == 0);
        P.fused_pupdate = fuse && sem_opgradt_fuses_pupdate(m);
    }
    if (P.precond && (m->dim == 3 || (c.pprecond == 0 && m->pprec.overlap))) {
        P.rz_part = op->d_part + 2 * m->E;
        P.rz_n = m->dim == 3 ? (int)((m->E + 3) / 4) : (int)((m->E * m->np2 + NT - 1) / NT);
    }
    // ---- residual projection (c.pproj): start from the A-orthogonal projection of the solution onto the span of the
    // previous increments of this matvec; the PCG then solves for the remainder
    hipStream_t st = m->ctx->stream;
    // (not in the fixed-iteration parity mode: there the iteration is the oracle's, run on the full right-hand side)
    Q.proj = c.pproj != 0 && op->prX != nullptr && c.fixed_iters_p <= 0;
    Q.nold = op->nproj;
    if (Q.proj && Q.nold > 0) {
        double *alpha = op->d_pc, *ppart = op->d_pc + 4 * PROJ_L;
        const int gp = red_grid(m->lpn);
        ProfScope ps(m->ctx, P_VECOPS);
        NLG_LAUNCH(k_proj_dots, lgrid(gp, nl), dim3(NT), 0, st, m->lpn, (const double *)op->prX, m->lps, Q.nold, (const double *)op->pr_r, ppart, ld);
        NLG_LAUNCH(k_proj_reduce, lgrid(1, nl), dim3(NT), 0, st, (const double *)ppart, gp, Q.nold, alpha, ld);
        for (int v = 0; v < nl; ++v) NLG_TRY(allreduce_sum(m->ctx, alpha + v * ld, Q.nold));                       // alpha = X^T b
        NLG_LAUNCH(k_proj_comb, lgrid(grid_for(m->lpn), nl), dim3(NT), 0, st, m->lpn, op->pr_r, (const double *)op->prB, m->lps,
                   Q.nold, (const double *)alpha, -1.0, ld);               // b <- b - B alpha
    }
    if (Q.proj)   // the right-hand side of the PCG, kept for the update of the projection space: A d = b - r_final
        NLG_HIP(hipMemcpy2DAsync(op->pr_b, sizeof(double) * (size_t)std::max<int64_t>(ld, m->lpn), op->pr_r, sizeof(double) * (size_t)std::max<int64_t>(ld, m->lpn),
                                 sizeof(double) * (size_t)m->lpn, (size_t)nl, hipMemcpyDeviceToDevice, st));
    return 0;
}

// gated: launches past convergence (the host only looks at the flags once per chunk) return at once
int pres_apply(const Lanes &L, const PresSolve &Q) {
    nlg_linop *op = L.op();
    // direction ring (deferred solution update): read direction it - 1, the fused update stores direction it one slot further
    double *p_in = op->pr_p, *p_out = op->pr_p;
    if (Q.P.hist.ph > 0) {
        const int ph = Q.P.hist.ph, so = Q.it % ph, si = (Q.it + ph - 1) % ph;
        ++Q.it;
        p_in = op->prh + (int64_t)si * Q.P.hist.stride;
        p_out = op->prh + (int64_t)so * Q.P.hist.stride;
    }
    if (L.nl == 1) {
        if (Q.P.fused_pupdate) {   // p <- (z - zmean) + beta p while the gradient kernel loads p
            nlg_pupd u;
            u.z = op->pr_z, u.beta = op->d_s + S_N + S_BETA, u.zmean = op->d_s + S_N + S_ZMEAN, u.p = p_out;
            return sem_cdabdtp(op->mesh, p_in, op->pr_w, Q.pw_part, op->d_s + S_N + S_DONE, &u);
        }
        return sem_cdabdtp(op->mesh, op->pr_p, op->pr_w, Q.pw_part, op->d_s + S_N + S_DONE);
    }
    // E p for all lanes: the gradient and divergence kernels take the lanes in one launch each, so does the gather-scatter
    const double *pp[kMaxLanes], *gg[kMaxLanes];
    double *ww[kMaxLanes], *pw[kMaxLanes];
    nlg_pupd pu[kMaxLanes];
    const int64_t ld = L.ld();
    for (int v = 0; v < L.nl; ++v) {
        pp[v] = p_in + v * ld, ww[v] = op->pr_w + v * ld, pw[v] = Q.pw_part + v * ld, gg[v] = op->d_s + S_N + S_DONE + v * ld;
        if (Q.P.fused_pupdate)
            pu[v].z = op->pr_z + v * ld, pu[v].beta = op->d_s + S_N + S_BETA + v * ld, pu[v].zmean = op->d_s + S_N + S_ZMEAN + v * ld, pu[v].p = p_out + v * ld;
    }
    return sem_cdabdtp_lanes(op->mesh, L.nl, pp, ww, pw, gg, Q.P.fused_pupdate ? pu : nullptr);
}

int pres_finish(const Lanes &L, const PresSolve &Q, const int *iters) {
    nlg_linop *op = L.op();
    nlg_mesh *m = op->mesh;
    hipStream_t st = m->ctx->stream;
    const int nl = L.nl;
    const int64_t ld = L.ld();
    for (int v = 0; v < nl; ++v) {
        nlg_linop *ln = L.ops[v];
        ln->st_piters += iters[v];
        ln->last_piters = iters[v];
        if ((int)ln->pit_hist.size() <= ln->istep) ln->pit_hist.resize(ln->istep + 1, 0);
        ln->pit_hist[ln->istep] = iters[v];
    }
    if (Q.proj) {
        double *alpha = op->d_pc, *beta = op->d_pc + PROJ_L, *nrm2 = op->d_pc + 2 * PROJ_L, *ppart = op->d_pc + 4 * PROJ_L;
        const int gp = red_grid(m->lpn);
        const int nold = Q.nold;
        auto dots = [&](const double *y, int nvec, const double *M, double *out) -> int {
            NLG_LAUNCH(k_proj_dots, lgrid(gp, nl), dim3(NT), 0, st, m->lpn, M, m->lps, nvec, y, ppart, ld);
            NLG_LAUNCH(k_proj_reduce, lgrid(1, nl), dim3(NT), 0, st, (const double *)ppart, gp, nvec, out, ld);
            for (int v = 0; v < nl; ++v) NLG_TRY(allreduce_sum(m->ctx, out + v * ld, nvec));
            return 0;
        };
        // new member from the increment d = pr_x: w = A d, A-orthogonalised against the old members, normalised.  A d is not
        // computed by another application of E (three launches, 4 % of a time step): the PCG started from r = b and updated
        // r <- r - alpha A p with x <- x + alpha p, so A d = b - r_final, mean-free like b and r (A = P E P)
        {
            F3 y = {{op->pr_w, nullptr, nullptr}};
            CF3 a = {{op->pr_b, nullptr, nullptr}}, b = {{op->pr_r, nullptr, nullptr}}, none = {{nullptr, nullptr, nullptr}};
            NLG_LAUNCH(k_lin3<1>, lgrid(grid_for(m->lpn), nl), dim3(NT), 0, st, m->lpn, y, a, b, -1.0, none, 0.0, ld);
        }
        ProfScope ps(m->ctx, P_VECOPS);
        NLG_HIP(hipMemcpy2DAsync(op->pr_z, sizeof(double) * (size_t)std::max<int64_t>(ld, m->lpn), op->pr_x, sizeof(double) * (size_t)std::max<int64_t>(ld, m->lpn),
                                 sizeof(double) * (size_t)m->lpn, (size_t)nl, hipMemcpyDeviceToDevice, st));   // v = d (all lanes)
        if (nold > 0) {
            NLG_TRY(dots(op->pr_w, nold, op->prX, beta));                    // beta = X^T A d
            NLG_LAUNCH(k_proj_comb, lgrid(grid_for(m->lpn), nl), dim3(NT), 0, st, m->lpn, op->pr_z, (const double *)op->prX, m->lps,
                       nold, (const double *)beta, -1.0, ld);
            NLG_LAUNCH(k_proj_comb, lgrid(grid_for(m->lpn), nl), dim3(NT), 0, st, m->lpn, op->pr_w, (const double *)op->prB, m->lps,
                       nold, (const double *)beta, -1.0, ld);
            // total solution: x = d + X alpha
            NLG_LAUNCH(k_proj_comb, lgrid(grid_for(m->lpn), nl), dim3(NT), 0, st, m->lpn, op->pr_x, (const double *)op->prX, m->lps,
                       nold, (const double *)alpha, 1.0, ld);
        }
        NLG_TRY(dots(op->pr_w, 1, op->pr_z, nrm2));                          // v^T A v
        const int slot = nold < PROJ_L ? nold : 0;                           // full: start over with the newest member
        NLG_LAUNCH(k_proj_store, lgrid(grid_for(m->lpn), nl), dim3(NT), 0, st, m->lpn, (const double *)op->pr_z, (const double *)op->pr_w,
                   (const double *)nrm2, op->prX + (size_t)slot * m->lps, op->prB + (size_t)slot * m->lps, ld);
        for (int v = 0; v < nl; ++v) L.ops[v]->nproj = nold < PROJ_L ? nold + 1 : 1;
        NLG_HIP(hipGetLastError());
    }
    return 0;
}

int pres_solve(const Lanes &L, double scale) {
    PresSolve Q;
    NLG_TRY(pres_problem(L, scale, Q));
    auto apply = [&](double *) -> int { return pres_apply(L, Q); };
    int iters[kMaxLanes] = {};
    NLG_TRY(run_pcg(L.op(), Q.P, apply, iters));
    if (Q.P.hist.ph > 0) {   // x = sum alpha_i p_i, in iteration order (k_add_hist without an addend)
        nlg_linop *op = L.op();
        nlg_mesh *m = op->mesh;
        F3 y = {{op->pr_x, nullptr, nullptr}};
        CF3 none = {{nullptr, nullptr, nullptr}}, x = {{op->pr_x, nullptr, nullptr}};
        NLG_LAUNCH(k_add_hist<1>, lgrid(grid_for(m->lpn), L.nl), dim3(NT), 0, m->ctx->stream, (const double *)Q.P.s, m->lpn, 1, (const int *)nullptr, y, none, x,
                   Q.P.hist, L.ld());
    }
    return pres_finish(L, Q, iters);
}

// rotate a set of three level pointers in every lane: the oldest becomes the newest
void rotate3(const Lanes &L, double *(nlg_linop::*buf)[3][3]) {
    for (int v = 0; v < L.nl; ++v) {
        double *(&b)[3][3] = L.ops[v]->*buf;
        double *t[3] = {b[2][0], b[2][1], b[2][2]};
        for (int c = 0; c < 3; ++c) {
            b[2][c] = b[1][c];
            b[1][c] = b[0][c];
            b[0][c] = t[c];
        }
    }
}

// one restated nek_advance step (perturbation mode), see oracle/lns.py ExptA.advance; three phases around the two solves
int adv_a(const Lanes &L) {
    nlg_linop *op = L.op();
    nlg_mesh *m = op->mesh;
    hipStream_t st = m->ctx->stream;
    const int dim = m->dim, nl = L.nl;
    const int64_t ld = L.ld();
    const double dt = op->dt, nu = 1.0 / op->cfg.re;
    for (int v = 0; v < nl; ++v) L.ops[v]->istep += 1;
    // gauge: keep the pressure mean-free (see oracle/lns.py advance)
    NLG_TRY(sem_ortho(m, op->p, nl, ld));
    const int k = std::min(op->istep, op->cfg.torder);
    const double b0 = BDF_B0[k];
    for (int v = 0; v < nl; ++v) L.ops[v]->adv_k = k, L.ops[v]->adv_b0 = b0, L.ops[v]->adv_h2 = b0 / dt;
    if (op->nonlinear) {   // the "base flow" is the current state (velocity, and temperature when coupled); one lane only
        NLG_TRY(sem_conv_setup(m, op->ubuf[0], op->Ur, op->GU));
        if (op->cfg.ifheat) NLG_TRY(sem_conv_scalar_setup(m, op->tbuf[0], op->GT));
    }
    if (op->cfg.ifheat) NLG_TRY(heat_step(L, k, b0));   // scalar first: the fluid sees the new temperature (Nek5000's order)
    // F = -N(u): written into the oldest forcing buffer, then the buffers rotate
    double **Fnew = op->fbuf[2];
    if (nl == 1) {
        NLG_TRY(sem_conv_apply(m, op->Ur, op->GU, op->ubuf[0], Fnew, op->nonlinear ? 0 : op->adjoint));
    } else {   // convective terms of all lanes against the shared base-flow fields: one launch
        double *const *ul[kMaxLanes], *const *ol[kMaxLanes];
        for (int v = 0; v < nl; ++v) ul[v] = L.ops[v]->ubuf[0], ol[v] = L.ops[v]->fbuf[2];
        NLG_TRY(sem_conv_apply_lanes(m, op->Ur, op->GU, nl, ul, ol, op->adjoint));
    }
    if (op->cfg.ifheat && op->adjoint && !op->nonlinear) {
        // adjoint momentum equation: - theta+ grad Theta with the new theta+ (stored F is +N: add the weak term); lane by lane
        for (int v = 0; v < nl; ++v) NLG_TRY(sem_scalar_grad_apply(m, op->GT, L.ops[v]->tbuf[0], L.ops[v]->fbuf[2], 1.0));
    } else if (op->cfg.ifheat) {
        const double bs = op->nonlinear ? 2.0 : 1.0;   // the nonlinear step halves the whole stored term (F holds 2 N there)
        for (int v = 0; v < nl; ++v)
            launch_nf(dim, k_buoyancy<1>, k_buoyancy<2>, k_buoyancy<3>, dim3(grid_for(m->lvn)), st, m->lvn, f3(L.ops[v]->fbuf[2], dim),
                      (const double *)m->d_bm1, (const double *)L.ops[v]->tbuf[0], bs * op->cfg.buoy[0], bs * op->cfg.buoy[1], bs * op->cfg.buoy[2]);
    }
    if (op->force_re) {
        // forcing of this step: evaluated at the time level the step starts from, (istep - 1) dt, like the explicit terms
        // (resolvent.f90:97-103: alpha = exp(sign i omega time) before nek_advance)
        const double ph = op->force_sign * op->force_omega * (op->istep - 1) * dt;
        CF3 fr = {{op->force_re->vel(0), op->force_re->vel(1), dim == 3 ? op->force_re->vel(2) : nullptr}};
        CF3 fi = {{nullptr, nullptr, nullptr}};
        if (op->force_im) fi = CF3{{op->force_im->vel(0), op->force_im->vel(1), dim == 3 ? op->force_im->vel(2) : nullptr}};
        launch_nf(dim, k_add_force<1>, k_add_force<2>, k_add_force<3>, dim3(grid_for(m->lvn)), st, m->lvn, f3(Fnew, dim),
                  (const double *)m->d_bm1, fr, fi, std::cos(ph), -std::sin(ph));
    }
    rotate3(L, &nlg_linop::fbuf);
    Hist h;
    h.k = k;
    for (int j = 0; j < 3; ++j) {
        h.ab[j] = -(op->nonlinear ? 0.5 : 1.0) * EXT_C[k][j];   // F = -N ; nonlinear: (u.grad)u = 1/2 [(U.grad)u + (u.grad)U] at U = u
        h.bd[j] = BDF_C[k][j];
        for (int c = 0; c < 3; ++c) {
            h.f[j][c] = op->fbuf[j][c];
            h.u[j][c] = op->ubuf[j][c];
        }
    }
    // residual form: res = mask QQ^T (rhs + D^T p - H u); the history sums and the two operator terms in ONE pass over the fields
    const double h2 = b0 / dt;
    if (nl == 1) {
        NLG_TRY(sem_opgradt(m, op->p, op->gp));
    } else {
        const double *pp[kMaxLanes];
        double *const *gl[kMaxLanes];
        for (int v = 0; v < nl; ++v) pp[v] = L.ops[v]->p, gl[v] = L.ops[v]->gp;
        NLG_TRY(sem_opgradt_lanes(m, nl, pp, gl, false, nullptr));
    }
    NLG_TRY(sem_axhelm(m, op->ubuf[0], op->w, dim, nu, h2, nullptr, nullptr, nullptr, nullptr, false, nl, ld));
    if (op->use_xp > 0) {
        // slab-permuted velocity solve: the right-hand side is born masked in that layout (no permutation pass in helm_problem, and
        // its gather-scatter moves the layout's 64-byte runs instead of the natural layout's single points)
        CF3 mk = {{m->d_mask[0], m->d_mask[1], m->d_mask[2]}};
        launch_nf(dim, k_rhs<1, true>, k_rhs<2, true>, k_rhs<3, true>, lgrid(grid_for(m->lvn), nl), st, m->lvn, h, (const double *)m->d_bm1, 1.0 / dt,
                  f3(op->rhs, dim), ld, cf3(op->gp, dim), cf3(op->w, dim), (const int *)m->d_slot_xp, m->np1, mk);
        NLG_TRY(sem_gs(m, op->rhs, dim, nullptr, LAYOUT_XP, nl, ld, 0));
        op->rhs_in_xp = true;
        return 0;
    }
    launch_nf(dim, k_rhs<1>, k_rhs<2>, k_rhs<3>, lgrid(grid_for(m->lvn), nl), st, m->lvn, h, (const double *)m->d_bm1, 1.0 / dt,
              f3(op->rhs, dim), ld, cf3(op->gp, dim), cf3(op->w, dim), (const int *)nullptr, 1, CF3{{nullptr, nullptr, nullptr}});
    NLG_TRY(sem_gs(m, op->rhs, dim, nullptr, LAYOUT_NAT, nl, ld, 0));
    if (op->use_xp <= 0) {   // (slab-permuted velocity solve: the mask is applied by the permutation of the right-hand side, helm_problem)
        CF3 mk = {{m->d_mask[0], m->d_mask[1], m->d_mask[2]}};
        launch_nf(dim, k_colmul_gated<1>, k_colmul_gated<2>, k_colmul_gated<3>, lgrid(grid_for(m->lvn), nl), st,
                  (const double *)nullptr, f3(op->rhs, dim), mk, m->lvn, ld);
    }
    return 0;
}

int adv_b(const Lanes &L) {
    nlg_linop *op = L.op();
    nlg_mesh *m = op->mesh;
    hipStream_t st = m->ctx->stream;
    const int dim = m->dim, nl = L.nl;
    const int64_t ld = L.ld();
    const double dt = op->dt, b0 = op->adv_b0;
    // uh = u + du -> into the oldest velocity buffer (slot 2), which becomes the new current after rotation
    double **unew = op->ubuf[2];
    if (op->ph > 0) {
        // deferred solution update of the velocity PCG: the increment is assembled from the direction ring here (k_add_hist)
        PHist Hh = {{nullptr, nullptr, nullptr}, (int64_t)dim * m->lvs, op->ph};
        for (int q = 0; q < dim; ++q) Hh.p0[q] = op->phist + (int64_t)q * m->lvs;
        launch_nf(dim, k_add_hist<1>, k_add_hist<2>, k_add_hist<3>, lgrid(grid_for(m->lvn), nl), st, (const double *)op->d_s, m->lvn, m->np1,
                  op->use_xp > 0 ? (const int *)m->d_slot_xp : (const int *)nullptr, f3(unew, dim), cf3(op->ubuf[0], dim), cf3(op->x, dim), Hh, ld);
    } else if (op->use_xp > 0) {
        launch_nf(dim, k_add_xp<1>, k_add_xp<2>, k_add_xp<3>, lgrid(grid_for(m->lvn), nl), st, m->lvn, m->np1, (const int *)m->d_slot_xp, f3(unew, dim),
                  cf3(op->ubuf[0], dim), cf3(op->x, dim), ld);
    } else {
        CF3 none = {{nullptr, nullptr, nullptr}};
        launch_nf(dim, k_lin3<1>, k_lin3<2>, k_lin3<3>, lgrid(grid_for(m->lvn), nl), st, m->lvn, f3(unew, dim), cf3(op->ubuf[0], dim),
                  cf3(op->x, dim), 1.0, none, 0.0, ld);
    }
    // pressure correction
    if (nl == 1) {
        NLG_TRY(sem_opdiv(m, unew, op->pr_r, -(b0 / dt)));
    } else {
        double *const *ul[kMaxLanes];
        double *ol[kMaxLanes], *none[kMaxLanes] = {};
        for (int v = 0; v < nl; ++v) ul[v] = L.ops[v]->ubuf[2], ol[v] = L.ops[v]->pr_r;
        NLG_TRY(sem_opdiv_lanes(m, nl, ul, ol, -(b0 / dt), nullptr, false, nullptr, none, nullptr));
    }
    NLG_TRY(sem_ortho(m, op->pr_r, nl, ld));
    return 0;
}

int adv_c(const Lanes &L) {
    nlg_linop *op = L.op();
    nlg_mesh *m = op->mesh;
    hipStream_t st = m->ctx->stream;
    const int dim = m->dim, nl = L.nl;
    const int64_t ld = L.ld();
    const double dt = op->dt, b0 = op->adv_b0;
    double **unew = op->ubuf[2];
    NLG_LAUNCH(k_axpy1, lgrid(grid_for(m->lpn), nl), dim3(NT), 0, st, m->lpn, op->p, (const double *)op->pr_x, 1.0, ld);
    // the gradient of the pressure increment in the face-grouped layout of the pressure operator's intermediates where that exists
    // (3-D): its gather-scatter then moves whole faces (50 us against 100 us in the natural layout at 10^4 elements), and the update
    // below reads it through the element-local slot table
    const bool fg = sem_opgradt_has_fg(m);
    if (nl == 1) {
        NLG_TRY(sem_opgradt(m, op->pr_x, op->gp, fg));
    } else {
        const double *pp[kMaxLanes];
        double *const *gl[kMaxLanes];
        for (int v = 0; v < nl; ++v) pp[v] = L.ops[v]->pr_x, gl[v] = L.ops[v]->gp;
        NLG_TRY(sem_opgradt_lanes(m, nl, pp, gl, fg, nullptr));
    }
    // u = uh + (dt / b0) mask binv QQ^T D^T dp: gather-scatter, then weights and update in one pass
    NLG_TRY(sem_gs(m, op->gp, dim, nullptr, fg ? LAYOUT_FG : LAYOUT_NAT, nl, ld, 0));
    {
        CF3 wt = {{m->d_mbinv[0], m->d_mbinv[1], m->d_mbinv[2]}};
        if (fg)
            launch_nf(dim, k_axpy_w<1, true>, k_axpy_w<2, true>, k_axpy_w<3, true>, lgrid(grid_for(m->lvn), nl), st, m->lvn, f3(unew, dim), cf3(op->gp, dim),
                      wt, dt / b0, ld, (const int *)m->d_slot_fg, m->np1);
        else
            launch_nf(dim, k_axpy_w<1>, k_axpy_w<2>, k_axpy_w<3>, lgrid(grid_for(m->lvn), nl), st, m->lvn, f3(unew, dim), cf3(op->gp, dim), wt, dt / b0,
                      ld, (const int *)nullptr, 1);
    }
    NLG_HIP(hipGetLastError());
    // rotate velocity history: new -> current, current -> lag1, lag1 -> lag2
    rotate3(L, &nlg_linop::ubuf);
    for (int v = 0; v < nl; ++v) L.ops[v]->st_steps += 1;
    return 0;
}

int advance(const Lanes &L) {
    NLG_TRY(adv_a(L));
    NLG_TRY(helm_solve(L, L.op()->adv_k, L.op()->adv_h2));
    NLG_TRY(adv_b(L));
    NLG_TRY(pres_solve(L, L.op()->dt / L.op()->adv_b0));
    return adv_c(L);
}
int advance(nlg_linop *op) {
    nlg_linop *one[1] = {op};
    return advance(Lanes{one, 1});
}

// integrator state of nl lanes <- 0: the rotating velocity / forcing (/ temperature) levels are the first buffers of a lane, in
// canonical order after the re-binding, so ONE strided memset clears them in every lane
int reset_state(nlg_linop *op, int nl) {
    nlg_mesh *m = op->mesh;
    lane_bind(op, op, 0);
    for (int v = 1; v < nl; ++v) lane_bind(op, op->lanes[v - 1], v);
    const int64_t nlev = (int64_t)(6 * m->dim + (op->cfg.ifheat ? 6 : 0)) * m->lvs;
    // one 1-D fill per lane: the 2-D fill of the runtime (hipMemset2DAsync over the slab pitch) ran at 0.2 TB/s -- 14.6 ms per block
    // step of four lanes, 8.5 % of it -- against 4.6 TB/s for the 1-D fill
    for (int v = 0; v < nl; ++v) NLG_HIP(hipMemsetAsync(op->slab + (size_t)v * (size_t)op->slab_ld, 0, sizeof(double) * (size_t)nlev, m->ctx->stream));
    return 0;
}

int load_state(nlg_linop *op, const nlg_vec *v, int irst) {
    nlg_mesh *m = op->mesh;
    hipStream_t st = m->ctx->stream;
    for (int c = 0; c < m->dim; ++c)
        NLG_HIP(hipMemcpyAsync(op->ubuf[0][c], v->vel(c, irst), sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToDevice, st));
    NLG_HIP(hipMemcpyAsync(op->p, v->pr(irst), sizeof(double) * (size_t)m->lpn, hipMemcpyDeviceToDevice, st));
    if (op->cfg.ifheat)
        NLG_HIP(hipMemcpyAsync(op->tbuf[0], v->theta(0, irst), sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToDevice, st));
    return 0;
}

int store_state(nlg_linop *op, nlg_vec *v, int irst) {
    nlg_mesh *m = op->mesh;
    hipStream_t st = m->ctx->stream;
    for (int c = 0; c < m->dim; ++c)
        NLG_HIP(hipMemcpyAsync(v->vel(c, irst), op->ubuf[0][c], sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToDevice, st));
    NLG_HIP(hipMemcpyAsync(v->pr(irst), op->p, sizeof(double) * (size_t)m->lpn, hipMemcpyDeviceToDevice, st));
    if (op->cfg.ifheat)
        NLG_HIP(hipMemcpyAsync(v->theta(0, irst), op->tbuf[0], sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToDevice, st));
    return 0;
}

// no-op unless nlg_linop_set_projection has been called.  `tab` holds the projection tables (the operator itself), `dat` the state that is
// projected: the operator again, or one of its lanes in a block step
int project_alpha(const nlg_linop *tab, nlg_linop *dat, int slot) {
    if (tab->proj_nlines == 0) return 0;
    nlg_mesh *m = tab->mesh;
    const unsigned grid = (unsigned)((tab->proj_nlines + NT / 64 - 1) / (NT / 64));
    F3 u = f3(dat->ubuf[slot], m->dim);
    if (tab->proj_gslot) {
        // several ranks: partial sums -> global slots -> all-reduce -> apply (the reference's planar_avg is a global
        // operation, exponential_propagator_proj.f90:146-169)
        hipStream_t st = m->ctx->stream;
        auto pass = [&](int nf, int64_t nl, const int *off, const int *idx, const int *gs, int64_t nglob, const double *wt,
                        const double *cv, const double *sv, const double *iden, F3 f) -> int {
            const unsigned g = (unsigned)((nl + NT / 64 - 1) / (NT / 64));
            const int64_t cnt = nglob * 2 * nf;
            NLG_HIP(hipMemsetAsync(tab->proj_glob, 0, sizeof(double) * (size_t)cnt, st));
            CF3 cf = {{f.p[0], f.p[1], f.p[2]}};
            if (nl > 0) {
                if (nf == 3)
                    NLG_LAUNCH(k_proj_sums<3>, dim3(g), dim3(NT), 0, st, nl, off, idx, gs, wt, cv, sv, cf, tab->proj_glob);
                else if (nf == 2)
                    NLG_LAUNCH(k_proj_sums<2>, dim3(g), dim3(NT), 0, st, nl, off, idx, gs, wt, cv, sv, cf, tab->proj_glob);
                else
                    NLG_LAUNCH(k_proj_sums<1>, dim3(g), dim3(NT), 0, st, nl, off, idx, gs, wt, cv, sv, cf, tab->proj_glob);
            }
            NLG_TRY(allreduce_sum(m->ctx, tab->proj_glob, (int)cnt));
            if (nl > 0) {
                if (nf == 3)
                    NLG_LAUNCH(k_proj_apply<3>, dim3(g), dim3(NT), 0, st, nl, off, idx, gs, cv, sv, iden, (const double *)tab->proj_glob, f);
                else if (nf == 2)
                    NLG_LAUNCH(k_proj_apply<2>, dim3(g), dim3(NT), 0, st, nl, off, idx, gs, cv, sv, iden, (const double *)tab->proj_glob, f);
                else
                    NLG_LAUNCH(k_proj_apply<1>, dim3(g), dim3(NT), 0, st, nl, off, idx, gs, cv, sv, iden, (const double *)tab->proj_glob, f);
            }
            return 0;
        };
        NLG_TRY(pass(m->dim, tab->proj_nlines, tab->proj_off, tab->proj_idx, tab->proj_gslot, tab->proj_nglob, m->d_bm1, tab->proj_cv, tab->proj_sv,
                     tab->proj_iden, u));
        if (tab->proj_gslot2 && slot == 0) {
            F3 pp = {{dat->p, nullptr, nullptr}};
            NLG_TRY(pass(1, tab->proj_nlines2, tab->proj_off2, tab->proj_idx2, tab->proj_gslot2, tab->proj_nglob2, m->d_bm2, tab->proj_cv2,
                         tab->proj_sv2, tab->proj_iden2, pp));
        }
        NLG_HIP(hipGetLastError());
        return 0;
    }
    if (m->dim == 3)
        NLG_LAUNCH(k_proj_alpha<3>, dim3(grid), dim3(NT), 0, m->ctx->stream, (int64_t)tab->proj_nlines, (const int *)tab->proj_off,
                           (const int *)tab->proj_idx, (const double *)m->d_bm1, (const double *)tab->proj_cv, (const double *)tab->proj_sv,
                           (const double *)tab->proj_iden, u);
    else
        NLG_LAUNCH(k_proj_alpha<2>, dim3(grid), dim3(NT), 0, m->ctx->stream, (int64_t)tab->proj_nlines, (const int *)tab->proj_off,
                           (const int *)tab->proj_idx, (const double *)m->d_bm1, (const double *)tab->proj_cv, (const double *)tab->proj_sv,
                           (const double *)tab->proj_iden, u);
    if (tab->proj_nlines2 > 0 && slot == 0) {
        // the pressure is part of the state the integrator starts from (lagged pressure of the correction scheme) but not
        // of the inner product: left unprojected it is a subspace the Arnoldi norm cannot see (observed: a spurious
        // |mu| = 1.41 for plane Poiseuille flow at alpha = 2 instead of 0.945)
        const unsigned grid2 = (unsigned)((tab->proj_nlines2 + NT / 64 - 1) / (NT / 64));
        F3 pp = {{dat->p, nullptr, nullptr}};
        NLG_LAUNCH(k_proj_alpha<1>, dim3(grid2), dim3(NT), 0, m->ctx->stream, (int64_t)tab->proj_nlines2, (const int *)tab->proj_off2,
                           (const int *)tab->proj_idx2, (const double *)m->d_bm2, (const double *)tab->proj_cv2, (const double *)tab->proj_sv2,
                           (const double *)tab->proj_iden2, pp);
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

int project_alpha(nlg_linop *op, int slot = 0) { return project_alpha(op, op, slot); }

int do_matvec(nlg_linop *op, const nlg_vec *vin, nlg_vec *vout, int adjoint) {
    NLG_CHECK(op && vin && vout, "exptA matvec: NULL argument");
    NLG_CHECK(op->inited, "exptA matvec: nlg_linop_init has not been called (reference: exptA%%init(), 1cyl.usr:20)");
    nlg_mesh *m = op->mesh;
    NLG_CHECK(vin->mesh == m && vout->mesh == m,
              "exptA matvec: vector on a different mesh (reference: type_error, exponential_propagator.f90:53-58)");
    NLG_CHECK(vin->nscal == (op->cfg.ifheat ? 1 : 0) && vout->nscal == vin->nscal,
              "exptA matvec: the vectors carry %d scalar(s), the operator expects %d (cfg.ifheat)", vin->nscal, op->cfg.ifheat ? 1 : 0);
    const bool nohist = op->cfg.no_history != 0;
    NLG_CHECK(nohist || (vin->lorder >= op->cfg.torder && vout->lorder >= op->cfg.torder),
              "exptA matvec: vector lorder %d < time order %d", vin->lorder, op->cfg.torder);
    NLG_CHECK(vin != vout, "exptA matvec: vec_in and vec_out must be distinct (intent(in) / intent(out))");
    const int nrst = nohist ? 0 : op->cfg.torder - 1;   // no_history: impulsive start, no history steps (include/neklab_gpu.h)
    NLG_TRY(reset_state(op, 1));
    op->istep = 0;
    op->adjoint = adjoint;
    // the projection space belongs to one matvec: the result must not depend on earlier calls.  (Keeping it across matvecs, as a
    // Nek5000 run does across time steps, was measured in round 4: 11.76 -> 11.40 pressure iterations per time step over 844 matvecs
    // of a real Arnoldi / Krylov-Schur run -- successive Krylov vectors are orthogonal, their pressure increments share little --
    // while bench.py, which re-applies the operator to the SAME column every step, would show 12.5 -> 6.5: an artefact, not adopted.)
    op->nproj = 0;
    NLG_TRY(load_state(op, vin, 0));
    NLG_TRY(project_alpha(op));                       // exptA_proj_matvec: initial condition, exponential_propagator_proj.f90:51
    for (int istep = 1; istep <= op->nsteps; ++istep) {
        NLG_TRY(advance(op));
        if (istep <= nrst && vin->nrst > 0) {
            NLG_TRY(load_state(op, vin, istep));   // get_rst, :129-142
            NLG_TRY(project_alpha(op));            // (projected operator: replayed states are projected as well, see below)
        }
    }
    // ... and the final state, :66.  The reference projects the initial condition and the final state only.  Here the
    // replayed history states and the lagged states of the multistep scheme are projected too: otherwise the extended map
    // (state, history) that the Arnoldi process iterates has a spurious unstable mode -- observed for plane Poiseuille flow
    // at alpha = 2, Re = 7500: a "converged" |mu| = 1.41 in front of the Orr-Sommerfeld pair |mu| = 0.9448; with the
    // consistent projection the leading pair is 0.94454.
    NLG_TRY(project_alpha(op));
    NLG_TRY(project_alpha(op, 1));
    NLG_TRY(project_alpha(op, 2));
    // vec_out is intent(out): default-initialised (history cleared, nrst = 0), then filled
    NLG_TRY(nlg_vec_zero(vout));
    NLG_TRY(store_state(op, vout, 0));
    for (int irst = 1; irst <= nrst; ++irst) {   // compute_rst, :109-127
        NLG_TRY(advance(op));
        NLG_TRY(store_state(op, vout, irst));
        vout->nrst = std::max(vout->nrst, irst);
    }
    op->st_matvecs += 1;
    return 0;
}

// =====================================================================================================================
// Multi-vector (block) propagator: s <= 4 perturbations advanced together, time step by time step (the reference advances
// several perturbations together through Nek5000's lpert / npert, src/neklab_nek_setup.f90:39-247, src/neklab_otd.f90:37-49).
// Every vector has a LANE: its own integrator state, PCG work vectors and device scalars, all lanes carved from one slab at a
// constant stride (struct nlg_linop); the base-flow data (fine-mesh fields of the convective term, preconditioners, weights)
// is shared.  The time step is the single-vector one (advance(Lanes)) with every kernel launched ONCE for all lanes
// (gridDim.y = lanes, or a lane loop inside the fused element kernels), own alpha / beta / convergence flag per lane -- each
// lane performs exactly the iteration of the single-vector path -- and, across ranks, ONE halo exchange / all-reduce per
// gather-scatter / reduction carrying all lanes.
// =====================================================================================================================
// lane v of the block stepper: lane 0 is the operator itself, lanes 1 .. 3 are created on first use
nlg_linop *lane_get(nlg_linop *op0, int v);
int lane_refresh(nlg_linop *op0, nlg_linop *ln);

// vec_out = state after the nsteps of one application started from `ic` (null: rest) under the time-harmonic body force
// Re[(f_re + i f_im) exp(i s omega t)], s = -1 for the adjoint equations: evaluate_rhs / evaluate_imaginary_part of the
// resolvent (src/linops/resolvent.f90:80-111, :133-166).  No restart-history replay, no history in the result.
int do_integrate_forced(nlg_linop *op, const nlg_vec *ic, const nlg_vec *f_re, const nlg_vec *f_im, double omega, int adjoint, nlg_vec *vout) {
    NLG_CHECK(op && f_re && vout, "integrate_forced: NULL argument");
    NLG_CHECK(op->inited, "integrate_forced: nlg_linop_init has not been called");
    nlg_mesh *m = op->mesh;
    NLG_CHECK(f_re->mesh == m && vout->mesh == m && (!ic || ic->mesh == m) && (!f_im || f_im->mesh == m), "integrate_forced: vector on a different mesh");
    NLG_CHECK(vout != f_re && vout != f_im && vout != ic, "integrate_forced: the output must be distinct from the inputs");
    NLG_CHECK(f_re->nscal == 0 && vout->nscal == 0, "integrate_forced: scalar (temperature) coupling is not built yet");
    hipStream_t st = m->ctx->stream;
    NLG_TRY(reset_state(op, 1));
    NLG_HIP(hipMemsetAsync(op->p, 0, sizeof(double) * (size_t)m->lps, st));
    op->istep = 0;
    op->adjoint = adjoint;
    op->nproj = 0;
    if (ic) NLG_TRY(load_state(op, ic, 0));
    op->force_re = f_re;
    op->force_im = f_im;
    op->force_omega = omega;
    op->force_sign = adjoint ? -1.0 : 1.0;
    int rc = 0;
    for (int istep = 1; istep <= op->nsteps && rc == 0; ++istep) rc = advance(op);
    op->force_re = op->force_im = nullptr;
    if (rc) return rc;
    NLG_TRY(nlg_vec_zero(vout));
    NLG_TRY(store_state(op, vout, 0));
    return 0;
}

// vec_out = Phi_tau(vec_in) - vec_in with the NONLINEAR integrator (reference: nonlinear_map, src/systems/fixed_point.f90:4-38):
// dt from the CFL number of vec_in itself, no restart-history replay, no history in the result
int do_nonlinear_map(nlg_linop *op, const nlg_vec *vin, nlg_vec *vout) {
    NLG_CHECK(op && vin && vout, "nonlinear_map: NULL argument");
    nlg_mesh *m = op->mesh;
    NLG_CHECK(vin->mesh == m && vout->mesh == m, "nonlinear_map: vector on a different mesh (reference: type_error, fixed_point.f90:31-36)");
    NLG_CHECK(vin->nscal == (op->cfg.ifheat ? 1 : 0) && vout->nscal == vin->nscal,
              "nonlinear_map: the vectors carry %d scalar(s), the operator expects %d (cfg.ifheat)", vin->nscal, op->cfg.ifheat ? 1 : 0);
    NLG_CHECK(vin != vout, "nonlinear_map: vec_in and vec_out must be distinct");
    hipStream_t st = m->ctx->stream;
    // "setup_nonlinear_solver(recompute_dt = .true.)": the time step follows the state that is integrated
    NLG_TRY(nlg_vec_copy(op->baseflow, vin));
    NLG_TRY(nlg_linop_init(op));
    NLG_TRY(reset_state(op, 1));
    op->istep = 0;
    op->adjoint = 0;
    op->nproj = 0;
    op->nonlinear = 1;
    int rc = load_state(op, vin, 0);
    for (int istep = 1; istep <= op->nsteps && rc == 0; ++istep) rc = advance(op);
    op->nonlinear = 0;
    if (rc) return rc;
    NLG_TRY(nlg_vec_zero(vout));
    NLG_TRY(store_state(op, vout, 0));
    NLG_TRY(nlg_vec_axpby(-1.0, vin, 1.0, vout));   // vec_out%sub(vec_in), fixed_point.f90:29
    // the base-flow dependent set-up now belongs to vec_in: a later linear matvec needs nlg_linop_set_baseflow
    return 0;
}

}  // namespace

extern "C" {

int nlg_exptA_config_default(nlg_exptA_config *c) {
    NLG_CHECK(c, "nlg_exptA_config_default: NULL");
    memset(c, 0, sizeof(*c));
    c->tau = 1.0;
    c->re = 100.0;
    c->cfl_limit = 0.5;
    c->vtol = 1e-9;
    c->ptol = 1e-7;
    c->dt = 0.0;
    c->torder = 3;
    c->maxit_v = 200;
    c->maxit_p = 2000;
    c->ifheat = 0;
    c->conductivity = 1.0;
    c->rhocp = 1.0;
    c->pproj = 1;   // residualProj = yes for the pressure, as in the reference's cylinder case (1cyl.par:23)
    c->no_history = 0;
    return 0;
}

int nlg_linop_create(nlg_mesh *mesh, const nlg_exptA_config *cfg, const nlg_vec *baseflow, nlg_linop **out) {
    NLG_CHECK(mesh && cfg && baseflow && out, "nlg_linop_create: NULL argument");
    NLG_CHECK(baseflow->mesh == mesh, "nlg_linop_create: baseflow lives on a different mesh");
    NLG_CHECK(cfg->torder >= 1 && cfg->torder <= 3, "nlg_linop_create: torder %d unsupported (1..3)", cfg->torder);
    NLG_CHECK(cfg->tau > 0.0 && cfg->re > 0.0, "nlg_linop_create: tau and re must be positive");
    NLG_CHECK(!cfg->ifheat || (baseflow->nscal >= 1 && cfg->conductivity > 0.0 && cfg->rhocp > 0.0),
              "nlg_linop_create: ifheat needs a base flow with its temperature (nscal >= 1) and positive conductivity / rhocp");
    nlg_linop *op = new nlg_linop();
    op->mesh = mesh;
    op->cfg = *cfg;
    NLG_TRY(nlg_vec_clone(baseflow, &op->baseflow));
    *out = op;
    return 0;
}

int nlg_linop_destroy(nlg_linop *op) {
    if (!op) return 0;
    hipDeviceSynchronize();
    for (nlg_linop *&ln : op->lanes) {
        if (ln) nlg_linop_destroy(ln);
        ln = nullptr;
    }
    if (op->is_lane) {   // a lane owns nothing: its work buffers are the owner's slab, the base-flow data is shared
        delete op;
        return 0;
    }
    auto fr = [](double *p) {
        if (p) hipFree(p);
    };
    fr(op->slab);        // every per-lane work buffer (lane_buffers)
    fr(op->d_red);
    for (int c = 0; c < 3; ++c) {
        fr(op->Ur[c]);
        for (int k = 0; k < 4; ++k) fr(op->pcv[k][c]);
        for (int k = 0; k < 4; ++k) fr(op->pcv_xp[k][c]);
    }
    for (int k = 0; k < 4; ++k) fr(op->pci[k]);
    if (op->maskb_v) hipFree(op->maskb_v);
    for (int q = 0; q < 9; ++q) fr(op->GU[q]);
    fr(op->pce);
    for (int q = 0; q < 3; ++q) fr(op->GT[q]);
    for (int k = 0; k < 4; ++k) fr(op->pct[k]);
    fr(op->proj_cv);
    fr(op->proj_sv);
    fr(op->proj_iden);
    fr(op->proj_cv2);
    fr(op->proj_sv2);
    fr(op->proj_iden2);
    if (op->proj_off) hipFree(op->proj_off);
    if (op->proj_idx) hipFree(op->proj_idx);
    if (op->proj_off2) hipFree(op->proj_off2);
    if (op->proj_idx2) hipFree(op->proj_idx2);
    if (op->proj_gslot) hipFree(op->proj_gslot);
    if (op->proj_gslot2) hipFree(op->proj_gslot2);
    fr(op->proj_glob);
    fr(op->nwv);
    fr(op->nwv_xp);
    fr(op->nwp);
    if (op->h_s) hipHostFree(op->h_s);
    if (op->baseflow) nlg_vec_destroy(op->baseflow);
    delete op;
    return 0;
}

int nlg_linop_init(nlg_linop *op) {
    NLG_CHECK(op, "nlg_linop_init: NULL");
    nlg_mesh *m = op->mesh;
    nlg_ctx *ctx = m->ctx;
    hipStream_t st = ctx->stream;
    const int dim = m->dim;
    NLG_HIP(hipSetDevice(ctx->device));
    if (!op->slab) {
        for (int c = 0; c < dim; ++c) {
            NLG_HIP(hipMalloc(&op->Ur[c], sizeof(double) * (size_t)m->lfn));
            for (int k = 1; k <= op->cfg.torder; ++k) NLG_TRY(lalloc(op, &op->pcv[k][c], m->lvs));
        }
        for (int q = 0; q < dim * dim; ++q) NLG_HIP(hipMalloc(&op->GU[q], sizeof(double) * (size_t)m->lfn));
        NLG_TRY(lalloc(op, &op->pce, m->lps));
        NLG_TRY(lalloc(op, &op->nwv, m->lvs));
        NLG_TRY(lalloc(op, &op->nwp, m->lps));
        NLG_HIP(hipHostMalloc(&op->h_s, sizeof(double) * kMaxLanes * S_N, hipHostMallocDefault));
        NLG_TRY(reduce_ws_reserve(ctx, 4));
        // single-reduction PCG for the pointwise-preconditioned solves, opt-in (NLG_PCG_SINGLE_RED=1; decided before the slab is laid
        // out: it needs one more vector per solve).  One all-reduce per iteration instead of two, for two more vector streams per
        // iteration: measured +9 % per time step on one rank at 10,240 elements, +3 % at 1,300, and 23 of 454 collectives per time step
        // fewer on 2 ranks -- it pays only where an all-reduce costs more than ~15 us, which this pool cannot measure
        op->use_sr = getenv("NLG_PCG_SINGLE_RED") && atoi(getenv("NLG_PCG_SINGLE_RED")) != 0;
        // deferred solution update of the velocity PCG (k_add_hist): depth of the direction ring, NLG_PCG_DEFER_X=0 switches it off
        // (default 16 slots, fewer where 16 would take more than 8 GB per lane: a solve that outlasts the ring pays one k_x_flush per
        // ring length, which still moves fewer bytes than updating x in every iteration as long as the ring holds >= 3 directions)
        const int64_t slot_bytes = (int64_t)sizeof(double) * dim * m->lvs;
        const int ph_auto = (int)std::max<int64_t>(3, std::min<int64_t>(16, ((int64_t)8 << 30) / std::max<int64_t>(slot_bytes, 1)));
        op->ph = op->use_sr ? 0 : std::max(0, std::min(kAlphaRing, getenv("NLG_PCG_DEFER_X") ? atoi(getenv("NLG_PCG_DEFER_X")) : ph_auto));
        // ... and of the pressure PCG, where the gradient kernel performs the direction update (3-D, lx1 = 8 .. 10).  Opt-in
        // (NLG_PCG_DEFER_XP=16): measured neutral at 10^4 elements -- the update kernel of the preconditioner drops 3 of its 9 streams
        // (preconditioner class 7.06 -> 6.83 ms per step), the assembly of x and the colder directions give it back (44.5 ms either way)
        op->php = (dim == 3 && sem_opgradt_fuses_pupdate(m) && getenv("NLG_PCG_DEFER_XP")) ? std::max(0, std::min(kAlphaRing, atoi(getenv("NLG_PCG_DEFER_XP")))) : 0;
        NLG_TRY(slab_ensure(op, 1));   // the work buffers of one lane; a block matvec grows the slab on first use
    }
    double *U[3] = {op->baseflow->vel(0), op->baseflow->vel(1), dim == 3 ? op->baseflow->vel(2) : nullptr};
    // dt / nsteps (reference: neklab_nek_setup.f90:195-198)
    if (op->cfg.dt > 0.0) {
        op->nsteps = (int)std::ceil(op->cfg.tau / op->cfg.dt - 1e-12);
        op->dt = op->cfg.tau / op->nsteps;
        NLG_TRY(sem_cfl(m, U, op->dt, &op->cfl));
    } else {
        double c1 = 0.0;
        NLG_TRY(sem_cfl(m, U, 1.0, &c1));
        NLG_CHECK(c1 > 0.0, "nlg_linop_init: base flow has zero CFL; give cfg.dt explicitly");
        const double dt0 = op->cfg.cfl_limit / c1;
        op->nsteps = (int)std::ceil(op->cfg.tau / dt0);
        op->dt = op->cfg.tau / op->nsteps;
        op->cfl = c1 * op->dt;
    }
    // convective-term precomputation
    NLG_TRY(sem_conv_setup(m, U, op->Ur, op->GU));
    // preconditioners
    const double nu = 1.0 / op->cfg.re;
    for (int k = 1; k <= op->cfg.torder; ++k) {
        double *dg = sem_scratch1(m, 3);
        NLG_CHECK(dg, "nlg_linop_init: scratch allocation failed");
        NLG_TRY(sem_helm_diag(m, dg, nu, BDF_B0[k] / op->dt));
        double *f[1] = {dg};
        NLG_TRY(sem_gs(m, f, 1));
        for (int c = 0; c < dim; ++c)
            NLG_LAUNCH(k_recipmask, dim3(grid_for(m->lvn)), dim3(NT), 0, st, m->lvn, op->pcv[k][c], (const double *)dg,
                               (const double *)m->d_mask[c]);
    }
    if (op->use_xp < 0) {
        const char *ev = getenv("NLG_XP");
        op->use_xp = (dim == 3 && !(m->n > 8 && getenv("NLG_AXHELM_CUBE") && atoi(getenv("NLG_AXHELM_CUBE")) != 0) && m->d_slot_xp && (m->gs.d_indices_xp || m->gs.ngroups == 0) && !(ev && atoi(ev) == 0)) ? 1 : 0;
    }
    if (op->use_xp > 0) {
        for (int k = 1; k <= op->cfg.torder; ++k) {
            for (int c = 0; c < dim; ++c)
                if (!op->pcv_xp[k][c]) NLG_TRY(lalloc(op, &op->pcv_xp[k][c], m->lvs));
            NLG_TRY(sem_to_xp(m, op->pcv[k], op->pcv_xp[k], dim));
        }
    }
    // compact form for the streaming kernels of the PCG: needs masks of zeros and ones (sem.hip checked that when it built the byte masks
    // of the pressure operator: m->d_maskb_fg exists) -- NLG_PC_MASKB=0 keeps the three arrays
    if (dim == 3 && m->d_maskb_fg && !op->use_sr && !(getenv("NLG_PC_MASKB") && atoi(getenv("NLG_PC_MASKB")) == 0)) {
        double *t0 = sem_scratch1(m, 3), *t1 = sem_scratch1(m, 4);
        NLG_CHECK(t0 && t1, "nlg_linop_init: scratch allocation failed");
        const bool xp = op->use_xp > 0;
        for (int k = 1; k <= op->cfg.torder; ++k) {
            NLG_TRY(sem_helm_diag(m, t0, nu, BDF_B0[k] / op->dt));
            double *f[1] = {t0};
            NLG_TRY(sem_gs(m, f, 1));
            if (!op->pci[k]) NLG_TRY(lalloc(op, &op->pci[k], m->lvs));
            NLG_LAUNCH(k_recip1, dim3(grid_for(m->lvn)), dim3(NT), 0, st, m->lvn, xp ? t1 : op->pci[k], (const double *)t0);
            if (xp) {
                double *a[1] = {t1}, *b[1] = {op->pci[k]};
                NLG_TRY(sem_to_xp(m, a, b, 1));
            }
            for (int c = 0; c < 3; ++c) op->pci_l[k][c] = op->pci[k];
        }
        NLG_LAUNCH(k_maskbits, dim3(grid_for(m->lvn)), dim3(NT), 0, st, m->lvn, t0, (const double *)m->d_mask[0], (const double *)m->d_mask[1],
                           (const double *)m->d_mask[2]);
        if (xp) {
            double *a[1] = {t0}, *b[1] = {t1};
            NLG_TRY(sem_to_xp(m, a, b, 1));
        }
        if (!op->maskb_v) {
            NLG_HIP(hipMalloc(&op->maskb_v, (size_t)m->lvs));
            NLG_HIP(hipMemsetAsync(op->maskb_v, 0, (size_t)m->lvs, st));
        }
        NLG_LAUNCH(k_to_bytes, dim3(grid_for(m->lvn)), dim3(NT), 0, st, m->lvn, op->maskb_v, (const double *)(xp ? t1 : t0));
    }
    if (op->cfg.ifheat) {
        if (!op->pct[1]) {
            for (int k = 1; k <= op->cfg.torder; ++k) NLG_TRY(lalloc(op, &op->pct[k], m->lvs));
            for (int q = 0; q < dim; ++q) NLG_HIP(hipMalloc(&op->GT[q], sizeof(double) * (size_t)m->lfn));
        }
        NLG_TRY(sem_conv_scalar_setup(m, op->baseflow->theta(0), op->GT));
        for (int k = 1; k <= op->cfg.torder; ++k) {
            double *dg = sem_scratch1(m, 3);
            NLG_CHECK(dg, "nlg_linop_init: scratch allocation failed");
            NLG_TRY(sem_helm_diag(m, dg, op->cfg.conductivity, op->cfg.rhocp * BDF_B0[k] / op->dt));
            double *f[1] = {dg};
            NLG_TRY(sem_gs(m, f, 1));
            NLG_LAUNCH(k_recipmask, dim3(grid_for(m->lvn)), dim3(NT), 0, st, m->lvn, op->pct[k], (const double *)dg,
                               (const double *)m->d_tmask);
        }
    }
    {
        double *ed = sem_scratch2(m, 5);
        NLG_CHECK(ed, "nlg_linop_init: scratch allocation failed");
        NLG_TRY(sem_ediag(m, ed));
        NLG_LAUNCH(k_recip1, dim3(grid_for(m->lpn)), dim3(NT), 0, st, m->lpn, op->pce, (const double *)ed);
    }
    NLG_LAUNCH(k_mul3, dim3(grid_for(m->lvn)), dim3(NT), 0, st, m->lvn, op->nwv, (const double *)m->d_binvm1,
                       (const double *)m->d_vmult, 1.0 / m->volvm1);
    NLG_LAUNCH(k_scale1, dim3(grid_for(m->lpn)), dim3(NT), 0, st, m->lpn, op->nwp, (const double *)m->d_bm2inv,
                       1.0 / m->volvm2);
    if (op->use_xp > 0) {
        if (!op->nwv_xp) NLG_TRY(lalloc(op, &op->nwv_xp, m->lvs));
        double *a[1] = {op->nwv}, *b[1] = {op->nwv_xp};
        NLG_TRY(sem_to_xp(m, a, b, 1));
    }
    NLG_HIP(hipGetLastError());
    NLG_HIP(hipStreamSynchronize(st));
    op->inited = true;
    return 0;
}

int nlg_linop_matvec(nlg_linop *op, const nlg_vec *vec_in, nlg_vec *vec_out) { return do_matvec(op, vec_in, vec_out, 0); }
int nlg_linop_rmatvec(nlg_linop *op, const nlg_vec *vec_in, nlg_vec *vec_out) { return do_matvec(op, vec_in, vec_out, 1); }

int nlg_linop_nonlinear_map(nlg_linop *op, const nlg_vec *vec_in, nlg_vec *vec_out) { return do_nonlinear_map(op, vec_in, vec_out); }

int nlg_linop_integrate_forced(nlg_linop *op, const nlg_vec *ic, const nlg_vec *f_re, const nlg_vec *f_im, double omega, int adjoint,
                               nlg_vec *vec_out) {
    return do_integrate_forced(op, ic, f_re, f_im, omega, adjoint, vec_out);
}

int nlg_linop_set_baseflow(nlg_linop *op, const nlg_vec *baseflow) {
    NLG_CHECK(op && baseflow && baseflow->mesh == op->mesh, "nlg_linop_set_baseflow: bad argument");
    NLG_TRY(nlg_vec_copy(op->baseflow, baseflow));
    return nlg_linop_init(op);
}

int nlg_linop_set_projection(nlg_linop *op, double alpha, int idir, const int64_t *line_label, const int64_t *line_label2,
                             const double *x2) {
    NLG_CHECK(op && line_label, "nlg_linop_set_projection: NULL argument");
    nlg_mesh *m = op->mesh;
    NLG_CHECK(idir >= 1 && idir <= m->dim, "nlg_linop_set_projection: idir %d out of range", idir);
    NLG_CHECK(op->inited, "nlg_linop_set_projection: call init first");
    NLG_CHECK((line_label2 == nullptr) == (x2 == nullptr), "nlg_linop_set_projection: pressure-mesh labels and coordinates go together");
    hipStream_t st = m->ctx->stream;
    int64_t proj_glob_cap = 0;
    if (op->proj_glob) {
        hipFree(op->proj_glob);
        op->proj_glob = nullptr;
    }
    // one set of lists per mesh: lines = groups of local dofs with the same label, ordered by label then by index
    auto build = [&](int64_t n, const int64_t *lab, const double *d_w, const double *d_x, const double *h_x, int *nl, int **d_off, int **d_idx,
                     double **d_cv, double **d_sv, double **d_iden, int **d_gslot, int64_t *nglob) -> int {
        std::vector<int> order((size_t)n);
        for (int64_t i = 0; i < n; ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [lab](int a, int b) { return lab[a] < lab[b] || (lab[a] == lab[b] && a < b); });
        std::vector<int> off{0};
        for (int64_t q = 1; q <= n; ++q)
            if (q == n || lab[order[q]] != lab[order[q - 1]]) off.push_back((int)q);
        const int nlines = (int)off.size() - 1;
        std::vector<double> wt((size_t)n), iden((size_t)nlines);
        NLG_HIP(hipMemcpyAsync(wt.data(), d_w, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        NLG_HIP(hipStreamSynchronize(st));
        for (int g = 0; g < nlines; ++g) {
            double sum = 0.0;
            for (int q = off[g]; q < off[g + 1]; ++q) sum += wt[order[q]];
            NLG_CHECK(sum > 0.0, "nlg_linop_set_projection: empty line");
            iden[g] = 1.0 / sum;
        }
        if (*d_off) hipFree(*d_off);
        if (*d_idx) hipFree(*d_idx);
        if (*d_cv) hipFree(*d_cv);
        if (*d_sv) hipFree(*d_sv);
        if (*d_iden) hipFree(*d_iden);
        NLG_HIP(hipMalloc(d_off, sizeof(int) * off.size()));
        NLG_HIP(hipMalloc(d_idx, sizeof(int) * (size_t)n));
        NLG_HIP(hipMalloc(d_iden, sizeof(double) * (size_t)nlines));
        NLG_HIP(hipMalloc(d_cv, sizeof(double) * (size_t)n));
        NLG_HIP(hipMalloc(d_sv, sizeof(double) * (size_t)n));
        NLG_HIP(hipMemcpy(*d_off, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice));
        NLG_HIP(hipMemcpy(*d_idx, order.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
        NLG_HIP(hipMemcpy(*d_iden, iden.data(), sizeof(double) * (size_t)nlines, hipMemcpyHostToDevice));
        const double *xs = d_x;
        double *tmp = nullptr;
        if (!xs) {   // coordinates given on the host (pressure mesh)
            NLG_HIP(hipMalloc(&tmp, sizeof(double) * (size_t)n));
            NLG_HIP(hipMemcpy(tmp, h_x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
            xs = tmp;
        }
        NLG_LAUNCH(k_cossin, dim3(grid_for(n)), dim3(NT), 0, st, n, xs, alpha, *d_cv, *d_sv);
        NLG_HIP(hipGetLastError());
        NLG_HIP(hipStreamSynchronize(st));
        if (tmp) hipFree(tmp);
        *nl = nlines;
        if (*d_gslot) hipFree(*d_gslot);
        *d_gslot = nullptr;
        *nglob = 0;
        if (m->ctx->distributed()) {
            // the labels are global line names: gather every rank's distinct labels, number the union (identically on all
            // ranks), and sum the weights of the parts of a line over the ranks for the denominators
            nlg_ctx *ctx = m->ctx;
            const int nr = ctx->nranks;
            double *d_cnt = nullptr;
            NLG_HIP(hipMalloc(&d_cnt, sizeof(double) * (nr + 1)));
            const double mine = (double)nlines;
            NLG_HIP(hipMemcpy(d_cnt + nr, &mine, sizeof(double), hipMemcpyHostToDevice));
            NLG_TRY(allgather_f64(ctx, d_cnt + nr, d_cnt, 1));
            std::vector<double> cnts(nr);
            NLG_HIP(hipMemcpyAsync(cnts.data(), d_cnt, sizeof(double) * nr, hipMemcpyDeviceToHost, st));
            NLG_HIP(hipStreamSynchronize(st));
            hipFree(d_cnt);
            int64_t maxc = 1;
            for (double c : cnts) maxc = std::max<int64_t>(maxc, (int64_t)c);
            std::vector<int64_t> mylab((size_t)maxc, -1), all((size_t)maxc * nr);
            for (int g = 0; g < nlines; ++g) mylab[g] = lab[order[off[g]]];
            int64_t *d_lab = nullptr;
            NLG_HIP(hipMalloc(&d_lab, sizeof(int64_t) * (size_t)maxc * (nr + 1)));
            NLG_HIP(hipMemcpy(d_lab + (size_t)maxc * nr, mylab.data(), sizeof(int64_t) * (size_t)maxc, hipMemcpyHostToDevice));
            NLG_TRY(allgather_f64(ctx, reinterpret_cast<const double *>(d_lab + (size_t)maxc * nr), reinterpret_cast<double *>(d_lab), maxc));
            NLG_HIP(hipMemcpyAsync(all.data(), d_lab, sizeof(int64_t) * (size_t)maxc * nr, hipMemcpyDeviceToHost, st));
            NLG_HIP(hipStreamSynchronize(st));
            hipFree(d_lab);
            std::vector<int64_t> uni;
            for (int q = 0; q < nr; ++q)
                for (int64_t g = 0; g < (int64_t)cnts[q]; ++g) uni.push_back(all[(size_t)q * maxc + g]);
            std::sort(uni.begin(), uni.end());
            uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
            std::vector<int> gs((size_t)std::max(nlines, 1));
            for (int g = 0; g < nlines; ++g) gs[g] = (int)(std::lower_bound(uni.begin(), uni.end(), mylab[g]) - uni.begin());
            NLG_HIP(hipMalloc(d_gslot, sizeof(int) * gs.size()));
            NLG_HIP(hipMemcpy(*d_gslot, gs.data(), sizeof(int) * gs.size(), hipMemcpyHostToDevice));
            *nglob = (int64_t)uni.size();
            const int64_t need = *nglob * 2 * 3;
            if (need > proj_glob_cap) {
                if (op->proj_glob) hipFree(op->proj_glob);
                NLG_HIP(hipMalloc(&op->proj_glob, sizeof(double) * (size_t)need));
                proj_glob_cap = need;
            }
            NLG_HIP(hipMemsetAsync(op->proj_glob, 0, sizeof(double) * (size_t)*nglob, st));
            const unsigned g1 = (unsigned)((nlines + NT / 64 - 1) / (NT / 64));
            if (nlines > 0)
                NLG_LAUNCH(k_proj_wsum, dim3(g1), dim3(NT), 0, st, (int64_t)nlines, (const int *)*d_off, (const int *)*d_idx,
                                   (const int *)*d_gslot, d_w, op->proj_glob);
            NLG_TRY(allreduce_sum(ctx, op->proj_glob, (int)*nglob));
            if (nlines > 0)
                NLG_LAUNCH(k_proj_iden, dim3(grid_for(nlines)), dim3(NT), 0, st, (int64_t)nlines, (const int *)*d_gslot,
                                   (const double *)op->proj_glob, *d_iden);
            NLG_HIP(hipGetLastError());
            NLG_HIP(hipStreamSynchronize(st));
        }
        return 0;
    };
    NLG_TRY(build(m->lvn, line_label, m->d_bm1, m->d_x[idir - 1], nullptr, &op->proj_nlines, &op->proj_off, &op->proj_idx, &op->proj_cv,
                  &op->proj_sv, &op->proj_iden, &op->proj_gslot, &op->proj_nglob));
    op->proj_nlines2 = 0;
    if (line_label2)
        NLG_TRY(build(m->lpn, line_label2, m->d_bm2, nullptr, x2, &op->proj_nlines2, &op->proj_off2, &op->proj_idx2, &op->proj_cv2,
                      &op->proj_sv2, &op->proj_iden2, &op->proj_gslot2, &op->proj_nglob2));
    return 0;
}

int nlg_linop_project(nlg_linop *op, nlg_vec *v) {
    NLG_CHECK(op && v && v->mesh == op->mesh, "nlg_linop_project: bad argument");
    NLG_CHECK(op->proj_nlines > 0, "nlg_linop_project: no projection set (nlg_linop_set_projection)");
    NLG_TRY(load_state(op, v, 0));
    NLG_TRY(project_alpha(op));
    NLG_TRY(store_state(op, v, 0));
    return 0;
}

int nlg_linop_set_tolerances(nlg_linop *op, double vtol, double ptol) {
    NLG_CHECK(op && vtol > 0.0 && ptol > 0.0, "nlg_linop_set_tolerances: bad argument");
    op->cfg.vtol = vtol;
    op->cfg.ptol = ptol;
    return 0;
}

int nlg_linop_set_tau(nlg_linop *op, double tau) {
    NLG_CHECK(op && tau > 0.0, "nlg_linop_set_tau: bad argument");
    if (tau != op->cfg.tau) {
        op->cfg.tau = tau;
        if (op->inited) return nlg_linop_init(op);
    }
    return 0;
}

int nlg_linop_get_info(const nlg_linop *op, double *tau, double *dt, int *nsteps, double *cfl) {
    NLG_CHECK(op, "nlg_linop_get_info: NULL");
    if (tau) *tau = op->cfg.tau;
    if (dt) *dt = op->dt;
    if (nsteps) *nsteps = op->nsteps;
    if (cfl) *cfl = op->cfl;
    return 0;
}

int nlg_linop_get_stats(const nlg_linop *op, int64_t *steps, int64_t *v_iters, int64_t *p_iters, int64_t *matvecs) {
    NLG_CHECK(op, "nlg_linop_get_stats: NULL");
    if (steps) *steps = op->st_steps;
    if (v_iters) *v_iters = op->st_viters;
    if (p_iters) *p_iters = op->st_piters;
    if (matvecs) *matvecs = op->st_matvecs;
    return 0;
}

}  // extern "C"

namespace {

// lane v >= 1: an operator object whose work buffers are lane v of the owner's slab; base-flow data shared with the owner
nlg_linop *lane_get(nlg_linop *op0, int v) {
    if (v == 0) return op0;
    if (op0->lanes[v - 1]) return op0->lanes[v - 1];
    nlg_linop *ln = new nlg_linop();
    ln->mesh = op0->mesh;
    ln->is_lane = true;
    ln->cfg = op0->cfg;
    op0->lanes[v - 1] = ln;
    return ln;
}

// base-flow data, time step and tolerances follow the owner (it may have been re-initialised since the lane was made)
int lane_refresh(nlg_linop *op0, nlg_linop *ln) {
    if (ln == op0) return 0;
    ln->cfg = op0->cfg;
    ln->baseflow = op0->baseflow;
    ln->inited = op0->inited;
    ln->dt = op0->dt, ln->cfl = op0->cfl, ln->nsteps = op0->nsteps;
    for (int c = 0; c < 3; ++c) {
        ln->Ur[c] = op0->Ur[c];
        for (int k = 0; k < 4; ++k) ln->pcv[k][c] = op0->pcv[k][c], ln->pcv_xp[k][c] = op0->pcv_xp[k][c];
    }
    for (int q = 0; q < 9; ++q) ln->GU[q] = op0->GU[q];
    for (int q = 0; q < 3; ++q) ln->GT[q] = op0->GT[q];          // Boussinesq coupling: base-temperature gradient and the scalar's
    for (int k = 0; k < 4; ++k) ln->pct[k] = op0->pct[k];        // Jacobi preconditioners are the owner's
    ln->pce = op0->pce, ln->nwv = op0->nwv, ln->nwv_xp = op0->nwv_xp, ln->nwp = op0->nwp;
    ln->use_xp = op0->use_xp;
    ln->h_s = nullptr;   // the PCG reads every lane's scalars through the owner's pinned buffer
    return 0;
}

int do_matvec_block(nlg_linop *op, int s, const nlg_vec *const *vin, nlg_vec *const *vout, int adjoint) {
    NLG_CHECK(op && vin && vout, "exptA block matvec: NULL argument");
    NLG_CHECK(s >= 1 && s <= kMaxLanes, "exptA block matvec: %d vectors unsupported (1..%d)", s, kMaxLanes);
    NLG_CHECK(op->inited, "exptA block matvec: nlg_linop_init has not been called");
    NLG_CHECK(!op->is_lane, "exptA block matvec: called on a lane");
    nlg_mesh *m = op->mesh;
    const int want_scal = op->cfg.ifheat ? 1 : 0;
    for (int v = 0; v < s; ++v) {
        NLG_CHECK(vin[v] && vout[v] && vin[v]->mesh == m && vout[v]->mesh == m, "exptA block matvec: vector %d NULL or on a different mesh", v);
        NLG_CHECK(vin[v]->nscal == want_scal && vout[v]->nscal == want_scal, "exptA block matvec: vector %d carries %d scalars, the operator %d", v,
                  vin[v]->nscal, want_scal);
        NLG_CHECK(op->cfg.no_history || (vin[v]->lorder >= op->cfg.torder && vout[v]->lorder >= op->cfg.torder), "exptA block matvec: vector lorder < time order");
        for (int u = 0; u < s; ++u) NLG_CHECK(vin[v] != vout[u], "exptA block matvec: an input vector is also an output vector");
        for (int u = 0; u < v; ++u) NLG_CHECK(vout[v] != vout[u], "exptA block matvec: the same output vector twice");
    }
    nlg_linop *ops[kMaxLanes];
    for (int v = 0; v < s; ++v) {
        ops[v] = lane_get(op, v);
        NLG_CHECK(ops[v], "exptA block matvec: lane allocation failed");
        NLG_TRY(lane_refresh(op, ops[v]));
    }
    NLG_TRY(slab_ensure(op, s));
    NLG_TRY(reset_state(op, s));
    const int nrst = op->cfg.no_history ? 0 : op->cfg.torder - 1;
    for (int v = 0; v < s; ++v) {
        nlg_linop *ln = ops[v];
        ln->istep = 0;
        ln->adjoint = adjoint;
        ln->nproj = 0;
        NLG_TRY(load_state(ln, vin[v], 0));
        NLG_TRY(project_alpha(op, ln, 0));   // exptA_proj_linop: the projections of do_matvec, lane by lane against the owner's tables
    }
    const Lanes L{ops, s};
    for (int istep = 1; istep <= op->nsteps; ++istep) {
        NLG_TRY(advance(L));
        if (istep <= nrst)
            for (int v = 0; v < s; ++v)
                if (vin[v]->nrst > 0) {
                    NLG_TRY(load_state(ops[v], vin[v], istep));   // get_rst, exponential_propagator.f90:129-142
                    NLG_TRY(project_alpha(op, ops[v], 0));
                }
    }
    for (int v = 0; v < s; ++v) {
        for (int slot = 0; slot < 3; ++slot) NLG_TRY(project_alpha(op, ops[v], slot));   // final state and the lagged levels (see do_matvec)
        NLG_TRY(nlg_vec_zero(vout[v]));
        NLG_TRY(store_state(ops[v], vout[v], 0));
    }
    for (int irst = 1; irst <= nrst; ++irst) {   // compute_rst, :109-127
        NLG_TRY(advance(L));
        for (int v = 0; v < s; ++v) {
            NLG_TRY(store_state(ops[v], vout[v], irst));
            vout[v]->nrst = std::max(vout[v]->nrst, irst);
        }
    }
    // the lanes' counters are part of the operator's statistics
    for (int v = 1; v < s; ++v) {
        op->st_steps += ops[v]->st_steps, op->st_viters += ops[v]->st_viters, op->st_piters += ops[v]->st_piters, op->st_titers += ops[v]->st_titers;
        ops[v]->st_steps = ops[v]->st_viters = ops[v]->st_piters = ops[v]->st_titers = 0;
    }
    op->st_matvecs += s;
    return 0;
}

}  // namespace

namespace nlg {
bool linop_can_block(const nlg_linop *op) { return op && !op->is_lane; }
}

extern "C" {

int nlg_linop_matvec_block(nlg_linop *op, int s, const nlg_vec *const *vec_in, nlg_vec *const *vec_out, int transpose) {
    return do_matvec_block(op, s, vec_in, vec_out, transpose ? 1 : 0);
}

int nlg_op_conv(nlg_mesh *m, const nlg_vec *base, const nlg_vec *in, nlg_vec *out, int adjoint) {
    NLG_CHECK(m && base && in && out && base->mesh == m && in->mesh == m && out->mesh == m, "nlg_op_conv: bad arguments");
    const int dim = m->dim;
    double *Ur[3] = {}, *GU[9] = {};
    for (int c = 0; c < dim; ++c) NLG_HIP(hipMalloc(&Ur[c], sizeof(double) * (size_t)m->lfn));
    for (int q = 0; q < dim * dim; ++q) NLG_HIP(hipMalloc(&GU[q], sizeof(double) * (size_t)m->lfn));
    double *U[3] = {base->vel(0), base->vel(1), dim == 3 ? base->vel(2) : nullptr};
    double *u[3] = {in->vel(0), in->vel(1), dim == 3 ? in->vel(2) : nullptr};
    double *o[3] = {out->vel(0), out->vel(1), dim == 3 ? out->vel(2) : nullptr};
    int rc = sem_conv_setup(m, U, Ur, GU);
    if (!rc) rc = sem_conv_apply(m, Ur, GU, u, o, adjoint);
    hipStreamSynchronize(m->ctx->stream);
    for (int c = 0; c < dim; ++c) hipFree(Ur[c]);
    for (int q = 0; q < dim * dim; ++q) hipFree(GU[q]);
    return rc;
}

}  // extern "C"
