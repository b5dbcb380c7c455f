// Spectral-element discretisation on device: mesh set-up, gather-scatter and the element-local
// operators of the matvec (gfx950).
//
// What the reference pins (file:line under /root/reference):
//   - Laplacian = grad -> metric -> grad^T with rxm1..tzm1, jacmi   src/linops/neklab_linops.f90:332-366
//   - pressure gradient / divergence on the lx2 = lx1-2 mesh         src/linops/neklab_linops.f90:368-380
//   - linearised convective terms, direct + adjoint                  src/linops/neklab_linops.f90:268-313
//   - dssum / multiplicity / masks                                   src/vectors/real_vectors.f90:100-108
// The Nek5000 routines behind those calls are restated from the published algorithm.
//
// Kernel / roofline summary (fp64, all HBM-bound; algorithmic bytes per element, n = lx1):
//   k_axhelm3   : (16*NF + 56) n^3 B   (u in, w out, 6 metric factors + mass)          12n^4+20n^3 flop/field
//   k_gs        : 20 B per shared local dof and field (value in/out + 4-byte index)
//   k_opgradt3  : 8 n2^3 (1 + 9) + 24 n^3 B ; k_opdiv3 : 24 n^3 + 8 n2^3 (9 + 1) B
#include <algorithm>
#include <cmath>
#include <numeric>

#include <type_traits>

#include "internal.h"

using namespace nlg;

// =================================================================================================
// host: 1-D operators (same formulas as oracle/sem.py, barycentric Lagrange form)
// =================================================================================================
namespace {

void legendre(int N, double x, double &pN, double &pNm1) {
    double p0 = 1.0, p1 = x;
    if (N == 0) {
        pN = 1.0;
        pNm1 = 0.0;
        return;
    }
    for (int k = 2; k <= N; ++k) {
        const double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
        p0 = p1;
        p1 = p2;
    }
    pN = p1;
    pNm1 = p0;
}

void gll_nodes(int n, std::vector<double> &x, std::vector<double> &w) {
    const int N = n - 1;
    x.resize(n);
    w.resize(n);
    for (int i = 0; i < n; ++i) x[i] = -cos(M_PI * i / N);
    for (int it = 0; it < 100; ++it) {
        double mx = 0.0;
        for (int i = 1; i < n - 1; ++i) {
            double pN, pNm1;
            legendre(N, x[i], pN, pNm1);
            const double f = N * (pNm1 - x[i] * pN);
            const double df = -(double)N * (N + 1) * pN;
            const double dx = f / df;
            x[i] -= dx;
            mx = std::max(mx, std::fabs(dx));
        }
        if (mx < 1e-16) break;
    }
    x[0] = -1.0;
    x[n - 1] = 1.0;
    std::vector<double> xs(x);
    for (int i = 0; i < n; ++i) x[i] = 0.5 * (xs[i] - xs[n - 1 - i]);
    for (int i = 0; i < n; ++i) {
        double pN, pNm1;
        legendre(N, x[i], pN, pNm1);
        w[i] = 2.0 / (N * (N + 1) * pN * pN);
    }
}

void gl_nodes(int n, std::vector<double> &x, std::vector<double> &w) {
    x.resize(n);
    w.resize(n);
    for (int k = 1; k <= n; ++k) x[k - 1] = -cos(M_PI * (k - 0.25) / (n + 0.5));
    for (int it = 0; it < 100; ++it) {
        double mx = 0.0;
        for (int i = 0; i < n; ++i) {
            double pn, pnm1;
            legendre(n, x[i], pn, pnm1);
            const double dpn = n * (x[i] * pn - pnm1) / (x[i] * x[i] - 1.0);
            const double dx = pn / dpn;
            x[i] -= dx;
            mx = std::max(mx, std::fabs(dx));
        }
        if (mx < 1e-16) break;
    }
    std::vector<double> xs(x);
    for (int i = 0; i < n; ++i) x[i] = 0.5 * (xs[i] - xs[n - 1 - i]);
    for (int i = 0; i < n; ++i) {
        double pn, pnm1;
        legendre(n, x[i], pn, pnm1);
        const double dpn = n * (x[i] * pn - pnm1) / (x[i] * x[i] - 1.0);
        w[i] = 2.0 / ((1.0 - x[i] * x[i]) * dpn * dpn);
    }
}

std::vector<double> bary(const std::vector<double> &x) {
    const int n = (int)x.size();
    std::vector<double> w(n, 1.0);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k)
            if (k != j) w[j] /= (x[j] - x[k]);
    return w;
}

// M[k*nf + j] = l_j(xto_k)
std::vector<double> interp_mat(const std::vector<double> &xf, const std::vector<double> &xt) {
    const int nf = (int)xf.size(), nt = (int)xt.size();
    std::vector<double> bw = bary(xf), M((size_t)nt * nf, 0.0);
    for (int k = 0; k < nt; ++k) {
        int hit = -1;
        for (int j = 0; j < nf; ++j)
            if (std::fabs(xt[k] - xf[j]) < 1e-15) hit = j;
        if (hit >= 0) {
            M[(size_t)k * nf + hit] = 1.0;
        } else {
            double s = 0.0;
            for (int j = 0; j < nf; ++j) {
                M[(size_t)k * nf + j] = bw[j] / (xt[k] - xf[j]);
                s += M[(size_t)k * nf + j];
            }
            for (int j = 0; j < nf; ++j) M[(size_t)k * nf + j] /= s;
        }
    }
    return M;
}

std::vector<double> deriv_mat(const std::vector<double> &x) {
    const int n = (int)x.size();
    std::vector<double> bw = bary(x), D((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j)
            if (i != j) {
                D[(size_t)i * n + j] = (bw[j] / bw[i]) / (x[i] - x[j]);
                s += D[(size_t)i * n + j];
            }
        D[(size_t)i * n + i] = -s;
    }
    return D;
}

std::vector<double> matmul(const std::vector<double> &A, const std::vector<double> &B, int m, int k, int n) {
    std::vector<double> C((size_t)m * n, 0.0);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += A[(size_t)i * k + l] * B[(size_t)l * n + j];
            C[(size_t)i * n + j] = s;
        }
    return C;
}

std::vector<double> transpose(const std::vector<double> &A, int m, int n) {
    std::vector<double> T((size_t)m * n);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) T[(size_t)j * m + i] = A[(size_t)i * n + j];
    return T;
}

int upload(const std::vector<double> &h, double **d) {
    NLG_HIP(hipMalloc(d, sizeof(double) * std::max<size_t>(h.size(), 1)));
    NLG_HIP(hipMemcpy(*d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    return 0;
}

int dalloc(double **d, int64_t n, hipStream_t s) {
    NLG_HIP(hipMalloc(d, sizeof(double) * (size_t)std::max<int64_t>(n, 1)));
    NLG_HIP(hipMemsetAsync(*d, 0, sizeof(double) * (size_t)std::max<int64_t>(n, 1), s));
    return 0;
}

struct F3 {
    double *p[3];
};
struct CF3 {
    const double *p[3];
};
struct CP4 {    // one read-only array per lane
    const double *p[4];
};
struct P4 {
    double *p[4];
};
struct CF3L {   // up to four lanes (vectors of a block step) of three fields
    const double *p[4][3];
};
struct F3L {
    double *p[4][3];
};
struct CF9 {
    const double *p[9];
};
struct F9 {
    double *p[9];
};

constexpr int NT = 256;

// =================================================================================================
// generic (runtime-size) set-up kernels: speed is irrelevant here
// =================================================================================================

// out = (Mz x My x Mx) in per element, optional pointwise tensor weight (wt[a]*wt[b]*wt[c]) on the output.
// M* are nout x nin row-major. One block per element, dynamic LDS: 2 * max(nin,nout)^dim doubles.
__global__ void k_tensor_generic(const double *in, double *out, int dim, int nin, int nout, const double *Mx,
                                 const double *My, const double *Mz, const double *wt, int64_t E) {
    extern __shared__ double sh[];
    const int64_t e = blockIdx.x;
    int nmax = nin > nout ? nin : nout;
    int cap = nmax * nmax * (dim == 3 ? nmax : 1);
    double *A = sh, *B = sh + cap;
    int s0 = nin, s1 = nin, s2 = (dim == 3 ? nin : 1);
    const int npin = s0 * s1 * s2;
    for (int p = threadIdx.x; p < npin; p += blockDim.x) A[p] = in[e * npin + p];
    __syncthreads();
    // x
    {
        const int o0 = nout;
        for (int p = threadIdx.x; p < o0 * s1 * s2; p += blockDim.x) {
            const int a = p % o0, bc = p / o0;
            double s = 0.0;
            for (int l = 0; l < s0; ++l) s += Mx[a * nin + l] * A[l + s0 * bc];
            B[p] = s;
        }
        s0 = o0;
        __syncthreads();
    }
    // y
    {
        const int o1 = nout;
        for (int p = threadIdx.x; p < s0 * o1 * s2; p += blockDim.x) {
            const int a = p % s0, b = (p / s0) % o1, c = p / (s0 * o1);
            double s = 0.0;
            for (int l = 0; l < s1; ++l) s += My[b * nin + l] * B[a + s0 * (l + s1 * c)];
            A[p] = s;
        }
        s1 = o1;
        __syncthreads();
    }
    double *res = A;
    if (dim == 3) {
        const int o2 = nout;
        for (int p = threadIdx.x; p < s0 * s1 * o2; p += blockDim.x) {
            const int ab = p % (s0 * s1), c = p / (s0 * s1);
            double s = 0.0;
            for (int l = 0; l < s2; ++l) s += Mz[c * nin + l] * A[ab + s0 * s1 * l];
            B[p] = s;
        }
        s2 = o2;
        res = B;
        __syncthreads();
    }
    const int npout = s0 * s1 * s2;
    for (int p = threadIdx.x; p < npout; p += blockDim.x) {
        double v = res[p];
        if (wt) {
            const int a = p % s0, b = (p / s0) % s1, c = p / (s0 * s1);
            v *= wt[a] * wt[b] * (dim == 3 ? wt[c] : 1.0);
        }
        out[e * npout + p] = v;
    }
}

// geometry from coordinates: rst (J-scaled), jac, bm1, G. One block per element.
__global__ void k_geom(int dim, int n, const double *D, const double *w1, CF3 X, F9 rst, double *jac, double *bm1,
                       double *G0, double *G1, double *G2, double *G3, double *G4, double *G5, int *bad) {
    extern __shared__ double sh[];
    const int np = n * n * (dim == 3 ? n : 1);
    const int64_t e = blockIdx.x;
    double *sx = sh, *sy = sh + np, *sz = sh + 2 * np;
    for (int p = threadIdx.x; p < np; p += blockDim.x) {
        sx[p] = X.p[0][e * np + p];
        sy[p] = X.p[1][e * np + p];
        if (dim == 3) sz[p] = X.p[2][e * np + p];
    }
    __syncthreads();
    for (int p = threadIdx.x; p < np; p += blockDim.x) {
        const int i = p % n, j = (p / n) % n, k = p / (n * n);
        double d[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // d[c][a] = d x_c / d r_a
        const double *sc[3] = {sx, sy, sz};
        for (int c = 0; c < dim; ++c) {
            double r = 0, s = 0, t = 0;
            for (int l = 0; l < n; ++l) {
                r += D[i * n + l] * sc[c][l + n * (j + n * k)];
                s += D[j * n + l] * sc[c][i + n * (l + n * k)];
                if (dim == 3) t += D[k * n + l] * sc[c][i + n * (j + n * l)];
            }
            d[c][0] = r;
            d[c][1] = s;
            d[c][2] = t;
        }
        double a[3][3];   // a[j][i] = J dr_j/dx_i
        double J;
        double w = w1[i] * w1[j] * (dim == 3 ? w1[k] : 1.0);
        if (dim == 2) {
            const double xr = d[0][0], xs = d[0][1], yr = d[1][0], ys = d[1][1];
            J = xr * ys - xs * yr;
            a[0][0] = ys;
            a[0][1] = -xs;
            a[1][0] = -yr;
            a[1][1] = xr;
        } else {
            const double xr = d[0][0], xs = d[0][1], xt = d[0][2];
            const double yr = d[1][0], ys = d[1][1], yt = d[1][2];
            const double zr = d[2][0], zs = d[2][1], zt = d[2][2];
            a[0][0] = ys * zt - yt * zs;
            a[0][1] = xt * zs - xs * zt;
            a[0][2] = xs * yt - xt * ys;
            a[1][0] = yt * zr - yr * zt;
            a[1][1] = xr * zt - xt * zr;
            a[1][2] = xt * yr - xr * yt;
            a[2][0] = yr * zs - ys * zr;
            a[2][1] = xs * zr - xr * zs;
            a[2][2] = xr * ys - xs * yr;
            J = xr * a[0][0] + xs * a[1][0] + xt * a[2][0];
        }
        if (!(J > 0.0)) atomicExch(bad, 1);
        const int64_t q = e * np + p;
        for (int jj = 0; jj < dim; ++jj)
            for (int ii = 0; ii < dim; ++ii) rst.p[jj * dim + ii][q] = a[jj][ii];
        jac[q] = J;
        bm1[q] = w * J;
        const double s = w / J;
        auto dotr = [&](int r1, int r2) {
            double v = 0;
            for (int m = 0; m < dim; ++m) v += a[r1][m] * a[r2][m];
            return v * s;
        };
        if (dim == 2) {
            G0[q] = dotr(0, 0);
            G1[q] = dotr(0, 1);
            G2[q] = dotr(1, 1);
        } else {
            G0[q] = dotr(0, 0);
            G1[q] = dotr(0, 1);
            G2[q] = dotr(0, 2);
            G3[q] = dotr(1, 1);
            G4[q] = dotr(1, 2);
            G5[q] = dotr(2, 2);
        }
    }
}

__global__ void k_set(double *x, double v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}
__global__ void k_recip(double *y, const double *x, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = 1.0 / x[i];
}
__global__ void k_addto(double *a, const double *b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] += b[i];
}
__global__ void k_mul(double *y, const double *a, const double *b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a[i] * b[i];
}

// =================================================================================================
// gather-scatter: one thread per group of local copies of a shared global dof
// =================================================================================================
// The groups are stored pairs first (a shared face interior: two copies, ~3/4 of all groups in 3-D), then quads (a
// shared edge: four copies), then the rest: a thread of the pair / quad range reads its indices as one int2 / int4 and
// needs neither the offset array nor a loop with dependent loads; the remaining groups (corners, irregular valences)
// go through the general CSR path.  Sums run over ascending local index in every class.
// Two pairs per thread where they are neighbours in memory (the face-grouped and slab-permuted layouts make the copies of
// a shared face contiguous in both elements, and the pairs are ordered by their first index): one int4 of indices and
// 16-byte loads / stores instead of two int2 and 8-byte accesses.  (In the natural layout the same idea lost -- 72 -> 86 us,
// DESIGN.md -- because there neighbouring pairs are 64 bytes apart; the check below falls back to the scalar path then.)
template <int NF>
__global__ __launch_bounds__(NT) void k_gs(const int *__restrict__ off, const int *__restrict__ idx, int64_t ngroups,
                                           int64_t npairs, int64_t nquads, F3 f, const double *__restrict__ gate, int64_t ld, int64_t ldg) {
    // blockIdx.y = lane of a block step: its fields sit ld doubles, its gate ldg doubles behind lane 0's (internal.h kMaxLanes)
    if (gate && gate[(int64_t)blockIdx.y * ldg] != 0.0) return;
    {
        const int64_t lo = (int64_t)blockIdx.y * ld;
#pragma unroll
        for (int c = 0; c < NF; ++c) f.p[c] += lo;
    }
    // Thread order: the general groups FIRST (corners and irregular valences: the longest chain of dependent loads in the kernel --
    // offsets, indices, values), then the quads, then the pairs.  At the end of the grid, where they used to be, their chain was the
    // kernel's tail: 17 us of a 50-us launch at 10^4 elements and the whole of a launch at 1300 (profiles/r04_E1300_counters.txt).
    // (round 4: an XCD-contiguous block order -- cdna_hip_programming.md T1, every XCD walking one eighth of the list so that blocks which
    //  share the lines at the seams of their runs share an L2 -- was measured: 54.5 -> 83 us per launch.  The dispatcher's round-robin
    //  order spreads the eight XCDs over all memory channels at every moment; an eighth of the list per XCD is an eighth of the address
    //  range per XCD, and the channels behind it saturate.  Not kept.)
    int64_t t = blockIdx.x * (int64_t)NT + threadIdx.x;
    const int64_t nrest = ngroups - npairs - nquads;
    if (t < nrest) {
        // indices of up to eight copies at once, then all their values, then the sum in ascending order of the local index (the order
        // of the one-at-a-time loop this replaces: same bits).  Eight covers a corner of a structured mesh; longer groups take
        // another round.  One at a time, every copy cost two serialised round trips (index, value).
        const int64_t g = npairs + nquads + t;
        const int b = off[g], e = off[g + 1];
        constexpr int CH = 8;   // (90 registers = five waves per SIMD for every thread of the kernel; CH = 4 gives 54 registers and eight waves -- and a SLOWER
                                // kernel, 4.91 -> 5.16 ms of gather-scatter per step: more waves in flight evict each other's lines)
        int i0[CH];
        double s[NF];
#pragma unroll
        for (int c = 0; c < NF; ++c) s[c] = 0.0;
        for (int q0 = b; q0 < e; q0 += CH) {
            int ii[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) ii[u] = q0 + u < e ? idx[q0 + u] : -1;
            double v[CH][NF];
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int c = 0; c < NF; ++c) v[u][c] = ii[u] >= 0 ? f.p[c][ii[u]] : 0.0;
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int c = 0; c < NF; ++c) s[c] += v[u][c];   // s is never -0.0 here, so adding the +0.0 of an unused slot changes nothing
            if (q0 == b) {
#pragma unroll
                for (int u = 0; u < CH; ++u) i0[u] = ii[u];
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (i0[u] >= 0) {
#pragma unroll
                for (int c = 0; c < NF; ++c) f.p[c][i0[u]] = s[c];
            }
        for (int q = b + CH; q < e; ++q) {
            const int i = idx[q];
#pragma unroll
            for (int c = 0; c < NF; ++c) f.p[c][i] = s[c];
        }
        return;
    }
    t -= nrest;
    if (t < nquads) {
        // 2 * npairs is even, so the quad block starts 8-byte aligned; read it as two int2
        const int2 *q2 = reinterpret_cast<const int2 *>(idx + 2 * npairs) + 2 * t;
        const int2 ab = q2[0], cd = q2[1];
        double s[NF];
#pragma unroll
        for (int c = 0; c < NF; ++c) s[c] = ((f.p[c][ab.x] + f.p[c][ab.y]) + f.p[c][cd.x]) + f.p[c][cd.y];
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            f.p[c][ab.x] = s[c];
            f.p[c][ab.y] = s[c];
            f.p[c][cd.x] = s[c];
            f.p[c][cd.y] = s[c];
        }
        return;
    }
    t -= nquads;
    const int64_t np2 = npairs >> 1;
    if (t < np2) {
        const int4 q = reinterpret_cast<const int4 *>(idx)[t];
        // (all fields are loaded before the first store: field by field, the compiler has to keep every later load behind the
        //  earlier stores -- the field pointers may alias -- and a thread pays NF serialised round trips)
        if (q.z == q.x + 1 && q.w == q.y + 1 && !((q.x | q.y) & 1)) {
            double2 a[NF], b[NF];
#pragma unroll
            for (int c = 0; c < NF; ++c) {
                a[c] = *reinterpret_cast<const double2 *>(f.p[c] + q.x);
                b[c] = *reinterpret_cast<const double2 *>(f.p[c] + q.y);
            }
#pragma unroll
            for (int c = 0; c < NF; ++c) {
                double2 s;
                s.x = a[c].x + b[c].x;
                s.y = a[c].y + b[c].y;
                *reinterpret_cast<double2 *>(f.p[c] + q.x) = s;
                *reinterpret_cast<double2 *>(f.p[c] + q.y) = s;
            }
        } else {
            double s0[NF], s1[NF];
#pragma unroll
            for (int c = 0; c < NF; ++c) {
                s0[c] = f.p[c][q.x] + f.p[c][q.y];
                s1[c] = f.p[c][q.z] + f.p[c][q.w];
            }
#pragma unroll
            for (int c = 0; c < NF; ++c) {
                f.p[c][q.x] = s0[c];
                f.p[c][q.y] = s0[c];
                f.p[c][q.z] = s1[c];
                f.p[c][q.w] = s1[c];
            }
        }
        return;
    }
    if (t == np2 && (npairs & 1)) {   // the odd pair
        const int2 ab = reinterpret_cast<const int2 *>(idx)[npairs - 1];
        double s[NF];
#pragma unroll
        for (int c = 0; c < NF; ++c) s[c] = f.p[c][ab.x] + f.p[c][ab.y];
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            f.p[c][ab.x] = s[c];
            f.p[c][ab.y] = s[c];
        }
    }
}

// w_c <- wt_c * w_c  (opbinv after dssum; also mask application)
template <int NF>
__global__ __launch_bounds__(NT) void k_colmul(F3 w, CF3 wt, int64_t n, int64_t ld = 0) {
#pragma unroll
    for (int c = 0; c < NF; ++c) w.p[c] += (int64_t)blockIdx.y * ld;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
#pragma unroll
        for (int c = 0; c < NF; ++c) w.p[c][i] *= wt.p[c][i];
    }
}

// =================================================================================================
// Helmholtz operator, element-local:  w = h1 * D^T G D u + h2 * B u
// 3-D: (N x N) threads per element sweep the k-slabs; u column and w column live in registers,
// the r/s contractions go through an LDS slab, geometric factors are read once for NF fields.
// =================================================================================================
// One thread per (i, j, field), runtime sweep over the k-slabs.  The element's u and the three metric-weighted
// derivative fields live in LDS cubes, so no register array is indexed by k (a fully unrolled register-column
// version made hipcc allocate 256 VGPRs and spill ~200 more: 1.9 ms per launch at E = 10k instead of ~0.15 ms).
// The fields of one element sit in different waves of the same block: the metric factors come from HBM once and
// are served to the other fields by L1.
template <int N>
__global__ __launch_bounds__(512) void k_axhelm3(int64_t E, int nf, int epb, const double *__restrict__ Dg,
                                                const double *__restrict__ G0, const double *__restrict__ G1,
                                                const double *__restrict__ G2, const double *__restrict__ G3,
                                                const double *__restrict__ G4, const double *__restrict__ G5,
                                                const double *__restrict__ bm1, CF3 u, F3 w, double h1, double h2,
                                                double *__restrict__ pw_part, CF3 zf, const double *__restrict__ beta_p,
                                                const double *__restrict__ done_p, int64_t uoff) {
    constexpr int NP = N * N * N, NS = N * N;
    extern __shared__ double smem[];
    __shared__ double sred[8];
    double *sD = smem;                   // N*N
    const int tid = threadIdx.x;
    const int slot = tid / NS;           // (element, field) slot inside the block
    const int ij = tid % NS;
    const int i = ij % N, j = ij / N;
    double *sU = smem + NS + (size_t)slot * 4 * NP;
    double *sR = sU + NP, *sS = sR + NP, *sT = sS + NP;
    for (int p = tid; p < NS; p += blockDim.x) sD[p] = Dg[p];
    const int64_t gslot = (int64_t)blockIdx.x * epb + slot;   // epb = (element, field) slots per block
    const int64_t e = gslot / nf;
    const int c = (int)(gslot % nf);
    const bool act = e < E;
    const int64_t base = (act ? e : 0) * NP;
    const double *uc = c == 0 ? u.p[0] : (c == 1 ? u.p[1] : u.p[2]);
    double *wc = c == 0 ? w.p[0] : (c == 1 ? w.p[1] : w.p[2]);
    // fused direction update of the surrounding PCG (beta_p != null): u <- z + beta u before the operator is applied,
    // unless the solver has converged (the separate update kernel is gated the same way)
    if (done_p && done_p[0] != 0.0) return;   // converged: nothing consumes w any more
    const bool upd = beta_p != nullptr && done_p[0] == 0.0;
    const double beta = upd ? beta_p[0] : 0.0;
    const double *zc = c == 0 ? zf.p[0] : (c == 1 ? zf.p[1] : zf.p[2]);
#pragma unroll 1
    for (int k = 0; k < N; ++k) {
        double v = act ? uc[base + ij + k * NS] : 0.0;
        if (upd && act) {
            v = zc[base + ij + k * NS] + beta * v;
            const_cast<double *>(uc)[uoff + base + ij + k * NS] = v;   // uoff: the updated direction goes to the next slot of the direction history (0 = in place)
        }
        sU[ij + k * NS] = v;
    }
    __syncthreads();
    double di[N], dj[N], dti[N], dtj[N];
#pragma unroll
    for (int l = 0; l < N; ++l) {
        di[l] = sD[i * N + l];
        dj[l] = sD[j * N + l];
        dti[l] = sD[l * N + i];
        dtj[l] = sD[l * N + j];
    }
#pragma unroll 1
    for (int k = 0; k < N; ++k) {
        const int64_t q = base + ij + k * NS;
        const double g0 = G0[q], g1 = G1[q], g2 = G2[q], g3 = G3[q], g4 = G4[q], g5 = G5[q];
        double ur = 0.0, us = 0.0, ut = 0.0;
#pragma unroll
        for (int l = 0; l < N; ++l) {
            ur += di[l] * sU[l + N * j + k * NS];
            us += dj[l] * sU[i + N * l + k * NS];
            ut += sD[k * N + l] * sU[ij + l * NS];
        }
        sR[ij + k * NS] = h1 * (g0 * ur + g1 * us + g2 * ut);
        sS[ij + k * NS] = h1 * (g1 * ur + g3 * us + g4 * ut);
        sT[ij + k * NS] = h1 * (g2 * ur + g4 * us + g5 * ut);
    }
    __syncthreads();
    double pw = 0.0;
#pragma unroll 1
    for (int k = 0; k < N; ++k) {
        const int64_t q = base + ij + k * NS;
        double a = h2 * bm1[q] * sU[ij + k * NS];
#pragma unroll
        for (int l = 0; l < N; ++l)
            a += dti[l] * sR[l + N * j + k * NS] + dtj[l] * sS[i + N * l + k * NS] + sD[l * N + k] * sT[ij + l * NS];
        if (act) {
            wc[q] = a;
            pw += a * sU[ij + k * NS];
        }
    }
    if (pw_part) {
        // first-stage sum of u . w_local of the surrounding PCG: for a continuous u this is (u, QQ^T w_local) with
        // the inverse-multiplicity weight, so the solver needs no separate pass over p and w
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pw += __shfl_down(pw, o, 64);
        if ((tid & 63) == 0) sred[tid >> 6] = pw;
        __syncthreads();
        if (tid == 0) {
            double a = 0.0;
            for (int q = 0; q < (int)((blockDim.x + 63) >> 6); ++q) a += sred[q];
            pw_part[blockIdx.x] = a;
        }
    }
}

// LDS hand-over between the lanes of ONE wave: LDS operations of a wave execute in order, so no s_barrier is needed;
// the fence keeps the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of `v` held by lane `srclane` (wave-uniform result in scalar registers)
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), srclane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Register-column variant for N <= 8: one wave (N*N active lanes) per (element, field).  Lane (i, j) keeps its k-column of
// u and of w in registers; per k-slab the r/s contractions go through three N x N LDS slabs (rows padded to N+1), the
// t contraction stays in registers.  A wave only ever touches its own slabs and LDS operations of one wave execute in
// order, so the kernel needs no barrier at all; ~2.5 KB of LDS per wave, occupancy set by registers alone.
// XP: u, zf and w live in the x-planes-first layout (xp_slot); the metric factors stay natural.
template <int N, int WPB, bool XP>
__global__ __launch_bounds__(64 * WPB) void k_axhelm3r(int64_t E, int nf, const double *__restrict__ Dg,
                                                       const double *__restrict__ G0, const double *__restrict__ G1,
                                                       const double *__restrict__ G2, const double *__restrict__ G3,
                                                       const double *__restrict__ G4, const double *__restrict__ G5,
                                                       const double *__restrict__ bm1, CF3 u, F3 w, double h1, double h2,
                                                       double *__restrict__ pw_part, CF3 zf, const double *__restrict__ beta_p,
                                                       const double *__restrict__ done_p, const int *__restrict__ xptab, int64_t ld, int64_t uoff) {
    static_assert(N * N <= 64, "one lane per (i, j)");
    constexpr int NP = N * N * N, NS = N * N, NQ = N + 1;
    __shared__ double sD[N * N];
    __shared__ double sU[WPB][N * NQ], sR[WPB][N * NQ], sS[WPB][N * NQ];
    __shared__ double sred[WPB];
    {   // blockIdx.y = lane of a block step (all per-lane arguments ld doubles apart, internal.h)
        const int64_t lo = (int64_t)blockIdx.y * ld;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (u.p[q]) u.p[q] += lo;
            if (w.p[q]) w.p[q] += lo;
            if (zf.p[q]) zf.p[q] += lo;
        }
        if (pw_part) pw_part += lo;
        if (beta_p) beta_p += lo;
        if (done_p) done_p += lo;
    }
    if (done_p && done_p[0] != 0.0) return;   // the surrounding PCG has converged: nothing consumes w any more
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: element, field and every base pointer stay scalar
    for (int p = tid; p < NS; p += 64 * WPB) sD[p] = Dg[p];
    __syncthreads();   // the only block-wide barrier: the derivative matrix
    const int64_t gslot = (int64_t)blockIdx.x * WPB + wv;
    const int64_t e = gslot / nf;
    const int c = (int)(gslot % nf);
    const bool act = e < E && lane < NS;
    const int ij = lane < NS ? lane : 0;
    const int i = ij % N, j = ij / N;
    const int64_t eoff = (e < E ? e : 0) * NP;
    const int base = ij;   // offset inside the element, natural layout (metric factors)
    // field offsets inside the element: vk[k]
    int vk[N];   // field offsets inside the element; slab-permuted layout: from the element's slot table (internal.h xp_slot), any permutation
    const int gen = XP ? xptab[N * N * N] : 0;   // entry N^3 of the table: 1 = general permutation, 0 = every slab permuted alike (one entry per lane)
    const int v0 = XP ? xptab[ij] : ij;
#pragma unroll
    for (int k = 0; k < N; ++k) vk[k] = gen ? xptab[ij + NS * k] : v0 + NS * k;
    const double *uc = (c == 0 ? u.p[0] : (c == 1 ? u.p[1] : u.p[2])) + eoff;
    double *wc = (c == 0 ? w.p[0] : (c == 1 ? w.p[1] : w.p[2])) + eoff;
    const double *zc = (c == 0 ? zf.p[0] : (c == 1 ? zf.p[1] : zf.p[2])) + eoff;
    G0 += eoff, G1 += eoff, G2 += eoff, G3 += eoff, G4 += eoff, G5 += eoff, bm1 += eoff;
    const bool upd = beta_p != nullptr && done_p[0] == 0.0;
    const double beta = upd ? beta_p[0] : 0.0;
    double uk[N], wk[N], di[N], dj[N], dti[N], dtj[N];
    // all loads of the column first, then the stores of the fused direction update: with load / store alternating per point the
    // compiler cannot hoist the later loads over the earlier stores (same array) and the wave pays N serialised round trips
    {
        double zk[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            uk[k] = act ? uc[vk[k]] : 0.0;
            zk[k] = (upd && act) ? zc[vk[k]] : 0.0;
            wk[k] = 0.0;
        }
        if (upd && act) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                uk[k] = zk[k] + beta * uk[k];
                const_cast<double *>(uc)[uoff + vk[k]] = uk[k];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < N; ++l) {
        di[l] = sD[i * N + l];
        dj[l] = sD[j * N + l];
        dti[l] = sD[l * N + i];
        dtj[l] = sD[l * N + j];
    }
    double *mU = sU[wv], *mR = sR[wv], *mS = sS[wv];
    double pw = 0.0;
    // metric factors of slab k+1 are requested while slab k is computed; the compiler barrier at the end of every slab
    // keeps it from hoisting ALL slabs' loads to the top (which costs 400 registers and the occupancy)
    double gn[7];
    gn[0] = G0[base], gn[1] = G1[base], gn[2] = G2[base], gn[3] = G3[base], gn[4] = G4[base], gn[5] = G5[base], gn[6] = bm1[base];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double g0 = gn[0], g1 = gn[1], g2 = gn[2], g3 = gn[3], g4 = gn[4], g5 = gn[5], bm = gn[6];
        if (k + 1 < N) {
            const int q = base + (k + 1) * NS;
            gn[0] = G0[q], gn[1] = G1[q], gn[2] = G2[q], gn[3] = G3[q], gn[4] = G4[q], gn[5] = G5[q], gn[6] = bm1[q];
        }
        if (lane < NS) mU[i + NQ * j] = uk[k];
        wave_lds_sync();
        // row k of D, wave-uniform: lane k (i = k, j = 0) holds it in di[] -> scalar registers, no LDS, no vector registers
        double dk[N];
#pragma unroll
        for (int l = 0; l < N; ++l) dk[l] = readlane_f64(di[l], k);
        double ur = 0.0, us = 0.0, ut = 0.0;
#pragma unroll
        for (int l = 0; l < N; ++l) {
            ur += di[l] * mU[l + NQ * j];
            us += dj[l] * mU[i + NQ * l];
            ut += dk[l] * uk[l];
        }
        const double gr = h1 * (g0 * ur + g1 * us + g2 * ut);
        const double gs = h1 * (g1 * ur + g3 * us + g4 * ut);
        const double gt = h1 * (g2 * ur + g4 * us + g5 * ut);
        if (lane < NS) {
            mR[i + NQ * j] = gr;
            mS[i + NQ * j] = gs;
        }
        wave_lds_sync();
#pragma unroll
        for (int l = 0; l < N; ++l) wk[l] += dk[l] * gt;
        double a = h2 * bm * uk[k];
#pragma unroll
        for (int l = 0; l < N; ++l) a += dti[l] * mR[l + NQ * j] + dtj[l] * mS[i + NQ * l];
        wk[k] += a;
        // pin the accumulators here: otherwise the compiler sinks these sums into the guarded store at the end and keeps
        // every slab's LDS operands alive until then (400 registers, one wave per SIMD)
#pragma unroll
        for (int l = 0; l < N; ++l) asm volatile("" : "+v"(wk[l]));
        asm volatile("" ::: "memory");
    }
    if (act) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            wc[vk[k]] = wk[k];
            pw += wk[k] * uk[k];
        }
    }
    if (pw_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pw += __shfl_down(pw, o, 64);
        if (lane == 0) sred[wv] = pw;
        __syncthreads();
        if (tid == 0) {
            double a = 0.0;
            for (int q = 0; q < WPB; ++q) a += sred[q];
            pw_part[blockIdx.x] = a;
        }
    }
}

// The same for the NL <= 4 lanes of a block step: one block = one element, 3 NL waves = (lane, component) pairs, so the six
// metric factors and the mass matrix of the element are fetched from HBM once and served to the other waves by L1.  Every
// lane has its own direction update (beta), done flag and (p, w) sums -- one sum per (lane, element).
struct HelmLanes {
    const double *u[4][3];
    double *w[4][3];
    const double *z[4][3];
    const double *beta[4];
    const double *done[4];
    double *pw[4];
};
template <int N, int NL, bool XP>
__global__ __launch_bounds__(64 * 3 * NL) void k_axhelm3rb(int64_t E, const double *__restrict__ Dg,
                                                       const double *__restrict__ G0, const double *__restrict__ G1,
                                                       const double *__restrict__ G2, const double *__restrict__ G3,
                                                       const double *__restrict__ G4, const double *__restrict__ G5,
                                                       const double *__restrict__ bm1, HelmLanes L, double h1, double h2,
                                                       const int *__restrict__ xptab) {
    constexpr int WPB = 3 * NL;
    static_assert(N * N <= 64, "one lane per (i, j)");
    constexpr int NP = N * N * N, NS = N * N, NQ = N + 1;
    __shared__ double sD[N * N];
    __shared__ double sU[WPB][N * NQ], sR[WPB][N * NQ], sS[WPB][N * NQ];
    __shared__ double sred[WPB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: element, field and every base pointer stay scalar
    for (int p = tid; p < NS; p += 64 * WPB) sD[p] = Dg[p];
    __syncthreads();   // the only block-wide barrier: the derivative matrix
    const int64_t e = blockIdx.x;
    const int lv = wv / 3, c = wv - 3 * lv;
    const double *done_p = L.done[lv], *beta_p = L.beta[lv];
    const bool gated = done_p && done_p[0] != 0.0;   // this lane's PCG has converged: nothing consumes its w any more
    const bool act = e < E && lane < NS && !gated;
    const int ij = lane < NS ? lane : 0;
    const int i = ij % N, j = ij / N;
    const int64_t eoff = (e < E ? e : 0) * NP;
    const int base = ij;   // offset inside the element, natural layout (metric factors)
    // field offsets inside the element: vk[k]
    int vk[N];   // field offsets inside the element; slab-permuted layout: from the element's slot table (internal.h xp_slot), any permutation
    const int gen = XP ? xptab[N * N * N] : 0;   // entry N^3 of the table: 1 = general permutation, 0 = every slab permuted alike (one entry per lane)
    const int v0 = XP ? xptab[ij] : ij;
#pragma unroll
    for (int k = 0; k < N; ++k) vk[k] = gen ? xptab[ij + NS * k] : v0 + NS * k;
    const double *uc = L.u[lv][c] + eoff;
    double *wc = L.w[lv][c] + eoff;
    const double *zc = L.z[lv][c] + eoff;
    G0 += eoff, G1 += eoff, G2 += eoff, G3 += eoff, G4 += eoff, G5 += eoff, bm1 += eoff;
    const bool upd = beta_p != nullptr && !gated;
    const double beta = upd ? beta_p[0] : 0.0;
    double pw = 0.0;
    if (!gated) {   // (a converged lane skips the work; its waves still take part in the block reduction below)
    double uk[N], wk[N], di[N], dj[N], dti[N], dtj[N];
    // all loads of the column first, then the stores of the fused direction update: with load / store alternating per point the
    // compiler cannot hoist the later loads over the earlier stores (same array) and the wave pays N serialised round trips
    {
        double zk[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            uk[k] = act ? uc[vk[k]] : 0.0;
            zk[k] = (upd && act) ? zc[vk[k]] : 0.0;
            wk[k] = 0.0;
        }
        if (upd && act) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                uk[k] = zk[k] + beta * uk[k];
                const_cast<double *>(uc)[vk[k]] = uk[k];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < N; ++l) {
        di[l] = sD[i * N + l];
        dj[l] = sD[j * N + l];
        dti[l] = sD[l * N + i];
        dtj[l] = sD[l * N + j];
    }
    double *mU = sU[wv], *mR = sR[wv], *mS = sS[wv];
    // metric factors of slab k+1 are requested while slab k is computed; the compiler barrier at the end of every slab
    // keeps it from hoisting ALL slabs' loads to the top (which costs 400 registers and the occupancy)
    double gn[7];
    gn[0] = G0[base], gn[1] = G1[base], gn[2] = G2[base], gn[3] = G3[base], gn[4] = G4[base], gn[5] = G5[base], gn[6] = bm1[base];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double g0 = gn[0], g1 = gn[1], g2 = gn[2], g3 = gn[3], g4 = gn[4], g5 = gn[5], bm = gn[6];
        if (k + 1 < N) {
            const int q = base + (k + 1) * NS;
            gn[0] = G0[q], gn[1] = G1[q], gn[2] = G2[q], gn[3] = G3[q], gn[4] = G4[q], gn[5] = G5[q], gn[6] = bm1[q];
        }
        if (lane < NS) mU[i + NQ * j] = uk[k];
        wave_lds_sync();
        // row k of D, wave-uniform: lane k (i = k, j = 0) holds it in di[] -> scalar registers, no LDS, no vector registers
        double dk[N];
#pragma unroll
        for (int l = 0; l < N; ++l) dk[l] = readlane_f64(di[l], k);
        double ur = 0.0, us = 0.0, ut = 0.0;
#pragma unroll
        for (int l = 0; l < N; ++l) {
            ur += di[l] * mU[l + NQ * j];
            us += dj[l] * mU[i + NQ * l];
            ut += dk[l] * uk[l];
        }
        const double gr = h1 * (g0 * ur + g1 * us + g2 * ut);
        const double gs = h1 * (g1 * ur + g3 * us + g4 * ut);
        const double gt = h1 * (g2 * ur + g4 * us + g5 * ut);
        if (lane < NS) {
            mR[i + NQ * j] = gr;
            mS[i + NQ * j] = gs;
        }
        wave_lds_sync();
#pragma unroll
        for (int l = 0; l < N; ++l) wk[l] += dk[l] * gt;
        double a = h2 * bm * uk[k];
#pragma unroll
        for (int l = 0; l < N; ++l) a += dti[l] * mR[l + NQ * j] + dtj[l] * mS[i + NQ * l];
        wk[k] += a;
        // pin the accumulators here: otherwise the compiler sinks these sums into the guarded store at the end and keeps
        // every slab's LDS operands alive until then (400 registers, one wave per SIMD)
#pragma unroll
        for (int l = 0; l < N; ++l) asm volatile("" : "+v"(wk[l]));
        asm volatile("" ::: "memory");
    }
    if (act) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            wc[vk[k]] = wk[k];
            pw += wk[k] * uk[k];
        }
    }
    }
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pw += __shfl_down(pw, o, 64);
        if (lane == 0) sred[wv] = pw;
        __syncthreads();
        if (tid < NL && L.pw[tid] && !(L.done[tid] && L.done[tid][0] != 0.0)) L.pw[tid][e] = (sred[3 * tid] + sred[3 * tid + 1]) + sred[3 * tid + 2];
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global load (s_waitcnt vmcnt(0)
// before s_barrier), which turns a prefetch issued ahead of it into a blocking load; with the fences restricted to the LDS address
// space only the LDS counter is drained.  (An inline-asm barrier with a "memory" clobber does the same to the waits but makes the
// compiler fetch the wave-uniform matrix rows with VECTOR loads: a clobber between two loads forbids the scalar path.)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Register-column variant for N > 8: one block of ceil(N N / 64) waves per (element, field); thread (i, j) keeps its
// k-column of u and of w in registers as in k_axhelm3r, the three N x N slabs are shared by the block's waves, so the two
// hand-overs per slab are block barriers (two or three waves: cheap) instead of the wave-level LDS ordering.  Row k of D
// comes from LDS at a block-uniform address (broadcast).  Replaces the LDS-cube kernel k_axhelm3, which ran at 25 % of
// the HBM roofline at lx1 = 10 (452 us for 912 MB at 6000 elements) against 58 % for k_axhelm3r at lx1 = 8.
// PPB (element, field) pairs per block (round 3): N N threads fill only 75 - 78 % of the lanes of their waves at N = 10, 12; three pairs
// side by side (thread -> pair tid / (N N)) fill 94 - 96 %.  Each pair has its own three slabs in LDS; the barriers are the block's.
template <int N, bool XP = false, int PPB = 1>
__global__ __launch_bounds__(((PPB * N * N + 63) / 64) * 64) void k_axhelm3c(int64_t E, int nf, const double *__restrict__ Dg,
                                                                        const double *__restrict__ G0, const double *__restrict__ G1,
                                                                        const double *__restrict__ G2, const double *__restrict__ G3,
                                                                        const double *__restrict__ G4, const double *__restrict__ G5,
                                                                        const double *__restrict__ bm1, CF3 u, F3 w, double h1, double h2,
                                                                        double *__restrict__ pw_part, CF3 zf, const double *__restrict__ beta_p,
                                                                        const double *__restrict__ done_p, const int *__restrict__ xptab, int64_t ld, int64_t uoff, int xcd_map) {
    constexpr int NP = N * N * N, NS = N * N, NQ = N + 1, NTB = ((PPB * NS + 63) / 64) * 64, NWB = NTB / 64;
    __shared__ double sD[NS];
    __shared__ double mUa[PPB][N * NQ], mRa[PPB][N * NQ], mSa[PPB][N * NQ];
    __shared__ double sred[NWB];
    {
        const int64_t lo = (int64_t)blockIdx.y * ld;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (u.p[q]) u.p[q] += lo;
            if (w.p[q]) w.p[q] += lo;
            if (zf.p[q]) zf.p[q] += lo;
        }
        if (pw_part) pw_part += lo;
        if (beta_p) beta_p += lo;
        if (done_p) done_p += lo;
    }
    if (done_p && done_p[0] != 0.0) return;
    const int tid = threadIdx.x;
    for (int p = tid; p < NS; p += NTB) sD[p] = Dg[p];
    lds_barrier();
    const int pb = tid / NS;                                   // pair of this thread within the block
    const int64_t pair = (int64_t)blockIdx.x * PPB + (pb < PPB ? pb : 0);
    const bool act = pb < PPB && pair < E * nf;
    const int64_t pr_ = act ? pair : 0;
    int64_t e = pr_ / nf;
    int c = (int)(pr_ % nf);
    if (PPB == 1 && nf == 3 && xcd_map) {
        // one pair per block (lx1 = 10): blocks go to the XCDs round-robin, so the three components of an element -- which read the same seven
        // metric arrays -- are given block numbers that are equal modulo 8: they run on ONE XCD, close in time, and share its L2
        // (groups of 8 elements x 3 components = 24 consecutive blocks; the elements behind the last full group keep the plain order)
        const int64_t full = (E / 8) * 24;
        if (pr_ < full) {
            const int64_t q = pr_ / 24;
            const int r = (int)(pr_ % 24);
            e = 8 * q + (r & 7);
            c = r >> 3;
        }
    }
    const int ij = act ? tid - pb * NS : 0;
    double *const mU = mUa[pb < PPB ? pb : 0], *const mR = mRa[pb < PPB ? pb : 0], *const mS = mSa[pb < PPB ? pb : 0];
    const int i = ij % N, j = ij / N;
    int pk[N];   // slab-permuted layout (any permutation of the element, internal.h xp_slot): the vectors of the PCG; the metric arrays stay natural
    const int gen = XP ? xptab[N * N * N] : 0;   // entry N^3 of the table: 1 = general permutation, 0 = every slab permuted alike
    const int p0 = XP ? xptab[ij] : ij;
#pragma unroll
    for (int k = 0; k < N; ++k) pk[k] = gen ? xptab[ij + NS * k] : p0 + NS * k;
    const int64_t eoff = e * NP;
    const double *uc = (c == 0 ? u.p[0] : (c == 1 ? u.p[1] : u.p[2])) + eoff;
    double *wc = (c == 0 ? w.p[0] : (c == 1 ? w.p[1] : w.p[2])) + eoff;
    const double *zc = (c == 0 ? zf.p[0] : (c == 1 ? zf.p[1] : zf.p[2])) + eoff;
    G0 += eoff, G1 += eoff, G2 += eoff, G3 += eoff, G4 += eoff, G5 += eoff, bm1 += eoff;
    const bool upd = beta_p != nullptr && done_p[0] == 0.0;
    const double beta = upd ? beta_p[0] : 0.0;
    double uk[N], wk[N], di[N], dj[N], dti[N], dtj[N];
    {   // loads first, then the stores of the fused direction update (see k_axhelm3r)
        double zk[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            uk[k] = act ? uc[pk[k]] : 0.0;
            zk[k] = (upd && act) ? zc[pk[k]] : 0.0;
            wk[k] = 0.0;
        }
        if (upd && act) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                uk[k] = zk[k] + beta * uk[k];
                const_cast<double *>(uc)[uoff + pk[k]] = uk[k];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < N; ++l) {
        di[l] = sD[i * N + l];
        dj[l] = sD[j * N + l];
        dti[l] = sD[l * N + i];
        dtj[l] = sD[l * N + j];
    }
    double gn[7];
    gn[0] = G0[ij], gn[1] = G1[ij], gn[2] = G2[ij], gn[3] = G3[ij], gn[4] = G4[ij], gn[5] = G5[ij], gn[6] = bm1[ij];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double g0 = gn[0], g1 = gn[1], g2 = gn[2], g3 = gn[3], g4 = gn[4], g5 = gn[5], bm = gn[6];
        if (k + 1 < N) {
            const int q = ij + (k + 1) * NS;
            gn[0] = G0[q], gn[1] = G1[q], gn[2] = G2[q], gn[3] = G3[q], gn[4] = G4[q], gn[5] = G5[q], gn[6] = bm1[q];
        }
        if (act) mU[i + NQ * j] = uk[k];
        lds_barrier();
        double ur = 0.0, us = 0.0, ut = 0.0;
#pragma unroll
        for (int l = 0; l < N; ++l) {
            ur += di[l] * mU[l + NQ * j];
            us += dj[l] * mU[i + NQ * l];
            ut += Dg[k * N + l] * uk[l];   // row k of D: compile-time index on a restrict argument = scalar operand (a third of the LDS reads of a slab)
        }
        const double gr = h1 * (g0 * ur + g1 * us + g2 * ut);
        const double gs = h1 * (g1 * ur + g3 * us + g4 * ut);
        const double gt = h1 * (g2 * ur + g4 * us + g5 * ut);
        if (act) {
            mR[i + NQ * j] = gr;
            mS[i + NQ * j] = gs;
        }
        lds_barrier();
#pragma unroll
        for (int l = 0; l < N; ++l) wk[l] += Dg[k * N + l] * gt;
        double a = h2 * bm * uk[k];
#pragma unroll
        for (int l = 0; l < N; ++l) a += dti[l] * mR[l + NQ * j] + dtj[l] * mS[i + NQ * l];
        wk[k] += a;
#pragma unroll
        for (int l = 0; l < N; ++l) asm volatile("" : "+v"(wk[l]));   // see k_axhelm3r: keeps the sums from sinking
        asm volatile("" ::: "memory");
    }
    double pw = 0.0;
    if (act) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            wc[pk[k]] = wk[k];
            pw += wk[k] * uk[k];
        }
    }
    if (pw_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pw += __shfl_down(pw, o, 64);
        if ((tid & 63) == 0) sred[tid >> 6] = pw;
        lds_barrier();
        if (tid == 0) {
            double a = 0.0;
            for (int q = 0; q < NWB; ++q) a += sred[q];
            pw_part[blockIdx.x] = a;
        }
    }
}

// 2-D: one thread per point.
template <int N, int NF>
__global__ __launch_bounds__(((NT / (N * N)) > 0 ? (NT / (N * N)) : 1) * N * N) void k_axhelm2(
    int64_t E, const double *__restrict__ Dg, const double *__restrict__ G0, const double *__restrict__ G1,
    const double *__restrict__ G2, const double *__restrict__ bm1, CF3 u, F3 w, double h1, double h2,
    double *__restrict__ pw_part, CF3 zf, const double *__restrict__ beta_p, const double *__restrict__ done_p, int64_t uoff) {
    constexpr int EPB = (NT / (N * N)) > 0 ? (NT / (N * N)) : 1;
    constexpr int NP = N * N;
    __shared__ double sD[N * N];
    __shared__ double sU[EPB][NF][NP];
    __shared__ double sR[EPB][NF][NP];
    __shared__ double sS[EPB][NF][NP];
    __shared__ double sred[8];
    // same PCG fusions as the 3-D kernel: skip after convergence, u <- z + beta u on load, (u, w_local) sums
    if (done_p && done_p[0] != 0.0) return;
    const bool upd = beta_p != nullptr;
    const double beta = upd ? beta_p[0] : 0.0;
    const int tid = threadIdx.x;
    const int le = tid / NP, ij = tid % NP, i = ij % N, j = ij / N;
    for (int p = tid; p < N * N; p += EPB * NP) sD[p] = Dg[p];
    const int64_t e = (int64_t)blockIdx.x * EPB + le;
    const bool act = e < E;
    const int64_t q = (act ? e : 0) * NP + ij;
    double uu[NF];
#pragma unroll
    for (int c = 0; c < NF; ++c) {
        uu[c] = act ? u.p[c][q] : 0.0;
        if (upd && act) {
            uu[c] = zf.p[c][q] + beta * uu[c];
            const_cast<double *>(u.p[c])[uoff + q] = uu[c];
        }
        sU[le][c][ij] = uu[c];
    }
    const double g0 = G0[q], g1 = G1[q], g2 = G2[q], bm = bm1[q];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NF; ++c) {
        double ur = 0.0, us = 0.0;
#pragma unroll
        for (int l = 0; l < N; ++l) {
            ur += sD[i * N + l] * sU[le][c][l + N * j];
            us += sD[j * N + l] * sU[le][c][i + N * l];
        }
        sR[le][c][ij] = g0 * ur + g1 * us;
        sS[le][c][ij] = g1 * ur + g2 * us;
    }
    __syncthreads();
    double pw = 0.0;
    if (act) {
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) a += sD[l * N + i] * sR[le][c][l + N * j] + sD[l * N + j] * sS[le][c][i + N * l];
            const double wv = h1 * a + h2 * bm * uu[c];
            w.p[c][q] = wv;
            pw += wv * uu[c];
        }
    }
    if (pw_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pw += __shfl_down(pw, o, 64);
        if ((tid & 63) == 0) sred[tid >> 6] = pw;
        __syncthreads();
        if (tid == 0) {
            double a = 0.0;
            for (int q2 = 0; q2 < (int)((blockDim.x + 63) >> 6); ++q2) a += sred[q2];
            pw_part[blockIdx.x] = a;
        }
    }
}

// exact diagonal of the local Helmholtz operator (cross metric terms included), generic sizes
__global__ void k_helm_diag(int dim, int n, int64_t E, const double *D, const double *G0, const double *G1,
                            const double *G2, const double *G3, const double *G4, const double *G5,
                            const double *bm1, double *out, double h1, double h2) {
    const int np = n * n * (dim == 3 ? n : 1);
    const int64_t tot = E * np;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < tot; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = q / np;
        const int p = (int)(q % np);
        const int i = p % n, j = (p / n) % n, k = p / (n * n);
        const int64_t b = e * np;
        double s = 0.0;
        if (dim == 2) {
            for (int l = 0; l < n; ++l) {
                s += D[l * n + i] * D[l * n + i] * G0[b + l + n * j];
                s += D[l * n + j] * D[l * n + j] * G2[b + i + n * l];
            }
            s += 2.0 * G1[q] * D[i * n + i] * D[j * n + j];
        } else {
            for (int l = 0; l < n; ++l) {
                s += D[l * n + i] * D[l * n + i] * G0[b + l + n * (j + n * k)];
                s += D[l * n + j] * D[l * n + j] * G3[b + i + n * (l + n * k)];
                s += D[l * n + k] * D[l * n + k] * G5[b + i + n * (j + n * l)];
            }
            const double di = D[i * n + i], dj = D[j * n + j], dk = D[k * n + k];
            s += 2.0 * (G1[q] * di * dj + G2[q] * di * dk + G4[q] * dj * dk);
        }
        out[q] = h1 * s + h2 * bm1[q];
    }
}

// =================================================================================================
// tensor contraction helper on LDS arrays: contract axis AX of in[S2][S1][S0] with M (NOUT x NIN,
// row-major) ; out has that axis replaced by NOUT.
// =================================================================================================
template <int S0, int S1, int S2, int AX, int NOUT, bool ACC>
__device__ __forceinline__ void contract(const double *__restrict__ in, double *__restrict__ out,
                                         const double *__restrict__ M, int tid, int nth) {
    constexpr int NIN = AX == 0 ? S0 : (AX == 1 ? S1 : S2);
    constexpr int O0 = AX == 0 ? NOUT : S0, O1 = AX == 1 ? NOUT : S1, O2 = AX == 2 ? NOUT : S2;
    for (int p = tid; p < O0 * O1 * O2; p += nth) {
        const int a = p % O0, b = (p / O0) % O1, c = p / (O0 * O1);
        const int o = AX == 0 ? a : (AX == 1 ? b : c);
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < NIN; ++l) {
            const int q = AX == 0 ? (l + S0 * (b + S1 * c)) : (AX == 1 ? (a + S0 * (l + S1 * c)) : (a + S0 * (b + S1 * l)));
            s += M[o * NIN + l] * in[q];
        }
        if (ACC)
            out[p] += s;
        else
            out[p] = s;
    }
}

// ---- pressure-mesh <-> velocity-mesh operators, 3-D ------------------------------------------------------------
// Each thread owns whole 1-D columns: it loads the N2 (or N) entries of a column once into registers and produces
// all outputs of the contraction from them; the small interpolation / derivative matrices arrive as BY-VALUE kernel
// arguments, i.e. in the scalar kernarg segment, so the matrix operand of every FMA is an SGPR pair and costs no
// LDS or vector-memory traffic.  NC = velocity components processed per pass (3 when the LDS image of all three
// fits, else 1).
// elem_slot: natural ix-fastest index (FG = false) or the corner/edge/face-grouped slot fg_slot of internal.h.
template <int N, bool FG>
__device__ __forceinline__ int elem_slot(int a, int j, int k) {
    if (!FG) return a + N * (j + N * k);
    return fg_slot(N, a, j, k);
}

template <int N>
struct PMats {
    double It[N * (N - 2)];   // I12^T  (N x N2 row-major)
    double Dt[N * (N - 2)];   // D12^T
    double Im[(N - 2) * N];   // I12    (N2 x N row-major)
    double Dm[(N - 2) * N];   // D12
};

// Block size of the two pressure-mesh kernels: with all three velocity components in flight (NC = 3, lx1 <= 8) the stages
// keep 144 - 216 of 256 threads busy; with one component at a time (NC = 1, lx1 > 8: the LDS budget) only 64 - 128 of them
// have work, so those instantiations run with 128 threads per block (and twice the blocks per CU).
template <int N, int NC>
struct PBlock {
    static constexpr int NTB = (NC == 1 && N > 8) ? 128 : 256;
};

// opgradt: w_i = sum_j T_j^T (g_ji o p),  T_j = (D12 along r_j, I12 otherwise)
template <int N, int NC, bool FG, bool ML = true>
__global__ __launch_bounds__(((NC == 1 && N > 8) ? 128 : 256)) void k_opgradt3(int64_t E, PMats<N> M, CF9 g, CP4 pl, F3L wl, CP4 gatel, int nl) {
    constexpr int N2 = N - 2, NS2 = N2 * N2;
    constexpr int NT = PBlock<N, NC>::NTB;   // (shadows the file-level block size: 128 threads when one component is in flight)
    constexpr int NP2 = N2 * N2 * N2, NP1 = N * N * N;
    constexpr int SA = N2 * N2 * N, SB = N2 * N * N;
    __shared__ double sA[NC * 3][SA];
    __shared__ double sB[NC * 2][SB];
    const int tid = threadIdx.x;
    const int64_t e = blockIdx.x;
    // lanes (the vectors of a block step) one after the other: the nine metric arrays of the element (15.5 KB at lx1 = 8) come
    // from HBM for the first lane and from L1 / L2 for the others
    for (int lv = 0; lv < (ML ? nl : 1); ++lv) {   // ML = false: the single-vector kernel (a runtime lane loop costs it 25 - 30 %)
    if (gatel.p[lv] && gatel.p[lv][0] != 0.0) continue;   // this lane's PCG has converged (device-side done flag, block-uniform)
    const double *pe = pl.p[lv] + e * NP2;
    for (int c0 = 0; c0 < 3; c0 += NC) {
        if (c0 > 0 || lv > 0) __syncthreads();
        // z stage, arrays j = 0,1 (interpolation along z)
        for (int t = tid; t < NC * 2 * NS2; t += NT) {
            const int arr = t / NS2, col = t % NS2;
            const int ci = arr >> 1, j = arr & 1, i = c0 + ci;
            const double *gp = (i == 0 ? g.p[j * 3 + 0] : (i == 1 ? g.p[j * 3 + 1] : g.p[j * 3 + 2])) + e * NP2;
            double q[N2];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) q[k2] = gp[col + NS2 * k2] * pe[col + NS2 * k2];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                double a = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < N2; ++k2) a += M.It[k * N2 + k2] * q[k2];
                sA[ci * 3 + j][col + NS2 * k] = a;
            }
        }
        // z stage, array j = 2 (derivative along z)
        for (int t = tid; t < NC * NS2; t += NT) {
            const int ci = t / NS2, col = t % NS2, i = c0 + ci;
            const double *gp = (i == 0 ? g.p[6] : (i == 1 ? g.p[7] : g.p[8])) + e * NP2;
            double q[N2];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) q[k2] = gp[col + NS2 * k2] * pe[col + NS2 * k2];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                double a = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < N2; ++k2) a += M.Dt[k * N2 + k2] * q[k2];
                sA[ci * 3 + 2][col + NS2 * k] = a;
            }
        }
        __syncthreads();
        // y stage: B0 = I^T_y A0 ; B1 = D^T_y A1 + I^T_y A2     (columns over (i2, k))
        for (int t = tid; t < NC * N2 * N; t += NT) {
            const int ci = t / (N2 * N), col = t % (N2 * N);
            const int i2 = col % N2, k = col / N2;
            double a0[N2], a1[N2], a2[N2];
#pragma unroll
            for (int j2 = 0; j2 < N2; ++j2) {
                const int q = i2 + N2 * (j2 + N2 * k);
                a0[j2] = sA[ci * 3 + 0][q];
                a1[j2] = sA[ci * 3 + 1][q];
                a2[j2] = sA[ci * 3 + 2][q];
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double b0 = 0.0, b1 = 0.0;
#pragma unroll
                for (int j2 = 0; j2 < N2; ++j2) {
                    b0 += M.It[j * N2 + j2] * a0[j2];
                    b1 += M.Dt[j * N2 + j2] * a1[j2] + M.It[j * N2 + j2] * a2[j2];
                }
                sB[ci * 2 + 0][i2 + N2 * (j + N * k)] = b0;
                sB[ci * 2 + 1][i2 + N2 * (j + N * k)] = b1;
            }
        }
        __syncthreads();
        // x stage: w = D^T_x B0 + I^T_x B1, written straight to HBM (N contiguous doubles per thread)
        for (int t = tid; t < NC * N * N; t += NT) {
            const int ci = t / (N * N), bc = t % (N * N), i = c0 + ci;
            double b0[N2], b1[N2];
#pragma unroll
            for (int i2 = 0; i2 < N2; ++i2) {
                b0[i2] = sB[ci * 2 + 0][i2 + N2 * bc];
                b1[i2] = sB[ci * 2 + 1][i2 + N2 * bc];
            }
            double *wp = wl.p[lv][i] + e * NP1;
            const int jj = bc % N, kk = bc / N;
#pragma unroll
            for (int a = 0; a < N; ++a) {
                double v = 0.0;
#pragma unroll
                for (int i2 = 0; i2 < N2; ++i2) v += M.Dt[a * N2 + i2] * b0[i2] + M.It[a * N2 + i2] * b1[i2];
                wp[elem_slot<N, FG>(a, jj, kk)] = v;
            }
        }
    }
    }
}

// opdiv: out = scale * sum_i sum_j g_ji o (T_j (wt_i o u_i)); wt (may hold nulls) fuses mask * binvm1 into the load
// `pdot`/`part` (may be null): first-stage sums of the surrounding PCG, part[e] = sum_q pdot_q out_q and
// part[E + e] = sum_q out_q over the element -- saves a separate pass over two pressure-mesh vectors.
template <int N, int NC, bool FG, bool ML = true>
__global__ __launch_bounds__(((NC == 1 && N > 8) ? 128 : 256)) void k_opdiv3(int64_t E, PMats<N> M, CF9 g, CF3L ul, CF3 wt, P4 outl,
                                               double scale, CP4 pdotl, P4 partl, CP4 gatel, int nl) {
    constexpr int N2 = N - 2, NS2 = N2 * N2;
    constexpr int NT = PBlock<N, NC>::NTB;
    constexpr int NP2 = N2 * N2 * N2, NP1 = N * N * N;
    constexpr int SB = N2 * N * N, SC = N2 * N2 * N;
    constexpr int SBTOT = NC * 2 * SB, SPTOT = NC * 3 * NP2;
    __shared__ double sBP[SBTOT > SPTOT ? SBTOT : SPTOT];   // x-stage output, later the per-array products
    __shared__ double sC[NC * 3][SC];
    const int tid = threadIdx.x;
    const int64_t e = blockIdx.x;
    constexpr int NACC = (NP2 + NT - 1) / NT;
    // lanes one after the other: weights (mask * binvm1) and metric arrays of the element are re-read from L1 / L2
    for (int lv = 0; lv < (ML ? nl : 1); ++lv) {   // ML = false: the single-vector kernel (a runtime lane loop costs it 25 - 30 %)
    if (gatel.p[lv] && gatel.p[lv][0] != 0.0) continue;
    double *__restrict__ out = outl.p[lv];
    const double *__restrict__ pdot = pdotl.p[lv];
    double *__restrict__ part = partl.p[lv];
    double acc[NACC];
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[r] = 0.0;
    for (int c0 = 0; c0 < 3; c0 += NC) {
        __syncthreads();
        // x stage from HBM: B0 = D_x u, B1 = I_x u   (columns over (j, k); N contiguous doubles per thread)
        for (int t = tid; t < NC * N * N; t += NT) {
            const int ci = t / (N * N), bc = t % (N * N), i = c0 + ci;
            const double *up = ul.p[lv][i] + e * NP1;
            const double *wp = (i == 0 ? wt.p[0] : (i == 1 ? wt.p[1] : wt.p[2]));
            const int jj = bc % N, kk = bc / N;
            int sl[N];
#pragma unroll
            for (int a = 0; a < N; ++a) sl[a] = elem_slot<N, FG>(a, jj, kk);
            double uu[N];
#pragma unroll
            for (int a = 0; a < N; ++a) uu[a] = up[sl[a]];
            if (wp) {
                wp += e * NP1;
#pragma unroll
                for (int a = 0; a < N; ++a) uu[a] *= wp[sl[a]];
            }
#pragma unroll
            for (int i2 = 0; i2 < N2; ++i2) {
                double b0 = 0.0, b1 = 0.0;
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    b0 += M.Dm[i2 * N + a] * uu[a];
                    b1 += M.Im[i2 * N + a] * uu[a];
                }
                sBP[(ci * 2 + 0) * SB + i2 + N2 * bc] = b0;
                sBP[(ci * 2 + 1) * SB + i2 + N2 * bc] = b1;
            }
        }
        __syncthreads();
        // y stage: C0 = I_y B0 ; C1 = D_y B1 ; C2 = I_y B1     (columns over (i2, k))
        for (int t = tid; t < NC * N2 * N; t += NT) {
            const int ci = t / (N2 * N), col = t % (N2 * N);
            const int i2 = col % N2, k = col / N2;
            double b0[N], b1[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                b0[j] = sBP[(ci * 2 + 0) * SB + i2 + N2 * (j + N * k)];
                b1[j] = sBP[(ci * 2 + 1) * SB + i2 + N2 * (j + N * k)];
            }
#pragma unroll
            for (int j2 = 0; j2 < N2; ++j2) {
                double c0v = 0.0, c1v = 0.0, c2v = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    c0v += M.Im[j2 * N + j] * b0[j];
                    c1v += M.Dm[j2 * N + j] * b1[j];
                    c2v += M.Im[j2 * N + j] * b1[j];
                }
                const int q = i2 + N2 * (j2 + N2 * k);
                sC[ci * 3 + 0][q] = c0v;
                sC[ci * 3 + 1][q] = c1v;
                sC[ci * 3 + 2][q] = c2v;
            }
        }
        __syncthreads();
        // z stage fused with the metric product: arrays j = 0,1 use I_z, array j = 2 uses D_z; products go to LDS
        for (int t = tid; t < NC * 2 * NS2; t += NT) {
            const int arr = t / NS2, col = t % NS2;
            const int ci = arr >> 1, j = arr & 1, i = c0 + ci;
            const double *gp = (i == 0 ? g.p[j * 3 + 0] : (i == 1 ? g.p[j * 3 + 1] : g.p[j * 3 + 2])) + e * NP2;
            double cc[N];
#pragma unroll
            for (int k = 0; k < N; ++k) cc[k] = sC[ci * 3 + j][col + NS2 * k];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) a += M.Im[k2 * N + k] * cc[k];
                sBP[(ci * 3 + j) * NP2 + col + NS2 * k2] = gp[col + NS2 * k2] * a;
            }
        }
        for (int t = tid; t < NC * NS2; t += NT) {
            const int ci = t / NS2, col = t % NS2, i = c0 + ci;
            const double *gp = (i == 0 ? g.p[6] : (i == 1 ? g.p[7] : g.p[8])) + e * NP2;
            double cc[N];
#pragma unroll
            for (int k = 0; k < N; ++k) cc[k] = sC[ci * 3 + 2][col + NS2 * k];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) a += M.Dm[k2 * N + k] * cc[k];
                sBP[(ci * 3 + 2) * NP2 + col + NS2 * k2] = gp[col + NS2 * k2] * a;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int q = tid + r * NT;
            if (q < NP2) {
                double a = 0.0;
#pragma unroll
                for (int arr = 0; arr < NC * 3; ++arr) a += sBP[arr * NP2 + q];
                acc[r] += a;
            }
        }
    }
    double spw = 0.0, sw = 0.0;
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
        const int q = tid + r * NT;
        if (q < NP2) {
            const double v = scale * acc[r];
            out[e * NP2 + q] = v;
            if (part) {
                spw += pdot[e * NP2 + q] * v;
                sw += v;
            }
        }
    }
    if (part) {
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            spw += __shfl_down(spw, o, 64);
            sw += __shfl_down(sw, o, 64);
        }
        double *red = sBP;
        if ((tid & 63) == 0) {
            red[tid >> 6] = spw;
            red[4 + (tid >> 6)] = sw;
        }
        __syncthreads();
        if (tid == 0) {
            double a = 0.0, b = 0.0;
            for (int q = 0; q < NT / 64; ++q) {
                a += red[q];
                b += red[4 + q];
            }
            part[e] = a;
            part[E + e] = b;
        }
    }
    __syncthreads();
    }
}

// ---- lx1 > 8 variants: one velocity component at a time through ONE LDS array ------------------------------------
// With NC = 1 the kernels above need sA (3 N2^2 N) + sB (2 N2 N^2) doubles = 52 KB at lx1 = 12: three 128-thread blocks per CU,
// 1.5 waves per SIMD, nothing to hide the global-load -> LDS -> barrier chain behind (14 - 31 % of the HBM roofline measured).
// Here the y stage runs IN PLACE: the LDS array is organised in regions, one per (i2, k) column of the y stage, of 3 N2 slots
// (+1 pad).  The z stage of opgradt fills slots [j N2 + j2] (arrays A0 | A1 | A2); the thread that owns the region reads its
// 3 N2 values into registers and overwrites them with the 2 N <= 3 N2 values B0 | B1 (slots [j], [N + j]); the x stage gathers
// its columns across regions.  opdiv runs the same three stages backwards and keeps the sum over arrays and components of a
// pressure point in the registers of the thread that owns its z column, so the per-array products never go through LDS.
// 30 KB (lx1 = 12) / 16 KB (lx1 = 10) per block: 5 / 10 blocks per CU.  Also used at lx1 = 8 with ONE wave per element
// (64 x-stage columns = 64 lanes, 7.3 KB, 77 VGPRs: 21 waves per CU, no cross-wave barrier): 81 / 87 us instead of 86 / 93 us
// for k_opgradt3 / k_opdiv3<8, 3> with their 256-thread blocks.  The 1-D matrices come from global memory through
// wave-uniform scalar loads (compile-time indices on `const __restrict__` kernel arguments): as by-value arguments the 4 x 120
// doubles of lx1 = 12 need 960 SGPRs, i.e. 500 of them spilled to VGPR lanes and one v_readlane per FMA.
template <int N>
struct PBlockN {
    static constexpr int NTB = (N * N <= 64) ? 64 : ((N * N <= 128) ? 128 : 192);
    static constexpr int RS = (3 * (N - 2)) | 1;   // odd region stride: the columns of a stage fall on different banks
};

// PUpd: the direction update of the surrounding PCG, p <- (z - zmean) + beta p, performed while the kernel loads p (per lane;
// z == null: none).  The threads that own the z columns are the only readers of p, so they also write it back.
struct PUpd {
    const double *z[4], *beta[4], *zmean[4];
    double *p[4];
};
struct NoPUpd {};   // PU = false: the kernel has no trace of the update (its mere presence cost k_opgradt3n<12> 40 %)
template <int N, bool FG, bool ML = true, bool PU = false>
__global__ __launch_bounds__(PBlockN<N>::NTB) void k_opgradt3n(int64_t E, const double *__restrict__ mIt, const double *__restrict__ mDt, const int *__restrict__ fgtab, CF9 g, CP4 pl, F3L wl, CP4 gatel, int nl,
                                                               std::conditional_t<PU, PUpd, NoPUpd> pu) {
    constexpr int N2 = N - 2, NS2 = N2 * N2, NP2 = NS2 * N2, NP1 = N * N * N;
    constexpr int RS = PBlockN<N>::RS;
    static_assert(2 * N <= 3 * N2, "in-place y stage needs 2 N <= 3 N2");
    __shared__ double sR[N2 * N * RS];
    const int tid = threadIdx.x;
    const int64_t e = blockIdx.x;
    for (int lv = 0; lv < (ML ? nl : 1); ++lv) {   // ML = false: the single-vector kernel (a runtime lane loop costs it 25 - 30 %)
        if (gatel.p[lv] && gatel.p[lv][0] != 0.0) continue;
        const double *pe = pl.p[lv] + e * NP2;
        double pv[N2];
        if (tid < NS2) {
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) pv[k2] = pe[tid + NS2 * k2];
            if constexpr (PU) if (pu.z[lv]) {
                const double beta = pu.beta[lv][0], zmean = pu.zmean[lv][0];
                const double *ze = pu.z[lv] + e * NP2;
                double *po = pu.p[lv] + e * NP2;
                double zv[N2];   // loads first, stores afterwards (alternating, they serialise: the compiler must assume aliasing)
#pragma unroll
                for (int k2 = 0; k2 < N2; ++k2) zv[k2] = ze[tid + NS2 * k2];
#pragma unroll
                for (int k2 = 0; k2 < N2; ++k2) {
                    pv[k2] = (zv[k2] - zmean) + beta * pv[k2];
                    po[tid + NS2 * k2] = pv[k2];
                }
            }
        }
        // lx1 <= 10: the metric columns of the NEXT component are requested before the y and x stages of the current one, so that
        // their latency is hidden behind two LDS stages
        constexpr bool PF = N <= 10;   // (measured with LDS-only barriers: lx1 = 10 190 -> 175 us, lx1 = 12 347 -> 414 us)
        double gq[PF ? 3 : 1][PF ? N2 : 1];
        if (PF && tid < NS2) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k2 = 0; k2 < N2; ++k2) gq[PF ? j : 0][PF ? k2 : 0] = g.p[j * 3 + 0][e * NP2 + tid + NS2 * k2];
        }
        for (int i = 0; i < 3; ++i) {
            if (i > 0 || lv > 0) lds_barrier();   // the x stage of the previous pass has read its columns
            // z stage: thread = (i2, j2) column, all three arrays
            if (tid < NS2) {
                const int i2 = tid % N2, j2 = tid / N2;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double *gp = g.p[j * 3 + i] + e * NP2;
                    double q[N2];
#pragma unroll
                    for (int k2 = 0; k2 < N2; ++k2) q[k2] = (PF ? gq[PF ? j : 0][PF ? k2 : 0] : gp[tid + NS2 * k2]) * pv[k2];
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        double a = 0.0;
#pragma unroll
                        for (int k2 = 0; k2 < N2; ++k2) a += (j < 2 ? mIt[k * N2 + k2] : mDt[k * N2 + k2]) * q[k2];
                        sR[(i2 + N2 * k) * RS + j * N2 + j2] = a;
                    }
                }
                if (PF && i < 2) {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int k2 = 0; k2 < N2; ++k2) gq[PF ? j : 0][PF ? k2 : 0] = g.p[j * 3 + i + 1][e * NP2 + tid + NS2 * k2];
                }
            }
            lds_barrier();
            // y stage in place: B0 = I^T_y A0 ; B1 = D^T_y A1 + I^T_y A2
            if (tid < N2 * N) {
                double *r = sR + tid * RS;
                double a0[N2], a1[N2], a2[N2];
#pragma unroll
                for (int j2 = 0; j2 < N2; ++j2) {
                    a0[j2] = r[j2];
                    a1[j2] = r[N2 + j2];
                    a2[j2] = r[2 * N2 + j2];
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double b0 = 0.0, b1 = 0.0;
#pragma unroll
                    for (int j2 = 0; j2 < N2; ++j2) {
                        b0 += mIt[j * N2 + j2] * a0[j2];
                        b1 += mDt[j * N2 + j2] * a1[j2] + mIt[j * N2 + j2] * a2[j2];
                    }
                    r[j] = b0;
                    r[N + j] = b1;
                }
            }
            lds_barrier();
            // x stage: w = D^T_x B0 + I^T_x B1, straight to HBM
            if (tid < N * N) {
                const int jj = tid % N, kk = tid / N;
                double b0[N2], b1[N2];
#pragma unroll
                for (int i2 = 0; i2 < N2; ++i2) {
                    b0[i2] = sR[(i2 + N2 * kk) * RS + jj];
                    b1[i2] = sR[(i2 + N2 * kk) * RS + N + jj];
                }
                double *wp = wl.p[lv][i] + e * NP1;
                int sl[N];   // face-grouped slots from a table: computing them branches on the boundary class of every point
#pragma unroll
                for (int a = 0; a < N; ++a) sl[a] = FG ? fgtab[a + N * tid] : a + N * tid;
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    double v = 0.0;
#pragma unroll
                    for (int i2 = 0; i2 < N2; ++i2) v += mDt[a * N2 + i2] * b0[i2] + mIt[a * N2 + i2] * b1[i2];
                    wp[sl[a]] = v;
                }
            }
        }
    }
}

template <int N, bool FG, bool ML = true>
__global__ __launch_bounds__(PBlockN<N>::NTB) void k_opdiv3n(int64_t E, const double *__restrict__ mIm, const double *__restrict__ mDm, const int *__restrict__ fgtab, CF9 g, CF3L ul, CF3 wt, P4 outl, double scale,
                                                             CP4 pdotl, P4 partl, CP4 gatel, int nl, const unsigned char *__restrict__ mb) {
    constexpr int N2 = N - 2, NS2 = N2 * N2, NP2 = NS2 * N2, NP1 = N * N * N;
    constexpr int RS = PBlockN<N>::RS, NTB = PBlockN<N>::NTB;
    __shared__ double sR[N2 * N * RS];
    __shared__ double red[2 * (NTB / 64)];
    const int tid = threadIdx.x;
    const int64_t e = blockIdx.x;   // (elements in REVERSE order, to meet what the gradient kernel left in the Infinity Cache: this kernel 3.80 -> 3.47 ms per step, the step unchanged -- the next kernel pays; not adopted)
    for (int lv = 0; lv < (ML ? nl : 1); ++lv) {   // ML = false: the single-vector kernel (a runtime lane loop costs it 25 - 30 %)
        if (gatel.p[lv] && gatel.p[lv][0] != 0.0) continue;
        double acc[N2];
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) acc[k2] = 0.0;
        double wwk[N];        // compact weights (mb): binvm1 of the thread's row and its N mask nibbles, for all three passes
        uint64_t mkk = 0u;
        constexpr bool PF = N <= 8;   // lx1 <= 8: the metric columns of a pass are requested at its start, two LDS stages before their use (no gain at lx1 = 10, a loss at 12)
        for (int i = 0; i < 3; ++i) {
            if (i > 0 || lv > 0) lds_barrier();   // the z stage of the previous pass has read its columns
            double gq[PF ? 3 : 1][PF ? N2 : 1];
            if (PF && tid < NS2) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int k2 = 0; k2 < N2; ++k2) gq[PF ? j : 0][PF ? k2 : 0] = g.p[j * 3 + i][e * NP2 + tid + NS2 * k2];
            }
            // x stage from HBM: B0 = D_x u, B1 = I_x u
            if (tid < N * N) {
                const int jj = tid % N, kk = tid / N;
                const double *up = ul.p[lv][i] + e * NP1;
                const double *wp = mb ? wt.p[0] : wt.p[i];
                int sl[N];
#pragma unroll
                for (int a = 0; a < N; ++a) sl[a] = FG ? fgtab[a + N * tid] : a + N * tid;
                double uu[N];
#pragma unroll
                for (int a = 0; a < N; ++a) uu[a] = up[sl[a]];
                if (mb) {   // weight = binvm1 where bit i of the point's mask byte is set, 0 elsewhere (= mask_i * binvm1)
                    if (i == 0) {   // the row's weights and mask bytes are loaded once and kept for the three components
                        const double *wq = wp + e * NP1;
                        const unsigned char *mp = mb + e * NP1;
                        mkk = 0u;
#pragma unroll
                        for (int a = 0; a < N; ++a) {
                            wwk[a] = wq[sl[a]];
                            mkk |= (uint64_t)(mp[sl[a]] & 7u) << (4 * a);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < N; ++a) uu[a] *= ((mkk >> (4 * a + i)) & 1u) ? wwk[a] : 0.0;
                } else if (wp) {
                    wp += e * NP1;
#pragma unroll
                    for (int a = 0; a < N; ++a) uu[a] *= wp[sl[a]];
                }
#pragma unroll
                for (int i2 = 0; i2 < N2; ++i2) {
                    double b0 = 0.0, b1 = 0.0;
#pragma unroll
                    for (int a = 0; a < N; ++a) {
                        b0 += mDm[i2 * N + a] * uu[a];
                        b1 += mIm[i2 * N + a] * uu[a];
                    }
                    sR[(i2 + N2 * kk) * RS + jj] = b0;
                    sR[(i2 + N2 * kk) * RS + N + jj] = b1;
                }
            }
            lds_barrier();
            // y stage in place: C0 = I_y B0 ; C1 = D_y B1 ; C2 = I_y B1
            if (tid < N2 * N) {
                double *r = sR + tid * RS;
                double b0[N], b1[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    b0[j] = r[j];
                    b1[j] = r[N + j];
                }
#pragma unroll
                for (int j2 = 0; j2 < N2; ++j2) {
                    double c0v = 0.0, c1v = 0.0, c2v = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        c0v += mIm[j2 * N + j] * b0[j];
                        c1v += mDm[j2 * N + j] * b1[j];
                        c2v += mIm[j2 * N + j] * b1[j];
                    }
                    r[j2] = c0v;
                    r[N2 + j2] = c1v;
                    r[2 * N2 + j2] = c2v;
                }
            }
            lds_barrier();
            // z stage: thread = (i2, j2) column; the metric products of the three arrays are summed in registers
            if (tid < NS2) {
                const int i2 = tid % N2, j2 = tid / N2;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double *gp = g.p[j * 3 + i] + e * NP2;
                    double cc[N];
#pragma unroll
                    for (int k = 0; k < N; ++k) cc[k] = sR[(i2 + N2 * k) * RS + j * N2 + j2];
#pragma unroll
                    for (int k2 = 0; k2 < N2; ++k2) {
                        double a = 0.0;
#pragma unroll
                        for (int k = 0; k < N; ++k) a += (j < 2 ? mIm[k2 * N + k] : mDm[k2 * N + k]) * cc[k];
                        acc[k2] += (PF ? gq[PF ? j : 0][PF ? k2 : 0] : gp[tid + NS2 * k2]) * a;
                    }
                }
            }
        }
        double *__restrict__ out = outl.p[lv];
        const double *__restrict__ pdot = pdotl.p[lv];
        double *__restrict__ part = partl.p[lv];
        double spw = 0.0, sw = 0.0;
        if (tid < NS2) {
            double pd[N2];   // loads before the stores (see k_axhelm3r)
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) pd[k2] = part ? pdot[e * NP2 + tid + NS2 * k2] : 0.0;
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                const double v = scale * acc[k2];
                out[e * NP2 + tid + NS2 * k2] = v;
                if (part) {
                    spw += pd[k2] * v;
                    sw += v;
                }
            }
        }
        if (part) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                spw += __shfl_down(spw, o, 64);
                sw += __shfl_down(sw, o, 64);
            }
            if ((tid & 63) == 0) {
                red[tid >> 6] = spw;
                red[NTB / 64 + (tid >> 6)] = sw;
            }
            lds_barrier();
            if (tid == 0) {
                double a = 0.0, b = 0.0;
                for (int q = 0; q < NTB / 64; ++q) {
                    a += red[q];
                    b += red[NTB / 64 + q];
                }
                part[e] = a;
                part[E + e] = b;
            }
        }
    }
}

// ---- small-mesh variants (strong-scaling regime: config 3 on 8 GPUs is 1300 elements per GPU = 1.3 one-wave blocks per SIMD) ----
// k_opgradt3n / k_opdiv3n send ONE wave per element through the three velocity components one after the other: nine dependent
// LDS stages and three rounds of metric loads per element, and with one or two waves per SIMD nothing hides them (16.5 us per launch
// at 1300 elements for 38 - 54 MB, profiles/r04_E1300_counters.txt).  Here a block is THREE waves, wave c owns component c (its own
// LDS array): the chain is a third as long.  opgradt: every wave loads (and, with the fused direction update, forms) the same
// pressure column -- the element's 216 values come from L2 twice more, wave 0 alone writes the updated direction back.  opdiv: the
// three partial sums of a pressure point meet in LDS and are added in the fixed order (c = 0, 1, 2) by wave 0, which also writes the
// result and the element's contribution to (p, E p).  Single-vector launches only; selected by the local element count
// (sem_small_mesh; NLG_SMALL_E).  lx1 <= 8 (one wave per stage).
template <int N, bool FG, bool PU>
__global__ __launch_bounds__(192) void k_opgradt3w(int64_t E, const double *__restrict__ mIt, const double *__restrict__ mDt, const int *__restrict__ fgtab, CF9 g,
                                                   const double *__restrict__ p, F3 w, const double *__restrict__ gate, std::conditional_t<PU, PUpd, NoPUpd> pu) {
    constexpr int N2 = N - 2, NS2 = N2 * N2, NP2 = NS2 * N2, NP1 = N * N * N;
    constexpr int RS = PBlockN<N>::RS;
    static_assert(N * N <= 64 && 2 * N <= 3 * N2, "one wave per stage; in-place y stage needs 2 N <= 3 N2");
    __shared__ double sRR[3][N2 * N * RS];
    if (gate && gate[0] != 0.0) return;
    const int tid = threadIdx.x & 63, i = threadIdx.x >> 6;   // i = component of this wave
    double *sR = sRR[i];
    const int64_t e = blockIdx.x;
    const double *pe = p + e * NP2;
    double pv[N2];
    double gq[3][N2];
    if (tid < NS2) {
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) pv[k2] = pe[tid + NS2 * k2];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) gq[j][k2] = g.p[j * 3 + i][e * NP2 + tid + NS2 * k2];
        if constexpr (PU) if (pu.z[0]) {
            const double beta = pu.beta[0][0], zmean = pu.zmean[0][0];
            const double *ze = pu.z[0] + e * NP2;
            double zv[N2];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) zv[k2] = ze[tid + NS2 * k2];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) pv[k2] = (zv[k2] - zmean) + beta * pv[k2];
        }
    }
    if constexpr (PU) {
        // the updated direction is written IN PLACE over the array the other two waves are still reading: every wave's loads of p must have
        // RETURNED before wave 0 stores (a full barrier with the vector-memory counter drained, not an LDS-only one).  Without it the race
        // is silent -- a wave that reads the updated p applies the update twice, the PCG still converges, only more slowly: 12.75 -> 26.75
        // pressure iterations per time step in the 4-rank rehearsal, which is how it was found
        __syncthreads();
        if (pu.z[0] && i == 0 && tid < NS2) {
            double *po = pu.p[0] + e * NP2;
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) po[tid + NS2 * k2] = pv[k2];
        }
    }
    if (tid < NS2) {
        const int i2 = tid % N2, j2 = tid / N2;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double q[N2];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) q[k2] = gq[j][k2] * pv[k2];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                double a = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < N2; ++k2) a += (j < 2 ? mIt[k * N2 + k2] : mDt[k * N2 + k2]) * q[k2];
                sR[(i2 + N2 * k) * RS + j * N2 + j2] = a;
            }
        }
    }
    lds_barrier();
    if (tid < N2 * N) {
        double *r = sR + tid * RS;
        double a0[N2], a1[N2], a2[N2];
#pragma unroll
        for (int j2 = 0; j2 < N2; ++j2) {
            a0[j2] = r[j2];
            a1[j2] = r[N2 + j2];
            a2[j2] = r[2 * N2 + j2];
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double b0 = 0.0, b1 = 0.0;
#pragma unroll
            for (int j2 = 0; j2 < N2; ++j2) {
                b0 += mIt[j * N2 + j2] * a0[j2];
                b1 += mDt[j * N2 + j2] * a1[j2] + mIt[j * N2 + j2] * a2[j2];
            }
            r[j] = b0;
            r[N + j] = b1;
        }
    }
    lds_barrier();
    if (tid < N * N) {
        const int jj = tid % N, kk = tid / N;
        double b0[N2], b1[N2];
#pragma unroll
        for (int i2 = 0; i2 < N2; ++i2) {
            b0[i2] = sR[(i2 + N2 * kk) * RS + jj];
            b1[i2] = sR[(i2 + N2 * kk) * RS + N + jj];
        }
        double *wp = w.p[i] + e * NP1;
        int sl[N];
#pragma unroll
        for (int a = 0; a < N; ++a) sl[a] = FG ? fgtab[a + N * tid] : a + N * tid;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            double v = 0.0;
#pragma unroll
            for (int i2 = 0; i2 < N2; ++i2) v += mDt[a * N2 + i2] * b0[i2] + mIt[a * N2 + i2] * b1[i2];
            wp[sl[a]] = v;
        }
    }
}

template <int N, bool FG>
__global__ __launch_bounds__(192) void k_opdiv3w(int64_t E, const double *__restrict__ mIm, const double *__restrict__ mDm, const int *__restrict__ fgtab, CF9 g, CF3 u, CF3 wt,
                                                 double *__restrict__ out, double scale, const double *__restrict__ pdot, double *__restrict__ part, const double *__restrict__ gate) {
    constexpr int N2 = N - 2, NS2 = N2 * N2, NP2 = NS2 * N2, NP1 = N * N * N;
    constexpr int RS = PBlockN<N>::RS;
    static_assert(N * N <= 64, "one wave per stage");
    __shared__ double sRR[3][N2 * N * RS];
    __shared__ double sAcc[2][N2][64];
    if (gate && gate[0] != 0.0) return;
    const int tid = threadIdx.x & 63, i = threadIdx.x >> 6;
    double *sR = sRR[i];
    const int64_t e = blockIdx.x;
    double acc[N2];
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) acc[k2] = 0.0;
    double gq[3][N2];
    if (tid < NS2) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) gq[j][k2] = g.p[j * 3 + i][e * NP2 + tid + NS2 * k2];
    }
    double pd[N2];   // wave 0: the pressure direction for (p, E p), requested with everything else
    if (i == 0 && tid < NS2) {
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) pd[k2] = part ? pdot[e * NP2 + tid + NS2 * k2] : 0.0;
    }
    if (tid < N * N) {
        const int jj = tid % N, kk = tid / N;
        const double *up = u.p[i] + e * NP1;
        const double *wp = wt.p[i];
        int sl[N];
#pragma unroll
        for (int a = 0; a < N; ++a) sl[a] = FG ? fgtab[a + N * tid] : a + N * tid;
        double uu[N];
#pragma unroll
        for (int a = 0; a < N; ++a) uu[a] = up[sl[a]];
        if (wp) {
            wp += e * NP1;
#pragma unroll
            for (int a = 0; a < N; ++a) uu[a] *= wp[sl[a]];
        }
#pragma unroll
        for (int i2 = 0; i2 < N2; ++i2) {
            double b0 = 0.0, b1 = 0.0;
#pragma unroll
            for (int a = 0; a < N; ++a) {
                b0 += mDm[i2 * N + a] * uu[a];
                b1 += mIm[i2 * N + a] * uu[a];
            }
            sR[(i2 + N2 * kk) * RS + jj] = b0;
            sR[(i2 + N2 * kk) * RS + N + jj] = b1;
        }
    }
    lds_barrier();
    if (tid < N2 * N) {
        double *r = sR + tid * RS;
        double b0[N], b1[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            b0[j] = r[j];
            b1[j] = r[N + j];
        }
#pragma unroll
        for (int j2 = 0; j2 < N2; ++j2) {
            double c0v = 0.0, c1v = 0.0, c2v = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                c0v += mIm[j2 * N + j] * b0[j];
                c1v += mDm[j2 * N + j] * b1[j];
                c2v += mIm[j2 * N + j] * b1[j];
            }
            r[j2] = c0v;
            r[N2 + j2] = c1v;
            r[2 * N2 + j2] = c2v;
        }
    }
    lds_barrier();
    if (tid < NS2) {
        const int i2 = tid % N2, j2 = tid / N2;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double cc[N];
#pragma unroll
            for (int k = 0; k < N; ++k) cc[k] = sR[(i2 + N2 * k) * RS + j * N2 + j2];
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) a += (j < 2 ? mIm[k2 * N + k] : mDm[k2 * N + k]) * cc[k];
                acc[k2] += gq[j][k2] * a;
            }
        }
        if (i > 0) {
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) sAcc[i - 1][k2][tid] = acc[k2];
        }
    }
    lds_barrier();
    if (i != 0) return;
    double spw = 0.0, sw = 0.0;
    if (tid < NS2) {
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            const double v = scale * ((acc[k2] + sAcc[0][k2][tid]) + sAcc[1][k2][tid]);
            out[e * NP2 + tid + NS2 * k2] = v;
            spw += pd[k2] * v;
            sw += v;
        }
    }
    if (part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            spw += __shfl_down(spw, o, 64);
            sw += __shfl_down(sw, o, 64);
        }
        if (tid == 0) {
            part[e] = spw;
            part[E + e] = sw;
        }
    }
}

// ---- 2-D versions (several elements per block would be faster; correctness first) -------------------
template <int N>
__global__ __launch_bounds__(NT) void k_opgradt2(int64_t E, const double *__restrict__ Itg,
                                                 const double *__restrict__ Dtg, CF9 g, const double *__restrict__ p,
                                                 F3 w) {
    constexpr int N2 = N - 2;
    constexpr int NP2 = N2 * N2, NP1 = N * N, SA = N2 * N;
    constexpr int EPB = NT / NP1 > 0 ? NT / NP1 : 1;
    __shared__ double sI[N * N2], sDt[N * N2];
    __shared__ double sQ[EPB][2][NP2];
    __shared__ double sA[EPB][2][SA];
    const int tid = threadIdx.x;
    const int le = tid / NP1, lt = tid % NP1;
    const int64_t e = (int64_t)blockIdx.x * EPB + le;
    const bool act = (le < EPB) && (e < E);
    for (int q = tid; q < N * N2; q += NT) {
        sI[q] = Itg[q];
        sDt[q] = Dtg[q];
    }
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
        if (act && lt < NP2) {
            const double pv = p[e * NP2 + lt];
            sQ[le][0][lt] = g.p[0 * 2 + i][e * NP2 + lt] * pv;
            sQ[le][1][lt] = g.p[1 * 2 + i][e * NP2 + lt] * pv;
        }
        __syncthreads();
        // y stage: A0 = I^T_y q0 ; A1 = D^T_y q1   -> (a2, b)
        if (act && lt < SA) {
            const int a = lt % N2, b = lt / N2;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int l = 0; l < N2; ++l) {
                s0 += sI[b * N2 + l] * sQ[le][0][a + N2 * l];
                s1 += sDt[b * N2 + l] * sQ[le][1][a + N2 * l];
            }
            sA[le][0][lt] = s0;
            sA[le][1][lt] = s1;
        }
        __syncthreads();
        if (act) {
            const int a = lt % N, b = lt / N;
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < N2; ++l) s += sDt[a * N2 + l] * sA[le][0][l + N2 * b] + sI[a * N2 + l] * sA[le][1][l + N2 * b];
            w.p[i][e * NP1 + lt] = s;
        }
    }
}

template <int N>
__global__ __launch_bounds__(NT) void k_opdiv2(int64_t E, const double *__restrict__ Img,
                                               const double *__restrict__ Dmg, CF9 g, CF3 u, CF3 wt,
                                               double *__restrict__ out, double scale, const double *__restrict__ pdot,
                                               double *__restrict__ part, const double *__restrict__ gate) {
    constexpr int N2 = N - 2;
    constexpr int NP2 = N2 * N2, NP1 = N * N, SB = N2 * N;
    constexpr int EPB = NT / NP1 > 0 ? NT / NP1 : 1;
    __shared__ double sI[N2 * N], sD[N2 * N];
    __shared__ double spw[2][NT / 64];
    if (gate && gate[0] != 0.0) return;   // the surrounding PCG has converged
    __shared__ double sU[EPB][NP1];
    __shared__ double sB[EPB][2][SB];
    const int tid = threadIdx.x;
    const int le = tid / NP1, lt = tid % NP1;
    const int64_t e = (int64_t)blockIdx.x * EPB + le;
    const bool act = (le < EPB) && (e < E);
    for (int q = tid; q < N * N2; q += NT) {
        sI[q] = Img[q];
        sD[q] = Dmg[q];
    }
    double acc = 0.0;
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
        if (act) sU[le][lt] = u.p[i][e * NP1 + lt] * (wt.p[i] ? wt.p[i][e * NP1 + lt] : 1.0);
        __syncthreads();
        if (act && lt < SB) {
            const int a = lt % N2, b = lt / N2;   // (a2, b)
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) {
                s0 += sD[a * N + l] * sU[le][l + N * b];
                s1 += sI[a * N + l] * sU[le][l + N * b];
            }
            sB[le][0][lt] = s0;
            sB[le][1][lt] = s1;
        }
        __syncthreads();
        if (act && lt < NP2) {
            const int a = lt % N2, b = lt / N2;
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int l = 0; l < N; ++l) {
                t0 += sI[b * N + l] * sB[le][0][a + N2 * l];
                t1 += sD[b * N + l] * sB[le][1][a + N2 * l];
            }
            const int64_t gq = e * NP2 + lt;
            acc += g.p[0 * 2 + i][gq] * t0 + g.p[1 * 2 + i][gq] * t1;
        }
    }
    double s1 = 0.0, s2 = 0.0;
    if (act && lt < NP2) {
        const double o = scale * acc;
        out[e * NP2 + lt] = o;
        if (part) {
            s1 = pdot[e * NP2 + lt] * o;
            s2 = o;
        }
    }
    if (part) {   // first-stage sums of the surrounding PCG, as in the 3-D kernel: part[b] = sum pdot out, part[nb + b] = sum out
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_down(s1, o, 64);
            s2 += __shfl_down(s2, o, 64);
        }
        const int lane = tid & 63, wid = tid >> 6;
        if (lane == 0) {
            spw[0][wid] = s1;
            spw[1][wid] = s2;
        }
        __syncthreads();
        if (tid == 0) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < NT / 64; ++w) {
                a += spw[0][w];
                b += spw[1][w];
            }
            part[blockIdx.x] = a;
            part[gridDim.x + blockIdx.x] = b;
        }
    }
}

// ---- pointwise kernels of the convective term on the fine mesh ------------------------------------
// Ur_j = sum_m rstdw[j][m] Uf_m
__global__ void k_conv_ur(int dim, int64_t n, CF9 rd, CF3 Uf, F3 Ur) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        for (int j = 0; j < dim; ++j) {
            double s = 0.0;
            for (int m = 0; m < dim; ++m) s += rd.p[j * dim + m][q] * Uf.p[m][q];
            Ur.p[j][q] = s;
        }
    }
}
// GU[i][m] = sum_j rstdw[j][m] dU_i/dr_j   (dU holds d/dr_j of one component i: 3 fields)
__global__ void k_conv_gu(int dim, int64_t n, CF9 rd, CF3 dUi, F3 GUi) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        for (int m = 0; m < dim; ++m) {
            double s = 0.0;
            for (int j = 0; j < dim; ++j) s += rd.p[j * dim + m][q] * dUi.p[j][q];
            GUi.p[m][q] = s;
        }
    }
}
// acc = sgn * sum_j Ur_j du_j + sum_m uf_m GUsel_m
__global__ void k_conv_combine(int dim, int64_t n, CF3 Ur, CF3 du, CF3 uf, CF3 GUsel, double sgn, double *acc) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0, t = 0.0;
        for (int j = 0; j < dim; ++j) {
            s += Ur.p[j][q] * du.p[j][q];
            t += uf.p[j][q] * GUsel.p[j][q];
        }
        acc[q] = sgn * s + t;
    }
}

__global__ void k_conv_combine_adj(int dim, int64_t n, CF3 Ur, CF3 du, double *acc) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int j = 0; j < dim; ++j) s += Ur.p[j][q] * du.p[j][q];
        acc[q] = -s;
    }
}

// Fused weak linearised convective term, 3-D, one block per element (replaces 3 + 9 + 3 tensor launches and three
// combine launches that round-tripped ~20 fine-mesh fields through HBM):
//   out_i = J^T [ sgn * sum_j Ur_j (du_i/dr_j)_fine + sum_m uf_m Gsel_m ]      Gsel_m = GU[i][m] (direct), GU[m][i] (adjoint)
// The only fine-mesh HBM traffic left is the twelve precomputed base-flow fields (Ur, GU), read once.
// Phase A interpolates the three components to the fine mesh; each of the ND*ND threads that own a fine-mesh column
// (a, b, :) keeps its 3 x ND values in registers.  Phase B, per component: x/y stages through LDS, the z stage
// produces value and the three derivatives of the thread's column in registers, combines them with the base flow
// (coalesced HBM reads: consecutive threads = consecutive (a, b)) and immediately applies J_z^T; the y and x
// back-projections go through LDS again.
// Work split of the LDS stages: lanes over the input columns, waves over the output index, so that every matrix
// entry a wave needs has a wave-uniform address -> scalar loads straight into FMA operands (two 96-entry matrices
// do not fit the scalar register file as kernel arguments: 366 spilled SGPRs and 2 ms per launch at E = 10k).
// LDS leading dimensions are padded to odd values where a thread walks a row.
// lx1 > 8: the LDS arrays exceed the 64 KB a kernel may declare statically, so they are carved out of dynamic LDS (up to
// 160 KB per workgroup on gfx950: one block per CU): lx1 = 9, 10 keep u in LDS as before (104 KB at lx1 = 10); at lx1 = 12
// (ULDS = false) the stage arrays alone take 134 KB and the x stages read u from global memory (41 KB per element, L2).
// NTC threads: one per fine-mesh column along z (ND^2 = 324 at lx1 = 12 -> 384 threads).
// Two blocks per CU wherever the LDS image allows it (lx1 <= 10: 78 KB at lx1 = 10 with u read from global memory): the launch
// bound caps the registers at 256 -- 289 were allocated at lx1 = 10 without it, i.e. ONE block per CU -- at the price of 140 bytes
// of scratch per lane; 17.2 -> 11.0 ms per step at lx1 = 10.
template <int N, int ND, int NTC, bool ULDS, bool DYN, bool ML = true>
__global__ __launch_bounds__(NTC, (DYN && N > 10) ? 1 : 2) void k_conv3(int64_t E, const double *__restrict__ Jg, const double *__restrict__ DJg,
                                                 CF3 Ur, CF9 GU, CF3L ul, F3L outl, int nl, int adjoint) {
    constexpr int NP = N * N * N, NPD = ND * ND * ND;
    constexpr int NQ = N | 1, NDQ = ND | 1;          // padded (odd) leading dimensions
    constexpr int SU = NQ * N * N;                   // u:  (i | j, k), row stride NQ
    constexpr int SX = NDQ * N * N;                  // after the x stage: (a | j, k), row stride NDQ
    constexpr int SY = ND * ND * N;                  // after the y stage: (a, b, k)
    constexpr int NCOLZ = ND * ND;                   // fine columns along z
    constexpr int NT = NTC;                          // (shadows the file-level block size inside this kernel)
    constexpr int NW = NT / 64;
    static_assert(NCOLZ <= NT, "one thread per fine-mesh column");
    extern __shared__ double conv_dyn[];
    __shared__ double sU_st[DYN ? 1 : 3 * SU];
    __shared__ double sW_st[DYN ? 1 : 2 * SX + 3 * SY];
    double *sW = DYN ? conv_dyn : sW_st;
    double(*sU)[SU] = reinterpret_cast<double(*)[SU]>(DYN ? conv_dyn + 2 * SX + 3 * SY : sU_st);
    double *sA = sW, *sB = sW + SX, *sAA = sW + 2 * SX, *sAD = sAA + SY, *sBA = sAD + SY;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t e = blockIdx.x;
    if (e >= E) return;
    // lanes (the vectors of a block step) one after the other: the twelve base-flow fields of the element are then served by
    // L2 / the Infinity Cache for every lane after the first
    for (int lv = 0; lv < (ML ? nl : 1); ++lv) {   // ML = false: the single-vector kernel (a runtime lane loop costs it 25 - 30 %)
    if (ULDS)
        for (int t = tid; t < 3 * NP; t += NT) {
            const int c = t / NP, q = t % NP;
            sU[c][(q % N) + NQ * (q / N)] = ul.p[lv][c][e * NP + q];
        }
    double ufr[3][ND];
    constexpr int OD = (ND + NW - 1) / NW;   // fine-index outputs per wave
    constexpr int ON = (N + NW - 1) / NW;    // coarse-index outputs per wave
    __syncthreads();
    // ---- phase A: uf_m on the fine mesh, columns in registers
#pragma unroll
    for (int mcomp = 0; mcomp < 3; ++mcomp) {   // unrolled: ufr must be indexed statically to stay in registers
        for (int col = lane; col < N * N; col += 64) {
            double v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = ULDS ? sU[mcomp][i + NQ * col] : ul.p[lv][mcomp][e * NP + i + N * col];
#pragma unroll
            for (int o = 0; o < OD; ++o) {
                const int a = wave + NW * o;
                if (a < ND) {
                    const double *__restrict__ row = Jg + a * N;
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) acc += row[i] * v[i];
                    sA[a + NDQ * col] = acc;
                }
            }
        }
        __syncthreads();
        for (int col = lane; col < ND * N; col += 64) {
            const int a = col % ND, k = col / ND;
            double v[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = sA[a + NDQ * (j + N * k)];
#pragma unroll
            for (int o = 0; o < OD; ++o) {
                const int b = wave + NW * o;
                if (b < ND) {
                    const double *__restrict__ row = Jg + b * N;
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) acc += row[j] * v[j];
                    sAA[a + ND * (b + ND * k)] = acc;
                }
            }
        }
        __syncthreads();
        if (tid < NCOLZ) {
            double v[N];
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = sAA[tid + NCOLZ * k];
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) acc += Jg[c * N + k] * v[k];
                ufr[mcomp][c] = acc;
            }
        }
        __syncthreads();
    }
    // ---- phase B: one output component at a time
    const double sgn = adjoint ? -1.0 : 1.0;
#pragma unroll 1
    for (int ic = 0; ic < 3; ++ic) {
        // (round 4: requesting the base-flow values of the first fine levels HERE, two LDS stages ahead, and every later batch before
        //  the previous one is consumed -- with LDS-only barriers -- was built and measured: 2.93 -> 3.01 ms per step at 10^4 elements.
        //  The kernel sits at the 256-register cap already (36 bytes of scratch); the second batch of 18 values spills (60 - 84 bytes)
        //  and the spill traffic costs more than the exposed latency did.  DESIGN.md section 5.)
        constexpr int CB = ND % 4 == 0 ? 4 : (ND % 5 == 0 ? 5 : (ND % 3 == 0 ? 3 : 1));
        const double *g0 = adjoint ? GU.p[0 * 3 + ic] : GU.p[ic * 3 + 0];
        const double *g1 = adjoint ? GU.p[1 * 3 + ic] : GU.p[ic * 3 + 1];
        const double *g2 = adjoint ? GU.p[2 * 3 + ic] : GU.p[ic * 3 + 2];
        const int64_t qb = e * NPD + tid;
        // x stage: A = J_x u_i, B = DJ_x u_i
        for (int col = lane; col < N * N; col += 64) {
            double v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = ULDS ? sU[ic][i + NQ * col] : ul.p[lv][ic][e * NP + i + N * col];
#pragma unroll
            for (int o = 0; o < OD; ++o) {
                const int a = wave + NW * o;
                if (a < ND) {
                    const double *__restrict__ r0 = Jg + a * N, *__restrict__ r1 = DJg + a * N;
                    double x0 = 0.0, x1 = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        x0 += r0[i] * v[i];
                        x1 += r1[i] * v[i];
                    }
                    sA[a + NDQ * col] = x0;
                    sB[a + NDQ * col] = x1;
                }
            }
        }
        __syncthreads();
        // y stage: AA = J_y A, AD = DJ_y A, BA = J_y B
        for (int col = lane; col < ND * N; col += 64) {
            const int a = col % ND, k = col / ND;
            double va[N], vb[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                va[j] = sA[a + NDQ * (j + N * k)];
                vb[j] = sB[a + NDQ * (j + N * k)];
            }
#pragma unroll
            for (int o = 0; o < OD; ++o) {
                const int b = wave + NW * o;
                if (b < ND) {
                    const double *__restrict__ r0 = Jg + b * N, *__restrict__ r1 = DJg + b * N;
                    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        y0 += r0[j] * va[j];
                        y1 += r1[j] * va[j];
                        y2 += r0[j] * vb[j];
                    }
                    sAA[a + ND * (b + ND * k)] = y0;
                    sAD[a + ND * (b + ND * k)] = y1;
                    sBA[a + ND * (b + ND * k)] = y2;
                }
            }
        }
        __syncthreads();
        // z stage + combination with the base flow + J_z^T, per fine column
        double T[N];
#pragma unroll
        for (int k = 0; k < N; ++k) T[k] = 0.0;
        if (tid < NCOLZ) {
            double v0[N], v1[N], v2[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                v0[k] = sAA[tid + NCOLZ * k];
                v1[k] = sAD[tid + NCOLZ * k];
                v2[k] = sBA[tid + NCOLZ * k];
            }
            // the six base-flow values of CB fine levels are requested together (6 CB loads in flight per lane); level by level
            // the compiler issued six loads and waited for them, i.e. ND serialised round trips to HBM per component
#pragma unroll
            for (int c0 = 0; c0 < ND; c0 += CB) {
                double bb[CB][6];
#pragma unroll
                for (int cc = 0; cc < CB; ++cc) {
                    const int64_t q = qb + (int64_t)NCOLZ * (c0 + cc);
                    bb[cc][0] = Ur.p[0][q], bb[cc][1] = Ur.p[1][q], bb[cc][2] = Ur.p[2][q];
                    bb[cc][3] = g0[q], bb[cc][4] = g1[q], bb[cc][5] = g2[q];
                }
#pragma unroll
                for (int cc = 0; cc < CB; ++cc) {
                    const int c = c0 + cc;
                    double ut = 0.0, us = 0.0, ur = 0.0;
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        ut += DJg[c * N + k] * v0[k];
                        us += Jg[c * N + k] * v1[k];
                        ur += Jg[c * N + k] * v2[k];
                    }
                    const double acc = sgn * (bb[cc][0] * ur + bb[cc][1] * us + bb[cc][2] * ut) +
                                       (ufr[0][c] * bb[cc][3] + ufr[1][c] * bb[cc][4] + ufr[2][c] * bb[cc][5]);
#pragma unroll
                    for (int k = 0; k < N; ++k) T[k] += Jg[c * N + k] * acc;
                }
            }
        }
        __syncthreads();   // every thread is done with sAA/sAD/sBA and sA/sB
        // T -> LDS as (a, b, kz), reusing sAA
        if (tid < NCOLZ) {
#pragma unroll
            for (int k = 0; k < N; ++k) sAA[tid + NCOLZ * k] = T[k];
        }
        __syncthreads();
        // back y stage: S(a, j, kz) = sum_b J[b][j] T(a, b, kz), into sA
        for (int col = lane; col < ND * N; col += 64) {
            const int a = col % ND, k = col / ND;
            double v[ND];
#pragma unroll
            for (int b = 0; b < ND; ++b) v[b] = sAA[a + ND * (b + ND * k)];
#pragma unroll
            for (int o = 0; o < ON; ++o) {
                const int j = wave + NW * o;
                if (j < N) {
                    double acc = 0.0;
#pragma unroll
                    for (int b = 0; b < ND; ++b) acc += Jg[b * N + j] * v[b];
                    sA[a + NDQ * (j + N * k)] = acc;
                }
            }
        }
        __syncthreads();
        // back x stage: out(i, j, kz) = sum_a J[a][i] S(a, j, kz) -> LDS (reusing sB) -> coalesced store
        for (int col = lane; col < N * N; col += 64) {
            double v[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) v[a] = sA[a + NDQ * col];
#pragma unroll
            for (int o = 0; o < ON; ++o) {
                const int i = wave + NW * o;
                if (i < N) {
                    double acc = 0.0;
#pragma unroll
                    for (int a = 0; a < ND; ++a) acc += Jg[a * N + i] * v[a];
                    sB[i + NQ * col] = acc;
                }
            }
        }
        __syncthreads();
        {
            double *op = outl.p[lv][ic] + e * NP;
            for (int q = tid; q < NP; q += NT) op[q] = sB[(q % N) + NQ * (q / N)];
        }
        __syncthreads();
    }
    }
}

// Plane-sweep form of the fused convective term for lx1 >= 9 (round 3).  k_conv3 above keeps the whole fine-mesh image of one
// component in LDS (134 KB at lx1 = 12: ONE six-wave block per CU, every stage waiting on the one before it, eighteen
// serialised round trips to HBM per element).  Here the z direction goes FIRST and LAST: for every fine level c
//   S1  Z0_m = (J_z u_m)(c), Z1_m = (DJ_z u_m)(c) on the coarse (i, j) plane         thread (i, j, m) holds its column u_m(i, j, :)
//   S2  x stage: J_x Z0, DJ_x Z0, J_x Z1 -> (a, j)                                    lanes (j, m), waves over a
//   S3  y stage: value and three derivatives at (a, b), all three components; combination with the twelve base-flow
//       values of the level (prefetched into registers one level ahead) -> acc_m(a, b)   lanes (a, m), waves over b
//   S4  J_y^T acc -> (a, j)                                                           lanes (a, m), waves over j
//   S5  J_x^T     -> (i, j)                                                           lanes (j, m), waves over i
//   S6  out_m(i, j, :) += J[c][:] P_m(i, j)                                           thread (i, j, m), 12 accumulators
// so that the LDS image is the planes of ONE level (40 KB at lx1 = 12), the value interpolation of phase A is a by-product
// of S3, and the base-flow stream is requested a level ahead of its use.  In every stage the matrix entry a wave needs has a
// wave-uniform address (scalar operand) and a lane reuses the row it read for all the outputs of its wave.
template <int N, int ND, int NW, int MINB>
__global__ __launch_bounds__(NW * 64, MINB) void k_conv3s(int64_t E, const double *__restrict__ Jg, const double *__restrict__ DJg, const double *__restrict__ Jt, CF3 Ur, CF9 GU,
                                                          CF3L ul, F3L outl, int nl, int adjoint) {
    constexpr int NN = N * N, NP = NN * N, NDD = ND * ND, NPD = NDD * ND;
    constexpr int NQ = N | 1, NDQ = ND | 1;   // odd leading dimensions where lanes walk the slow index
    constexpr int OB = (ND + NW - 1) / NW, OJ = (N + NW - 1) / NW;
    static_assert(3 * NN <= NW * 64 && 3 * ND <= 64, "one thread per coarse column and component; (a, m) in one wave");
    __shared__ double sZ[2][3][NQ * N];
    __shared__ double sX[3][3][NDQ * N];
    __shared__ double sAc[3][NDQ * ND];
    __shared__ double sY[3][NDQ * N];
    __shared__ double sP[3][NN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the vectors of a block step are consecutive blocks of the same element (the base flow of the element is then served by L2 for
    // every lane after the first); no lane loop: with the stores of one lane ahead of the loads of the next the compiler could not
    // prove the matrix rows unclobbered and fetched them with vector loads
    const int64_t e = blockIdx.x / nl;
    const int lv = (int)(blockIdx.x - e * nl);
    if (e >= E) return;
    const bool r1 = tid < 3 * NN;                    // coarse column (i, j) of component m1
    const int m1 = r1 ? tid / NN : 0, ij = r1 ? tid % NN : 0;
    const int zidx = (ij % N) + NQ * (ij / N);
    const bool r2 = lane < 3 * N;                    // (j, m)
    const int j2 = r2 ? lane % N : 0, m2 = r2 ? lane / N : 0;
    const bool r3 = lane < 3 * ND;                   // (a, m)
    const int a3 = r3 ? lane % ND : 0, m3 = r3 ? lane / ND : 0;
    const double sgn = adjoint ? -1.0 : 1.0;
    const double *__restrict__ g0 = adjoint ? GU.p[0 * 3 + m3] : GU.p[m3 * 3 + 0];
    const double *__restrict__ g1 = adjoint ? GU.p[1 * 3 + m3] : GU.p[m3 * 3 + 1];
    const double *__restrict__ g2 = adjoint ? GU.p[2 * 3 + m3] : GU.p[m3 * 3 + 2];
    const double *__restrict__ u0 = Ur.p[0], *__restrict__ u1 = Ur.p[1], *__restrict__ u2 = Ur.p[2];
    double uc[N], oacc[N];
    {
        const double *__restrict__ up = ul.p[lv][m1] + e * NP + ij;   // (threads without a column read column 0: in bounds, unused)
#pragma unroll
        for (int k = 0; k < N; ++k) {
            uc[k] = up[NN * k];
            oacc[k] = 0.0;
        }
    }
    // base-flow values of TWO levels in flight (one level = 31 KB per CU at lx1 = 12 covers a third of the memory latency: measured, the
    // wait in S3 was 70 % of the kernel): lane (a, m) holds Ur_m and the three gradient entries of its row; the other two Ur come by
    // lane exchange at the use
    double bf[2][OB][4];
    const double *__restrict__ um = m3 == 0 ? u0 : (m3 == 1 ? u1 : u2);
    // (branch-free: every lane of every wave issues all its loads, out-of-range rows, levels and idle lanes clamped to valid addresses --
    //  with the loads under wave-uniform branches the compiler's counter bookkeeping merges the paths conservatively and drains BOTH
    //  levels at the first use)
    auto request = [&](auto slot, int c) {
        constexpr int S = decltype(slot)::value;
        const int64_t qc = e * NPD + (int64_t)(c < ND ? c : ND - 1) * NDD + a3;
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            const int b = wave + NW * o < ND ? wave + NW * o : ND - 1;
            const int64_t q = qc + ND * b;
            bf[S][o][0] = um[q], bf[S][o][1] = g0[q], bf[S][o][2] = g1[q], bf[S][o][3] = g2[q];
        }
    };
    request(std::integral_constant<int, 0>{}, 0);
    request(std::integral_constant<int, 1>{}, 1);
    // the column must have arrived BEFORE the loop: a use inside it makes the compiler drain the load counter at the top of every
    // level, and with it the base-flow prefetch of the level before
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" ::"v"(uc[k]));

    auto S1 = [&](int c) {   // z stage of level c
        if (r1) {
            const double *__restrict__ r0 = Jg + c * N, *__restrict__ rd = DJg + c * N;
            double z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                z0 += r0[k] * uc[k];
                z1 += rd[k] * uc[k];
            }
            sZ[0][m1][zidx] = z0;
            sZ[1][m1][zidx] = z1;
        }
    };
    auto S2 = [&]() {        // x stage
        if (r2) {
            double z0[N], z1[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                z0[i] = sZ[0][m2][i + NQ * j2];
                z1[i] = sZ[1][m2][i + NQ * j2];
            }
#pragma unroll
            for (int o = 0; o < OB; ++o) {
                const int a = wave + NW * o;
                if (a < ND) {
                    const double *__restrict__ r0 = Jg + a * N, *__restrict__ rd = DJg + a * N;
                    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        x0 += r0[i] * z0[i];
                        x1 += rd[i] * z0[i];
                        x2 += r0[i] * z1[i];
                    }
                    sX[0][m2][a + NDQ * j2] = x0;
                    sX[1][m2][a + NDQ * j2] = x1;
                    sX[2][m2][a + NDQ * j2] = x2;
                }
            }
        }
    };
    auto S3 = [&](auto slot, int c) {   // y stage, combination with the base flow of level c (slot c & 1), request of level c + 2
        constexpr int S = decltype(slot)::value;
        double val[OB], us[OB], ur[OB], ut[OB];
#pragma unroll
        for (int o = 0; o < OB; ++o) val[o] = us[o] = ur[o] = ut[o] = 0.0;
        if (r3) {   // the three rows stay in registers: ONE fetch of the matrix rows J[b], DJ[b] per output b (scalar-cache round trip)
            double x0[N], x1[N], x2[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                x0[j] = sX[0][m3][a3 + NDQ * j];
                x1[j] = sX[1][m3][a3 + NDQ * j];
                x2[j] = sX[2][m3][a3 + NDQ * j];
            }
#pragma unroll
            for (int o = 0; o < OB; ++o) {
                const int b = wave + NW * o;
                if (b < ND) {
                    const double *__restrict__ r0 = Jg + b * N, *__restrict__ rd = DJg + b * N;
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        val[o] += r0[j] * x0[j];
                        us[o] += rd[j] * x0[j];
                        ur[o] += r0[j] * x1[j];
                        ut[o] += r0[j] * x2[j];
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            const int b = wave + NW * o;
            if (b < ND) {   // wave-uniform: every lane takes part in the exchanges
                const double v0 = __shfl(val[o], a3, 64), v1 = __shfl(val[o], a3 + ND, 64), v2 = __shfl(val[o], a3 + 2 * ND, 64);
                const double w0 = __shfl(bf[S][o][0], a3, 64), w1 = __shfl(bf[S][o][0], a3 + ND, 64), w2 = __shfl(bf[S][o][0], a3 + 2 * ND, 64);
                if (r3)
                    sAc[m3][a3 + NDQ * b] = sgn * (w0 * ur[o] + w1 * us[o] + w2 * ut[o]) + (v0 * bf[S][o][1] + v1 * bf[S][o][2] + v2 * bf[S][o][3]);
            }
        }
        // requested only after every value of this level has been used (loads issued between the uses make the compiler wait for ALL
        // outstanding loads at the next use: vmcnt counts in order)
        request(slot, c + 2);
    };
    auto S4 = [&]() {        // J_y^T
        if (r3) {
            double col[ND];
#pragma unroll
            for (int b = 0; b < ND; ++b) col[b] = sAc[m3][a3 + NDQ * b];
#pragma unroll
            for (int o = 0; o < OJ; ++o) {
                const int j = wave + NW * o;
                if (j < N) {
                    double y = 0.0;
#pragma unroll
                    for (int b = 0; b < ND; ++b) y += Jt[j * ND + b] * col[b];
                    sY[m3][a3 + NDQ * j] = y;
                }
            }
        }
    };
    auto S5 = [&]() {        // J_x^T
        if (r2) {
            double row[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) row[a] = sY[m2][a + NDQ * j2];
#pragma unroll
            for (int o = 0; o < OJ; ++o) {
                const int i = wave + NW * o;
                if (i < N) {
                    double p = 0.0;
#pragma unroll
                    for (int a = 0; a < ND; ++a) p += Jt[i * ND + a] * row[a];
                    sP[m2][i + N * j2] = p;
                }
            }
        }
    };
    auto S6 = [&](int c) {   // J_z^T of level c into the accumulators of the coarse column
        if (r1) {
            const double p = sP[m1][ij];
            const double *__restrict__ r0 = Jg + c * N;
#pragma unroll
            for (int k = 0; k < N; ++k) oacc[k] += r0[k] * p;
        }
    };
    // Software pipeline over the levels, three barrier intervals per level: the forward stages of level c share their intervals with
    // the backward stages of level c - 1 (independent instruction streams inside a wave, three barriers instead of five):
    //   [S1(c) S4(c-1)] | [S2(c) S5(c-1)] | [S3(c) S6(c-1)] |
    S1(0);
    lds_barrier();
    S2();
    lds_barrier();
    S3(std::integral_constant<int, 0>{}, 0);
    lds_barrier();
    auto level = [&](auto slot, int c) {
        S1(c);
        S4();
        lds_barrier();
        S2();
        S5();
        lds_barrier();
        S3(slot, c);
        S6(c - 1);
        lds_barrier();
    };
#pragma unroll 1
    for (int cv = 1; cv + 1 < ND; cv += 2) {
        const int c = __builtin_amdgcn_readfirstlane(cv);   // (the counter also feeds per-lane addresses: keep a scalar copy for the matrix rows)
        level(std::integral_constant<int, 1>{}, c);
        level(std::integral_constant<int, 0>{}, c + 1);
    }
    if (ND % 2 == 0) level(std::integral_constant<int, 1>{}, ND - 1);
    S4();
    lds_barrier();
    S5();
    lds_barrier();
    S6(ND - 1);
    if (r1) {
        double *__restrict__ op = outl.p[lv][m1] + e * NP + ij;
#pragma unroll
        for (int k = 0; k < N; ++k) op[NN * k] = oacc[k];
    }
}

// The scalar (temperature) transport term of the Boussinesq coupling as a plane sweep of the same kind (round 3; it went through seven
// launches of the generic tensor kernel and a combine pass, 7.7 % of a config-4-shaped step):
//   out = J^T [ sgn sum_j Ur_j (d theta / d r_j)_fine + gsel sum_m uf_m GT_m ]       direct: sgn = gsel = 1; adjoint: sgn = -1, gsel = 0
// Four fields are interpolated level by level -- the three velocity components (value only) and theta (three derivatives only); in
// the y stage lane (a, f), f < 3, forms the value of component f AND the f-th derivative of theta, so that its share
// sgn Ur_f dtheta_f + gsel uf_f GT_f of the sum is local and the three shares meet by lane exchange.  One field is projected back.
template <int N, int ND, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_conv3s_scalar(int64_t E, const double *__restrict__ Jg, const double *__restrict__ DJg, const double *__restrict__ Jt,
                                                              CF3 Ur, CF3 GT, CF3 u, const double *__restrict__ theta, double *__restrict__ out, int adjoint) {
    constexpr int NN = N * N, NP = NN * N, NDD = ND * ND, NPD = NDD * ND;
    constexpr int NQ = N | 1, NDQ = ND | 1;
    constexpr int OB = (ND + NW - 1) / NW, OJ = (N + NW - 1) / NW;
    static_assert(4 * NN <= NW * 64 && 3 * ND <= 64 && 4 * N <= 64, "one thread per coarse column and field; (a, f) and (j, f) in one wave");
    __shared__ double sZ0[4][NQ * N], sZ1[NQ * N];
    __shared__ double sXu[3][NDQ * N], sXt[3][NDQ * N];
    __shared__ double sAc[NDQ * ND], sY[NDQ * N], sP[NN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t e = blockIdx.x;
    if (e >= E) return;
    const bool r1 = tid < 4 * NN;                    // coarse column (i, j) of field f1 (0 .. 2: velocity, 3: theta)
    const int f1 = r1 ? tid / NN : 0, ij = r1 ? tid % NN : 0;
    const bool rt = r1 && f1 == 3;                   // ... the theta columns also carry the accumulators of the result
    const int zidx = (ij % N) + NQ * (ij / N);
    const bool r2 = lane < 4 * N;                    // (j, f)
    const int j2 = r2 ? lane % N : 0, f2 = r2 ? lane / N : 0;
    const bool r3 = lane < 3 * ND;                   // (a, f), f < 3
    const int a3 = r3 ? lane % ND : 0, f3 = r3 ? lane / ND : 0;
    const int qf = f3 == 0 ? 1 : (f3 == 1 ? 0 : 2);  // the stage array of theta this lane differentiates: d/dr from DJ_x, d/ds from J_x (then DJ_y), d/dt from J_x DJ_z
    const bool r4 = lane < ND, r5 = lane < N;
    const double sgn = adjoint ? -1.0 : 1.0, gsel = adjoint ? 0.0 : 1.0;
    const double *__restrict__ um = f3 == 0 ? Ur.p[0] : (f3 == 1 ? Ur.p[1] : Ur.p[2]);
    const double *__restrict__ gm = f3 == 0 ? GT.p[0] : (f3 == 1 ? GT.p[1] : GT.p[2]);
    double uc[N], oacc[N];
    {
        const double *__restrict__ up = (f1 == 0 ? u.p[0] : (f1 == 1 ? u.p[1] : (f1 == 2 ? u.p[2] : theta))) + e * NP + ij;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            uc[k] = up[NN * k];
            oacc[k] = 0.0;
        }
    }
    double bf[2][OB][2];
    auto request = [&](auto slot, int c) {   // branch-free, clamped (see k_conv3s)
        constexpr int S = decltype(slot)::value;
        const int64_t qc = e * NPD + (int64_t)(c < ND ? c : ND - 1) * NDD + a3;
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            const int b = wave + NW * o < ND ? wave + NW * o : ND - 1;
            const int64_t q = qc + ND * b;
            bf[S][o][0] = um[q], bf[S][o][1] = gm[q];
        }
    };
    request(std::integral_constant<int, 0>{}, 0);
    request(std::integral_constant<int, 1>{}, 1);
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("" ::"v"(uc[k]));

    auto S1 = [&](int c) {
        if (r1) {
            const double *__restrict__ r0 = Jg + c * N, *__restrict__ rd = DJg + c * N;
            double z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                z0 += r0[k] * uc[k];
                z1 += rd[k] * uc[k];
            }
            sZ0[f1][zidx] = z0;
            if (rt) sZ1[zidx] = z1;
        }
    };
    auto S2 = [&]() {
        if (r2) {
            double z0[N], z1[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                z0[i] = sZ0[f2][i + NQ * j2];
                z1[i] = sZ1[i + NQ * j2];
            }
#pragma unroll
            for (int o = 0; o < OB; ++o) {
                const int a = wave + NW * o;
                if (a < ND) {
                    const double *__restrict__ r0 = Jg + a * N, *__restrict__ rd = DJg + a * N;
                    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        x0 += r0[i] * z0[i];
                        x1 += rd[i] * z0[i];
                        x2 += r0[i] * z1[i];
                    }
                    if (f2 < 3) {
                        sXu[f2][a + NDQ * j2] = x0;
                    } else {
                        sXt[0][a + NDQ * j2] = x0;
                        sXt[1][a + NDQ * j2] = x1;
                        sXt[2][a + NDQ * j2] = x2;
                    }
                }
            }
        }
    };
    auto S3 = [&](auto slot, int c) {
        constexpr int S = decltype(slot)::value;
        double val[OB], dth[OB];
#pragma unroll
        for (int o = 0; o < OB; ++o) val[o] = dth[o] = 0.0;
        if (r3) {
            double xu[N], yt[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                xu[j] = sXu[f3][a3 + NDQ * j];
                yt[j] = sXt[qf][a3 + NDQ * j];
            }
#pragma unroll
            for (int o = 0; o < OB; ++o) {
                const int b = wave + NW * o;
                if (b < ND) {
                    const double *__restrict__ r0 = Jg + b * N, *__restrict__ rd = DJg + b * N;
                    double tj = 0.0, td = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        val[o] += r0[j] * xu[j];
                        tj += r0[j] * yt[j];
                        td += rd[j] * yt[j];
                    }
                    dth[o] = f3 == 1 ? td : tj;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            const int b = wave + NW * o;
            if (b < ND) {   // wave-uniform
                const double sh = sgn * bf[S][o][0] * dth[o] + gsel * (val[o] * bf[S][o][1]);
                const double t0 = __shfl(sh, a3, 64), t1 = __shfl(sh, a3 + ND, 64), t2 = __shfl(sh, a3 + 2 * ND, 64);
                if (r3 && f3 == 0) sAc[a3 + NDQ * b] = (t0 + t1) + t2;
            }
        }
        request(slot, c + 2);
    };
    auto S4 = [&]() {
        if (r4) {
            double col[ND];
#pragma unroll
            for (int b = 0; b < ND; ++b) col[b] = sAc[lane + NDQ * b];
#pragma unroll
            for (int o = 0; o < OJ; ++o) {
                const int j = wave + NW * o;
                if (j < N) {
                    double y = 0.0;
#pragma unroll
                    for (int b = 0; b < ND; ++b) y += Jt[j * ND + b] * col[b];
                    sY[lane + NDQ * j] = y;
                }
            }
        }
    };
    auto S5 = [&]() {
        if (r5) {
            double row[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) row[a] = sY[a + NDQ * lane];
#pragma unroll
            for (int o = 0; o < OJ; ++o) {
                const int i = wave + NW * o;
                if (i < N) {
                    double p = 0.0;
#pragma unroll
                    for (int a = 0; a < ND; ++a) p += Jt[i * ND + a] * row[a];
                    sP[i + N * lane] = p;
                }
            }
        }
    };
    auto S6 = [&](int c) {
        if (rt) {
            const double p = sP[ij];
            const double *__restrict__ r0 = Jg + c * N;
#pragma unroll
            for (int k = 0; k < N; ++k) oacc[k] += r0[k] * p;
        }
    };
    S1(0);
    lds_barrier();
    S2();
    lds_barrier();
    S3(std::integral_constant<int, 0>{}, 0);
    lds_barrier();
    auto level = [&](auto slot, int c) {
        S1(c);
        S4();
        lds_barrier();
        S2();
        S5();
        lds_barrier();
        S3(slot, c);
        S6(c - 1);
        lds_barrier();
    };
#pragma unroll 1
    for (int cv = 1; cv + 1 < ND; cv += 2) {
        const int c = __builtin_amdgcn_readfirstlane(cv);
        level(std::integral_constant<int, 1>{}, c);
        level(std::integral_constant<int, 0>{}, c + 1);
    }
    if (ND % 2 == 0) level(std::integral_constant<int, 1>{}, ND - 1);
    S4();
    lds_barrier();
    S5();
    lds_barrier();
    S6(ND - 1);
    if (rt) {
        double *__restrict__ op = out + e * NP + ij;
#pragma unroll
        for (int k = 0; k < N; ++k) op[NN * k] = oacc[k];
    }
}

// =================================================================================================
// Dealiasing interpolation on the matrix cores: one velocity-mesh field -> its value and its three reference-space
// derivatives on the fine (Gauss) mesh,
//   uf = (J x J x J) u,   dx = (J x J x DJ) u,   dy = (J x DJ x J) u,   dz = (DJ x J x J) u        (z x y x x factors)
// as three passes of small GEMMs  out(ND x cols) = M(ND x N) in(N x cols)  on v_mfma_f64_16x16x4_f64: M = a 16-row tile
// (ND = 12 real rows, 4 of padding), K = N = 8 in two steps of 4, cols in tiles of 16.  Operand layout (the f64 form has
// its own C/D map, cdna_hip_programming.md): A: lane l holds M[l & 15][l >> 4 (+ 4 ks)]; B: lane l holds in[l >> 4 (+ 4 ks)]
// [col l & 15]; D: register r of lane l holds out[(l >> 4) + 4 r][col l & 15] -- r = 3 is the padding rows.
// One element per block, four waves; a wave owns column tiles.  Shared stages: the x pass feeds two arrays (J u, DJ u),
// the y pass three (JJ, J DJ, DJ J), the z pass the four results: 16 + 36 + 72 = 124 MFMAs per element.
// Used for the base-flow side of the convective term (sem_conv_setup: once per operator for exptA, once per TIME STEP in
// the nonlinear map of the Newton-Krylov solver), where the generic tensor kernel took twelve launches per field triple.
// fp64 MFMA has the same peak rate as the fp64 vector pipe on MI355X (and the 12 -> 16 row padding costs a quarter of it),
// so this is not about flops: it removes the LDS operand traffic and the instruction count of the scalar-FMA form; the
// kernel is bound by its 55 KB of output per element.
// =================================================================================================
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int N, int ND>
__global__ __launch_bounds__(256) void k_interp4_mfma(int64_t E, const double *__restrict__ Jg, const double *__restrict__ DJg,
                                                      const double *__restrict__ u, double *__restrict__ uf, double *__restrict__ dx,
                                                      double *__restrict__ dy, double *__restrict__ dz) {
    static_assert(N == 8 && ND <= 16 && ND > 8, "two k-steps of 4, one 16-row tile, three result registers");
    constexpr int NP = N * N * N, NPD = ND * ND * ND, NDQ = ND + 1;
    constexpr int CY = ND * N, CZ = ND * ND;                 // columns of the y and z passes
    __shared__ double sA[NDQ * N * N], sB[NDQ * N * N];      // x pass: (a | j, k), J u and DJ u
    __shared__ double sAA[CZ * N], sAD[CZ * N], sBA[CZ * N]; // y pass: (a, b | k)
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t e = blockIdx.x;
    if (e >= E) return;
    double ja[2], da[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        ja[ks] = l15 < ND ? Jg[l15 * N + lg + 4 * ks] : 0.0;
        da[ks] = l15 < ND ? DJg[l15 * N + lg + 4 * ks] : 0.0;
    }
    const v4f64 zero = {0.0, 0.0, 0.0, 0.0};
    // ---- x pass: columns (j, k), 64 = 4 tiles, one per wave
    {
        const int col = 16 * wave + l15;
        const double b0 = u[e * NP + lg + N * col], b1 = u[e * NP + lg + 4 + N * col];
        v4f64 aj = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[0], b0, zero, 0, 0, 0);
        aj = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[1], b1, aj, 0, 0, 0);
        v4f64 ad = __builtin_amdgcn_mfma_f64_16x16x4f64(da[0], b0, zero, 0, 0, 0);
        ad = __builtin_amdgcn_mfma_f64_16x16x4f64(da[1], b1, ad, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int a = lg + 4 * r;
            if (a < ND) {
                sA[a + NDQ * col] = aj[r];
                sB[a + NDQ * col] = ad[r];
            }
        }
    }
    __syncthreads();
    // ---- y pass: columns (a, k), ND * N = 96 = 6 tiles
    for (int t = wave; t * 16 < CY; t += 4) {
        const int col = 16 * t + l15;
        const int a = col % ND, kz = col / ND;
        const bool ok = col < CY;
        double va[2], vb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int j = lg + 4 * ks;
            va[ks] = ok ? sA[a + NDQ * (j + N * kz)] : 0.0;
            vb[ks] = ok ? sB[a + NDQ * (j + N * kz)] : 0.0;
        }
        v4f64 aa = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[0], va[0], zero, 0, 0, 0);
        aa = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[1], va[1], aa, 0, 0, 0);
        v4f64 ad = __builtin_amdgcn_mfma_f64_16x16x4f64(da[0], va[0], zero, 0, 0, 0);
        ad = __builtin_amdgcn_mfma_f64_16x16x4f64(da[1], va[1], ad, 0, 0, 0);
        v4f64 ba = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[0], vb[0], zero, 0, 0, 0);
        ba = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[1], vb[1], ba, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int b = lg + 4 * r;
            if (ok && b < ND) {
                const int q = a + ND * (b + ND * kz);
                sAA[q] = aa[r];
                sAD[q] = ad[r];
                sBA[q] = ba[r];
            }
        }
    }
    __syncthreads();
    // ---- z pass: columns (a, b), ND * ND = 144 = 9 tiles; results straight to HBM (16 consecutive doubles per row)
    for (int t = wave; t * 16 < CZ; t += 4) {
        const int col = 16 * t + l15;
        const bool ok = col < CZ;
        double v0[2], v1[2], v2[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = lg + 4 * ks;
            v0[ks] = ok ? sAA[col + CZ * k] : 0.0;
            v1[ks] = ok ? sAD[col + CZ * k] : 0.0;
            v2[ks] = ok ? sBA[col + CZ * k] : 0.0;
        }
        v4f64 r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[0], v0[0], zero, 0, 0, 0);
        r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[1], v0[1], r0, 0, 0, 0);
        v4f64 r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(da[0], v0[0], zero, 0, 0, 0);
        r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(da[1], v0[1], r1, 0, 0, 0);
        v4f64 r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[0], v1[0], zero, 0, 0, 0);
        r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[1], v1[1], r2, 0, 0, 0);
        v4f64 r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[0], v2[0], zero, 0, 0, 0);
        r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ja[1], v2[1], r3, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int c = lg + 4 * r;
            if (ok && c < ND) {
                const int64_t q = e * NPD + col + (int64_t)CZ * c;
                uf[q] = r0[r];
                dz[q] = r1[r];
                dy[q] = r2[r];
                dx[q] = r3[r];
            }
        }
    }
}

// =================================================================================================
// The fused dealiased convective term on the matrix cores (lx1 = 8, lxd = 12; round 4).  Same mathematics as k_conv3:
//   out_i = J^T [ sgn * sum_j Ur_j (du_i/dr_j)_fine + sum_m uf_m Gsel_m ],   Gsel_m = GU[i][m] (direct), GU[m][i] (adjoint)
// but every one of the 21 tensor stages of an element is a set of v_mfma_f64_16x16x4_f64 tiles (operand layouts: k_interp4_mfma):
//   forward, per velocity component m:  x pass (J u, DJ u: 16 MFMA) -> LDS -> y pass (36) -> LDS -> z pass (72): value and the three
//     derivatives of u_m arrive in the D registers of the wave that owns the column tile -- row c = (lane >> 4) + 4 r, column (a, b) --
//     and are folded at once into the three running sums acc_i of the tile with the base-flow values of the same points (loaded in
//     that very layout: 16 consecutive doubles per row), so no fine-mesh quantity of the perturbation ever goes to LDS or HBM;
//   backward, per output component i:  the z pass contracts over the fine index c -- and register r of the D layout IS the B operand
//     of k-step r (B wants in[k = (lane >> 4) + 4 ks][column]), so acc_i is consumed from registers (27 MFMA) -> LDS -> y pass (18)
//     -> LDS -> x pass (12) -> LDS -> coalesced store.
// 543 MFMA per element (372 forward, 171 backward), 15 block barriers (33 in k_conv3), LDS 41 KB, the matrix entries live in 7
// registers per lane instead of going through the scalar cache.  fp64 MFMA has the vector pipe's peak rate on this chip: what
// the form buys is issue slots -- one instruction per 1024 multiply-adds instead of one per 64 plus its operand traffic -- in a kernel
// that ran at 33 % of the HBM peak with the vector pipe as the bound.  Lanes of a block step: blockIdx.y.
// =================================================================================================
template <int N, int ND, bool DYN>
__global__ __launch_bounds__(256, (ND > 16) ? 1 : 2) void k_conv3m(int64_t E, const double *__restrict__ Jg, const double *__restrict__ DJg, CF3 Ur, CF9 GU,
                                                   CF3L ul, F3L outl, int adjoint) {
    static_assert(ND <= 32 && N <= 16 && ND > N, "fine rows in one or two 16-row tiles: lx1 = 8 (lxd 12), 10 (15), 12 (18: rows 16, 17 in a second tile)");
    constexpr int NP = N * N * N, NPD = ND * ND * ND, NDQ = ND + 1, NQ = N + 1;
    constexpr int CX = N * N, CY = ND * N, CZ = ND * ND;     // columns of the x, y and z passes
    constexpr int KF = (N + 3) / 4, KB = (ND + 3) / 4;       // k-steps of a forward (contraction over a coarse index) / backward pass
    constexpr int RF = KB, RB = KF;                          // D registers that hold real rows: fine rows forward, coarse rows backward
    constexpr int RT = (ND + 15) / 16;                       // 16-row tiles of a forward result; register r = 4 rt + r' holds fine row lg + 4 r
    constexpr int NTZ = (CZ + 63) / 64;                      // z-pass column tiles per wave (lx1 = 8: 9 tiles on 4 waves: 3, 2, 2, 2)
    constexpr bool XF = CX % 16 == 0, YF = CY % 16 == 0, ZF = CZ % 16 == 0;   // full column tiles (lx1 = 8): no column guards
    constexpr bool RFF = ND % 4 == 0, RBF = N % 4 == 0;                       // full register rows: no row guards
    // forward x pass: (a | j, k) in sA, sB; y pass: (a, b | k) in sAA, sAD, sBA.  Backward: sAA = after the z pass, sA = after the y
    // pass, sB = the result (i | j, k).  lx1 = 10: 80 KB, more than a kernel may declare statically -> dynamic LDS, two blocks per CU still
    constexpr int LA = NDQ * CX, LY = CZ * N;
    extern __shared__ double conv3m_dyn[];
    __shared__ double st_lds[DYN ? 1 : 2 * LA + 3 * LY];
    double *sA = DYN ? conv3m_dyn : st_lds, *sB = sA + LA, *sAA = sB + LA, *sAD = sAA + LY, *sBA = sAD + LY;
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t e = blockIdx.x;
    const int lv = blockIdx.y;
    if (e >= E) return;
    double ja[RT][KF], da[RT][KF], jt[KB];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ks = 0; ks < KF; ++ks) {
            const bool ok = 16 * rt + l15 < ND && (RBF || lg + 4 * ks < N);
            ja[rt][ks] = ok ? Jg[(16 * rt + l15) * N + lg + 4 * ks] : 0.0;       // A[row = fine index][k = coarse index]
            da[rt][ks] = ok ? DJg[(16 * rt + l15) * N + lg + 4 * ks] : 0.0;
        }
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) jt[ks] = (l15 < N && (RFF || lg + 4 * ks < ND)) ? Jg[(lg + 4 * ks) * N + l15] : 0.0;   // A[row = coarse][k = fine] = J^T
    const v4f64 zero = {0.0, 0.0, 0.0, 0.0};
    const double sgn = adjoint ? -1.0 : 1.0;
    // one small GEMM: D = sum_ks A[ks] B[ks]
    auto mm = [&](const auto &A, const auto &B, auto nks) {
        v4f64 d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0], B[0], zero, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < decltype(nks)::value; ++ks) d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ks], B[ks], d, 0, 0, 0);
        return d;
    };
    constexpr std::integral_constant<int, KF> kf{};
    constexpr std::integral_constant<int, KB> kb{};
    // forward GEMM: all fine rows of the result, out[r] = row lg + 4 r
    auto fw = [&](const double (&A)[RT][KF], const double (&B)[KF], double (&out)[RF]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const v4f64 d = mm(A[rt], B, kf);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * rt + q < RF) out[4 * rt + q] = d[q];
        }
    };
    // KEEPU: keep the transporting base flow Ur_j of this wave's column tiles in registers across the three velocity components m instead of
    // re-reading it for every m (the PMC traffic of this kernel is 1.40 x its algorithmic bytes, and those re-reads are the difference).
    // Measured at lx1 = 8 (194 registers, no spills): 420 - 432 -> 443 us -- the re-reads come from the Infinity Cache at no cost worth 54
    // registers; at lxd = 15 it spills (78 registers).  Off.
    constexpr bool KEEPU = false;
    double buk[KEEPU ? NTZ : 1][3][RF];
    double acc[NTZ][3][RF];   // [column tile of this wave][output component][D register = fine level (lane >> 4) + 4 r]
#pragma unroll
    for (int ti = 0; ti < NTZ; ++ti)
#pragma unroll
        for (int ic = 0; ic < 3; ++ic)
#pragma unroll
            for (int r = 0; r < RF; ++r) acc[ti][ic][r] = 0.0;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const double *__restrict__ u = ul.p[lv][m] + e * NP;
        // ---- x pass: columns (j, k)
        for (int t = wave; t * 16 < CX; t += 4) {
            const int col = 16 * t + l15;
            const bool okc = XF || col < CX;
            double bv[KF];
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) bv[ks] = (okc && (RBF || lg + 4 * ks < N)) ? u[lg + 4 * ks + N * col] : 0.0;
            double aj[RF], ad[RF];
            fw(ja, bv, aj);
            fw(da, bv, ad);
#pragma unroll
            for (int r = 0; r < RF; ++r)
                if (okc && (RFF || lg + 4 * r < ND)) {
                    sA[lg + 4 * r + NDQ * col] = aj[r];
                    sB[lg + 4 * r + NDQ * col] = ad[r];
                }
        }
        lds_barrier();   // (also: every wave has left the z pass of the previous component, whose operands the y pass overwrites)
        // ---- y pass: columns (a, k)
        for (int t = wave; t * 16 < CY; t += 4) {
            const int col = 16 * t + l15;
            const bool okc = YF || col < CY;
            const int a = col % ND, kz = col / ND;
            double va[KF], vb[KF];
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) {
                const int j = lg + 4 * ks;
                const bool ok = okc && (RBF || j < N);
                va[ks] = ok ? sA[a + NDQ * (j + N * kz)] : 0.0;
                vb[ks] = ok ? sB[a + NDQ * (j + N * kz)] : 0.0;
            }
            double aa[RF], ad[RF], ba[RF];
            fw(ja, va, aa);
            fw(da, va, ad);
            fw(ja, vb, ba);
#pragma unroll
            for (int r = 0; r < RF; ++r)
                if (okc && (RFF || lg + 4 * r < ND)) {
                    const int q = a + ND * (lg + 4 * r + ND * kz);
                    sAA[q] = aa[r];
                    sAD[q] = ad[r];
                    sBA[q] = ba[r];
                }
        }
        lds_barrier();
        // ---- z pass: columns (a, b); results stay in registers and meet the base flow there
#pragma unroll
        for (int ti = 0; ti < NTZ; ++ti) {
            const int t = wave + 4 * ti;
            if (t * 16 < CZ) {   // wave-uniform
                const int col = 16 * t + l15;
                const bool okc = ZF || col < CZ;
                // base-flow values of the tile's points, requested before the matrix work: Ur_j and the three G of this component
                const int64_t qb = e * NPD + col;
                double bg[3][RF];
#pragma unroll
                for (int r = 0; r < RF; ++r) {
                    const bool ok = okc && (RFF || lg + 4 * r < ND);
                    const int64_t q = qb + (int64_t)CZ * (lg + 4 * r);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (KEEPU ? m == 0 : true) buk[KEEPU ? ti : 0][j][r] = ok ? Ur.p[j][q] : 0.0;   // KEEPU: the same for every component m, loaded once and kept
                        bg[j][r] = ok ? (adjoint ? GU.p[m * 3 + j][q] : GU.p[j * 3 + m][q]) : 0.0;   // multiplies uf_m in output component j
                    }
                }
                const double (&bu)[3][RF] = buk[KEEPU ? ti : 0];
                double v0[KF], v1[KF], v2[KF];
#pragma unroll
                for (int ks = 0; ks < KF; ++ks) {
                    const int k = lg + 4 * ks;
                    const bool ok = okc && (RBF || k < N);
                    v0[ks] = ok ? sAA[col + CZ * k] : 0.0;
                    v1[ks] = ok ? sAD[col + CZ * k] : 0.0;
                    v2[ks] = ok ? sBA[col + CZ * k] : 0.0;
                }
                double r0[RF], r1[RF], r2[RF], r3[RF];
                fw(ja, v0, r0);   // uf_m
                fw(da, v0, r1);   // d/dr_2 (z)
                fw(ja, v1, r2);   // d/dr_1 (y)
                fw(ja, v2, r3);   // d/dr_0 (x)
#pragma unroll
                for (int r = 0; r < RF; ++r) {
                    acc[ti][m][r] += sgn * (bu[0][r] * r3[r] + bu[1][r] * r2[r] + bu[2][r] * r1[r]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[ti][j][r] += r0[r] * bg[j][r];
                }
            }
        }
    }
    // ---- backward: J_z^T from registers, J_y^T and J_x^T through LDS, one output component at a time
#pragma unroll
    for (int ic = 0; ic < 3; ++ic) {
        lds_barrier();   // sAA: the last z pass / the y pass of the previous component has read it; sB: the previous store has read it
#pragma unroll
        for (int ti = 0; ti < NTZ; ++ti) {
            const int t = wave + 4 * ti;
            if (t * 16 < CZ) {
                const int col = 16 * t + l15;
                const v4f64 tt = mm(jt, acc[ti][ic], kb);      // (rows beyond the fine mesh and columns beyond (a, b) carry zeros)
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    if ((ZF || col < CZ) && (RBF || lg + 4 * r < N)) sAA[col + CZ * (lg + 4 * r)] = tt[r];   // T(a, b | k)
            }
        }
        lds_barrier();
        for (int t = wave; t * 16 < CY; t += 4) {      // S(a | j, k) = sum_b J[b][j] T(a, b, k): columns (a, k)
            const int col = 16 * t + l15;
            const bool okc = YF || col < CY;
            const int a = col % ND, kz = col / ND;
            double vb[KB];
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) vb[ks] = (okc && (RFF || lg + 4 * ks < ND)) ? sAA[a + ND * (lg + 4 * ks) + CZ * kz] : 0.0;
            const v4f64 ss = mm(jt, vb, kb);
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (okc && (RBF || lg + 4 * r < N)) sA[a + NDQ * (lg + 4 * r + N * kz)] = ss[r];
        }
        lds_barrier();
        for (int t = wave; t * 16 < CX; t += 4) {      // out(i | j, k) = sum_a J[a][i] S(a, j, k): columns (j, k)
            const int col = 16 * t + l15;
            const bool okc = XF || col < CX;
            double va[KB];
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) va[ks] = (okc && (RFF || lg + 4 * ks < ND)) ? sA[lg + 4 * ks + NDQ * col] : 0.0;
            const v4f64 oo = mm(jt, va, kb);
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (okc && (RBF || lg + 4 * r < N)) sB[lg + 4 * r + NQ * col] = oo[r];
        }
        lds_barrier();
        {
            double *__restrict__ op = outl.p[lv][ic] + e * NP;
            for (int q = threadIdx.x; q < NP; q += 256) op[q] = sB[(q % N) + NQ * (q / N)];
        }
    }
}

// The scalar (temperature) transport term of the Boussinesq coupling in the same matrix-pipe form (see k_conv3s_scalar for the term):
//   out = J^T [ sgn sum_j Ur_j (d theta / d r_j)_fine + gsel sum_m uf_m GT_m ]       direct: sgn = gsel = 1; adjoint: sgn = -1, gsel = 0
// Four fields go through the forward passes -- the velocity components for their VALUE only (one matrix per pass; skipped in the adjoint),
// theta for its three derivatives -- into ONE running sum per column tile; one field is projected back.
template <int N, int ND, bool DYN>
__global__ __launch_bounds__(256, 2) void k_conv3m_scalar(int64_t E, const double *__restrict__ Jg, const double *__restrict__ DJg, CF3 Ur, CF3 GT, CF3 uv,
                                                          const double *__restrict__ theta, double *__restrict__ out, int adjoint) {
    static_assert(ND <= 16 && N <= 12 && ND > N, "one 16-row tile per matrix: lx1 = 8 (lxd 12), 10 (15)");
    constexpr int NP = N * N * N, NPD = ND * ND * ND, NDQ = ND + 1, NQ = N + 1;
    constexpr int CX = N * N, CY = ND * N, CZ = ND * ND;
    constexpr int KF = (N + 3) / 4, KB = (ND + 3) / 4, RF = KB, RB = KF;
    constexpr int NTZ = (CZ + 63) / 64;
    constexpr bool XF = CX % 16 == 0, YF = CY % 16 == 0, ZF = CZ % 16 == 0, RFF = ND % 4 == 0, RBF = N % 4 == 0;
    constexpr int LA = NDQ * CX, LY = CZ * N;
    extern __shared__ double conv3ms_dyn[];
    __shared__ double st_lds[DYN ? 1 : 2 * LA + 3 * LY];
    double *sA = DYN ? conv3ms_dyn : st_lds, *sB = sA + LA, *sAA = sB + LA, *sAD = sAA + LY, *sBA = sAD + LY;
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t e = blockIdx.x;
    if (e >= E) return;
    double ja[KF], da[KF], jt[KB];
#pragma unroll
    for (int ks = 0; ks < KF; ++ks) {
        const bool ok = l15 < ND && (RBF || lg + 4 * ks < N);
        ja[ks] = ok ? Jg[l15 * N + lg + 4 * ks] : 0.0;
        da[ks] = ok ? DJg[l15 * N + lg + 4 * ks] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) jt[ks] = (l15 < N && (RFF || lg + 4 * ks < ND)) ? Jg[(lg + 4 * ks) * N + l15] : 0.0;
    const v4f64 zero = {0.0, 0.0, 0.0, 0.0};
    const double sgn = adjoint ? -1.0 : 1.0;
    auto mm = [&](const auto &A, const auto &B, auto nks) {
        v4f64 d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0], B[0], zero, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < decltype(nks)::value; ++ks) d = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ks], B[ks], d, 0, 0, 0);
        return d;
    };
    constexpr std::integral_constant<int, KF> kf{};
    constexpr std::integral_constant<int, KB> kb{};
    double acc[NTZ][RF];
#pragma unroll
    for (int ti = 0; ti < NTZ; ++ti)
#pragma unroll
        for (int r = 0; r < RF; ++r) acc[ti][r] = 0.0;
    // one field through the three forward passes; TH: theta (derivatives), else a velocity component (value, times GT_f)
    auto field = [&](auto th, const double *__restrict__ u, const double *__restrict__ gt) {
        constexpr bool TH = decltype(th)::value;
        for (int t = wave; t * 16 < CX; t += 4) {
            const int col = 16 * t + l15;
            const bool okc = XF || col < CX;
            double bv[KF];
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) bv[ks] = (okc && (RBF || lg + 4 * ks < N)) ? u[lg + 4 * ks + N * col] : 0.0;
            const v4f64 aj = mm(ja, bv, kf);
            v4f64 ad = zero;
            if constexpr (TH) ad = mm(da, bv, kf);
#pragma unroll
            for (int r = 0; r < RF; ++r)
                if (okc && (RFF || lg + 4 * r < ND)) {
                    sA[lg + 4 * r + NDQ * col] = aj[r];
                    if constexpr (TH) sB[lg + 4 * r + NDQ * col] = ad[r];
                }
        }
        lds_barrier();
        for (int t = wave; t * 16 < CY; t += 4) {
            const int col = 16 * t + l15;
            const bool okc = YF || col < CY;
            const int a = col % ND, kz = col / ND;
            double va[KF], vb[KF];
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) {
                const int j = lg + 4 * ks;
                const bool ok = okc && (RBF || j < N);
                va[ks] = ok ? sA[a + NDQ * (j + N * kz)] : 0.0;
                vb[ks] = (TH && ok) ? sB[a + NDQ * (j + N * kz)] : 0.0;
            }
            const v4f64 aa = mm(ja, va, kf);
            v4f64 ad = zero, ba = zero;
            if constexpr (TH) {
                ad = mm(da, va, kf);
                ba = mm(ja, vb, kf);
            }
#pragma unroll
            for (int r = 0; r < RF; ++r)
                if (okc && (RFF || lg + 4 * r < ND)) {
                    const int q = a + ND * (lg + 4 * r + ND * kz);
                    sAA[q] = aa[r];
                    if constexpr (TH) {
                        sAD[q] = ad[r];
                        sBA[q] = ba[r];
                    }
                }
        }
        lds_barrier();
#pragma unroll
        for (int ti = 0; ti < NTZ; ++ti) {
            const int t = wave + 4 * ti;
            if (t * 16 < CZ) {
                const int col = 16 * t + l15;
                const bool okc = ZF || col < CZ;
                const int64_t qb = e * NPD + col;
                double b0[RF], b1[RF], b2[RF];
#pragma unroll
                for (int r = 0; r < RF; ++r) {
                    const bool ok = okc && (RFF || lg + 4 * r < ND);
                    const int64_t q = qb + (int64_t)CZ * (lg + 4 * r);
                    if constexpr (TH) {
                        b0[r] = ok ? Ur.p[0][q] : 0.0, b1[r] = ok ? Ur.p[1][q] : 0.0, b2[r] = ok ? Ur.p[2][q] : 0.0;
                    } else {
                        b0[r] = ok ? gt[q] : 0.0;
                    }
                }
                double v0[KF], v1[KF], v2[KF];
#pragma unroll
                for (int ks = 0; ks < KF; ++ks) {
                    const int k = lg + 4 * ks;
                    const bool ok = okc && (RBF || k < N);
                    v0[ks] = ok ? sAA[col + CZ * k] : 0.0;
                    v1[ks] = (TH && ok) ? sAD[col + CZ * k] : 0.0;
                    v2[ks] = (TH && ok) ? sBA[col + CZ * k] : 0.0;
                }
                if constexpr (TH) {
                    const v4f64 r1 = mm(da, v0, kf), r2 = mm(ja, v1, kf), r3 = mm(ja, v2, kf);   // d/dr_2, d/dr_1, d/dr_0
#pragma unroll
                    for (int r = 0; r < RF; ++r) acc[ti][r] += sgn * (b0[r] * r3[r] + b1[r] * r2[r] + b2[r] * r1[r]);
                } else {
                    const v4f64 r0 = mm(ja, v0, kf);
#pragma unroll
                    for (int r = 0; r < RF; ++r) acc[ti][r] += r0[r] * b0[r];
                }
            }
        }
        lds_barrier();   // the next field's passes overwrite the stage arrays
    };
    if (!adjoint) {
        field(std::false_type{}, uv.p[0] + e * NP, GT.p[0]);
        field(std::false_type{}, uv.p[1] + e * NP, GT.p[1]);
        field(std::false_type{}, uv.p[2] + e * NP, GT.p[2]);
    }
    field(std::true_type{}, theta + e * NP, nullptr);
    // ---- backward
#pragma unroll
    for (int ti = 0; ti < NTZ; ++ti) {
        const int t = wave + 4 * ti;
        if (t * 16 < CZ) {
            const int col = 16 * t + l15;
            const v4f64 tt = mm(jt, acc[ti], kb);
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if ((ZF || col < CZ) && (RBF || lg + 4 * r < N)) sAA[col + CZ * (lg + 4 * r)] = tt[r];
        }
    }
    lds_barrier();
    for (int t = wave; t * 16 < CY; t += 4) {
        const int col = 16 * t + l15;
        const bool okc = YF || col < CY;
        const int a = col % ND, kz = col / ND;
        double vb[KB];
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) vb[ks] = (okc && (RFF || lg + 4 * ks < ND)) ? sAA[a + ND * (lg + 4 * ks) + CZ * kz] : 0.0;
        const v4f64 ss = mm(jt, vb, kb);
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (okc && (RBF || lg + 4 * r < N)) sA[a + NDQ * (lg + 4 * r + N * kz)] = ss[r];
    }
    lds_barrier();
    for (int t = wave; t * 16 < CX; t += 4) {
        const int col = 16 * t + l15;
        const bool okc = XF || col < CX;
        double va[KB];
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) va[ks] = (okc && (RFF || lg + 4 * ks < ND)) ? sA[lg + 4 * ks + NDQ * col] : 0.0;
        const v4f64 oo = mm(jt, va, kb);
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (okc && (RBF || lg + 4 * r < N)) sB[lg + 4 * r + NQ * col] = oo[r];
    }
    lds_barrier();
    {
        double *__restrict__ op = out + e * NP;
        for (int q = threadIdx.x; q < NP; q += 256) op[q] = sB[(q % N) + NQ * (q / N)];
    }
}

// CFL (Nek compute_cfl): max over points of dt * sum_j |u_rj| * rdr
__global__ __launch_bounds__(NT) void k_cfl(int dim, int n, int64_t E, CF9 rst, const double *jac, const double *rdr,
                                            CF3 U, double dt, double *partial) {
    __shared__ double sm[NT];
    const int np = n * n * (dim == 3 ? n : 1);
    const int64_t tot = E * np;
    double mx = 0.0;
    for (int64_t q = blockIdx.x * (int64_t)NT + threadIdx.x; q < tot; q += (int64_t)gridDim.x * NT) {
        const int p = (int)(q % np);
        const int ijk[3] = {p % n, (p / n) % n, p / (n * n)};
        double s = 0.0;
        for (int j = 0; j < dim; ++j) {
            double ur = 0.0;
            for (int m = 0; m < dim; ++m) ur += rst.p[j * dim + m][q] * U.p[m][q];
            s += fabs(ur / jac[q] * rdr[ijk[j]]);
        }
        mx = fmax(mx, s * dt);
    }
    sm[threadIdx.x] = mx;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
__global__ void k_max_final(const double *partial, int n, double *out) {
    double m = 0.0;
    for (int i = 0; i < n; ++i) m = fmax(m, partial[i]);
    out[0] = m;
}

// sum of a pressure-mesh vector (ortho).  blockIdx.y = lane of a block step: x is ld doubles, the partial sums NPART doubles apart.
constexpr int NPART = 256;
__global__ __launch_bounds__(NT) void k_sum_partial(const double *x, int64_t n, double *partial, int64_t ld) {
    __shared__ double sm[NT];
    x += (int64_t)blockIdx.y * ld, partial += (int64_t)blockIdx.y * NPART;
    double a = 0.0;
    for (int64_t q = blockIdx.x * (int64_t)NT + threadIdx.x; q < n; q += (int64_t)gridDim.x * NT) a += x[q];
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}
// second stage, one block per lane: out[lane] = sum of the lane's NPART partials (fixed tree: every rank, every run the same)
__global__ __launch_bounds__(NT) void k_sum_final(const double *partial, int n, double *out) {
    __shared__ double sm[NT];
    partial += (int64_t)blockIdx.x * NPART;
    sm[threadIdx.x] = (int)threadIdx.x < n ? partial[threadIdx.x] : 0.0;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}
// x -= mean.  sum != null: the (all-reduced) sums, one per lane; otherwise every block re-sums the lane's partials itself with the
// tree of k_sum_final (one launch less on a single rank; all blocks obtain the same bits)
__global__ __launch_bounds__(NT) void k_sub_mean(double *x, int64_t n, const double *sum, const double *partial, int npart, double inv_count, int64_t ld) {
    __shared__ double sm[NT];
    x += (int64_t)blockIdx.y * ld;
    double tot;
    if (sum) {
        tot = sum[blockIdx.y];
    } else {
        partial += (int64_t)blockIdx.y * NPART;
        sm[threadIdx.x] = (int)threadIdx.x < npart ? partial[threadIdx.x] : 0.0;
        __syncthreads();
        for (int o = NT / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
            __syncthreads();
        }
        tot = sm[0];
    }
    const double mean = tot * inv_count;
    for (int64_t q = blockIdx.x * (int64_t)NT + threadIdx.x; q < n; q += (int64_t)gridDim.x * NT) x[q] -= mean;
}

// nek_drand noise (reference: real_vectors.f90:52-98 + neklab_vectors.f90:305-314), counter-based RNG
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void k_rand_add(int dim, int n, int64_t E, const int64_t *lglel, CF3 X, double *field, int field_id,
                           uint64_t seed) {
    const int np = n * n * (dim == 3 ? n : 1);
    const int64_t tot = E * np;
    const uint64_t sk = splitmix64(seed);
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < tot; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = q / np;
        const int p = (int)(q % np);
        const int ix = p % n + 1, iy = (p / n) % n + 1, iz = (dim == 3) ? p / (n * n) + 1 : 1;
        const int64_t ieg = lglel[e] + 1;
        const uint64_t base = (((uint64_t)((ieg - 1) * np + p)) * 8ull + (uint64_t)field_id) * 4ull;
        double fc[3];
        for (int c = 0; c < 3; ++c)
            fc[c] = (double)(splitmix64((base + (uint64_t)c) ^ sk) >> 11) * (1.0 / 9007199254740992.0) * 1.0e4;
        // mth_rand, neklab_vectors.f90:305-314.  The two nested 1e3 * sin() amplify a rounding difference in the
        // argument a million-fold, so the expression is evaluated operation by operation (no fused multiply-add), left
        // to right as written: the value then depends on sin / cos alone and agrees with the oracle's to ~1e-9.
        const double x = X.p[0][q], y = X.p[1][q];
        double r = __dadd_rn(__dadd_rn(__dmul_rn(fc[0], __dadd_rn((double)ieg, __dmul_rn(x, sin(y)))), __dmul_rn(__dmul_rn(fc[1], (double)ix), (double)iy)),
                             __dmul_rn(fc[2], (double)ix));
        if (dim == 3)
            r = __dadd_rn(__dadd_rn(__dmul_rn(fc[0], __dadd_rn((double)ieg, __dmul_rn(X.p[2][q], sin(r)))), __dmul_rn(__dmul_rn(fc[1], (double)iz), (double)ix)),
                          __dmul_rn(fc[2], (double)iz));
        r = __dmul_rn(1.0e3, sin(r));
        r = __dmul_rn(1.0e3, sin(r));
        field[q] += cos(r);
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + NT - 1) / NT;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

// =================================================================================================
// dispatch on N
// =================================================================================================
#define NLG_FOR_N(MACRO) \
    switch (m->n) {      \
        case 4: MACRO(4); break;   \
        case 5: MACRO(5); break;   \
        case 6: MACRO(6); break;   \
        case 7: MACRO(7); break;   \
        case 8: MACRO(8); break;   \
        case 9: MACRO(9); break;   \
        case 10: MACRO(10); break; \
        case 12: MACRO(12); break; \
        default: set_error("unsupported lx1 = %d (built for 4..10, 12)", m->n); return 1; \
    }

namespace nlg {

double *sem_scratch1(nlg_mesh *m, int i) {
    while ((int)m->scratch1.size() <= i) {
        double *p = nullptr;
        if (hipMalloc(&p, sizeof(double) * (size_t)m->lvs) != hipSuccess) return nullptr;
        hipMemsetAsync(p, 0, sizeof(double) * (size_t)m->lvs, m->ctx->stream);
        m->scratch1.push_back(p);
    }
    return m->scratch1[i];
}
double *sem_scratchd(nlg_mesh *m, int i) {
    while ((int)m->scratchd.size() <= i) {
        double *p = nullptr;
        if (hipMalloc(&p, sizeof(double) * (size_t)m->lfn) != hipSuccess) return nullptr;
        m->scratchd.push_back(p);
    }
    return m->scratchd[i];
}
double *sem_scratch2(nlg_mesh *m, int i) {
    while ((int)m->scratch2.size() <= i) {
        double *p = nullptr;
        if (hipMalloc(&p, sizeof(double) * (size_t)m->lps) != hipSuccess) return nullptr;
        hipMemsetAsync(p, 0, sizeof(double) * (size_t)m->lps, m->ctx->stream);
        m->scratch2.push_back(p);
    }
    return m->scratch2[i];
}

static int gs_launch(nlg_mesh *m, const int *goff, const int *gidx, int64_t ngroups, int64_t npairs, int64_t nquads, double *const *fields,
                     int nf, const double *gate, int nl = 1, int64_t ld = 0, int64_t ldg = 0) {
    if (ngroups == 0) return 0;
    F3 f = {{fields[0], nf > 1 ? fields[1] : nullptr, nf > 2 ? fields[2] : nullptr}};
    const dim3 grid((unsigned)((ngroups + NT - 1) / NT), (unsigned)nl);
    if (nf == 1)
        NLG_LAUNCH(k_gs<1>, grid, dim3(NT), 0, m->ctx->stream, goff, gidx, ngroups, npairs, nquads, f, gate, ld, ldg);
    else if (nf == 2)
        NLG_LAUNCH(k_gs<2>, grid, dim3(NT), 0, m->ctx->stream, goff, gidx, ngroups, npairs, nquads, f, gate, ld, ldg);
    else
        NLG_LAUNCH(k_gs<3>, grid, dim3(NT), 0, m->ctx->stream, goff, gidx, ngroups, npairs, nquads, f, gate, ld, ldg);
    NLG_HIP(hipGetLastError());
    return 0;
}

int sem_gs(nlg_mesh *m, double *const *fields, int nf, const double *gate, int layout, int nl, int64_t ld, int64_t ldg) {
    if (m->gs.ngroups == 0 && !m->halo.active) return 0;
    ++g_collectives;   // one halo exchange (carrying all lanes) when the mesh is partitioned
    // (the timed class "gs" is the dim-field kernel of the two PCGs; scalar-field calls go to "vec_ops" so that the
    //  class average is the duration of ONE kernel with ONE algorithmic byte count)
    ProfScope ps(m->ctx, nf == m->dim ? P_GS : P_VECOPS);
    const int *goff = layout == LAYOUT_XP ? m->gs.d_offsets_xp : (layout == LAYOUT_FG ? m->gs.d_offsets_fg : m->gs.d_offsets);
    const int *gidx = layout == LAYOUT_XP ? m->gs.d_indices_xp : (layout == LAYOUT_FG ? m->gs.d_indices_fg : m->gs.d_indices);
    NLG_CHECK(layout >= LAYOUT_NAT && layout <= LAYOUT_XP && (gidx || m->gs.ngroups == 0), "sem_gs: layout %d has no tables", layout);
    if (nf < 1 || nf > 3) {
        set_error("sem_gs: nf=%d unsupported", nf);
        return 1;
    }
    if (m->halo.active && m->gs.split && m->halo.overlap) {
        // several ranks, NLG_HALO_OVERLAP=1: first the groups that hold a dof another rank shares, so that their sums can be packed and sent,
        // then all other groups (disjoint dofs) while the exchange is under way on the side stream, then the received sums (halo.hip).
        // Without the overlap the exchange is in-stream and the split buys nothing: one pass over all groups, then pack / exchange /
        // unpack (one launch less per gather-scatter; the sums are the same)
        const nlg_gs_tab &th = m->gs.tab_halo[layout], &tr = m->gs.tab_rest[layout];
        NLG_TRY(gs_launch(m, th.d_off, th.d_idx, th.ngroups, th.npairs, th.nquads, fields, nf, gate, nl, ld, ldg));
        NLG_TRY(halo_begin(m, fields, nf, layout, nl, ld));
        NLG_TRY(gs_launch(m, tr.d_off, tr.d_idx, tr.ngroups, tr.npairs, tr.nquads, fields, nf, gate, nl, ld, ldg));
        return halo_finish(m, fields, nf, layout, nl, ld);
    }
    NLG_TRY(gs_launch(m, goff, gidx, m->gs.ngroups, m->gs.npairs, m->gs.nquads, fields, nf, gate, nl, ld, ldg));
    return halo_exchange(m, fields, nf, layout, nl, ld);   // no-op on a single rank
}

// natural <-> x-planes-first, out of place, one thread per point
template <int NF, bool TO>
__global__ __launch_bounds__(NT) void k_xp_perm(int64_t n, int np, const int *__restrict__ slot, CF3 src, F3 dst, int64_t ld, CF3 wt) {
    // wt (natural layout, TO only; may be null): dst = wt * src -- the Dirichlet mask of a right-hand side rides in its permutation
#pragma unroll
    for (int c = 0; c < NF; ++c) src.p[c] += (int64_t)blockIdx.y * ld, dst.p[c] += (int64_t)blockIdx.y * ld;
    for (int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const int64_t e = i / np;
        const int64_t q = e * np + slot[(int)(i - e * np)];
#pragma unroll
        for (int c = 0; c < NF; ++c) {
            if (TO)
                dst.p[c][q] = wt.p[c] ? wt.p[c][i] * src.p[c][i] : src.p[c][i];
            else
                dst.p[c][i] = src.p[c][q];
        }
    }
}

static int xp_perm(nlg_mesh *m, double *const *src, double *const *dst, int nf, bool to, int nl, int64_t ld, double *const *wts = nullptr) {
    NLG_CHECK(m->d_slot_xp && nf >= 1 && nf <= 3, "sem_to_xp: no x-planes-first table (3-D only) or bad field count");
    CF3 a = {{src[0], nf > 1 ? src[1] : nullptr, nf > 2 ? src[2] : nullptr}};
    F3 b = {{dst[0], nf > 1 ? dst[1] : nullptr, nf > 2 ? dst[2] : nullptr}};
    const dim3 g(grid_for(m->lvn), nl), t(NT);
    hipStream_t st = m->ctx->stream;
    CF3 wt = {{wts ? wts[0] : nullptr, (wts && nf > 1) ? wts[1] : nullptr, (wts && nf > 2) ? wts[2] : nullptr}};
#define XPL(NF_)                                                                                               \
    if (to)                                                                                                    \
        NLG_LAUNCH((k_xp_perm<NF_, true>), g, t, 0, st, m->lvn, m->np1, (const int *)m->d_slot_xp, a, b, ld, wt);  \
    else                                                                                                       \
        NLG_LAUNCH((k_xp_perm<NF_, false>), g, t, 0, st, m->lvn, m->np1, (const int *)m->d_slot_xp, a, b, ld, wt);
    if (nf == 1) {
        XPL(1)
    } else if (nf == 2) {
        XPL(2)
    } else {
        XPL(3)
    }
#undef XPL
    NLG_HIP(hipGetLastError());
    return 0;
}
int sem_to_xp(nlg_mesh *m, double *const *src, double *const *dst, int nf, int nl, int64_t ld, double *const *wts) { return xp_perm(m, src, dst, nf, true, nl, ld, wts); }
int sem_from_xp(nlg_mesh *m, double *const *src, double *const *dst, int nf, int nl, int64_t ld) { return xp_perm(m, src, dst, nf, false, nl, ld); }

// (element, field) slots per block of k_axhelm3: bounded by 512 threads and by 64 KB of dynamic LDS
// waves = (element, field) slots per block of k_axhelm3r.  Three: the three components of ONE element share a block, hence an XCD and its
// L2 -- the seven metric arrays are 37 % of the kernel's bytes, and with four slots per block two thirds of the elements had their
// components in two blocks, i.e. on two XCDs (measured traffic 1.18 x algorithmic).  Same box: 5.62 (four) -> 5.46 (three) ms per step,
// 6.46 with six.  NLG_AXHELM_WPB = 4 / 6 for A/B runs.
static int axhelm3_wpb() {
    static const int w = getenv("NLG_AXHELM_WPB") ? atoi(getenv("NLG_AXHELM_WPB")) : 3;
    return (w == 4 || w == 6) ? w : 3;
}
static int axhelm3_nslot(int N) {
    if (N <= 8) return axhelm3_wpb();   // k_axhelm3r: one wave per (element, field) slot
    int nslot = 512 / (N * N);
    const int lds_cap = (int)((64 * 1024 / 8 - N * N) / (4 * N * N * N));
    if (nslot > lds_cap) nslot = lds_cap;
    if (nslot > 6) nslot = 6;
    if (nslot < 1) nslot = 1;
    return nslot;
}

int sem_gs_pairs(nlg_mesh *m, double *w, const double *gate, int nl, int64_t ld, int64_t ldg) {
    if (m->gs.npairs == 0) return 0;
    F3 f = {{w, nullptr, nullptr}};
    const dim3 grid((unsigned)((m->gs.npairs + NT - 1) / NT), (unsigned)nl);
    NLG_LAUNCH(k_gs<1>, grid, dim3(NT), 0, m->ctx->stream, m->gs.d_offsets, m->gs.d_indices, m->gs.npairs,
                       m->gs.npairs, (int64_t)0, f, gate, ld, ldg);
    NLG_HIP(hipGetLastError());
    return 0;
}

int sem_gs_pairs_fg(nlg_mesh *m, double *w, const double *gate, int nl, int64_t ld, int64_t ldg) {
    NLG_CHECK(m->gs.d_indices_fg, "sem_gs_pairs_fg: no face-grouped tables (3-D only)");
    if (m->gs.npairs == 0) return 0;
    F3 f = {{w, nullptr, nullptr}};
    const dim3 grid((unsigned)((m->gs.npairs + NT - 1) / NT), (unsigned)nl);
    NLG_LAUNCH(k_gs<1>, grid, dim3(NT), 0, m->ctx->stream, m->gs.d_offsets_fg, m->gs.d_indices_fg, m->gs.npairs,
                       m->gs.npairs, (int64_t)0, f, gate, ld, ldg);
    NLG_HIP(hipGetLastError());
    return 0;
}

// pairs per block of k_axhelm3c (NLG_AXHELM_PPB=1: one pair per block, the round-2 form)
static int axhelm3c_ppb(int n) {
    static const int env = getenv("NLG_AXHELM_PPB") ? atoi(getenv("NLG_AXHELM_PPB")) : 0;
    if (env == 1) return 1;
    return n == 12 ? 3 : 1;   // measured at 10^4 elements: lx1 = 12 838 -> 778 us (144 of 192 lanes -> 432 of 448), lx1 = 10 397 -> 519 us
}

int sem_axhelm_blocks(nlg_mesh *m, int nf) {
    if (m->dim == 2) {
        const int epb = NT / (m->n * m->n) > 0 ? NT / (m->n * m->n) : 1;
        return (int)((m->E + epb - 1) / epb);
    }
    static const bool use_cube = getenv("NLG_AXHELM_CUBE") && atoi(getenv("NLG_AXHELM_CUBE")) != 0;
    if (m->n > 8 && !use_cube) return (int)((m->E * nf + axhelm3c_ppb(m->n) - 1) / axhelm3c_ppb(m->n));   // k_axhelm3c: PPB (element, field) pairs per block
    const int nslot = axhelm3_nslot(m->n);
    return (int)((m->E * nf + nslot - 1) / nslot);
}

int sem_axhelm(nlg_mesh *m, double *const *u, double *const *w, int nf, double h1, double h2, double *pw_part,
               double *const *zf, const double *beta_p, const double *done_p, bool xp, int nl, int64_t ld, int64_t uoff) {
    NLG_CHECK(nf >= 1 && nf <= 3, "sem_axhelm: nf=%d unsupported", nf);
    NLG_CHECK(uoff == 0 || beta_p, "sem_axhelm: an output offset for the direction without the fused direction update");
    static const bool use_cube0 = getenv("NLG_AXHELM_CUBE") && atoi(getenv("NLG_AXHELM_CUBE")) != 0;
    if (nl > 1 && (m->dim == 2 || (m->n > 8 && use_cube0))) {
        // kernels without the lane dimension (2-D, the LDS-cube variant): one launch per lane at the lane's offsets
        for (int v = 0; v < nl; ++v) {
            double *uu[3], *ww[3], *zz[3];
            for (int c = 0; c < nf; ++c) uu[c] = u[c] + v * ld, ww[c] = w[c] + v * ld, zz[c] = zf ? zf[c] + v * ld : nullptr;
            NLG_TRY(sem_axhelm(m, uu, ww, nf, h1, h2, pw_part ? pw_part + v * ld : nullptr, zf ? zz : nullptr, beta_p ? beta_p + v * ld : nullptr,
                               done_p ? done_p + v * ld : nullptr, xp, 1, 0, uoff));
        }
        return 0;
    }
    NLG_CHECK(!xp || (m->dim == 3 && m->d_slot_xp), "sem_axhelm: the slab-permuted layout exists in 3-D only");
    ProfScope ps(m->ctx, P_AXHELM);
    CF3 cu = {{u[0], nf > 1 ? u[1] : nullptr, nf > 2 ? u[2] : nullptr}};
    F3 cw = {{w[0], nf > 1 ? w[1] : nullptr, nf > 2 ? w[2] : nullptr}};
    hipStream_t s = m->ctx->stream;
    NLG_CHECK(!beta_p || (zf && done_p), "sem_axhelm: the fused direction update needs zf and the done flag");
    CF3 cz = {{zf ? zf[0] : nullptr, (zf && nf > 1) ? zf[1] : nullptr, (zf && nf > 2) ? zf[2] : nullptr}};
    static const bool use_cube = getenv("NLG_AXHELM_CUBE") && atoi(getenv("NLG_AXHELM_CUBE")) != 0;   // the LDS-cube kernel (lx1 > 8), for A/B runs
    if (m->dim == 3) {
#define AX3(N_)                                                                                                       \
    {                                                                                                                 \
        const int nslot = axhelm3_nslot(N_);                                                                          \
        const int64_t tot = m->E * nf;                                                                                \
        const int grid = (int)((tot + nslot - 1) / nslot);                                                            \
        const size_t lds = sizeof(double) * (size_t)(N_ * N_ + nslot * 4 * N_ * N_ * N_);                             \
        if constexpr (N_ <= 8) {                                                                                      \
            if (xp && nslot == 3)                                                                                     \
            NLG_LAUNCH((k_axhelm3r<N_, 3, true>), dim3(grid, nl), dim3(192), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], \
                               m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)m->d_slot_xp, ld, uoff); \
            else if (xp && nslot == 6)                                                                                \
            NLG_LAUNCH((k_axhelm3r<N_, 6, true>), dim3(grid, nl), dim3(384), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], \
                               m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)m->d_slot_xp, ld, uoff); \
            else if (xp)                                                                                              \
            NLG_LAUNCH((k_axhelm3r<N_, 4, true>), dim3(grid, nl), dim3(256), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], \
                               m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)m->d_slot_xp, ld, uoff); \
            else if (nslot == 3)                                                                                      \
            NLG_LAUNCH((k_axhelm3r<N_, 3, false>), dim3(grid, nl), dim3(192), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], \
                               m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)nullptr, ld, uoff); \
            else if (nslot == 6)                                                                                      \
            NLG_LAUNCH((k_axhelm3r<N_, 6, false>), dim3(grid, nl), dim3(384), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], \
                               m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)nullptr, ld, uoff); \
            else                                                                                                      \
            NLG_LAUNCH((k_axhelm3r<N_, 4, false>), dim3(grid, nl), dim3(256), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], \
                               m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)nullptr, ld, uoff); \
        } else if (use_cube)                                                                                          \
        NLG_LAUNCH((k_axhelm3<N_>), dim3(grid), dim3(nslot * N_ * N_), lds, s, m->E, nf, nslot, m->d_D,      \
                           m->d_G[0], m->d_G[1], m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, uoff); \
        else                                                                                                          \
        {                                                                                                             \
            constexpr int PPB_ = 3;                                                                    \
            static const int xcd3c = !(getenv("NLG_AXHELM_XCD") && atoi(getenv("NLG_AXHELM_XCD")) == 0);              \
            const bool one = axhelm3c_ppb(N_) == 1;                                                                   \
            const unsigned gb = (unsigned)((tot + (one ? 1 : PPB_) - 1) / (one ? 1 : PPB_));                          \
            if (one) {                                                                                                \
                if (xp)                                                                                               \
                    NLG_LAUNCH((k_axhelm3c<N_, true, 1>), dim3(gb, nl), dim3(((N_ * N_ + 63) / 64) * 64), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)m->d_slot_xp, ld, uoff, xcd3c); \
                else                                                                                                  \
                    NLG_LAUNCH((k_axhelm3c<N_, false, 1>), dim3(gb, nl), dim3(((N_ * N_ + 63) / 64) * 64), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)nullptr, ld, uoff, xcd3c); \
            } else {                                                                                                  \
                if (xp)                                                                                               \
                    NLG_LAUNCH((k_axhelm3c<N_, true, PPB_>), dim3(gb, nl), dim3(((PPB_ * N_ * N_ + 63) / 64) * 64), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)m->d_slot_xp, ld, uoff, xcd3c); \
                else                                                                                                  \
                    NLG_LAUNCH((k_axhelm3c<N_, false, PPB_>), dim3(gb, nl), dim3(((PPB_ * N_ * N_ + 63) / 64) * 64), 0, s, m->E, nf, m->d_D, m->d_G[0], m->d_G[1], m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, (const int *)nullptr, ld, uoff, xcd3c); \
            }                                                                                                         \
        }                                                                                                             \
    }
        NLG_FOR_N(AX3)
#undef AX3
    } else {
#define AX2(N_)                                                                                                       \
    {                                                                                                                 \
        constexpr int EPB = (NT / (N_ * N_)) > 0 ? (NT / (N_ * N_)) : 1;                                              \
        const int grid = (int)((m->E + EPB - 1) / EPB);                                                               \
        if (nf == 1)                                                                                                  \
            NLG_LAUNCH((k_axhelm2<N_, 1>), dim3(grid), dim3(EPB * N_ * N_), 0, s, m->E, m->d_D, m->d_G[0],     \
                               m->d_G[1], m->d_G[2], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, uoff);                                 \
        else if (nf == 2)                                                                                             \
            NLG_LAUNCH((k_axhelm2<N_, 2>), dim3(grid), dim3(EPB * N_ * N_), 0, s, m->E, m->d_D, m->d_G[0],     \
                               m->d_G[1], m->d_G[2], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, uoff);                                 \
        else                                                                                                          \
            NLG_LAUNCH((k_axhelm2<N_, 3>), dim3(grid), dim3(EPB * N_ * N_), 0, s, m->E, m->d_D, m->d_G[0],     \
                               m->d_G[1], m->d_G[2], m->d_bm1, cu, cw, h1, h2, pw_part, cz, beta_p, done_p, uoff);                                 \
    }
        NLG_FOR_N(AX2)
#undef AX2
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

// nl <= 4 lanes of the velocity PCG of a block step in one launch (3-D, lx1 <= 8); otherwise lane by lane.  pw[v]: E sums of
// p . w per lane (one per element); zf / beta / done: the fused direction update and the done flag of every lane.
int sem_axhelm_lanes(nlg_mesh *m, int nl, double *const *const *u, double *const *const *w, double h1, double h2, double *const *pw,
                     double *const *const *zf, const double *const *beta, const double *const *done, bool xp) {
    if (!(m->dim == 3 && m->n <= 8 && nl >= 2 && nl <= 4)) {
        for (int v = 0; v < nl; ++v) NLG_TRY(sem_axhelm(m, u[v], w[v], m->dim, h1, h2, pw[v], zf[v], beta[v], done[v], xp));
        return 0;
    }
    ProfScope ps(m->ctx, P_AXHELM);
    HelmLanes L;
    for (int v = 0; v < 4; ++v) {
        for (int c = 0; c < 3; ++c) {
            L.u[v][c] = v < nl ? u[v][c] : nullptr;
            L.w[v][c] = v < nl ? w[v][c] : nullptr;
            L.z[v][c] = v < nl ? zf[v][c] : nullptr;
        }
        L.beta[v] = v < nl ? beta[v] : nullptr;
        L.done[v] = v < nl ? done[v] : nullptr;
        L.pw[v] = v < nl ? pw[v] : nullptr;
    }
    hipStream_t s = m->ctx->stream;
    const int *tab = xp ? (const int *)m->d_slot_xp : nullptr;
#define AXB(N_, NL_)                                                                                                         \
    if (xp)                                                                                                                  \
        NLG_LAUNCH((k_axhelm3rb<N_, NL_, true>), dim3((unsigned)m->E), dim3(64 * 3 * NL_), 0, s, m->E, m->d_D, m->d_G[0], m->d_G[1], \
                           m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, L, h1, h2, tab);                          \
    else                                                                                                                     \
        NLG_LAUNCH((k_axhelm3rb<N_, NL_, false>), dim3((unsigned)m->E), dim3(64 * 3 * NL_), 0, s, m->E, m->d_D, m->d_G[0], m->d_G[1], \
                           m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, L, h1, h2, tab);
#define AXBN(N_)                        \
    case N_:                            \
        if (nl == 2) {                  \
            AXB(N_, 2)                  \
        } else if (nl == 3) {           \
            AXB(N_, 3)                  \
        } else {                        \
            AXB(N_, 4)                  \
        }                               \
        break;
    switch (m->n) {
        AXBN(4) AXBN(5) AXBN(6) AXBN(7) AXBN(8)
        default: set_error("sem_axhelm_lanes: lx1 = %d", m->n); return 1;
    }
#undef AXBN
#undef AXB
    NLG_HIP(hipGetLastError());
    return 0;
}

int sem_helm_diag(nlg_mesh *m, double *out, double h1, double h2) {
    NLG_LAUNCH(k_helm_diag, dim3(grid_for(m->lvn)), dim3(NT), 0, m->ctx->stream, m->dim, m->n, m->E, m->d_D,
                       m->d_G[0], m->d_G[1], m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], m->d_bm1, out, h1, h2);
    NLG_HIP(hipGetLastError());
    return 0;
}

static bool pkern_old() {   // NLG_PKERN_OLD=1: the two-array kernels k_opgradt3 / k_opdiv3<N, 1> for lx1 > 8 (A/B runs)
    static const bool v = getenv("NLG_PKERN_OLD") && atoi(getenv("NLG_PKERN_OLD")) != 0;
    return v;
}

template <int N>
static void fill_pmats(const nlg_mesh *m, PMats<N> &M) {
    const int n2 = N - 2;
    const nlg_ops1d &o = m->ops;
    for (int k = 0; k < n2; ++k)
        for (int j = 0; j < N; ++j) {
            M.Im[k * N + j] = o.I12[(size_t)k * N + j];
            M.Dm[k * N + j] = o.D12[(size_t)k * N + j];
            M.It[j * n2 + k] = o.I12[(size_t)k * N + j];
            M.Dt[j * n2 + k] = o.D12[(size_t)k * N + j];
        }
}

// the small-mesh (strong-scaling) variants of the element kernels are chosen below this many local elements; NLG_SMALL_E=0 switches
// them off, NLG_SMALL_E=<count> moves the threshold (A/B runs).  Default: 4 waves per SIMD of one-wave-per-element blocks
bool sem_small_mesh(const nlg_mesh *m) {
    static const int64_t lim = getenv("NLG_SMALL_E") ? atoll(getenv("NLG_SMALL_E")) : 4096;
    return m->E < lim;
}

static CF9 rst2w_ptrs(const nlg_mesh *m) {
    CF9 g;
    for (int q = 0; q < 9; ++q) g.p[q] = m->d_rst2w[q];
    return g;
}

int sem_opgradt(nlg_mesh *m, const double *p, double *const *w, bool face_grouped, const double *gate, const nlg_pupd *upd) {
    const double *pl[1] = {p}, *gl[1] = {gate};
    double *const *wl[1] = {w};
    return sem_opgradt_lanes(m, 1, pl, wl, face_grouped, gl, upd);
}

// does sem_opgradt perform the PCG direction update itself when asked to (nlg_pupd)?  3-D, lx1 >= 8: the in-place kernels
// (lx1 = 12: the extra 2 x 10 memory operations per z column push k_opgradt3n<12> from 354 to 550 us -- more than the separate
//  update kernel costs -- so the fusion stops at lx1 = 10)
bool sem_opgradt_fuses_pupdate(const nlg_mesh *m) { return m->dim == 3 && m->n >= 8 && m->n <= 10 && !pkern_old(); }

// nl <= 4 pressure fields -> nl velocity-mesh field triples in one launch (block stepper); gates: per-lane done flags (may be null)
int sem_opgradt_lanes(nlg_mesh *m, int nl, const double *const *p, double *const *const *w, bool face_grouped, const double *const *gate,
                      const nlg_pupd *upd) {
    NLG_CHECK(!upd || sem_opgradt_fuses_pupdate(m), "sem_opgradt: the fused direction update exists for 3-D, lx1 >= 8 only");
    PUpd pu;
    for (int v = 0; v < 4; ++v) {
        const bool on = upd && v < nl && upd[v].z;
        pu.z[v] = on ? upd[v].z : nullptr;
        pu.beta[v] = on ? upd[v].beta : nullptr;
        pu.zmean[v] = on ? upd[v].zmean : nullptr;
        pu.p[v] = on ? upd[v].p : nullptr;
    }
    // (the timed class is ONE kernel instantiation -- the face-grouped variant of the pressure operator in 3-D; the natural-layout
    //  launches of the right-hand sides, a few per time step, are booked under "vec_ops")
    ProfScope ps(m->ctx, (m->dim == 2 || face_grouped) ? P_OPGRADT : P_VECOPS);
    CF9 g = rst2w_ptrs(m);
    hipStream_t s = m->ctx->stream;
    if (m->dim == 3) {
        const bool old_big = pkern_old();
        static const bool n8new = !(getenv("NLG_PKERN_N8") && atoi(getenv("NLG_PKERN_N8")) == 0);   // lx1 = 8: one wave per element through the in-place kernels (5 - 6 % faster than <8, 3>; NLG_PKERN_N8=0 for A/B)
        CP4 pl, gl;
        F3L wl;
        for (int v = 0; v < 4; ++v) {
            pl.p[v] = v < nl ? p[v] : nullptr;
            gl.p[v] = (v < nl && gate) ? gate[v] : nullptr;
            for (int c = 0; c < 3; ++c) wl.p[v][c] = v < nl ? w[v][c] : nullptr;
        }
        if (m->n == 8 && nl == 1 && n8new && !old_big && sem_small_mesh(m)) {   // strong-scaling regime: three waves per element
            const F3 w3 = {{w[0][0], w[0][1], w[0][2]}};
            const double *g0 = gate ? gate[0] : nullptr;
            if (upd) {
                if (face_grouped)
                    NLG_LAUNCH((k_opgradt3w<8, true, true>), dim3((unsigned)m->E), dim3(192), 0, s, m->E, (const double *)m->d_I12t, (const double *)m->d_D12t, (const int *)m->d_slot_fg, g, p[0], w3, g0, pu);
                else
                    NLG_LAUNCH((k_opgradt3w<8, false, true>), dim3((unsigned)m->E), dim3(192), 0, s, m->E, (const double *)m->d_I12t, (const double *)m->d_D12t, (const int *)m->d_slot_fg, g, p[0], w3, g0, pu);
            } else {
                if (face_grouped)
                    NLG_LAUNCH((k_opgradt3w<8, true, false>), dim3((unsigned)m->E), dim3(192), 0, s, m->E, (const double *)m->d_I12t, (const double *)m->d_D12t, (const int *)m->d_slot_fg, g, p[0], w3, g0, NoPUpd{});
                else
                    NLG_LAUNCH((k_opgradt3w<8, false, false>), dim3((unsigned)m->E), dim3(192), 0, s, m->E, (const double *)m->d_I12t, (const double *)m->d_D12t, (const int *)m->d_slot_fg, g, p[0], w3, g0, NoPUpd{});
            }
            NLG_HIP(hipGetLastError());
            return 0;
        }
#define GT3_(N_, ML_)                                                                                                        \
    {                                                                                                                  \
        PMats<N_> M;                                                                                                   \
        fill_pmats<N_>(m, M);                                                                                          \
        if (N_ <= 8 && face_grouped && !(N_ == 8 && n8new))                                                            \
            NLG_LAUNCH((k_opgradt3<N_, 3, true, ML_>), dim3((unsigned)m->E), dim3(NT), 0, s, m->E, M, g, pl, wl, gl, nl);    \
        else if (N_ <= 8 && !(N_ == 8 && n8new))                                                                       \
            NLG_LAUNCH((k_opgradt3<N_, 3, false, ML_>), dim3((unsigned)m->E), dim3(NT), 0, s, m->E, M, g, pl, wl, gl, nl);   \
        else if (!old_big && face_grouped)                                                                             \
            {                                                                                                          \
                if (upd)                                                                                               \
                    NLG_LAUNCH((k_opgradt3n<(N_ >= 8 && N_ <= 10 ? N_ : 9), true, ML_, true>), dim3((unsigned)m->E), dim3(PBlockN<N_>::NTB), 0, s, m->E, (const double *)m->d_I12t, (const double *)m->d_D12t, (const int *)m->d_slot_fg, g, pl, wl, gl, nl, pu);    \
                else                                                                                                   \
                    NLG_LAUNCH((k_opgradt3n<(N_ >= 8 ? N_ : 9), true, ML_, false>), dim3((unsigned)m->E), dim3(PBlockN<N_>::NTB), 0, s, m->E, (const double *)m->d_I12t, (const double *)m->d_D12t, (const int *)m->d_slot_fg, g, pl, wl, gl, nl, NoPUpd{});    \
            }                                                                                                          \
        else if (!old_big)                                                                                             \
            {                                                                                                          \
                if (upd)                                                                                               \
                    NLG_LAUNCH((k_opgradt3n<(N_ >= 8 && N_ <= 10 ? N_ : 9), false, ML_, true>), dim3((unsigned)m->E), dim3(PBlockN<N_>::NTB), 0, s, m->E, (const double *)m->d_I12t, (const double *)m->d_D12t, (const int *)m->d_slot_fg, g, pl, wl, gl, nl, pu);   \
                else                                                                                                   \
                    NLG_LAUNCH((k_opgradt3n<(N_ >= 8 ? N_ : 9), false, ML_, false>), dim3((unsigned)m->E), dim3(PBlockN<N_>::NTB), 0, s, m->E, (const double *)m->d_I12t, (const double *)m->d_D12t, (const int *)m->d_slot_fg, g, pl, wl, gl, nl, NoPUpd{});   \
            }                                                                                                          \
        else if (face_grouped)                                                                                         \
            NLG_LAUNCH((k_opgradt3<N_, 1, true, ML_>), dim3((unsigned)m->E), dim3(PBlock<N_, 1>::NTB), 0, s, m->E, M, g, pl, wl, gl, nl);    \
        else                                                                                                           \
            NLG_LAUNCH((k_opgradt3<N_, 1, false, ML_>), dim3((unsigned)m->E), dim3(PBlock<N_, 1>::NTB), 0, s, m->E, M, g, pl, wl, gl, nl);   \
    }
#define GT3(N_)           \
    if (nl == 1)          \
        GT3_(N_, false)   \
    else                  \
        GT3_(N_, true)
        NLG_FOR_N(GT3)
#undef GT3
#undef GT3_
    } else {
        for (int v = 0; v < nl; ++v) {
            F3 cw = {{w[v][0], w[v][1], nullptr}};
            const double *pv = p[v];
#define GT2(N_)                                                                                              \
    {                                                                                                        \
        constexpr int EPB = NT / (N_ * N_) > 0 ? NT / (N_ * N_) : 1;                                         \
        NLG_LAUNCH((k_opgradt2<N_>), dim3((unsigned)((m->E + EPB - 1) / EPB)), dim3(NT), 0, s, m->E, \
                           m->d_I12t, m->d_D12t, g, pv, cw);                                                 \
    }
            NLG_FOR_N(GT2)
#undef GT2
        }
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

// number of first-stage sums sem_opdiv writes per reduction (pw_part holds 2 x this)
int sem_opdiv_blocks(const nlg_mesh *m) {
    if (m->dim == 3) return (int)m->E;
    const int epb = NT / (m->n * m->n) > 0 ? NT / (m->n * m->n) : 1;
    return (int)((m->E + epb - 1) / epb);
}

int sem_opdiv(nlg_mesh *m, double *const *u, double *out, double scale, double *const *wts, bool face_grouped,
              const double *pdot, double *pw_part, const double *gate) {
    double *const *ul[1] = {u};
    double *ol[1] = {out}, *pl[1] = {pw_part};
    const double *dl[1] = {pdot}, *gl[1] = {gate};
    return sem_opdiv_lanes(m, 1, ul, ol, scale, wts, face_grouped, dl, pl, gl);
}

// nl <= 4 velocity-mesh field triples -> nl pressure fields in one launch (block stepper)
int sem_opdiv_lanes(nlg_mesh *m, int nl, double *const *const *u, double *const *out, double scale, double *const *wts, bool face_grouped,
                    const double *const *pdot, double *const *pw_part, const double *const *gate) {
    ProfScope ps(m->ctx, (m->dim == 2 || face_grouped) ? P_OPDIV : P_VECOPS);
    CF3 wt = {{wts ? wts[0] : nullptr, wts ? wts[1] : nullptr, (wts && m->dim == 3) ? wts[2] : nullptr}};
    // face-grouped weights mask_i * binvm1: ONE array (binvm1) and one byte per point (bit i = mask_i) instead of three arrays
    // (8.5 instead of 24 bytes per point: the three weight arrays were 29 % of this kernel's bytes); same products, same bits
    static const bool use_mb = !(getenv("NLG_OPDIV_MASKB") && atoi(getenv("NLG_OPDIV_MASKB")) == 0);
    const unsigned char *mb = nullptr;
    CF3 wtn = wt;   // (k_opdiv3n only: the three-wave kernel of small meshes works out of the Infinity Cache, the older kernels keep their arrays)
    if (use_mb && m->dim == 3 && face_grouped && wts == m->d_mbinv_fg && m->d_maskb_fg && m->d_binv_fg) {
        mb = m->d_maskb_fg;
        wtn.p[0] = m->d_binv_fg;
        wtn.p[1] = wtn.p[2] = nullptr;
    }
    CF9 g = rst2w_ptrs(m);
    hipStream_t s = m->ctx->stream;
    if (m->dim == 3) {
        const bool old_big = pkern_old();
        static const bool n8new = !(getenv("NLG_PKERN_N8") && atoi(getenv("NLG_PKERN_N8")) == 0);   // lx1 = 8: one wave per element through the in-place kernels (5 - 6 % faster than <8, 3>; NLG_PKERN_N8=0 for A/B)
        CF3L ul;
        P4 ol, pl;
        CP4 dl, gl;
        for (int v = 0; v < 4; ++v) {
            for (int c = 0; c < 3; ++c) ul.p[v][c] = v < nl ? u[v][c] : nullptr;
            ol.p[v] = v < nl ? out[v] : nullptr;
            pl.p[v] = (v < nl && pw_part) ? pw_part[v] : nullptr;
            dl.p[v] = (v < nl && pdot) ? pdot[v] : nullptr;
            gl.p[v] = (v < nl && gate) ? gate[v] : nullptr;
        }
        if (m->n == 8 && nl == 1 && n8new && !old_big && sem_small_mesh(m)) {   // strong-scaling regime: three waves per element
            const CF3 u3 = {{u[0][0], u[0][1], u[0][2]}};
            const double *g0 = gate ? gate[0] : nullptr, *d0 = pdot ? pdot[0] : nullptr;
            double *p0 = pw_part ? pw_part[0] : nullptr;
            if (face_grouped)
                NLG_LAUNCH((k_opdiv3w<8, true>), dim3((unsigned)m->E), dim3(192), 0, s, m->E, (const double *)m->d_I12, (const double *)m->d_D12, (const int *)m->d_slot_fg, g, u3, wt, out[0], scale, d0, p0, g0);
            else
                NLG_LAUNCH((k_opdiv3w<8, false>), dim3((unsigned)m->E), dim3(192), 0, s, m->E, (const double *)m->d_I12, (const double *)m->d_D12, (const int *)m->d_slot_fg, g, u3, wt, out[0], scale, d0, p0, g0);
            NLG_HIP(hipGetLastError());
            return 0;
        }
#define DV3_(N_, ML_)                                                                                                        \
    {                                                                                                                  \
        PMats<N_> M;                                                                                                   \
        fill_pmats<N_>(m, M);                                                                                          \
        if (N_ <= 8 && face_grouped && !(N_ == 8 && n8new))                                                            \
            NLG_LAUNCH((k_opdiv3<N_, 3, true, ML_>), dim3((unsigned)m->E), dim3(NT), 0, s, m->E, M, g, ul, wt, ol, scale, dl, pl, gl, nl);  \
        else if (N_ <= 8 && !(N_ == 8 && n8new))                                                                       \
            NLG_LAUNCH((k_opdiv3<N_, 3, false, ML_>), dim3((unsigned)m->E), dim3(NT), 0, s, m->E, M, g, ul, wt, ol, scale, dl, pl, gl, nl); \
        else if (!old_big && face_grouped)                                                                             \
            NLG_LAUNCH((k_opdiv3n<(N_ >= 8 ? N_ : 9), true, ML_>), dim3((unsigned)m->E), dim3(PBlockN<N_>::NTB), 0, s, m->E, (const double *)m->d_I12, (const double *)m->d_D12, (const int *)m->d_slot_fg, g, ul, wtn, ol, scale, dl, pl, gl, nl, mb);  \
        else if (!old_big)                                                                                             \
            NLG_LAUNCH((k_opdiv3n<(N_ >= 8 ? N_ : 9), false, ML_>), dim3((unsigned)m->E), dim3(PBlockN<N_>::NTB), 0, s, m->E, (const double *)m->d_I12, (const double *)m->d_D12, (const int *)m->d_slot_fg, g, ul, wtn, ol, scale, dl, pl, gl, nl, mb); \
        else if (face_grouped)                                                                                         \
            NLG_LAUNCH((k_opdiv3<N_, 1, true, ML_>), dim3((unsigned)m->E), dim3(PBlock<N_, 1>::NTB), 0, s, m->E, M, g, ul, wt, ol, scale, dl, pl, gl, nl);  \
        else                                                                                                           \
            NLG_LAUNCH((k_opdiv3<N_, 1, false, ML_>), dim3((unsigned)m->E), dim3(PBlock<N_, 1>::NTB), 0, s, m->E, M, g, ul, wt, ol, scale, dl, pl, gl, nl); \
    }
#define DV3(N_)           \
    if (nl == 1)          \
        DV3_(N_, false)   \
    else                  \
        DV3_(N_, true)
        NLG_FOR_N(DV3)
#undef DV3
#undef DV3_
    } else {
        for (int v = 0; v < nl; ++v) {
            CF3 cu = {{u[v][0], u[v][1], nullptr}};
            double *ov = out[v], *pv = pw_part ? pw_part[v] : nullptr;
            const double *dv = pdot ? pdot[v] : nullptr, *gv = gate ? gate[v] : nullptr;
#define DV2(N_)                                                                                            \
    {                                                                                                      \
        constexpr int EPB = NT / (N_ * N_) > 0 ? NT / (N_ * N_) : 1;                                       \
        NLG_LAUNCH((k_opdiv2<N_>), dim3((unsigned)((m->E + EPB - 1) / EPB)), dim3(NT), 0, s, m->E, \
                           m->d_I12, m->d_D12, g, cu, wt, ov, scale, dv, pv, gv);                          \
    }
            NLG_FOR_N(DV2)
#undef DV2
        }
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

int sem_opbinv(nlg_mesh *m, double *const *w, int nl, int64_t ld) {
    NLG_TRY(sem_gs(m, w, m->dim, nullptr, LAYOUT_NAT, nl, ld, 0));
    ProfScope ps(m->ctx, P_COLMUL);
    F3 cw = {{w[0], w[1], m->dim == 3 ? w[2] : nullptr}};
    CF3 wt = {{m->d_mbinv[0], m->d_mbinv[1], m->d_mbinv[2]}};
    if (m->dim == 3)
        NLG_LAUNCH(k_colmul<3>, dim3(grid_for(m->lvn), nl), dim3(NT), 0, m->ctx->stream, cw, wt, m->lvn, ld);
    else
        NLG_LAUNCH(k_colmul<2>, dim3(grid_for(m->lvn), nl), dim3(NT), 0, m->ctx->stream, cw, wt, m->lvn, ld);
    NLG_HIP(hipGetLastError());
    return 0;
}

// E applied to nl <= 4 pressure fields (block stepper): gradient, gather-scatter, divergence; the two element kernels take
// all lanes in one launch
// the intermediates of the pressure operator can be kept in the face-grouped element layout (tables, gather-scatter lists and, on
// several ranks, halo index lists of that layout exist)
bool sem_opgradt_has_fg(const nlg_mesh *m) { return m->dim == 3 && m->gs.d_indices_fg && m->d_slot_fg && (!m->halo.active || m->halo.d_send_idx_fg); }

int sem_cdabdtp_lanes(nlg_mesh *m, int nl, const double *const *p, double *const *out, double *const *pw_part, const double *const *gate,
                      const nlg_pupd *upd) {
    const bool fg = sem_opgradt_has_fg(m);
    // the intermediate velocity-mesh fields of all lanes in ONE allocation at a constant stride, so that their gather-scatter
    // (and its halo exchange) is one launch with gridDim.y = lanes
    const int64_t ldw = 3 * m->lvs;
    if (!m->d_wlanes) {
        NLG_HIP(hipMalloc(&m->d_wlanes, sizeof(double) * (size_t)(kMaxLanes * ldw)));
        NLG_HIP(hipMemsetAsync(m->d_wlanes, 0, sizeof(double) * (size_t)(kMaxLanes * ldw), m->ctx->stream));
    }
    double *w[4][3];
    double *const *wl[4];
    for (int v = 0; v < nl; ++v) {
        for (int c = 0; c < 3; ++c) w[v][c] = c < m->dim ? m->d_wlanes + v * ldw + c * m->lvs : nullptr;
        wl[v] = w[v];
    }
    NLG_TRY(sem_opgradt_lanes(m, nl, p, wl, fg, gate, upd));
    // the gates of the lanes are the done flags of their solver scalars: at a constant stride when the lanes share a slab
    bool strided = true;
    int64_t ldg = 0;
    if (gate && nl > 1) {
        ldg = gate[1] - gate[0];
        for (int v = 1; v < nl; ++v) strided = strided && gate[v] && (gate[v] - gate[0]) == v * ldg;
    }
    if (strided) {
        NLG_TRY(sem_gs(m, w[0], m->dim, gate ? gate[0] : nullptr, fg ? LAYOUT_FG : LAYOUT_NAT, nl, ldw, ldg));
    } else {
        for (int v = 0; v < nl; ++v) NLG_TRY(sem_gs(m, w[v], m->dim, gate ? gate[v] : nullptr, fg ? LAYOUT_FG : LAYOUT_NAT));
    }
    // (p, E p): p AFTER the fused direction update, which may have gone to another buffer than the one it was read from (direction ring)
    const double *pd[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int v = 0; v < nl; ++v) pd[v] = (upd && upd[v].z && upd[v].p) ? upd[v].p : p[v];
    return sem_opdiv_lanes(m, nl, wl, out, 1.0, fg ? m->d_mbinv_fg : m->d_mbinv, fg, pd, pw_part, gate);
}

int sem_cdabdtp(nlg_mesh *m, const double *p, double *out, double *pw_part, const double *gate, const nlg_pupd *upd) {
    double *w[3] = {sem_scratch1(m, 0), sem_scratch1(m, 1), m->dim == 3 ? sem_scratch1(m, 2) : nullptr};
    NLG_CHECK(w[0] && w[1], "sem_cdabdtp: scratch allocation failed");
    if (m->dim == 3 && m->gs.d_indices_fg && (!m->halo.active || m->halo.d_send_idx_fg)) {
        // 3-D: the intermediate velocity-mesh fields use the face-grouped element layout, in which the copies of a
        // shared face are contiguous runs -> coalesced gather-scatter; the rank halo uses index lists in that layout
        NLG_TRY(sem_opgradt(m, p, w, true, gate, upd));
        NLG_TRY(sem_gs(m, w, 3, gate, LAYOUT_FG));
        NLG_TRY(sem_opdiv(m, w, out, 1.0, m->d_mbinv_fg, true, (upd && upd->z && upd->p) ? upd->p : p, pw_part, gate));   // (p, E p) with the UPDATED p
        return 0;
    }
    NLG_TRY(sem_opgradt(m, p, w, false, gate, upd));
    NLG_TRY(sem_gs(m, w, m->dim, gate));
    NLG_TRY(sem_opdiv(m, w, out, 1.0, m->d_mbinv, false, p, pw_part, gate));   // mask * binvm1 fused into the load
    return 0;
}

int sem_ortho(nlg_mesh *m, double *p, int nl, int64_t ld) {
    if (m->has_outflow) return 0;
    nlg_ctx *ctx = m->ctx;
    NLG_CHECK(nl >= 1 && nl <= kMaxLanes, "sem_ortho: %d lanes", nl);
    NLG_LAUNCH(k_sum_partial, dim3(NPART, nl), dim3(NT), 0, ctx->stream, p, m->lpn, ctx->d_partial, ld);
    if (ctx->distributed()) {
        double *d_sum = ctx->d_scalars + 4000;
        NLG_LAUNCH(k_sum_final, dim3(nl), dim3(NT), 0, ctx->stream, ctx->d_partial, NPART, d_sum);
        NLG_TRY(allreduce_sum(ctx, d_sum, nl));
        NLG_LAUNCH(k_sub_mean, dim3(grid_for(m->lpn), nl), dim3(NT), 0, ctx->stream, p, m->lpn, (const double *)d_sum, (const double *)nullptr, 0,
                   1.0 / (double)m->lpn_global, ld);
    } else {
        ++g_collectives;
        NLG_LAUNCH(k_sub_mean, dim3(grid_for(m->lpn), nl), dim3(NT), 0, ctx->stream, p, m->lpn, (const double *)nullptr, (const double *)ctx->d_partial,
                   NPART, 1.0 / (double)m->lpn_global, ld);
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

int sem_cfl(nlg_mesh *m, double *const *U, double dt, double *cfl_host) {
    nlg_ctx *ctx = m->ctx;
    CF9 r;
    for (int q = 0; q < 9; ++q) r.p[q] = m->d_rst[q];
    CF3 cu = {{U[0], U[1], m->dim == 3 ? U[2] : nullptr}};
    const int nb = 256;
    NLG_LAUNCH(k_cfl, dim3(nb), dim3(NT), 0, ctx->stream, m->dim, m->n, m->E, r, m->d_jac, m->d_rdr, cu, dt,
                       ctx->d_partial);
    double *d_out = ctx->d_scalars + 4001;
    NLG_LAUNCH(k_max_final, dim3(1), dim3(1), 0, ctx->stream, ctx->d_partial, nb, d_out);
    NLG_HIP(hipGetLastError());
    NLG_TRY(allreduce_max(ctx, d_out, 1));
    return scalars_to_host(ctx, 4001, 1, cfl_host);
}

// generic tensor apply launcher (set-up and convection paths)
int sem_tensor(nlg_mesh *m, const double *in, double *out, int nin, int nout, const double *Mx, const double *My,
               const double *Mz, const double *wt) {
    const int nmax = std::max(nin, nout);
    const size_t cap = (size_t)nmax * nmax * (m->dim == 3 ? nmax : 1);
    NLG_LAUNCH(k_tensor_generic, dim3((unsigned)m->E), dim3(NT), 2 * cap * sizeof(double), m->ctx->stream, in, out,
                       m->dim, nin, nout, Mx, My, Mz, wt, m->E);
    NLG_HIP(hipGetLastError());
    return 0;
}

// Pre-computation of the base-flow part of the convective term on the fine mesh:
//   Ur[j]      = sum_m rstdw[j][m] Uf_m                     (dim fields)
//   GU[i*dim+m] = sum_j rstdw[j][m] dU_i/dr_j               (dim^2 fields)
int sem_conv_setup(nlg_mesh *m, double *const *U, double **Ur, double **GU) {
    const int dim = m->dim;
    CF9 rd;
    for (int q = 0; q < 9; ++q) rd.p[q] = m->d_rstdw[q];
    static const int use_mfma = getenv("NLG_MFMA") ? atoi(getenv("NLG_MFMA")) : 1;
    if (dim == 3 && m->n == 8 && m->nd == 12 && use_mfma) {
        // matrix-core path: per component one launch produces the fine-mesh value and the three derivatives
        double *ufb[3] = {sem_scratchd(m, 0), sem_scratchd(m, 1), sem_scratchd(m, 2)};
        double *du[3] = {sem_scratchd(m, 3), sem_scratchd(m, 4), sem_scratchd(m, 5)};
        NLG_CHECK(ufb[0] && ufb[1] && ufb[2] && du[0] && du[1] && du[2], "sem_conv_setup: scratch allocation failed");
        for (int i = 0; i < 3; ++i) {
            NLG_LAUNCH((k_interp4_mfma<8, 12>), dim3((unsigned)m->E), dim3(256), 0, m->ctx->stream, m->E, (const double *)m->d_Jd,
                               (const double *)m->d_DJd, (const double *)U[i], ufb[i], du[0], du[1], du[2]);
            CF3 dd = {{du[0], du[1], du[2]}};
            F3 gu = {{GU[i * 3 + 0], GU[i * 3 + 1], GU[i * 3 + 2]}};
            NLG_LAUNCH(k_conv_gu, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, dim, m->lfn, rd, dd, gu);
        }
        CF3 uf = {{ufb[0], ufb[1], ufb[2]}};
        F3 ur = {{Ur[0], Ur[1], Ur[2]}};
        NLG_LAUNCH(k_conv_ur, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, dim, m->lfn, rd, uf, ur);
        NLG_HIP(hipGetLastError());
        return 0;
    }
    double *t[3] = {sem_scratchd(m, 0), sem_scratchd(m, 1), dim == 3 ? sem_scratchd(m, 2) : nullptr};
    NLG_CHECK(t[0] && t[1], "sem_conv_setup: scratch allocation failed");
    for (int i = 0; i < dim; ++i) NLG_TRY(sem_tensor(m, U[i], t[i], m->n, m->nd, m->d_Jd, m->d_Jd, m->d_Jd, nullptr));
    {
        CF3 uf = {{t[0], t[1], t[2]}};
        F3 ur = {{Ur[0], Ur[1], dim == 3 ? Ur[2] : nullptr}};
        NLG_LAUNCH(k_conv_ur, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, dim, m->lfn, rd, uf, ur);
    }
    for (int i = 0; i < dim; ++i) {
        for (int j = 0; j < dim; ++j)
            NLG_TRY(sem_tensor(m, U[i], t[j], m->n, m->nd, j == 0 ? m->d_DJd : m->d_Jd, j == 1 ? m->d_DJd : m->d_Jd,
                               j == 2 ? m->d_DJd : m->d_Jd, nullptr));
        CF3 du = {{t[0], t[1], t[2]}};
        F3 gu = {{GU[i * dim + 0], GU[i * dim + 1], dim == 3 ? GU[i * dim + 2] : nullptr}};
        NLG_LAUNCH(k_conv_gu, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, dim, m->lfn, rd, du, gu);
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

// Scalar (temperature) transport of the Boussinesq coupling: GT[m] = sum_j rstdw[j][m] dTheta/dr_j of the base temperature
// (the gradient part of u . grad Theta on the fine mesh), precomputed like GU
int sem_conv_scalar_setup(nlg_mesh *m, const double *Theta, double **GT) {
    const int dim = m->dim;
    CF9 rd;
    for (int q = 0; q < 9; ++q) rd.p[q] = m->d_rstdw[q];
    double *t[3] = {sem_scratchd(m, 0), sem_scratchd(m, 1), dim == 3 ? sem_scratchd(m, 2) : nullptr};
    NLG_CHECK(t[0] && t[1], "sem_conv_scalar_setup: scratch allocation failed");
    for (int j = 0; j < dim; ++j)
        NLG_TRY(sem_tensor(m, Theta, t[j], m->n, m->nd, j == 0 ? m->d_DJd : m->d_Jd, j == 1 ? m->d_DJd : m->d_Jd,
                           j == 2 ? m->d_DJd : m->d_Jd, nullptr));
    CF3 du = {{t[0], t[1], t[2]}};
    F3 gt = {{GT[0], GT[1], dim == 3 ? GT[2] : nullptr}};
    NLG_LAUNCH(k_conv_gu, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, dim, m->lfn, rd, du, gt);
    NLG_HIP(hipGetLastError());
    return 0;
}

// out = J^T W [(U . grad) theta + (u . grad) Theta]   (weak, element-local), oracle: conv_weak(U, theta) + conv_weak(u, Theta)
// out_i += sgn * J^T [ theta_f GT_i ]: the temperature term of the ADJOINT momentum equation (- theta+ grad Theta), the transpose
// of the u . grad Theta part of sem_conv_scalar_apply; generic tensor kernels (the coupled adjoint is not a hot path)
__global__ void k_mul_fine(int64_t n, const double *a, const double *b, double *o) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) o[q] = a[q] * b[q];
}
__global__ void k_axpy_field(int64_t n, double *y, const double *x, double s) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) y[q] += s * x[q];
}
int sem_scalar_grad_apply(nlg_mesh *m, double *const *GT, const double *theta, double *const *out, double sgn) {
    ProfScope ps(m->ctx, P_CONV);
    const int dim = m->dim;
    double *tf = sem_scratchd(m, 0), *prod = sem_scratchd(m, 1), *back = sem_scratch1(m, 4);
    NLG_CHECK(tf && prod && back, "sem_scalar_grad_apply: scratch allocation failed");
    NLG_TRY(sem_tensor(m, theta, tf, m->n, m->nd, m->d_Jd, m->d_Jd, m->d_Jd, nullptr));
    for (int i = 0; i < dim; ++i) {
        NLG_LAUNCH(k_mul_fine, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, m->lfn, (const double *)tf, (const double *)GT[i], prod);
        NLG_TRY(sem_tensor(m, prod, back, m->nd, m->n, m->d_Jdt, m->d_Jdt, m->d_Jdt, nullptr));
        NLG_LAUNCH(k_axpy_field, dim3(grid_for(m->lvn)), dim3(NT), 0, m->ctx->stream, m->lvn, out[i], (const double *)back, sgn);
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

// adjoint != 0: out = - J^T [ Ur . grad theta ]  (the transport term of the adjoint temperature equation; no u . grad Theta)
int sem_conv_scalar_apply(nlg_mesh *m, double *const *Ur, double *const *GT, double *const *u, const double *theta, double *out, int adjoint) {
    ProfScope ps(m->ctx, P_CONV);
    const int dim = m->dim;
    static const bool sweep = !(getenv("NLG_CONVS_SWEEP") && atoi(getenv("NLG_CONVS_SWEEP")) == 0);   // A/B: the generic tensor kernels
    if (sweep && dim == 3 && m->n >= 8 && m->n <= 10 && m->nd == (3 * m->n) / 2) {
        CF3 cur = {{Ur[0], Ur[1], Ur[2]}}, cgt = {{GT[0], GT[1], GT[2]}}, cu = {{u[0], u[1], u[2]}};
        static const bool mfma = !(getenv("NLG_CONV_MFMA") && atoi(getenv("NLG_CONV_MFMA")) == 0);   // A/B: 0 = the plane-sweep kernel
        if (mfma && m->n == 8) {
            NLG_LAUNCH((k_conv3m_scalar<8, 12, false>), dim3((unsigned)m->E), dim3(256), 0, m->ctx->stream, m->E, (const double *)m->d_Jd, (const double *)m->d_DJd,
                       cur, cgt, cu, theta, out, adjoint);
            NLG_HIP(hipGetLastError());
            return 0;
        }
        if (mfma && m->n == 10) {
            constexpr size_t lds = sizeof(double) * (2 * 16 * 100 + 3 * 225 * 10);
            static bool attr_set = false;
            if (!attr_set) {
                NLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3m_scalar<10, 15, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_set = true;
            }
            NLG_LAUNCH((k_conv3m_scalar<10, 15, true>), dim3((unsigned)m->E), dim3(256), lds, m->ctx->stream, m->E, (const double *)m->d_Jd, (const double *)m->d_DJd,
                       cur, cgt, cu, theta, out, adjoint);
            NLG_HIP(hipGetLastError());
            return 0;
        }
#define CVS(N_, NW_)                                                                                                                         \
    NLG_LAUNCH((k_conv3s_scalar<N_, (3 * N_) / 2, NW_>), dim3((unsigned)m->E), dim3(NW_ * 64), 0, m->ctx->stream, m->E, (const double *)m->d_Jd, \
               (const double *)m->d_DJd, (const double *)m->d_Jdt, cur, cgt, cu, theta, out, adjoint);
        if (m->n == 8) CVS(8, 4) else if (m->n == 9) CVS(9, 6) else CVS(10, 7)
#undef CVS
        NLG_HIP(hipGetLastError());
        return 0;
    }
    double *uf[3] = {sem_scratchd(m, 0), sem_scratchd(m, 1), dim == 3 ? sem_scratchd(m, 2) : nullptr};
    double *dt[3] = {sem_scratchd(m, 3), sem_scratchd(m, 4), dim == 3 ? sem_scratchd(m, 5) : nullptr};
    double *acc = sem_scratchd(m, 6);
    NLG_CHECK(uf[0] && dt[0] && acc, "sem_conv_scalar_apply: scratch allocation failed");
    for (int i = 0; i < dim; ++i) NLG_TRY(sem_tensor(m, u[i], uf[i], m->n, m->nd, m->d_Jd, m->d_Jd, m->d_Jd, nullptr));
    for (int j = 0; j < dim; ++j)
        NLG_TRY(sem_tensor(m, theta, dt[j], m->n, m->nd, j == 0 ? m->d_DJd : m->d_Jd, j == 1 ? m->d_DJd : m->d_Jd,
                           j == 2 ? m->d_DJd : m->d_Jd, nullptr));
    CF3 cur = {{Ur[0], Ur[1], dim == 3 ? Ur[2] : nullptr}};
    CF3 cdt = {{dt[0], dt[1], dt[2]}}, cuf = {{uf[0], uf[1], uf[2]}};
    CF3 cgt = {{GT[0], GT[1], dim == 3 ? GT[2] : nullptr}};
    if (adjoint) {
        NLG_LAUNCH(k_conv_combine_adj, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, dim, m->lfn, cur, cdt, acc);
    } else {
        NLG_LAUNCH(k_conv_combine, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, dim, m->lfn, cur, cdt, cuf, cgt, 1.0, acc);
    }
    NLG_TRY(sem_tensor(m, acc, out, m->nd, m->n, m->d_Jdt, m->d_Jdt, m->d_Jdt, nullptr));
    NLG_HIP(hipGetLastError());
    return 0;
}

// out_i = weak linearised convective term (B-weighted, element-local), see oracle/sem.py lns_conv_weak
int sem_conv_apply(nlg_mesh *m, double *const *Ur, double *const *GU, double *const *u, double *const *out, int adjoint) {
    double *const *ul[1] = {u}, *const *ol[1] = {out};
    return sem_conv_apply_lanes(m, Ur, GU, 1, ul, ol, adjoint);
}

// nl <= 4 vectors against the same base flow (block stepper): one launch where the fused kernel exists
int sem_conv_apply_lanes(nlg_mesh *m, double *const *Ur, double *const *GU, int nl, double *const *const *ulanes, double *const *const *olanes, int adjoint) {
    const int dim = m->dim;
    if (!(dim == 3 && (m->n <= 10 || m->n == 12) && m->nd == (3 * m->n) / 2)) {
        for (int v = 0; v < nl; ++v) NLG_TRY(sem_conv_apply_generic(m, Ur, GU, ulanes[v], olanes[v], adjoint));
        return 0;
    }
    ProfScope ps(m->ctx, P_CONV);
    {
        // fused kernel: static LDS up to lx1 = 8, dynamic LDS (one block per CU) for lx1 = 9, 10, 12
        CF3 cur = {{Ur[0], Ur[1], Ur[2]}};
        CF3L cu;
        F3L co;
        for (int v = 0; v < 4; ++v)
            for (int c = 0; c < 3; ++c) {
                cu.p[v][c] = v < nl ? ulanes[v][c] : nullptr;
                co.p[v][c] = v < nl ? olanes[v][c] : nullptr;
            }
        CF9 cg;
        for (int q = 0; q < 9; ++q) cg.p[q] = GU[q];
#define CV3(N_)                                                                                                       \
    if (nl == 1)                                                                                                      \
        NLG_LAUNCH((k_conv3<N_, (3 * N_) / 2, NT, true, false, false>), dim3((unsigned)m->E), dim3(NT), 0, m->ctx->stream, m->E, \
                           (const double *)m->d_Jd, (const double *)m->d_DJd, cur, cg, cu, co, nl, adjoint);          \
    else                                                                                                              \
        NLG_LAUNCH((k_conv3<N_, (3 * N_) / 2, NT, true, false, true>), dim3((unsigned)m->E), dim3(NT), 0, m->ctx->stream, m->E, \
                           (const double *)m->d_Jd, (const double *)m->d_DJd, cur, cg, cu, co, nl, adjoint);
#define CV3D(N_, NTC_, ULDS_)                                                                                          \
    {                                                                                                                  \
        constexpr int ND_ = (3 * N_) / 2, NQ_ = N_ | 1, NDQ_ = ND_ | 1;                                                \
        constexpr size_t lds = sizeof(double) * (2 * NDQ_ * N_ * N_ + 3 * ND_ * ND_ * N_ + (ULDS_ ? 3 * NQ_ * N_ * N_ : 0)); \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            NLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3<N_, ND_, NTC_, ULDS_, true>),          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        NLG_LAUNCH((k_conv3<N_, ND_, NTC_, ULDS_, true>), dim3((unsigned)m->E), dim3(NTC_), lds, m->ctx->stream, m->E, \
                           (const double *)m->d_Jd, (const double *)m->d_DJd, cur, cg, cu, co, nl, adjoint);          \
    }
#define CV3S(N_, NW_, MINB_)                                                                                                    \
    NLG_LAUNCH((k_conv3s<N_, (3 * N_) / 2, NW_, MINB_>), dim3((unsigned)(m->E * nl)), dim3(NW_ * 64), 0, m->ctx->stream, m->E, (const double *)m->d_Jd, \
               (const double *)m->d_DJd, (const double *)m->d_Jdt, cur, cg, cu, co, nl, adjoint);
        static const int sweep = getenv("NLG_CONV_SWEEP") ? atoi(getenv("NLG_CONV_SWEEP")) : 1;   // A/B: 0 = the LDS-image kernel k_conv3
        switch (m->n) {
            case 4: CV3(4); break;
            case 5: CV3(5); break;
            case 6: CV3(6); break;
            case 7: CV3(7); break;
            case 8: {
                static const bool mfma = !(getenv("NLG_CONV_MFMA") && atoi(getenv("NLG_CONV_MFMA")) == 0);   // A/B: 0 = the vector-pipe kernel k_conv3
                if (sweep == 3)
                    CV3S(8, 3, 3)
                else if (mfma)
                    NLG_LAUNCH((k_conv3m<8, 12, false>), dim3((unsigned)m->E, (unsigned)nl), dim3(256), 0, m->ctx->stream, m->E, (const double *)m->d_Jd,
                               (const double *)m->d_DJd, cur, cg, cu, co, adjoint);
                else
                    CV3(8);   // (measured: see DESIGN.md section 5)
            } break;
            case 9: CV3D(9, 256, true); break;
            case 10: {
                static const bool mfma = !(getenv("NLG_CONV_MFMA") && atoi(getenv("NLG_CONV_MFMA")) == 0);   // A/B: 0 = the plane-sweep kernel k_conv3s
                if (mfma) {
                    constexpr size_t lds = sizeof(double) * (2 * 16 * 100 + 3 * 225 * 10);   // 79.6 KB: two blocks per CU
                    static bool attr_set = false;
                    if (!attr_set) {
                        NLG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3m<10, 15, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                        attr_set = true;
                    }
                    NLG_LAUNCH((k_conv3m<10, 15, true>), dim3((unsigned)m->E, (unsigned)nl), dim3(256), lds, m->ctx->stream, m->E, (const double *)m->d_Jd,
                               (const double *)m->d_DJd, cur, cg, cu, co, adjoint);
                } else if (sweep)
                    CV3S(10, 5, 2)
                else
                    CV3D(10, 256, false);   // (u from global memory: 78 KB of LDS instead of 104 KB, two blocks per CU)
            } break;
            default:
                // lx1 = 12 stays on the plane-sweep kernel: lxd = 18 needs a SECOND 16-row tile for fine rows 16, 17 in every forward
                // pass (3252 MFMA per element against 1320 at lx1 = 10) and 137 KB of LDS, i.e. one block per CU; built and measured
                // (k_conv3m<12, 18>, parity green): 3.10 ms per launch against 3.10 ms -- a tie, so the instantiation was dropped
                if (sweep) CV3S(12, 7, 1) else CV3D(12, 384, false);
                break;
        }
#undef CV3
#undef CV3D
#undef CV3S
        NLG_HIP(hipGetLastError());
        return 0;
    }
}

// the generic path: tensor-product kernels and fine-mesh intermediates (2-D, lx1 = 11, non-standard lxd)
int sem_conv_apply_generic(nlg_mesh *m, double *const *Ur, double *const *GU, double *const *u, double *const *out, int adjoint) {
    ProfScope ps(m->ctx, P_CONV);
    const int dim = m->dim;
    double *uf[3] = {sem_scratchd(m, 0), sem_scratchd(m, 1), dim == 3 ? sem_scratchd(m, 2) : nullptr};
    double *du[3] = {sem_scratchd(m, 3), sem_scratchd(m, 4), dim == 3 ? sem_scratchd(m, 5) : nullptr};
    double *acc = sem_scratchd(m, 6);
    NLG_CHECK(uf[0] && du[0] && acc, "sem_conv_apply: scratch allocation failed");
    for (int i = 0; i < dim; ++i) NLG_TRY(sem_tensor(m, u[i], uf[i], m->n, m->nd, m->d_Jd, m->d_Jd, m->d_Jd, nullptr));
    CF3 cur = {{Ur[0], Ur[1], dim == 3 ? Ur[2] : nullptr}};
    CF3 cuf = {{uf[0], uf[1], uf[2]}};
    for (int i = 0; i < dim; ++i) {
        for (int j = 0; j < dim; ++j)
            NLG_TRY(sem_tensor(m, u[i], du[j], m->n, m->nd, j == 0 ? m->d_DJd : m->d_Jd, j == 1 ? m->d_DJd : m->d_Jd,
                               j == 2 ? m->d_DJd : m->d_Jd, nullptr));
        CF3 cdu = {{du[0], du[1], du[2]}};
        CF3 gsel;
        for (int mm = 0; mm < 3; ++mm) gsel.p[mm] = nullptr;
        // direct: GU[i][m] ; adjoint: GU[m][i]
        for (int mm = 0; mm < dim; ++mm) gsel.p[mm] = adjoint ? GU[mm * dim + i] : GU[i * dim + mm];
        NLG_LAUNCH(k_conv_combine, dim3(grid_for(m->lfn)), dim3(NT), 0, m->ctx->stream, dim, m->lfn, cur, cdu, cuf,
                           gsel, adjoint ? -1.0 : 1.0, acc);
        NLG_TRY(sem_tensor(m, acc, out[i], m->nd, m->n, m->d_Jdt, m->d_Jdt, m->d_Jdt, nullptr));
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

int sem_ediag(nlg_mesh *m, double *out) {
    // diag_k = sum_i sum_{j,jj} g_ji,k g_jji,k [ (M_j o M_jj) applied to c_i ]_k , c_i = mask_i binvm1
    const int dim = m->dim, n = m->n, n2 = m->n2;
    const nlg_ops1d &o = m->ops;
    // elementwise-product matrices II, ID, DD (n2 x n)
    std::vector<double> II((size_t)n2 * n), ID((size_t)n2 * n), DD((size_t)n2 * n);
    for (int q = 0; q < n2 * n; ++q) {
        II[q] = o.I12[q] * o.I12[q];
        ID[q] = o.I12[q] * o.D12[q];
        DD[q] = o.D12[q] * o.D12[q];
    }
    double *dII, *dID, *dDD;
    NLG_TRY(upload(II, &dII));
    NLG_TRY(upload(ID, &dID));
    NLG_TRY(upload(DD, &dDD));
    double *tmp = sem_scratch2(m, 6), *acc = out;
    NLG_CHECK(tmp, "sem_ediag: scratch allocation failed");
    NLG_HIP(hipMemsetAsync(acc, 0, sizeof(double) * (size_t)m->lps, m->ctx->stream));
    double *prod = sem_scratch2(m, 7);
    for (int i = 0; i < dim; ++i)
        for (int j = 0; j < dim; ++j)
            for (int jj = 0; jj < dim; ++jj) {
                const double *M[3];
                for (int ax = 0; ax < 3; ++ax) {
                    const bool a = (ax == j), b = (ax == jj);
                    M[ax] = (a && b) ? dDD : ((a || b) ? dID : dII);
                }
                NLG_TRY(sem_tensor(m, m->d_mbinv[i], tmp, n, n2, M[0], M[1], M[2], nullptr));
                NLG_LAUNCH(k_mul, dim3(grid_for(m->lpn)), dim3(NT), 0, m->ctx->stream, prod, m->d_rst2w[j * dim + i],
                                   m->d_rst2w[jj * dim + i], m->lpn);
                // acc += prod * tmp  (reuse k_mul then add via colmul-less path)
                NLG_LAUNCH(k_mul, dim3(grid_for(m->lpn)), dim3(NT), 0, m->ctx->stream, tmp, prod, tmp, m->lpn);
                NLG_LAUNCH(k_addto, dim3(grid_for(m->lpn)), dim3(NT), 0, m->ctx->stream, acc, tmp, m->lpn);
            }
    NLG_HIP(hipStreamSynchronize(m->ctx->stream));
    hipFree(dII);
    hipFree(dID);
    hipFree(dDD);
    return 0;
}

}  // namespace nlg

// =================================================================================================
// C ABI: mesh
// =================================================================================================
extern "C" {

int nlg_mesh_create(nlg_ctx *ctx, const nlg_mesh_desc *d, nlg_mesh **out) {
    NLG_CHECK(ctx && d && out, "nlg_mesh_create: NULL argument");
    NLG_CHECK(d->dim == 2 || d->dim == 3, "nlg_mesh_create: dim must be 2 or 3 (got %d)", d->dim);
    NLG_CHECK(d->n >= 4 && d->n <= 12 && d->n != 11, "nlg_mesh_create: lx1 = %d unsupported (4..10, 12)", d->n);
    NLG_CHECK(d->nelv >= 1, "nlg_mesh_create: nelv must be >= 1 (got %lld)", (long long)d->nelv);
    NLG_CHECK(d->xm1 && d->ym1 && (d->dim == 2 || d->zm1), "nlg_mesh_create: coordinate arrays missing");
    NLG_CHECK(d->glo_num, "nlg_mesh_create: glo_num missing");
    NLG_CHECK(d->v1mask && d->v2mask && (d->dim == 2 || d->v3mask), "nlg_mesh_create: velocity masks missing");
    NLG_HIP(hipSetDevice(ctx->device));
    nlg_mesh *m = new nlg_mesh();
    m->ctx = ctx;
    m->dim = d->dim;
    m->n = d->n;
    m->n2 = d->n - 2;
    m->nd = d->lxd > 0 ? d->lxd : (3 * d->n) / 2;
    NLG_CHECK(m->nd >= m->n && m->nd <= 18, "nlg_mesh_create: lxd = %d out of range", m->nd);
    m->E = d->nelv;
    const int dim = m->dim, n = m->n, n2 = m->n2, nd = m->nd;
    m->np1 = n * n * (dim == 3 ? n : 1);
    m->np2 = n2 * n2 * (dim == 3 ? n2 : 1);
    m->npd = nd * nd * (dim == 3 ? nd : 1);
    m->lvn = m->E * m->np1;
    m->lpn = m->E * m->np2;
    m->lfn = m->E * m->npd;
    NLG_CHECK(m->lvn < (int64_t)2000000000, "nlg_mesh_create: local dof count %lld exceeds 32-bit gather-scatter indices",
              (long long)m->lvn);
    m->lvs = round_up(m->lvn, kAlign);
    m->lps = round_up(m->lpn, kAlign);
    m->has_outflow = d->has_outflow;
    hipStream_t s = ctx->stream;

    // ---- 1-D operators
    nlg_ops1d &o = m->ops;
    o.n = n;
    o.n2 = n2;
    o.nd = nd;
    gll_nodes(n, o.z1, o.w1);
    gl_nodes(n2, o.z2, o.w2);
    gl_nodes(nd, o.zd, o.wd);
    o.D = deriv_mat(o.z1);
    o.I12 = interp_mat(o.z1, o.z2);
    o.D12 = matmul(o.I12, o.D, n2, n, n);
    o.Jd = interp_mat(o.z1, o.zd);
    o.DJd = matmul(o.Jd, o.D, nd, n, n);
    o.rdr.resize(n);
    for (int i = 0; i < n; ++i) {
        double dr;
        if (i == 0)
            dr = o.z1[1] - o.z1[0];
        else if (i == n - 1)
            dr = o.z1[n - 1] - o.z1[n - 2];
        else
            dr = 0.5 * (o.z1[i + 1] - o.z1[i - 1]);
        o.rdr[i] = 1.0 / dr;
    }
    NLG_TRY(upload(o.D, &m->d_D));
    NLG_TRY(upload(transpose(o.D, n, n), &m->d_Dt));
    NLG_TRY(upload(o.I12, &m->d_I12));
    NLG_TRY(upload(transpose(o.I12, n2, n), &m->d_I12t));
    NLG_TRY(upload(o.D12, &m->d_D12));
    NLG_TRY(upload(transpose(o.D12, n2, n), &m->d_D12t));
    NLG_TRY(upload(o.Jd, &m->d_Jd));
    NLG_TRY(upload(transpose(o.Jd, nd, n), &m->d_Jdt));
    NLG_TRY(upload(o.DJd, &m->d_DJd));
    NLG_TRY(upload(transpose(o.DJd, nd, n), &m->d_DJdt));
    NLG_TRY(upload(o.rdr, &m->d_rdr));
    NLG_TRY(upload(o.w1, &m->d_w1));
    NLG_TRY(upload(o.w2, &m->d_w2));
    NLG_TRY(upload(o.wd, &m->d_wd));

    // ---- coordinates, masks
    const double *hx[3] = {d->xm1, d->ym1, d->zm1};
    const double *hm[3] = {d->v1mask, d->v2mask, d->v3mask};
    for (int c = 0; c < dim; ++c) {
        NLG_TRY(dalloc(&m->d_x[c], m->lvs, s));
        NLG_HIP(hipMemcpyAsync(m->d_x[c], hx[c], sizeof(double) * (size_t)m->lvn, hipMemcpyHostToDevice, s));
        NLG_TRY(dalloc(&m->d_mask[c], m->lvs, s));
        NLG_HIP(hipMemcpyAsync(m->d_mask[c], hm[c], sizeof(double) * (size_t)m->lvn, hipMemcpyHostToDevice, s));
        NLG_TRY(dalloc(&m->d_mbinv[c], m->lvs, s));
    }
    NLG_TRY(dalloc(&m->d_tmask, m->lvs, s));
    if (d->tmask)
        NLG_HIP(hipMemcpyAsync(m->d_tmask, d->tmask, sizeof(double) * (size_t)m->lvn, hipMemcpyHostToDevice, s));
    else
        NLG_LAUNCH(k_set, dim3(grid_for(m->lvn)), dim3(NT), 0, s, m->d_tmask, 1.0, m->lvn);
    m->h_lglel.resize(m->E);
    for (int64_t e = 0; e < m->E; ++e) m->h_lglel[e] = d->lglel ? d->lglel[e] : e;
    NLG_HIP(hipMalloc(&m->d_lglel, sizeof(int64_t) * (size_t)m->E));
    NLG_HIP(hipMemcpyAsync(m->d_lglel, m->h_lglel.data(), sizeof(int64_t) * (size_t)m->E, hipMemcpyHostToDevice, s));

    // ---- geometry
    for (int q = 0; q < dim * dim; ++q) NLG_TRY(dalloc(&m->d_rst[q], m->lvs, s));
    NLG_TRY(dalloc(&m->d_jac, m->lvs, s));
    NLG_TRY(dalloc(&m->d_bm1, m->lvs, s));
    NLG_TRY(dalloc(&m->d_binvm1, m->lvs, s));
    NLG_TRY(dalloc(&m->d_vmult, m->lvs, s));
    const int ng = dim == 3 ? 6 : 3;
    for (int q = 0; q < ng; ++q) NLG_TRY(dalloc(&m->d_G[q], m->lvs, s));
    int *d_bad;
    NLG_HIP(hipMalloc(&d_bad, sizeof(int)));
    NLG_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), s));
    {
        CF3 X = {{m->d_x[0], m->d_x[1], m->d_x[2]}};
        F9 r;
        for (int q = 0; q < 9; ++q) r.p[q] = m->d_rst[q];
        NLG_LAUNCH(k_geom, dim3((unsigned)m->E), dim3(NT), sizeof(double) * 3 * m->np1, s, dim, n, m->d_D, m->d_w1, X, r,
                           m->d_jac, m->d_bm1, m->d_G[0], m->d_G[1], m->d_G[2], m->d_G[3], m->d_G[4], m->d_G[5], d_bad);
        NLG_HIP(hipGetLastError());
    }
    int bad = 0;
    NLG_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s));
    NLG_HIP(hipStreamSynchronize(s));
    hipFree(d_bad);
    if (bad) {
        set_error("nlg_mesh_create: non-positive Jacobian in the mesh");
        return 1;
    }

    // ---- gather-scatter set-up (host): groups of local dofs sharing a label
    {
        std::vector<int> order((size_t)m->lvn);
        std::iota(order.begin(), order.end(), 0);
        const int64_t *glo = d->glo_num;
        std::sort(order.begin(), order.end(), [glo](int a, int b) { return glo[a] < glo[b] || (glo[a] == glo[b] && a < b); });
        std::vector<int> off, idx;
        off.push_back(0);
        // collect groups, then order groups by their first (smallest) local index for locality
        std::vector<std::pair<int, std::pair<int, int>>> groups;   // (first index, (begin, end) in `order`)
        int64_t b = 0;
        while (b < m->lvn) {
            int64_t e = b + 1;
            while (e < m->lvn && glo[order[e]] == glo[order[b]]) ++e;
            if (e - b >= 2) groups.push_back({order[b], {(int)b, (int)e}});
            b = e;
        }
        // pairs first, each class ordered by its first (smallest) local index
        auto cls = [](int sz) { return sz == 2 ? 0 : (sz == 4 ? 1 : 2); };
        auto gsz = [](const std::pair<int, std::pair<int, int>> &g) { return g.second.second - g.second.first; };
        std::sort(groups.begin(), groups.end(), [&](const auto &a, const auto &b) {
            const int ca = cls(gsz(a)), cb = cls(gsz(b));
            return ca != cb ? ca < cb : a < b;
        });
        m->gs.npairs = (int64_t)std::count_if(groups.begin(), groups.end(), [&](const auto &g) { return gsz(g) == 2; });
        m->gs.nquads = (int64_t)std::count_if(groups.begin(), groups.end(), [&](const auto &g) { return gsz(g) == 4; });
        for (auto &g : groups) {
            for (int q = g.second.first; q < g.second.second; ++q) idx.push_back(order[q]);
            off.push_back((int)idx.size());
        }
        m->gs.ngroups = (int64_t)groups.size();
        m->gs.nshared = (int64_t)idx.size();
        m->gs.h_groups.clear();
        if (m->ctx->distributed())   // halo_setup splits them by "holds a dof another rank shares"
            for (size_t gi = 0; gi + 1 < off.size(); ++gi) m->gs.h_groups.emplace_back(idx.begin() + off[gi], idx.begin() + off[gi + 1]);
        if (dim == 3 && !groups.empty()) {
            // the same groups in face-grouped numbering, re-sorted by their first index
            std::vector<int> slot((size_t)m->np1);
            for (int p = 0; p < m->np1; ++p) slot[p] = fg_slot(n, p % n, (p / n) % n, p / (n * n));
            m->h_slot = slot;
            NLG_HIP(hipMalloc(&m->d_slot_fg, sizeof(int) * slot.size()));
            NLG_HIP(hipMemcpy(m->d_slot_fg, slot.data(), sizeof(int) * slot.size(), hipMemcpyHostToDevice));
            std::vector<std::vector<int>> gl(groups.size());
            for (size_t gi = 0; gi + 1 < off.size(); ++gi) {
                for (int q = off[gi]; q < off[gi + 1]; ++q) gl[gi].push_back((idx[q] / m->np1) * m->np1 + slot[idx[q] % m->np1]);
                std::sort(gl[gi].begin(), gl[gi].end());
            }
            std::sort(gl.begin(), gl.end(), [](const std::vector<int> &a, const std::vector<int> &b) {
                const int ca = a.size() == 2 ? 0 : (a.size() == 4 ? 1 : 2), cb = b.size() == 2 ? 0 : (b.size() == 4 ? 1 : 2);
                return ca != cb ? ca < cb : a[0] < b[0];
            });
            std::vector<int> off2{0}, idx2;
            for (auto &v : gl) {
                idx2.insert(idx2.end(), v.begin(), v.end());
                off2.push_back((int)idx2.size());
            }
            NLG_HIP(hipMalloc(&m->gs.d_offsets_fg, sizeof(int) * off2.size()));
            NLG_HIP(hipMalloc(&m->gs.d_indices_fg, sizeof(int) * idx2.size()));
            NLG_HIP(hipMemcpy(m->gs.d_offsets_fg, off2.data(), sizeof(int) * off2.size(), hipMemcpyHostToDevice));
            NLG_HIP(hipMemcpy(m->gs.d_indices_fg, idx2.data(), sizeof(int) * idx2.size(), hipMemcpyHostToDevice));
        }
        if (dim == 3) {
            // ... and in the x-planes-first numbering (velocity PCG)
            std::vector<int> slot((size_t)m->np1);
            for (int p = 0; p < m->np1; ++p) slot[p] = xp_slot(n, p % n, (p / n) % n, p / (n * n));
            m->h_slot_xp = slot;
            {   // device copy with one more entry: 1 = the slabs of an element are NOT all permuted alike (the operator kernels then look every point up)
                std::vector<int> dev(slot);
                int general = 0;
                for (int p = 0; p < m->np1; ++p) general |= (slot[p] != slot[p % (n * n)] + (p / (n * n)) * n * n);
                dev.push_back(general);
                NLG_HIP(hipMalloc(&m->d_slot_xp, sizeof(int) * dev.size()));
                NLG_HIP(hipMemcpy(m->d_slot_xp, dev.data(), sizeof(int) * dev.size(), hipMemcpyHostToDevice));
            }
            if (!groups.empty()) {
                std::vector<std::vector<int>> gl(groups.size());
                for (size_t gi = 0; gi + 1 < off.size(); ++gi) {
                    for (int q = off[gi]; q < off[gi + 1]; ++q) gl[gi].push_back((idx[q] / m->np1) * m->np1 + slot[idx[q] % m->np1]);
                    std::sort(gl[gi].begin(), gl[gi].end());
                }
                // pairs ordered by (element, partner element, index): inside a slab the rows of the three faces alternate, and
                // ordering by the first index alone makes the partner side hop between three neighbour elements every
                // 6 - 8 pairs (measured: 67 us against 49 us for the face-by-face order of the face-grouped tables)
                const int np1 = m->np1;
                std::sort(gl.begin(), gl.end(), [np1](const std::vector<int> &a, const std::vector<int> &b) {
                    const int ca = a.size() == 2 ? 0 : (a.size() == 4 ? 1 : 2), cb = b.size() == 2 ? 0 : (b.size() == 4 ? 1 : 2);
                    if (ca != cb) return ca < cb;
                    if (ca == 0) {
                        const int ea = a[0] / np1, eb = b[0] / np1, pa = a[1] / np1, pb = b[1] / np1;
                        if (ea != eb) return ea < eb;
                        if (pa != pb) return pa < pb;
                    }
                    return a[0] < b[0];
                });
                // (round 4: the faces in CHAIN order -- face of e towards p, then the face of p opposite to it, ... -- so that the two users of
                //  a slab line, which holds the rows of two opposite faces, are lanes of one block: built from the pair list for any conforming
                //  mesh, parity green, 4.91 -> 4.84 ms of gather-scatter per step: the second fetch of such a line was coming from the Infinity
                //  Cache, not from HBM.  Not kept.)
                std::vector<int> off2{0}, idx2;
                for (auto &v : gl) {
                    idx2.insert(idx2.end(), v.begin(), v.end());
                    off2.push_back((int)idx2.size());
                }
                NLG_HIP(hipMalloc(&m->gs.d_offsets_xp, sizeof(int) * off2.size()));
                NLG_HIP(hipMalloc(&m->gs.d_indices_xp, sizeof(int) * idx2.size()));
                NLG_HIP(hipMemcpy(m->gs.d_offsets_xp, off2.data(), sizeof(int) * off2.size(), hipMemcpyHostToDevice));
                NLG_HIP(hipMemcpy(m->gs.d_indices_xp, idx2.data(), sizeof(int) * idx2.size(), hipMemcpyHostToDevice));
            }
        }
        NLG_HIP(hipMalloc(&m->gs.d_offsets, sizeof(int) * off.size()));
        NLG_HIP(hipMalloc(&m->gs.d_indices, sizeof(int) * std::max<size_t>(idx.size(), 1)));
        NLG_HIP(hipMemcpy(m->gs.d_offsets, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice));
        if (!idx.empty()) NLG_HIP(hipMemcpy(m->gs.d_indices, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    }
    NLG_TRY(halo_setup(m, d->glo_num));
    m->gs.h_groups.clear();
    m->gs.h_groups.shrink_to_fit();
    // ---- multiplicity, assembled inverse mass, fused opbinv weights
    {
        NLG_LAUNCH(k_set, dim3(grid_for(m->lvn)), dim3(NT), 0, s, m->d_vmult, 1.0, m->lvn);
        double *f[1] = {m->d_vmult};
        NLG_TRY(sem_gs(m, f, 1));
        NLG_LAUNCH(k_recip, dim3(grid_for(m->lvn)), dim3(NT), 0, s, m->d_vmult, m->d_vmult, m->lvn);
        NLG_HIP(hipMemcpyAsync(m->d_binvm1, m->d_bm1, sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToDevice, s));
        double *f2[1] = {m->d_binvm1};
        NLG_TRY(sem_gs(m, f2, 1));
        NLG_LAUNCH(k_recip, dim3(grid_for(m->lvn)), dim3(NT), 0, s, m->d_binvm1, m->d_binvm1, m->lvn);
        for (int c = 0; c < dim; ++c)
            NLG_LAUNCH(k_mul, dim3(grid_for(m->lvn)), dim3(NT), 0, s, m->d_mbinv[c], m->d_mask[c], m->d_binvm1, m->lvn);
        NLG_HIP(hipGetLastError());
        if (dim == 3 && m->d_slot_xp) {
            NLG_TRY(dalloc(&m->d_vmult_xp, m->lvs, s));
            double *a[1] = {m->d_vmult}, *b[1] = {m->d_vmult_xp};
            NLG_TRY(sem_to_xp(m, a, b, 1));
        }
        if (dim == 3 && !m->h_slot.empty()) {
            std::vector<double> h((size_t)m->lvn), hp((size_t)m->lvs, 0.0);
            for (int c = 0; c < dim; ++c) {
                NLG_HIP(hipMemcpyAsync(h.data(), m->d_mbinv[c], sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToHost, s));
                NLG_HIP(hipStreamSynchronize(s));
                for (int64_t e = 0; e < m->E; ++e)
                    for (int p = 0; p < m->np1; ++p) hp[(size_t)e * m->np1 + m->h_slot[p]] = h[(size_t)e * m->np1 + p];
                NLG_HIP(hipMalloc(&m->d_mbinv_fg[c], sizeof(double) * (size_t)m->lvs));
                NLG_HIP(hipMemcpy(m->d_mbinv_fg[c], hp.data(), sizeof(double) * (size_t)m->lvs, hipMemcpyHostToDevice));
            }
            // the same weights as ONE array and one byte per point (k_opdiv3n): binvm1 and the bits of the three masks, face-grouped.
            // Only where every mask entry is exactly 0 or 1 (it is for every boundary condition the mesh builder knows).
            std::vector<double> hb((size_t)m->lvn), hm((size_t)m->lvn);
            std::vector<unsigned char> bits((size_t)m->lvs, 0);
            bool binary = true;
            for (int c = 0; c < dim; ++c) {
                NLG_HIP(hipMemcpy(hm.data(), m->d_mask[c], sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToHost));
                for (int64_t e = 0; e < m->E; ++e)
                    for (int p = 0; p < m->np1; ++p) {
                        const double v = hm[(size_t)e * m->np1 + p];
                        binary = binary && (v == 0.0 || v == 1.0);
                        if (v != 0.0) bits[(size_t)e * m->np1 + m->h_slot[p]] |= (unsigned char)(1u << c);
                    }
            }
            if (binary) {
                NLG_HIP(hipMemcpy(hb.data(), m->d_binvm1, sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToHost));
                std::fill(hp.begin(), hp.end(), 0.0);
                for (int64_t e = 0; e < m->E; ++e)
                    for (int p = 0; p < m->np1; ++p) hp[(size_t)e * m->np1 + m->h_slot[p]] = hb[(size_t)e * m->np1 + p];
                NLG_HIP(hipMalloc(&m->d_binv_fg, sizeof(double) * (size_t)m->lvs));
                NLG_HIP(hipMemcpy(m->d_binv_fg, hp.data(), sizeof(double) * (size_t)m->lvs, hipMemcpyHostToDevice));
                NLG_HIP(hipMalloc(&m->d_maskb_fg, (size_t)m->lvs));
                NLG_HIP(hipMemcpy(m->d_maskb_fg, bits.data(), (size_t)m->lvs, hipMemcpyHostToDevice));
            }
        }
    }

    // ---- pressure-mesh and fine-mesh metrics
    for (int q = 0; q < dim * dim; ++q) {
        NLG_TRY(dalloc(&m->d_rst2w[q], m->lps, s));
        NLG_TRY(sem_tensor(m, m->d_rst[q], m->d_rst2w[q], n, n2, m->d_I12, m->d_I12, m->d_I12, m->d_w2));
        NLG_HIP(hipMalloc(&m->d_rstdw[q], sizeof(double) * (size_t)m->lfn));
        NLG_TRY(sem_tensor(m, m->d_rst[q], m->d_rstdw[q], n, nd, m->d_Jd, m->d_Jd, m->d_Jd, m->d_wd));
    }
    NLG_TRY(dalloc(&m->d_bm2, m->lps, s));
    NLG_TRY(dalloc(&m->d_bm2inv, m->lps, s));
    NLG_TRY(sem_tensor(m, m->d_jac, m->d_bm2, n, n2, m->d_I12, m->d_I12, m->d_I12, m->d_w2));
    NLG_LAUNCH(k_recip, dim3(grid_for(m->lpn)), dim3(NT), 0, s, m->d_bm2inv, m->d_bm2, m->lpn);

    // ---- volumes (host sums; set-up only)
    {
        std::vector<double> h((size_t)std::max(m->lvn, m->lpn));
        NLG_HIP(hipMemcpyAsync(h.data(), m->d_bm1, sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToHost, s));
        NLG_HIP(hipStreamSynchronize(s));
        double v = 0.0;
        for (int64_t q = 0; q < m->lvn; ++q) v += h[q];
        m->volvm1 = v;
        NLG_HIP(hipMemcpyAsync(h.data(), m->d_bm2, sizeof(double) * (size_t)m->lpn, hipMemcpyDeviceToHost, s));
        NLG_HIP(hipStreamSynchronize(s));
        v = 0.0;
        for (int64_t q = 0; q < m->lpn; ++q) v += h[q];
        m->volvm2 = v;
        m->lpn_global = m->lpn;
        if (ctx->distributed()) {   // global volumes and pressure dof count
            double hv[3] = {m->volvm1, m->volvm2, (double)m->lpn}, *dv = ctx->d_scalars + 4010;
            NLG_HIP(hipMemcpyAsync(dv, hv, sizeof(hv), hipMemcpyHostToDevice, s));
            NLG_TRY(allreduce_sum(ctx, dv, 3));
            NLG_TRY(scalars_to_host(ctx, 4010, 3, hv));
            m->volvm1 = hv[0];
            m->volvm2 = hv[1];
            m->lpn_global = (int64_t)(hv[2] + 0.5);
        }
    }

    // ---- names for read-back
    auto reg = [&](const char *nm, const double *p, int64_t len) {
        m->named[nm] = p;
        m->named_len[nm] = len;
    };
    reg("bm1", m->d_bm1, m->lvn);
    reg("binvm1", m->d_binvm1, m->lvn);
    reg("vmult", m->d_vmult, m->lvn);
    reg("jac", m->d_jac, m->lvn);
    reg("bm2", m->d_bm2, m->lpn);
    {
        const char *gn3[6] = {"g11", "g12", "g13", "g22", "g23", "g33"};
        const char *gn2[3] = {"g11", "g12", "g22"};
        for (int q = 0; q < ng; ++q) reg(dim == 3 ? gn3[q] : gn2[q], m->d_G[q], m->lvn);
        char nm[16];
        for (int j = 0; j < dim; ++j)
            for (int i = 0; i < dim; ++i) {
                snprintf(nm, sizeof(nm), "rst%d%d", j + 1, i + 1);
                reg(nm, m->d_rst[j * dim + i], m->lvn);
                snprintf(nm, sizeof(nm), "rst2w%d%d", j + 1, i + 1);
                reg(nm, m->d_rst2w[j * dim + i], m->lpn);
                snprintf(nm, sizeof(nm), "rstdw%d%d", j + 1, i + 1);
                reg(nm, m->d_rstdw[j * dim + i], m->lfn);
            }
    }
    NLG_HIP(hipStreamSynchronize(s));
    NLG_TRY(pprec_setup(m, d));
    NLG_HIP(hipStreamSynchronize(s));
    *out = m;
    return 0;
}

int nlg_mesh_destroy(nlg_mesh *m) {
    if (!m) return 0;
    hipDeviceSynchronize();
    nlg_vec_pool_trim(nullptr);   // released vectors (nlg_vec_release) may still point at this mesh
    double *ptrs[] = {m->d_D, m->d_Dt, m->d_I12, m->d_I12t, m->d_D12, m->d_D12t, m->d_Jd, m->d_Jdt, m->d_DJd, m->d_DJdt,
                      m->d_rdr, m->d_w1, m->d_w2, m->d_wd, m->d_jac, m->d_bm1, m->d_binvm1, m->d_vmult, m->d_tmask,
                      m->d_bm2, m->d_bm2inv};
    for (double *p : ptrs)
        if (p) hipFree(p);
    for (int c = 0; c < 3; ++c) {
        if (m->d_x[c]) hipFree(m->d_x[c]);
        if (m->d_mask[c]) hipFree(m->d_mask[c]);
        if (m->d_mbinv[c]) hipFree(m->d_mbinv[c]);
    }
    for (int q = 0; q < 9; ++q) {
        if (m->d_rst[q]) hipFree(m->d_rst[q]);
        if (m->d_rst2w[q]) hipFree(m->d_rst2w[q]);
        if (m->d_rstdw[q]) hipFree(m->d_rstdw[q]);
    }
    for (int q = 0; q < 6; ++q)
        if (m->d_G[q]) hipFree(m->d_G[q]);
    if (m->d_lglel) hipFree(m->d_lglel);
    halo_free(m);
    pprec_free(m);
    if (m->d_binv_fg) hipFree(m->d_binv_fg);
    if (m->d_maskb_fg) hipFree(m->d_maskb_fg);
    if (m->d_slot_fg) hipFree(m->d_slot_fg);
    if (m->gs.d_offsets_fg) hipFree(m->gs.d_offsets_fg);
    if (m->gs.d_indices_fg) hipFree(m->gs.d_indices_fg);
    if (m->gs.d_offsets_xp) hipFree(m->gs.d_offsets_xp);
    if (m->gs.d_indices_xp) hipFree(m->gs.d_indices_xp);
    if (m->d_slot_xp) hipFree(m->d_slot_xp);
    if (m->d_vmult_xp) hipFree(m->d_vmult_xp);
    if (m->d_wlanes) hipFree(m->d_wlanes);
    for (int c = 0; c < 3; ++c)
        if (m->d_mbinv_fg[c]) hipFree(m->d_mbinv_fg[c]);
    if (m->gs.d_offsets) hipFree(m->gs.d_offsets);
    if (m->gs.d_indices) hipFree(m->gs.d_indices);
    for (double *p : m->scratch1) hipFree(p);
    for (double *p : m->scratch2) hipFree(p);
    for (double *p : m->scratchd) hipFree(p);
    delete m;
    return 0;
}

int nlg_mesh_sizes(const nlg_mesh *m, int64_t *lvn, int64_t *lpn, int *dim, int *n) {
    NLG_CHECK(m, "nlg_mesh_sizes: NULL mesh");
    if (lvn) *lvn = m->lvn;
    if (lpn) *lpn = m->lpn;
    if (dim) *dim = m->dim;
    if (n) *n = m->n;
    return 0;
}

int nlg_mesh_get(const nlg_mesh *m, const char *name, double *out, int64_t count) {
    NLG_CHECK(m && name && out, "nlg_mesh_get: NULL argument");
    // derived-on-demand arrays (verification only): "ediag" = diag(E), "hdiag:<h1>:<h2>" = assembled diag(H)
    if (strcmp(name, "ediag") == 0) {
        NLG_CHECK(count == m->lpn, "nlg_mesh_get: count mismatch for ediag");
        nlg_mesh *mm = const_cast<nlg_mesh *>(m);
        double *ed = sem_scratch2(mm, 5);
        NLG_TRY(sem_ediag(mm, ed));
        NLG_HIP(hipMemcpyAsync(out, ed, sizeof(double) * (size_t)m->lpn, hipMemcpyDeviceToHost, m->ctx->stream));
        NLG_HIP(hipStreamSynchronize(m->ctx->stream));
        return 0;
    }
    if (strncmp(name, "hdiag:", 6) == 0) {
        double h1 = 0, h2 = 0;
        NLG_CHECK(sscanf(name + 6, "%lf:%lf", &h1, &h2) == 2, "nlg_mesh_get: bad hdiag spec '%s'", name);
        NLG_CHECK(count == m->lvn, "nlg_mesh_get: count mismatch for hdiag");
        nlg_mesh *mm = const_cast<nlg_mesh *>(m);
        double *dg = sem_scratch1(mm, 3);
        NLG_TRY(sem_helm_diag(mm, dg, h1, h2));
        double *f[1] = {dg};
        NLG_TRY(sem_gs(mm, f, 1));
        NLG_HIP(hipMemcpyAsync(out, dg, sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToHost, m->ctx->stream));
        NLG_HIP(hipStreamSynchronize(m->ctx->stream));
        return 0;
    }
    auto it = m->named.find(name);
    NLG_CHECK(it != m->named.end(), "nlg_mesh_get: unknown array '%s'", name);
    const int64_t len = m->named_len.at(name);
    NLG_CHECK(count == len, "nlg_mesh_get: count %lld != length %lld of '%s'", (long long)count, (long long)len, name);
    NLG_HIP(hipMemcpyAsync(out, it->second, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, m->ctx->stream));
    NLG_HIP(hipStreamSynchronize(m->ctx->stream));
    return 0;
}

// ---- rand (needs the gather-scatter, hence lives here) ---------------------------------------------
int nlg_vec_rand(nlg_vec *self, int ifnorm, uint64_t seed) {
    NLG_TRY(nlg_vec_rand_noise(self, seed));
    return nlg_vec_rand_finish(self, ifnorm);
}

// first half of nek_drand (real_vectors.f90:62-98): the mth_rand noise added point by point to every active field
int nlg_vec_rand_noise(nlg_vec *self, uint64_t seed) {
    NLG_CHECK(self, "nlg_vec_rand: NULL vector");
    nlg_mesh *m = self->mesh;
    hipStream_t s = m->ctx->stream;
    CF3 X = {{m->d_x[0], m->d_x[1], m->d_x[2]}};
    for (int f = 0; f < self->ncomp; ++f) {
        double *fld = f < m->dim ? self->vel(f) : self->theta(f - m->dim);
        NLG_LAUNCH(k_rand_add, dim3(grid_for(m->lvn)), dim3(NT), 0, s, m->dim, m->n, m->E, m->d_lglel, X, fld, f, seed);
    }
    NLG_HIP(hipGetLastError());
    return 0;
}

// second half (real_vectors.f90:100-122): continuity, Dirichlet masks, optional normalisation, history cleared
int nlg_vec_rand_finish(nlg_vec *self, int ifnorm) {
    NLG_CHECK(self, "nlg_vec_rand: NULL vector");
    nlg_mesh *m = self->mesh;
    hipStream_t s = m->ctx->stream;
    // opdssum, opcolv(vmult), dsavg, bcdirvc   (real_vectors.f90:100-105)
    double *v[3] = {self->vel(0), self->vel(1), m->dim == 3 ? self->vel(2) : nullptr};
    F3 fv = {{v[0], v[1], v[2]}};
    CF3 vm = {{m->d_vmult, m->d_vmult, m->d_vmult}};
    CF3 mk = {{m->d_mask[0], m->d_mask[1], m->d_mask[2]}};
    for (int pass = 0; pass < 2; ++pass) {
        NLG_TRY(sem_gs(m, v, m->dim));
        if (m->dim == 3)
            NLG_LAUNCH(k_colmul<3>, dim3(grid_for(m->lvn)), dim3(NT), 0, s, fv, vm, m->lvn);
        else
            NLG_LAUNCH(k_colmul<2>, dim3(grid_for(m->lvn)), dim3(NT), 0, s, fv, vm, m->lvn);
    }
    if (m->dim == 3)
        NLG_LAUNCH(k_colmul<3>, dim3(grid_for(m->lvn)), dim3(NT), 0, s, fv, mk, m->lvn);
    else
        NLG_LAUNCH(k_colmul<2>, dim3(grid_for(m->lvn)), dim3(NT), 0, s, fv, mk, m->lvn);
    for (int sc = 0; sc < self->nscal; ++sc) {
        double *t[1] = {self->theta(sc)};
        F3 ft = {{t[0], nullptr, nullptr}};
        CF3 vm1 = {{m->d_vmult, nullptr, nullptr}};
        CF3 tm = {{m->d_tmask, nullptr, nullptr}};
        NLG_TRY(sem_gs(m, t, 1));
        NLG_LAUNCH(k_colmul<1>, dim3(grid_for(m->lvn)), dim3(NT), 0, s, ft, vm1, m->lvn);
        NLG_LAUNCH(k_colmul<1>, dim3(grid_for(m->lvn)), dim3(NT), 0, s, ft, tm, m->lvn);
    }
    NLG_HIP(hipGetLastError());
    if (ifnorm) {
        double nrm = 0.0;
        NLG_TRY(nlg_vec_norm(self, &nrm));
        NLG_CHECK(nrm > 0.0, "nlg_vec_rand: zero norm");
        NLG_TRY(nlg_vec_scal(self, 1.0 / nrm));
    }
    self->nrst = 0;
    return 0;
}

// ---- outpost_dnek (src/neklab_utils.f90:305-333): one vector -> one Nek5000 "#std" field file -----------------
// Layout as written by Nek5000's mfo_outfld for a Pn-Pn-2 run: 132-byte header, endian tag, element map, then the groups
// X (optional), U, P (pressure interpolated GL -> GLL inside every element and averaged across elements: `mappr`),
// T (first scalar), element by element, component by component; 3-D files end with float32 min / max records.
// Host-side I/O on one rank; the only device work is the pressure map.
int nlg_vec_outpost(const nlg_vec *v, const char *path, int with_coords, double time, int istep) {
    NLG_CHECK(v && path, "nlg_vec_outpost: NULL argument");
    nlg_mesh *m = v->mesh;
    NLG_CHECK(!m->ctx->distributed(), "nlg_vec_outpost: single-rank writer (every rank would write its own file)");
    hipStream_t st = m->ctx->stream;
    const int dim = m->dim, n = m->n, n2 = m->n2, np1 = m->np1;
    const int64_t E = m->E;
    // GL(n2) -> GLL(n) Lagrange interpolation, row-major n x n2
    std::vector<double> I21((size_t)n * n2, 1.0);
    const auto &z1 = m->ops.z1, &z2 = m->ops.z2;
    for (int a = 0; a < n; ++a)
        for (int k = 0; k < n2; ++k) {
            double p = 1.0;
            for (int l = 0; l < n2; ++l)
                if (l != k) p *= (z1[a] - z2[l]) / (z2[k] - z2[l]);
            I21[(size_t)a * n2 + k] = p;
        }
    double *dM = nullptr;
    NLG_HIP(hipMalloc(&dM, sizeof(double) * I21.size()));
    NLG_HIP(hipMemcpyAsync(dM, I21.data(), sizeof(double) * I21.size(), hipMemcpyHostToDevice, st));
    double *p1 = sem_scratch1(m, 0);
    NLG_CHECK(p1, "nlg_vec_outpost: scratch allocation failed");
    NLG_TRY(sem_tensor(m, v->pr(), p1, n2, n, dM, dM, dM, nullptr));
    double *f1[1] = {p1};
    NLG_TRY(sem_gs(m, f1, 1));
    {
        F3 fw = {{p1, nullptr, nullptr}};
        CF3 vm = {{m->d_vmult, nullptr, nullptr}};
        NLG_LAUNCH(k_colmul<1>, dim3(grid_for(m->lvn)), dim3(NT), 0, st, fw, vm, m->lvn);
    }
    std::vector<std::vector<double>> grp;   // each [E][nc][np1]
    std::vector<int> ncs;
    std::string code;
    auto fetch = [&](const double *const *src, int nc) -> int {
        std::vector<double> h((size_t)E * np1), out((size_t)E * nc * np1);
        for (int c = 0; c < nc; ++c) {
            NLG_HIP(hipMemcpyAsync(h.data(), src[c], sizeof(double) * (size_t)E * np1, hipMemcpyDeviceToHost, st));
            NLG_HIP(hipStreamSynchronize(st));
            for (int64_t e = 0; e < E; ++e) memcpy(&out[((size_t)e * nc + c) * np1], &h[(size_t)e * np1], sizeof(double) * np1);
        }
        grp.push_back(std::move(out));
        ncs.push_back(nc);
        return 0;
    };
    if (with_coords) {
        const double *x[3] = {m->d_x[0], m->d_x[1], m->d_x[2]};
        NLG_TRY(fetch(x, dim));
        code += "X";
    }
    {
        const double *u[3] = {v->vel(0), v->vel(1), dim == 3 ? v->vel(2) : nullptr};
        NLG_TRY(fetch(u, dim));
        code += "U";
        const double *pp[1] = {p1};
        NLG_TRY(fetch(pp, 1));
        code += "P";
        if (v->nscal > 0) {
            const double *tt[1] = {v->theta(0)};
            NLG_TRY(fetch(tt, 1));
            code += "T";
        }
    }
    hipFree(dM);
    // time in Fortran's e20.13 form 0.dddddddddddddE+xx
    char mant[64];
    {
        char buf[64];
        snprintf(buf, sizeof(buf), "%.12E", time);        // d.ddddddddddddE+xx
        std::string b(buf);
        const size_t ep = b.find('E');
        std::string digits;
        for (char ch : b.substr(0, ep))
            if (ch >= '0' && ch <= '9') digits += ch;
        int ex = atoi(b.c_str() + ep + 1);
        if (time != 0.0) ex += 1;
        else ex = 0;
        snprintf(mant, sizeof(mant), "%s0.%sE%+03d", time < 0 ? "-" : "", digits.c_str(), ex);
    }
    char hdr[256];
    snprintf(hdr, sizeof(hdr), "#std %1d %2d %2d %2d %10lld %10lld %20s %9d %6d %6d %-10s%15s %s", 8, n, n, dim == 3 ? n : 1, (long long)E,
             (long long)E, mant, istep, 0, 1, code.c_str(), "1.0000000E+00", "F");
    std::string h132(hdr);
    h132.resize(132, ' ');
    FILE *f = fopen(path, "wb");
    NLG_CHECK(f, "nlg_vec_outpost: cannot open %s", path);
    fwrite(h132.data(), 1, 132, f);
    const float tag = 6.54321f;
    fwrite(&tag, 4, 1, f);
    std::vector<int32_t> elmap((size_t)E);
    for (int64_t e = 0; e < E; ++e) elmap[e] = (int32_t)(m->h_lglel.empty() ? e + 1 : m->h_lglel[e] + 1);
    fwrite(elmap.data(), 4, (size_t)E, f);
    for (auto &g : grp) fwrite(g.data(), 8, g.size(), f);
    if (dim == 3)
        for (size_t q = 0; q < grp.size(); ++q) {
            const int nc = ncs[q];
            std::vector<float> mm((size_t)E * nc * 2);
            for (int64_t e = 0; e < E; ++e)
                for (int c = 0; c < nc; ++c) {
                    const double *a = &grp[q][((size_t)e * nc + c) * np1];
                    double lo = a[0], hi = a[0];
                    for (int i = 1; i < np1; ++i) {
                        lo = a[i] < lo ? a[i] : lo;
                        hi = a[i] > hi ? a[i] : hi;
                    }
                    mm[((size_t)e * nc + c) * 2] = (float)lo;
                    mm[((size_t)e * nc + c) * 2 + 1] = (float)hi;
                }
            fwrite(mm.data(), 4, mm.size(), f);
        }
    const bool bad = ferror(f) != 0;
    fclose(f);
    NLG_CHECK(!bad, "nlg_vec_outpost: write error on %s", path);
    return 0;
}

int64_t nlg_vec_size_value(const nlg_vec *self) { return self ? (int64_t)self->ncomp * self->mesh->lvn + self->mesh->lpn : -1; }
int nlg_vec_has_rst_value(const nlg_vec *self) { return (self && self->nrst > 0) ? 1 : 0; }

// ---- operator-level entry points ------------------------------------------------------------------
static int vel_ptrs(const nlg_vec *v, double **p) {
    for (int i = 0; i < 3; ++i) p[i] = i < v->mesh->dim ? v->vel(i) : nullptr;
    return 0;
}

int nlg_op_helmholtz(nlg_mesh *m, const nlg_vec *in, nlg_vec *out, double h1, double h2, int assemble) {
    NLG_CHECK(m && in && out && in->mesh == m && out->mesh == m, "nlg_op_helmholtz: bad arguments");
    NLG_CHECK(in != out, "nlg_op_helmholtz: in-place application is not supported");
    double *u[3], *w[3];
    vel_ptrs(in, u);
    vel_ptrs(out, w);
    NLG_TRY(sem_axhelm(m, u, w, m->dim, h1, h2));
    if (assemble) {
        NLG_TRY(sem_gs(m, w, m->dim));
        F3 fw = {{w[0], w[1], w[2]}};
        CF3 mk = {{m->d_mask[0], m->d_mask[1], m->d_mask[2]}};
        if (m->dim == 3)
            NLG_LAUNCH(k_colmul<3>, dim3(grid_for(m->lvn)), dim3(NT), 0, m->ctx->stream, fw, mk, m->lvn);
        else
            NLG_LAUNCH(k_colmul<2>, dim3(grid_for(m->lvn)), dim3(NT), 0, m->ctx->stream, fw, mk, m->lvn);
        NLG_HIP(hipGetLastError());
    }
    return 0;
}

int nlg_op_dssum(nlg_mesh *m, nlg_vec *v) {
    NLG_CHECK(m && v && v->mesh == m, "nlg_op_dssum: bad arguments");
    double *w[3];
    vel_ptrs(v, w);
    return sem_gs(m, w, m->dim);
}

int nlg_op_cdabdtp(nlg_mesh *m, const nlg_vec *in, nlg_vec *out) {
    NLG_CHECK(m && in && out && in->mesh == m && out->mesh == m, "nlg_op_cdabdtp: bad arguments");
    return sem_cdabdtp(m, in->pr(), out->pr());
}

int nlg_op_pprec(nlg_mesh *m, const nlg_vec *in, nlg_vec *out, int overlap, int with_coarse) {
    NLG_CHECK(m && in && out && in->mesh == m && out->mesh == m && in != out, "nlg_op_pprec: bad arguments");
    const double *xc = nullptr;
    // the coarse call also packs the overlap layers, so it always runs; its result is dropped when not wanted
    NLG_TRY(pprec_coarse(m, m->ctx->stream, nullptr, in->pr(), &xc, overlap != 0));
    NLG_TRY(pprec_fine(m, m->ctx->stream, nullptr, in->pr(), with_coarse ? xc : nullptr, out->pr(), nullptr, overlap != 0));
    return 0;
}

int nlg_op_opdiv(nlg_mesh *m, const nlg_vec *in, nlg_vec *out) {
    NLG_CHECK(m && in && out && in->mesh == m && out->mesh == m, "nlg_op_opdiv: bad arguments");
    double *u[3];
    vel_ptrs(in, u);
    return sem_opdiv(m, u, out->pr(), 1.0, nullptr);
}

int nlg_op_opgradt(nlg_mesh *m, const nlg_vec *in, nlg_vec *out) {
    NLG_CHECK(m && in && out && in->mesh == m && out->mesh == m, "nlg_op_opgradt: bad arguments");
    double *w[3];
    vel_ptrs(out, w);
    return sem_opgradt(m, in->pr(), w);
}

int nlg_op_cfl(nlg_mesh *m, const nlg_vec *base, double dt, double *cfl) {
    NLG_CHECK(m && base && cfl && base->mesh == m, "nlg_op_cfl: bad arguments");
    double *u[3];
    vel_ptrs(base, u);
    return sem_cfl(m, u, dt, cfl);
}

}  // extern "C"
