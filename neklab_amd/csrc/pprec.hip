// Two-level preconditioner of the consistent Poisson operator E = D (mask B^-1 QQ^T) D^T (gfx950).
//
// Role in the reference: `preconditioner = semg_xxt` of the pressure solve inside nek_advance
// (/root/reference/examples/cylinder/stability/direct/1cyl.par:21); Nek5000's implementation is not in the
// reference tree.  This is a restatement of the published idea (Fischer 1997; Lottes & Fischer 2005) in its
// simplest robust form, chosen from a numpy prototype (docs/prototypes/precond_proto.py: Jacobi 680 iterations,
// element-wise FDM 192, FDM + exact piecewise-constant coarse grid 81, FDM + the V-cycle below 110, all at
// E = 512, independent of E for the two-level variants):
//   M^-1 r = sum_e R_e^T Etilde_e^-1 R_e r                     (element-wise fast diagonalisation, additive)
//          + R_1   V(A_c) R_1^T r                              (trilinear coarse space on the element vertices)
//   Etilde_e : E restricted to element e with the element replaced by a box of its mean edge lengths and the
//              neighbours' mass lumped on shared faces: separable, inverted exactly by 1-D generalised
//              eigen-decompositions (n2 x n2 per direction).
//   R_1     : columns = the trilinear (bilinear in 2-D) hat functions of the element vertices evaluated at the GL
//             pressure points; continuous across elements although the pressure space is not.  Prototype
//             (docs/prototypes/precond_proto2.py, precond_proto3.py, explicit sparse E): against the piecewise-constant
//             space used first (R_0) the vertex space halves the iteration count (73 -> 37 at lx1 = 8 with exact
//             element blocks; 30 at lx1 = 6), mesh-independent, and one V-cycle is as good as the exact solve.
//   A_c = R_1^T E R_1 : Galerkin, sparse nvert x nvert (125-point on structured meshes).  Assembled by probing E on
//             the device: elements are coloured so that no two of one colour share a neighbour, one E application
//             per (colour, corner) gives the 2^dim x 2^dim blocks phi_c'^T E_{e',e} phi_c of every neighbouring pair.
//   V(A_c)  : additive two-grid approximation of A_c^-1: omega diag(A_c)^-1 + P (P^T A_c P)^-1 P^T with P the
//             piecewise-constant prolongation from greedy aggregates of vertices (dense inverse on the aggregates).
//             In the prototype it needs as few PCG iterations as the exact coarse solve or a multiplicative V-cycle
//             (29-30), and it needs no product with A_c at run time.  When nvert is small the aggregates are the
//             vertices (exact solve, no Jacobi term).
//   With several ranks the coarse level is rank-local (E without the halo exchange): block-diagonal, still SPD.
// M is a fixed symmetric positive (semi-)definite operator, so plain PCG stays valid.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <unordered_map>

#include <rocsolver/rocsolver.h>

#include "internal.h"

using namespace nlg;

namespace {

constexpr int NT = 256;

struct Hat {
    double h1[12];   // (1 + z)/2 at the GL points; the lower-corner hat is 1 - h1
};

template <int S0, int S1, int S2, int AX, int NOUT, bool TRANS>
__device__ __forceinline__ void contract(const double *__restrict__ in, double *__restrict__ out,
                                         const double *__restrict__ M, int tid, int nth) {
    // out[.., o, ..] = sum_l Mop[o][l] in[.., l, ..] ; Mop = M (NOUT x NIN row-major) or its transpose
    constexpr int NIN = AX == 0 ? S0 : (AX == 1 ? S1 : S2);
    constexpr int O0 = AX == 0 ? NOUT : S0, O1 = AX == 1 ? NOUT : S1, O2 = AX == 2 ? NOUT : S2;
    for (int p = tid; p < O0 * O1 * O2; p += nth) {
        const int a = p % O0, b = (p / O0) % O1, c = p / (O0 * O1);
        const int o = AX == 0 ? a : (AX == 1 ? b : c);
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < NIN; ++l) {
            const int q = AX == 0 ? (l + S0 * (b + S1 * c)) : (AX == 1 ? (a + S0 * (l + S1 * c)) : (a + S0 * (b + S1 * l)));
            s += (TRANS ? M[l * NOUT + o] : M[o * NIN + l]) * in[q];
        }
        out[p] = s;
    }
}

// The same transform in place: a lane reads its whole column into registers before it writes the outputs, and the
// columns of one stage are disjoint, so one LDS array serves as input and output (half the LDS per element -> twice
// the resident elements per CU for a kernel that mostly waits).
template <int N2, bool FWD, int AX, int ST = 64>
__device__ __forceinline__ void fdm_stage_inplace3(double *buf, const double *S, int lane) {
    constexpr int NCOL = N2 * N2;
    for (int col = lane; col < NCOL; col += ST) {
        int base, stride;
        if (AX == 0) {
            base = col * N2;
            stride = 1;
        } else if (AX == 1) {
            base = (col % N2) + N2 * N2 * (col / N2);
            stride = N2;
        } else {
            base = col;
            stride = N2 * N2;
        }
        double v[N2];
#pragma unroll
        for (int l = 0; l < N2; ++l) v[l] = buf[base + stride * l];
        double o_[N2];
#pragma unroll
        for (int o = 0; o < N2; ++o) {
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < N2; ++l) a += (FWD ? S[l * N2 + o] : S[o * N2 + l]) * v[l];
            o_[o] = a;
        }
#pragma unroll
        for (int o = 0; o < N2; ++o) buf[base + stride * o] = o_[o];
    }
}

// z_e = (Sz x Sy x Sx) [ invden o ((Sz x Sy x Sx)^T r_e) ] + xc[e]      (S stored row-major [point][mode])
// One wave per element, four elements per block: the six 1-D transforms of an element are tiny (N2^3 points), so
// the kernel is bound by launch/barrier latency, not by bytes; each lane owns whole columns and keeps them in
// registers between the load and the N2 outputs.
template <int N2, int DIM, bool FWD, int AX>
__device__ __forceinline__ void fdm_stage(const double *__restrict__ in, double *__restrict__ out,
                                          const double *__restrict__ S, int lane) {
    constexpr int NZ = DIM == 3 ? N2 : 1;
    constexpr int NCOL = (N2 * N2 * NZ) / N2;
    for (int col = lane; col < NCOL; col += 64) {
        // column base index and stride along axis AX of an [NZ][N2][N2] array
        int base, stride;
        if (AX == 0) {
            base = col * N2;
            stride = 1;
        } else if (AX == 1) {
            base = (col % N2) + N2 * N2 * (col / N2);
            stride = N2;
        } else {
            base = col;
            stride = N2 * N2;
        }
        double v[N2];
#pragma unroll
        for (int l = 0; l < N2; ++l) v[l] = in[base + stride * l];
#pragma unroll
        for (int o = 0; o < N2; ++o) {
            double a = 0.0;
#pragma unroll
            for (int l = 0; l < N2; ++l) a += (FWD ? S[l * N2 + o] : S[o * N2 + l]) * v[l];
            out[base + stride * o] = a;
        }
    }
}

template <int N2, int DIM>
__global__ __launch_bounds__(NT) void k_fdm(const double *__restrict__ flag, int64_t E, const double *__restrict__ S,
                                            const double *__restrict__ invden, const double *__restrict__ r,
                                            const double *__restrict__ xc, const double *__restrict__ xa,
                                            const int *__restrict__ agg, const int *__restrict__ vg, Hat hat,
                                            double *__restrict__ z, double *__restrict__ part) {
    constexpr int NP = DIM == 3 ? N2 * N2 * N2 : N2 * N2;
    __shared__ double sS[4][3][N2 * N2];
    __shared__ double sA[4][NP], sB[4][NP];
    if (flag && flag[0] != 0.0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 4 + wv;
    const bool act = e < E;
    const int64_t ee = act ? e : 0;
    for (int q = lane; q < DIM * N2 * N2; q += 64) sS[wv][q / (N2 * N2)][q % (N2 * N2)] = S[ee * (3 * N2 * N2) + q];
    for (int q = lane; q < NP; q += 64) sA[wv][q] = r[ee * NP + q];
    __syncthreads();
    fdm_stage<N2, DIM, true, 0>(sA[wv], sB[wv], sS[wv][0], lane);
    __syncthreads();
    fdm_stage<N2, DIM, true, 1>(sB[wv], sA[wv], sS[wv][1], lane);
    __syncthreads();
    if constexpr (DIM == 3) {
        fdm_stage<N2, DIM, true, 2>(sA[wv], sB[wv], sS[wv][2], lane);
        __syncthreads();
        for (int q = lane; q < NP; q += 64) sB[wv][q] *= invden[ee * NP + q];
        __syncthreads();
        fdm_stage<N2, DIM, false, 2>(sB[wv], sA[wv], sS[wv][2], lane);
        __syncthreads();
    } else {
        for (int q = lane; q < NP; q += 64) sA[wv][q] *= invden[ee * NP + q];
        __syncthreads();
    }
    fdm_stage<N2, DIM, false, 1>(sA[wv], sB[wv], sS[wv][1], lane);
    __syncthreads();
    fdm_stage<N2, DIM, false, 0>(sB[wv], sA[wv], sS[wv][0], lane);
    __syncthreads();
    double srz = 0.0, sz = 0.0;
    if (act) {
        // coarse correction prolonged on the fly: trilinear interpolation of the 2^DIM vertex values of the element
        constexpr int NC = 1 << DIM;
        double cv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int v = xc ? vg[e * NC + c] : 0;
            cv[c] = xc ? xc[v] + xa[agg[v]] : 0.0;   // Jacobi term + aggregate-level correction
        }
        for (int q = lane; q < NP; q += 64) {
            const double ha = hat.h1[q % N2], hb = hat.h1[(q / N2) % N2];
            double c0 = (cv[0] + ha * (cv[1] - cv[0])), c1 = (cv[2] + ha * (cv[3] - cv[2]));
            double cc = c0 + hb * (c1 - c0);
            if constexpr (DIM == 3) {
                const double hc = hat.h1[q / (N2 * N2)];
                const double d0 = (cv[4] + ha * (cv[5] - cv[4])), d1 = (cv[6] + ha * (cv[7] - cv[6]));
                cc += hc * ((d0 + hb * (d1 - d0)) - cc);
            }
            const double zv = sA[wv][q] + cc;
            z[e * NP + q] = zv;
            if (part) {
                srz += r[e * NP + q] * zv;
                sz += zv;
            }
        }
    }
    if (part) {   // first-stage sums of the surrounding PCG: part[blk] = sum r.z, part[nblk + blk] = sum z
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            srz += __shfl_down(srz, o, 64);
            sz += __shfl_down(sz, o, 64);
        }
        __syncthreads();
        if (lane == 0) {
            sB[0][wv] = srz;
            sB[0][4 + wv] = sz;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            part[blockIdx.x] = sB[0][0] + sB[0][1] + sB[0][2] + sB[0][3];
            part[gridDim.x + blockIdx.x] = sB[0][4] + sB[0][5] + sB[0][6] + sB[0][7];
        }
    }
}

// aggregate restriction of the coarse chain as a device function: it rides in the launch of k_fdm_ext (merged launches, below)
struct AggArgs {
    int na;
    const int *ap, *am;
    const double *rr;
    double *ra;
    int64_t lv, la;
};
// (as a part of a merged launch: `bx` = block index inside this part, any multiple of 64 threads per block)
__device__ __forceinline__ void agg_restrict_body(int bx, const double *flag, int64_t ld, const AggArgs &g) {
    if (flag) flag += (int64_t)blockIdx.y * ld;
    if (flag && flag[0] != 0.0) return;
    const double *rr = g.rr + (int64_t)blockIdx.y * g.lv;
    double *ra = g.ra + (int64_t)blockIdx.y * g.la;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int a = bx * wpb + wid;
    if (a >= g.na) return;
    double s = 0.0;
    for (int q = g.ap[a] + lane; q < g.ap[a + 1]; q += 64) s += rr[g.am[q]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) ra[a] = s;
}

// ---- overlapping variant (3-D) ---------------------------------------------------------------------------
// Extended local problems: element e plus the layer of GL points of each face neighbour that is adjacent to the
// shared face -> an N^3 grid (N = N2 + 2, the size of the velocity mesh), solved by fast diagonalisation with 1-D
// operators built from the line of up to three elements (pprec_setup).  The ghost layers travel through the
// velocity-mesh gather-scatter, as Nek5000's Schwarz smoother does: a velocity-shaped work array W (face-grouped
// layout) carries, at the interior points of an element face, the values of the adjacent pressure layer; after
// QQ^T each face holds own + neighbour's, whatever the relative orientation of the two elements.
//   k_q1_restrict_local (pack)  W_face = r(adjacent layer)
//   gs (pairs only)             W_face = own + neighbour
//   k_fdm_ext                   ext = [r | W_face - own];  solve;  z = z_int - (own ghost values, folded back);
//                               W_face = own ghost values
//   gs (pairs only)             W_face = own ghosts + neighbour's ghosts (= neighbour's solve at MY adjacent layer)
//   k_sch_finish                z += W_face (+ coarse correction), r.z sums
// i.e. z = sum_e R_e^T Atilde_e^-1 R_e r with overlapping index sets R_e: symmetric, additive.
__device__ __forceinline__ int ext_slot(int N, int a, int b, int c) { return fg_slot(N, a, b, c); }

// WPB waves (= elements) per block.  WPB = 1 makes every __syncthreads a single-wave barrier: the stages of one element
// never wait for another element's.
// WPE > 1 (with WPB = 1): WPE waves share ONE element -- for lx1 > 8 a single wave has to sweep N N = 100 (144) columns
// with 64 lanes in two (three) rounds, 56 % (75 %) of them busy; two (three) waves take one column per lane.
template <int N, int WPB, int WPE = 1>
__global__ __launch_bounds__(64 * WPB * WPE) void k_fdm_ext(const double *__restrict__ flag, int64_t E, const double *__restrict__ S,
                                                const double *__restrict__ lam, double thr, const double *__restrict__ r,
                                                const double *__restrict__ wq, double *__restrict__ W,
                                                double *__restrict__ z, const int *__restrict__ tab, int64_t ld, int64_t lW, int nb_fdm, AggArgs ag) {
    constexpr int N2 = N - 2, NP = N * N * N, NP2 = N2 * N2 * N2;
    __shared__ double sL[WPB][3][N];
    __shared__ double sA[WPB][NP];   // the six transforms run in place (fdm_stage_inplace3)
    if ((int)blockIdx.x >= nb_fdm) {   // merged launch: the blocks behind the elements restrict the coarse residual to the aggregates
        agg_restrict_body((int)blockIdx.x - nb_fdm, flag, ld, ag);
        return;
    }
    {   // blockIdx.y = lane of a block step
        const int64_t lo = (int64_t)blockIdx.y * ld;
        if (flag) flag += lo;
        r += lo, z += lo, W += (int64_t)blockIdx.y * lW;
    }
    if (flag && flag[0] != 0.0) return;
    static_assert(WPB == 1 || WPE == 1, "either several elements per block or several waves per element");
    constexpr int ST = 64 * WPE;                                   // threads that share one element
    const int lane = WPE > 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63), wv = WPE > 1 ? 0 : (int)(threadIdx.x >> 6);
    const int64_t e = (int64_t)blockIdx.x * WPB + wv;
    const bool act = e < E;
    const int64_t ee = act ? e : 0;
    // the element index is wave-uniform: with the pointer made provably uniform the 1-D eigenvector matrices are read by
    // scalar loads straight into FMA operands instead of 384 broadcast LDS reads per lane
    const int eu = __builtin_amdgcn_readfirstlane((int)ee);
    const double *__restrict__ Sgg = S + (int64_t)eu * (3 * N * N);
    // lx1 <= 8: the three N x N matrices of the element go to LDS (three coalesced loads per lane) and are read from there at
    // wave-uniform addresses (broadcast).  As scalar-load operands they need 384 SGPRs: 100 of them were spilled to VGPR lanes
    // and a quarter of the kernel's instructions were v_readlane / v_writelane moves on the (saturated) vector pipe.
    constexpr bool SLDS = (N <= 8 && WPE == 1);
    __shared__ double sS[SLDS ? WPB : 1][SLDS ? 3 * N * N : 1];
    if (SLDS)
        for (int q = lane; q < 3 * N * N; q += ST) sS[wv][q] = Sgg[q];
    const double *Sg = SLDS ? sS[wv] : Sgg;
    for (int q = lane; q < 3 * N; q += ST) sL[wv][q / N][q % N] = lam[ee * (3 * N) + q];
    const double *re = r + ee * NP2;
    double *We = W + ee * NP;
    // packed per-point constants (pprec_setup): bits 0-1 boundary directions, 2-12 exchange slot, 13-22 pressure point,
    // 23-28 ghost-neighbour flags; kept in registers for the store phase
    constexpr int NQL = (NP + ST - 1) / ST;
    int te[NQL];
#pragma unroll
    for (int u = 0; u < NQL; ++u) {
        const int q = lane + ST * u;
        te[u] = q < NP ? tab[q] : 3;
    }
    // branch-free: every lane issues all its loads (clamped to valid addresses) before the first use, so a wave pays ONE
    // global-memory latency here; with the loads inside `if (nb <= 1)` / `nb == 0 ? :` the compiler emitted, per point, branch ->
    // two loads -> wait -> branch -> load -> wait: sixteen serialised round trips per wave
    {
        double own[NQL], gw[NQL];
#pragma unroll
        for (int u = 0; u < NQL; ++u) {
            const int nb = te[u] & 3;
            const int q2 = nb <= 1 ? ((te[u] >> 13) & 1023) : 0;
            const int sl = nb == 1 ? ((te[u] >> 2) & 2047) : 0;
            own[u] = re[q2] * wq[ee * NP2 + q2];
            gw[u] = We[sl];
        }
#pragma unroll
        for (int u = 0; u < NQL; ++u) {
            const int q = lane + ST * u;
            if (q >= NP) break;
            const int nb = te[u] & 3;
            const double v = nb == 0 ? own[u] : (nb == 1 ? gw[u] - own[u] : 0.0);
            sA[wv][q] = v;
        }
    }
    __syncthreads();
    fdm_stage_inplace3<N, true, 0, ST>(sA[wv], Sg + 0 * N * N, lane);
    __syncthreads();
    fdm_stage_inplace3<N, true, 1, ST>(sA[wv], Sg + 1 * N * N, lane);
    __syncthreads();
    fdm_stage_inplace3<N, true, 2, ST>(sA[wv], Sg + 2 * N * N, lane);
    __syncthreads();
    for (int q = lane; q < NP; q += ST) {
        const double den = sL[wv][0][q % N] + sL[wv][1][(q / N) % N] + sL[wv][2][q / (N * N)];
        sA[wv][q] = den > thr ? sA[wv][q] / den : 0.0;
    }
    __syncthreads();
    fdm_stage_inplace3<N, false, 2, ST>(sA[wv], Sg + 2 * N * N, lane);
    __syncthreads();
    fdm_stage_inplace3<N, false, 1, ST>(sA[wv], Sg + 1 * N * N, lane);
    __syncthreads();
    fdm_stage_inplace3<N, false, 0, ST>(sA[wv], Sg + 0 * N * N, lane);
    __syncthreads();
    if (act) {
#pragma unroll
        for (int u = 0; u < NQL; ++u) {
            const int q = lane + ST * u;
            if (q >= NP) break;
            const int nb = te[u] & 3;
            if constexpr (N <= 10) {
                // (LDS reads unconditional with clamped indices, selected afterwards: no divergent branches around them)
                const int fl = nb == 0 ? (te[u] >> 23) : 0;
                const double c0 = sA[wv][q];
                const double m1 = sA[wv][q >= 1 ? q - 1 : q], p1 = sA[wv][q + 1 < NP ? q + 1 : q];
                const double mN = sA[wv][q >= N ? q - N : q], pN = sA[wv][q + N < NP ? q + N : q];
                const double mM = sA[wv][q >= N * N ? q - N * N : q], pM = sA[wv][q + N * N < NP ? q + N * N : q];
                double v = c0;
                v -= (fl & 1) ? m1 : 0.0;
                v -= (fl & 2) ? p1 : 0.0;
                v -= (fl & 4) ? mN : 0.0;
                v -= (fl & 8) ? pN : 0.0;
                v -= (fl & 16) ? mM : 0.0;
                v -= (fl & 32) ? pM : 0.0;
                if (nb == 1)
                    We[(te[u] >> 2) & 2047] = c0;   // ghost value: belongs to the neighbour's adjacent layer
                else if (nb == 0)
                    z[e * NP2 + ((te[u] >> 13) & 1023)] = v;
            } else {   // lx1 = 12 (three waves per element): the branchy form is faster there (213 vs 240 us)
                if (nb == 1) {
                    We[(te[u] >> 2) & 2047] = sA[wv][q];
                } else if (nb == 0) {
                    const int fl = te[u] >> 23;
                    double v = sA[wv][q];
                    if (fl & 1) v -= sA[wv][q - 1];
                    if (fl & 2) v -= sA[wv][q + 1];
                    if (fl & 4) v -= sA[wv][q - N];
                    if (fl & 8) v -= sA[wv][q + N];
                    if (fl & 16) v -= sA[wv][q - N * N];
                    if (fl & 32) v -= sA[wv][q + N * N];
                    z[e * NP2 + ((te[u] >> 13) & 1023)] = v;
                }
            }
        }
    }
}

// ---- the same extended local solve with the six 1-D transforms on the matrix pipe (lx1 = 8) --------------------------------
// One wave per element.  A transform along one axis is the GEMM  out(8 x 64 columns) = S(8 x 8) in(8 x 64 columns), issued as
// v_mfma_f64_16x16x4_f64 tiles: M = output index (8 of the 16 rows used), N = 16 data columns, K = 8 in two steps -> 8 MFMA per
// stage and 48 per element instead of 384 vector FMAs per lane, with 8 LDS reads and 8 LDS writes per lane and stage as the only
// other work.  Operand layout (guide, "f64 MFMA"): A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15],
// D[row = (lane >> 4) + 4 reg][col = lane & 15]: registers 0 and 1 of a lane are outputs o = lane >> 4 and o + 4 of its column,
// registers 2 and 3 belong to the unused rows 8 .. 15.  The cube is stored with the x rows padded to 9 doubles (index
// a + 9 (b + 8 c)) so that the 16 columns of a tile fall into different LDS banks for all three axes.
typedef double v4f64_t __attribute__((ext_vector_type(4)));
template <int AX, bool FWD>
__device__ __forceinline__ void fdm_stage_mfma8(double *__restrict__ buf, const double *__restrict__ sS, int l15, int lg) {
    // A operand of this stage from the element's matrices in LDS: forward = S^T (out[o] = sum_l S[l][o] in[l]), backward = S;
    // rows 8 .. 15 of the tile are zero.  (Held in registers for all six stages they cost 24 VGPRs and a wave per SIMD.)
    const int o = l15 & 7;
    const double live = l15 < 8 ? 1.0 : 0.0;
    const double a0 = live * (FWD ? sS[AX * 64 + lg * 8 + o] : sS[AX * 64 + o * 8 + lg]);
    const double a1 = live * (FWD ? sS[AX * 64 + (lg + 4) * 8 + o] : sS[AX * 64 + o * 8 + lg + 4]);
    constexpr int str = AX == 0 ? 1 : (AX == 1 ? 9 : 72);
    const v4f64_t zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // two tiles (32 columns) at a time: half the operand / result registers
        int base[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = 16 * (2 * h + t) + l15;
            base[t] = AX == 0 ? 9 * col : (AX == 1 ? (col & 7) + 72 * (col >> 3) : (col & 7) + 9 * (col >> 3));
        }
        double b0[2], b1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            b0[t] = buf[base[t] + str * lg];
            b1[t] = buf[base[t] + str * (lg + 4)];
        }
        v4f64_t d[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            d[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0[t], zero, 0, 0, 0);
            d[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[t], d[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            buf[base[t] + str * lg] = d[t][0];
            buf[base[t] + str * (lg + 4)] = d[t][1];
        }
    }
}

__global__ __launch_bounds__(64, 4) void k_fdm_ext_mfma8(const double *__restrict__ flag, int64_t E, const double *__restrict__ S,
                                                      const double *__restrict__ lam, double thr, const double *__restrict__ r,
                                                      const double *__restrict__ wq, double *__restrict__ W,
                                                      double *__restrict__ z, const int *__restrict__ tab, int64_t ld, int64_t lW, int nb_fdm,
                                                      AggArgs ag) {
    constexpr int N = 8, N2 = 6, NP = 512, NP2 = 216, NPAD = 9 * 64;
    __shared__ double sL[3][N];
    __shared__ double sA[NPAD];
    if ((int)blockIdx.x >= nb_fdm) {   // merged launch: the blocks behind the elements restrict the coarse residual to the aggregates
        agg_restrict_body((int)blockIdx.x - nb_fdm, flag, ld, ag);
        return;
    }
    {   // blockIdx.y = lane of a block step
        const int64_t lo = (int64_t)blockIdx.y * ld;
        if (flag) flag += lo;
        r += lo, z += lo, W += (int64_t)blockIdx.y * lW;
    }
    if (flag && flag[0] != 0.0) return;
    const int lane = threadIdx.x, l15 = lane & 15, lg = lane >> 4;
    const int64_t e = blockIdx.x;
    const double *__restrict__ Sg = S + e * (3 * N * N);
    __shared__ double sS[3 * N * N];
#pragma unroll
    for (int m = 0; m < 3; ++m) sS[m * N * N + lane] = Sg[m * N * N + lane];
    if (lane < 3 * N) sL[lane / N][lane % N] = lam[e * (3 * N) + lane];
    const double *re = r + e * NP2;
    double *We = W + e * NP;
    int te[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) te[u] = tab[lane + 64 * u];
    {
        double own[8], gw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int nb = te[u] & 3;
            const int q2 = nb <= 1 ? ((te[u] >> 13) & 1023) : 0;
            const int sl = nb == 1 ? ((te[u] >> 2) & 2047) : 0;
            own[u] = re[q2] * wq[e * NP2 + q2];
            gw[u] = We[sl];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = lane + 64 * u;
            const int nb = te[u] & 3;
            sA[q + (q >> 3)] = nb == 0 ? own[u] : (nb == 1 ? gw[u] - own[u] : 0.0);
        }
    }
    __syncthreads();
    fdm_stage_mfma8<0, true>(sA, sS, l15, lg);
    __syncthreads();
    fdm_stage_mfma8<1, true>(sA, sS, l15, lg);
    __syncthreads();
    fdm_stage_mfma8<2, true>(sA, sS, l15, lg);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = lane + 64 * u;
        const double den = sL[0][q % N] + sL[1][(q / N) % N] + sL[2][q / (N * N)];
        const double v = sA[q + (q >> 3)];
        sA[q + (q >> 3)] = den > thr ? v / den : 0.0;
    }
    __syncthreads();
    fdm_stage_mfma8<2, false>(sA, sS, l15, lg);
    __syncthreads();
    fdm_stage_mfma8<1, false>(sA, sS, l15, lg);
    __syncthreads();
    fdm_stage_mfma8<0, false>(sA, sS, l15, lg);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = lane + 64 * u;
        const int nb = te[u] & 3;
        const int fl = nb == 0 ? (te[u] >> 23) : 0;
        const int pq = q + (q >> 3);
        // (LDS reads unconditional with clamped indices, selected afterwards: no divergent branches around them)
        const double c0 = sA[pq];
        const double m1 = sA[pq >= 1 ? pq - 1 : pq], p1 = sA[pq + 1 < NPAD ? pq + 1 : pq];
        const double mN = sA[pq >= 9 ? pq - 9 : pq], pN = sA[pq + 9 < NPAD ? pq + 9 : pq];
        const double mM = sA[pq >= 72 ? pq - 72 : pq], pM = sA[pq + 72 < NPAD ? pq + 72 : pq];
        double v = c0;
        v -= (fl & 1) ? m1 : 0.0;
        v -= (fl & 2) ? p1 : 0.0;
        v -= (fl & 4) ? mN : 0.0;
        v -= (fl & 8) ? pN : 0.0;
        v -= (fl & 16) ? mM : 0.0;
        v -= (fl & 32) ? pM : 0.0;
        if (nb == 1)
            We[(te[u] >> 2) & 2047] = c0;   // ghost value: belongs to the neighbour's adjacent layer
        else if (nb == 0)
            z[e * NP2 + ((te[u] >> 13) & 1023)] = v;
    }
}

// z += (own + neighbours' ghost values at this point, from W after QQ^T) + prolonged coarse correction; r.z and z sums
// relative weight of the coarse-level correction in the additive sum (1 = plain additive; NLG_COARSE_SCALE for experiments: PCG does not
// care about the overall scale of a preconditioner, but it does about the balance of its two terms)
__device__ __constant__ double c_coarse_scale = 1.0;

template <int N>
__global__ __launch_bounds__(NT) void k_sch_finish(const double *__restrict__ flag, int64_t E, const double *__restrict__ W,
                                                   const double *__restrict__ r, const double *__restrict__ wq,
                                                   const double *__restrict__ xc, const double *__restrict__ xa,
                                                   const int *__restrict__ agg, const int *__restrict__ vg, Hat hat,
                                                   double *__restrict__ z, double *__restrict__ part, const int *__restrict__ wslot,
                                                   int64_t ld, int64_t lW, int64_t lv, int64_t la) {
    constexpr int N2 = N - 2, NP = N * N * N, NP2 = N2 * N2 * N2;
    __shared__ double sred[8];
    {   // blockIdx.y = lane of a block step
        const int64_t lo = (int64_t)blockIdx.y * ld;
        if (flag) flag += lo;
        r += lo, z += lo, W += (int64_t)blockIdx.y * lW;
        if (part) part += lo;
        if (xc) xc += (int64_t)blockIdx.y * lv, xa += (int64_t)blockIdx.y * la;
    }
    if (flag && flag[0] != 0.0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 4 + wv;
    const bool act = e < E;
    double srz = 0.0, sz = 0.0;
    if (act) {
        // all loads of a group of points first, unconditional and with clamped indices (the face slots of W come from the
        // table of k_q1_restrict_local3s; a conditional load inside the point loop costs a branch and a full wait per point),
        // then the arithmetic with selects
        constexpr int NIT = 4;   // points per lane in flight (lx1 = 8: the whole element)
        const double *We = W + e * NP;
        double cv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) cv[c] = 0.0;
        for (int q0 = 0; q0 < NP2; q0 += 64 * NIT) {
            double zv[NIT], qv[NIT], rv[NIT], gh[NIT][3];
            int sl[NIT][3];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = q0 + lane + 64 * it;
                const int qc = q < NP2 ? q : 0;
                const int64_t i = e * NP2 + qc;
                zv[it] = z[i];
                qv[it] = wq[i];
                rv[it] = r[i];
#pragma unroll
                for (int d = 0; d < 3; ++d) sl[it][d] = wslot[3 * qc + d];
            }
            if (q0 == 0 && xc) {
                int vv[8], av[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) vv[c] = vg[e * 8 + c];
#pragma unroll
                for (int c = 0; c < 8; ++c) av[c] = agg[vv[c]];
#pragma unroll
                for (int c = 0; c < 8; ++c) cv[c] = c_coarse_scale * (xc[vv[c]] + xa[av[c]]);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int d = 0; d < 3; ++d) gh[it][d] = We[sl[it][d] >= 0 ? sl[it][d] : 0];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = q0 + lane + 64 * it;
                if (q >= NP2) continue;
                const int a = q % N2, b = (q / N2) % N2, c = q / (N2 * N2);
                double v = zv[it];
                v += sl[it][0] >= 0 ? gh[it][0] : 0.0;
                v += sl[it][1] >= 0 ? gh[it][1] : 0.0;
                v += sl[it][2] >= 0 ? gh[it][2] : 0.0;
                v *= qv[it];
                const double ha = hat.h1[a], hb = hat.h1[b], hc = hat.h1[c];
                const double c0 = cv[0] + ha * (cv[1] - cv[0]), c1 = cv[2] + ha * (cv[3] - cv[2]);
                const double d0 = cv[4] + ha * (cv[5] - cv[4]), d1 = cv[6] + ha * (cv[7] - cv[6]);
                double cc = c0 + hb * (c1 - c0);
                cc += hc * ((d0 + hb * (d1 - d0)) - cc);
                v += cc;
                z[e * NP2 + q] = v;
                srz += rv[it] * v;
                sz += v;
            }
        }
    }
    if (part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            srz += __shfl_down(srz, o, 64);
            sz += __shfl_down(sz, o, 64);
        }
        if (lane == 0) {
            sred[wv] = srz;
            sred[4 + wv] = sz;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            part[blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
            part[gridDim.x + blockIdx.x] = sred[4] + sred[5] + sred[6] + sred[7];
        }
    }
}

// ---- 2-D twins: the exchange array W is a velocity-shaped field in the natural layout (N x N per element), the ghost
// layers are the interior points of the element edges, one lane per extended point.
template <int N, bool FWD, int AX>
__device__ __forceinline__ void fdm_stage2(const double *__restrict__ in, double *__restrict__ out,
                                           const double *__restrict__ S, int p) {
    // out[o, b] = sum_l Sop[o][l] in[l, b] along axis AX; Sop = S^T (FWD) or S;  S row-major [point][mode]
    const int a = p % N, b = p / N;
    const int o = AX == 0 ? a : b;
    double acc = 0.0;
#pragma unroll
    for (int l = 0; l < N; ++l) {
        const double sv = FWD ? S[l * N + o] : S[o * N + l];
        acc += sv * (AX == 0 ? in[l + N * b] : in[a + N * l]);
    }
    out[p] = acc;
}

template <int N>
__global__ __launch_bounds__(NT) void k_fdm_ext2(const double *__restrict__ flag, int64_t E, const double *__restrict__ S,
                                                 const double *__restrict__ lam, double thr, const double *__restrict__ r,
                                                 const double *__restrict__ wq, double *__restrict__ W,
                                                 double *__restrict__ z, int nb_fdm, AggArgs ag) {
    static_assert(N * N <= 64, "one lane per extended point");
    constexpr int N2 = N - 2, NP = N * N, NP2 = N2 * N2;
    __shared__ double sS[4][2][N * N];
    __shared__ double sA[4][NP], sB[4][NP];
    if ((int)blockIdx.x >= nb_fdm) {   // merged launch: the aggregate restriction of the coarse chain (see k_fdm_ext)
        agg_restrict_body((int)blockIdx.x - nb_fdm, flag, 0, ag);
        return;
    }
    if (flag && flag[0] != 0.0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 4 + wv;
    const bool act = e < E;
    const int64_t ee = act ? e : 0;
    for (int q = lane; q < 2 * N * N; q += 64) sS[wv][q / (N * N)][q % (N * N)] = S[ee * (3 * N * N) + q];
    const bool on = lane < NP;
    const int p = on ? lane : 0;
    const int a = p % N, b = p / N;
    const int nb = (a == 0 || a == N - 1) + (b == 0 || b == N - 1);
    const int a2 = min(max(a, 1), N - 2) - 1, b2 = min(max(b, 1), N - 2) - 1;
    const int q2 = a2 + N2 * b2;
    double v = 0.0;
    if (nb <= 1) {
        const double own = r[ee * NP2 + q2] * wq[ee * NP2 + q2];
        v = nb == 0 ? own : W[ee * NP + p] - own;
    }
    if (on) sA[wv][p] = v;
    __syncthreads();
    double t = 0.0;
    if (on) fdm_stage2<N, true, 0>(sA[wv], sB[wv], sS[wv][0], p);
    __syncthreads();
    if (on) fdm_stage2<N, true, 1>(sB[wv], sA[wv], sS[wv][1], p);
    __syncthreads();
    if (on) {
        const double den = lam[ee * (3 * N) + a] + lam[ee * (3 * N) + N + b];
        sA[wv][p] = den > thr ? sA[wv][p] / den : 0.0;
    }
    __syncthreads();
    if (on) fdm_stage2<N, false, 1>(sA[wv], sB[wv], sS[wv][1], p);
    __syncthreads();
    if (on) fdm_stage2<N, false, 0>(sB[wv], sA[wv], sS[wv][0], p);
    __syncthreads();
    (void)t;
    if (act && on) {
        if (nb == 1) {
            W[e * NP + p] = sA[wv][p];
        } else if (nb == 0) {
            double o = sA[wv][p];
            if (a == 1) o -= sA[wv][p - 1];
            if (a == N - 2) o -= sA[wv][p + 1];
            if (b == 1) o -= sA[wv][p - N];
            if (b == N - 2) o -= sA[wv][p + N];
            z[e * NP2 + q2] = o;
        }
    }
}

template <int N>
__global__ __launch_bounds__(NT) void k_sch_finish2(const double *__restrict__ flag, int64_t E, const double *__restrict__ W,
                                                    const double *__restrict__ r, const double *__restrict__ wq,
                                                    const double *__restrict__ xc, const double *__restrict__ xa,
                                                    const int *__restrict__ agg, const int *__restrict__ vg, Hat hat,
                                                    double *__restrict__ z, double *__restrict__ part) {
    constexpr int N2 = N - 2, NP = N * N, NP2 = N2 * N2;
    __shared__ double srz[2][NT / 64];
    if (flag && flag[0] != 0.0) return;
    const int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x;
    double srzv = 0.0, szv = 0.0;
    if (i < E * NP2) {
    const int64_t e = i / NP2;
    const int q = (int)(i % NP2), a = q % N2, b = q / N2;
    const double *We = W + e * NP;
    double v = z[i];
    if (a == 0) v += We[0 + N * (b + 1)];
    if (a == N2 - 1) v += We[(N - 1) + N * (b + 1)];
    if (b == 0) v += We[(a + 1) + N * 0];
    if (b == N2 - 1) v += We[(a + 1) + N * (N - 1)];
    v *= wq[i];
    if (xc) {
        double cv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int vv = vg[e * 4 + c];
            cv[c] = xc[vv] + xa[agg[vv]];
        }
        const double ha = hat.h1[a], hb = hat.h1[b];
        const double c0 = cv[0] + ha * (cv[1] - cv[0]), c1 = cv[2] + ha * (cv[3] - cv[2]);
        v += c0 + hb * (c1 - c0);
    }
    z[i] = v;
    srzv = r[i] * v;
    szv = v;
    }
    if (part) {   // first-stage sums of the PCG: part[b] = sum r z, part[nb + b] = sum z
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            srzv += __shfl_down(srzv, o, 64);
            szv += __shfl_down(szv, o, 64);
        }
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        if (lane == 0) {
            srz[0][wid] = srzv;
            srz[1][wid] = szv;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < NT / 64; ++w) {
                a += srz[0][w];
                b += srz[1][w];
            }
            part[blockIdx.x] = a;
            part[gridDim.x + blockIdx.x] = b;
        }
    }
}

// t[e][c] = sum_q phi_c(q) r_e(q): the element-local part of R_1^T r, one wave per element
template <int DIM>
__global__ __launch_bounds__(NT) void k_q1_restrict_local(const double *__restrict__ flag, int64_t E, int n2, Hat hat,
                                                          double *__restrict__ r, double *__restrict__ t,
                                                          double *__restrict__ W, const double *__restrict__ wq,
                                                          nlg_pcg_upd u, int64_t ld = 0, int64_t lt = 0, int64_t lW = 0) {
    __shared__ double srr[4];
    {   // blockIdx.y = lane of a block step: solver fields ld doubles apart (the integrator's slab), the preconditioner's own scratch at its strides
        const int64_t lo = (int64_t)blockIdx.y * ld;
        if (flag) flag += lo;
        r += lo;
        t += (int64_t)blockIdx.y * lt;
        if (W) W += (int64_t)blockIdx.y * lW;
        if (u.alpha) u.alpha += lo, u.wmean += lo, u.w += lo, u.rr_part += lo;
        if (u.x) u.x += lo, u.p += lo;   // (x == null: deferred solution update, the directions stay in the ring of the solver)
    }
    if (flag && flag[0] != 0.0) return;
    constexpr int NC = 1 << DIM;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 4 + wid;
    const bool act = e < E;
    const int np2 = DIM == 3 ? n2 * n2 * n2 : n2 * n2;
    const bool upd = u.alpha != nullptr;
    const double alpha = upd ? u.alpha[0] : 0.0, wmean = upd ? u.wmean[0] : 0.0;
    double rr = 0.0;
    double a[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) a[c] = 0.0;
    for (int q = lane; act && q < np2; q += 64) {
        double v = r[e * np2 + q];
        if (upd) {   // the PCG update of this point: the new residual is what gets restricted
            const int64_t i = e * np2 + q;
            if (u.x) u.x[i] += alpha * u.p[i];
            v -= alpha * (u.w[i] - wmean);
            r[i] = v;
            rr += v * v * u.nw[i];
        }
        const double ha = hat.h1[q % n2], hb = hat.h1[(q / n2) % n2];
        const double hc = DIM == 3 ? hat.h1[q / (n2 * n2)] : 0.0;
        if (DIM == 2 && W) {
            const int N = n2 + 2, a = q % n2, b = q / n2;
            double *We = W + e * (int64_t)(N * N);
            const double vw = v * wq[e * np2 + q];
            if (a == 0) We[0 + N * (b + 1)] = vw;
            if (a == n2 - 1) We[(N - 1) + N * (b + 1)] = vw;
            if (b == 0) We[(a + 1)] = vw;
            if (b == n2 - 1) We[(a + 1) + N * (N - 1)] = vw;
        }
        if (DIM == 3 && W) {
            // overlapping Schwarz, pack: the layers adjacent to the element faces go to the face points of W
            const int N = n2 + 2, a = q % n2, b = (q / n2) % n2, c = q / (n2 * n2);
            double *We = W + e * (int64_t)(N * N * N);
            const double vw = v * wq[e * np2 + q];
            if (a == 0) We[fg_slot(N, 0, b + 1, c + 1)] = vw;
            if (a == n2 - 1) We[fg_slot(N, N - 1, b + 1, c + 1)] = vw;
            if (b == 0) We[fg_slot(N, a + 1, 0, c + 1)] = vw;
            if (b == n2 - 1) We[fg_slot(N, a + 1, N - 1, c + 1)] = vw;
            if (c == 0) We[fg_slot(N, a + 1, b + 1, 0)] = vw;
            if (c == n2 - 1) We[fg_slot(N, a + 1, b + 1, N - 1)] = vw;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double w = ((c & 1) ? ha : 1.0 - ha) * ((c & 2) ? hb : 1.0 - hb);
            if (DIM == 3) w *= (c & 4) ? hc : 1.0 - hc;
            a[c] += w * v;
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[c] += __shfl_down(a[c], o, 64);
    }
    if (lane == 0 && act) {
#pragma unroll
        for (int c = 0; c < NC; ++c) t[e * NC + c] = a[c];
    }
    if (upd) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rr += __shfl_down(rr, o, 64);
        if (lane == 0) srr[wid] = rr;
        __syncthreads();
        if (threadIdx.x == 0) u.rr_part[blockIdx.x] = (srr[0] + srr[1]) + (srr[2] + srr[3]);
    }
}

// The same for 3-D with the pressure-mesh size known at compile time (lx1 = 8, 10, 12): the point loop is unrolled, so all loads
// of a wave are in flight at once, the index arithmetic is constant-folded, and the face slots of W come from a table
// (wslot[q][d]: slot of the face point next to pressure point q in direction d, -1 if q is not in that boundary layer)
// instead of fg_slot's branches.  Same summation order as the generic kernel: bit-identical results.
template <int N2>
__global__ __launch_bounds__(NT) void k_q1_restrict_local3s(const double *__restrict__ flag, int64_t E, Hat hat, double *__restrict__ r,
                                                            double *__restrict__ t, double *__restrict__ W, const double *__restrict__ wq,
                                                            const int *__restrict__ wslot, nlg_pcg_upd u, int64_t ld, int64_t lt, int64_t lW) {
    __shared__ double srr[4];
    {   // blockIdx.y = lane of a block step: solver fields ld doubles apart (the integrator's slab), the preconditioner's own scratch at its strides
        const int64_t lo = (int64_t)blockIdx.y * ld;
        if (flag) flag += lo;
        r += lo;
        t += (int64_t)blockIdx.y * lt;
        if (W) W += (int64_t)blockIdx.y * lW;
        if (u.alpha) u.alpha += lo, u.wmean += lo, u.w += lo, u.rr_part += lo;
        if (u.x) u.x += lo, u.p += lo;   // (x == null: deferred solution update, the directions stay in the ring of the solver)
    }
    if (flag && flag[0] != 0.0) return;
    constexpr int NP2 = N2 * N2 * N2, N = N2 + 2, NIT = 4;   // four points per lane in flight (lx1 = 8: the whole element)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 4 + wid;
    const bool act = e < E;
    const bool upd = u.alpha != nullptr;
    const double alpha = upd ? u.alpha[0] : 0.0, wmean = upd ? u.wmean[0] : 0.0;
    double rr = 0.0;
    double a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = 0.0;
    for (int q0 = 0; q0 < NP2; q0 += 64 * NIT) {
        double v[NIT], pv[NIT], wv[NIT], xv[NIT], nv[NIT], qv[NIT], hh[NIT][3];
        int sl[NIT][3];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = q0 + lane + 64 * it;
            const int qc = q < NP2 ? q : 0;                 // (unconditional loads at clamped addresses: no branch, no wait per point)
            const int64_t i = (act ? e : 0) * NP2 + qc;
            v[it] = r[i];
            if (upd) {
                wv[it] = u.w[i];
                nv[it] = u.nw[i];
                if (u.x) pv[it] = u.p[i], xv[it] = u.x[i];
            }
            if (W) {
                qv[it] = wq[i];
#pragma unroll
                for (int d = 0; d < 3; ++d) sl[it][d] = wslot[3 * qc + d];
            }
            hh[it][0] = hat.h1[qc % N2], hh[it][1] = hat.h1[(qc / N2) % N2], hh[it][2] = hat.h1[qc / (N2 * N2)];   // (a load too)
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = q0 + lane + 64 * it;
            if (!(act && q < NP2)) continue;
            const int64_t i = e * NP2 + q;
            double vv = v[it];
            if (upd) {
                if (u.x) u.x[i] = xv[it] + alpha * pv[it];
                vv -= alpha * (wv[it] - wmean);
                r[i] = vv;
                rr += vv * vv * nv[it];
            }
            const double ha = hh[it][0], hb = hh[it][1], hc = hh[it][2];
            if (W) {
                double *We = W + e * (int64_t)(N * N * N);
                const double vw = vv * qv[it];
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    if (sl[it][d] >= 0) We[sl[it][d]] = vw;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                double w = ((c & 1) ? ha : 1.0 - ha) * ((c & 2) ? hb : 1.0 - hb);
                w *= (c & 4) ? hc : 1.0 - hc;
                a[c] += w * vv;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[c] += __shfl_down(a[c], o, 64);
    }
    if (lane == 0 && act) {
#pragma unroll
        for (int c = 0; c < 8; ++c) t[e * 8 + c] = a[c];
    }
    if (upd) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rr += __shfl_down(rr, o, 64);
        if (lane == 0) srr[wid] = rr;
        __syncthreads();
        if (threadIdx.x == 0) u.rr_part[blockIdx.x] = (srr[0] + srr[1]) + (srr[2] + srr[3]);
    }
}

// rc[v] = sum of t over the (element, corner) entries incident to vertex v (fixed order), and the first damped-Jacobi
// sweep from a zero guess
struct GatherArgs {
    int nvert;
    const int *vp, *vi;
    const double *t;
    double *rc;
    const double *dinv;
    double om;
    double *x;
    int64_t lt, lv;
};
__device__ __forceinline__ void q1_gather_body(int bx, const double *flag, int64_t ld, const GatherArgs &g) {
    if (flag) flag += (int64_t)blockIdx.y * ld;
    if (flag && flag[0] != 0.0) return;
    const double *t = g.t + (int64_t)blockIdx.y * g.lt;
    double *rc = g.rc + (int64_t)blockIdx.y * g.lv, *x = g.x + (int64_t)blockIdx.y * g.lv;
    const int v = bx * NT + threadIdx.x;
    if (v >= g.nvert) return;
    double a = 0.0;
    for (int q = g.vp[v]; q < g.vp[v + 1]; ++q) a += t[g.vi[q]];
    rc[v] = a;
    x[v] = g.om * g.dinv[v] * a;
}
__global__ __launch_bounds__(NT) void k_q1_gather(const double *__restrict__ flag, int nvert, const int *__restrict__ vp,
                                                  const int *__restrict__ vi, const double *__restrict__ t,
                                                  double *__restrict__ rc, const double *__restrict__ dinv, double om,
                                                  double *__restrict__ x, int64_t ld = 0, int64_t lt = 0, int64_t lv = 0) {
    if (flag) flag += (int64_t)blockIdx.y * ld;
    t += (int64_t)blockIdx.y * lt, rc += (int64_t)blockIdx.y * lv, x += (int64_t)blockIdx.y * lv;
    if (flag && flag[0] != 0.0) return;
    const int v = blockIdx.x * NT + threadIdx.x;
    if (v >= nvert) return;
    double a = 0.0;
    for (int q = vp[v]; q < vp[v + 1]; ++q) a += t[vi[q]];
    rc[v] = a;
    x[v] = om * dinv[v] * a;
}

// probing vector of the coarse-operator assembly: p = phi_c on the elements of colour `col`, 0 elsewhere
template <int DIM>
__global__ __launch_bounds__(NT) void k_q1_probe(int64_t E, int n2, Hat hat, const int *__restrict__ colour, int col, int c,
                                                 double *__restrict__ p) {
    const int np2 = DIM == 3 ? n2 * n2 * n2 : n2 * n2;
    const int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x;
    if (i >= E * np2) return;
    const int64_t e = i / np2;
    const int q = (int)(i % np2);
    double w = 0.0;
    if (colour[e] == col) {
        const double ha = hat.h1[q % n2], hb = hat.h1[(q / n2) % n2];
        w = ((c & 1) ? ha : 1.0 - ha) * ((c & 2) ? hb : 1.0 - hb);
        if (DIM == 3) {
            const double hc = hat.h1[q / (n2 * n2)];
            w *= (c & 4) ? hc : 1.0 - hc;
        }
    }
    p[i] = w;
}

// probing vector of the GLOBAL aggregate operator (several ranks): p = R_1 P_a e_a, the prolongation of the indicator
// of aggregate `a` (a < 0: this rank is not the probing one, p = 0)
template <int DIM>
__global__ __launch_bounds__(NT) void k_agg_probe(int64_t E, int n2, Hat hat, const int *__restrict__ vg,
                                                  const int *__restrict__ agg, int a, double *__restrict__ p) {
    constexpr int NC = 1 << DIM;
    const int np2 = DIM == 3 ? n2 * n2 * n2 : n2 * n2;
    const int64_t i = blockIdx.x * (int64_t)NT + threadIdx.x;
    if (i >= E * np2) return;
    double w = 0.0;
    if (a >= 0) {
        const int64_t e = i / np2;
        const int q = (int)(i % np2);
        const double ha = hat.h1[q % n2], hb = hat.h1[(q / n2) % n2];
        const double hc = DIM == 3 ? hat.h1[q / (n2 * n2)] : 0.0;
        for (int c = 0; c < NC; ++c)
            if (agg[vg[e * NC + c]] == a) {
                double t = ((c & 1) ? ha : 1.0 - ha) * ((c & 2) ? hb : 1.0 - hb);
                if (DIM == 3) t *= (c & 4) ? hc : 1.0 - hc;
                w += t;
            }
    }
    p[i] = w;
}

// column `col` of this rank's rows of the global aggregate operator
__global__ void k_store_col(int na, const double *__restrict__ ra, double *__restrict__ rows, int64_t ncols, int64_t col) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) rows[(size_t)i * ncols + col] = ra[i];
}

// gathered operator -> invertible: unit diagonal on the padding rows, symmetrised, constant null space shifted
__global__ void k_glob_fix(int64_t n, int na_max, const int *__restrict__ na_of, double alpha, double *__restrict__ A) {
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= n || j < i) return;
    const bool ri = (int)(i % na_max) < na_of[i / na_max], rj = (int)(j % na_max) < na_of[j / na_max];
    double v;
    if (ri && rj)
        v = 0.5 * (A[(size_t)i * n + j] + A[(size_t)j * n + i]) + alpha;
    else
        v = i == j ? 1.0 : 0.0;
    A[(size_t)i * n + j] = v;
    A[(size_t)j * n + i] = v;
}

__global__ void k_mirror_upper(int n, double *__restrict__ A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < i) A[(size_t)i * n + j] = A[(size_t)j * n + i];
}

// ra[a] = sum of rr over the members of aggregate a, one wave per aggregate
__global__ __launch_bounds__(NT) void k_agg_restrict(const double *flag, int na, const int *__restrict__ ap,
                                                     const int *__restrict__ am, const double *__restrict__ rr,
                                                     double *__restrict__ ra, int64_t ld = 0, int64_t lv = 0, int64_t la = 0) {
    if (flag) flag += (int64_t)blockIdx.y * ld;
    rr += (int64_t)blockIdx.y * lv, ra += (int64_t)blockIdx.y * la;
    if (flag && flag[0] != 0.0) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int a = blockIdx.x * 4 + wid;
    if (a >= na) return;
    double s = 0.0;
    for (int q = ap[a] + lane; q < ap[a + 1]; q += 64) s += rr[am[q]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) ra[a] = s;
}

// xa = Ainv ra (dense, row-major), one wave per row.  T = float for the global aggregate level of several ranks: the
// rows of the inverse are the largest per-iteration read there (2.4k x 19.6k at 8 x 10^4 elements), a preconditioner
// needs no more than single precision, and the symmetric matrix is rounded entry by entry, so it stays symmetric
// across the ranks that hold its rows; sums in double.
template <typename T>
struct GemvArgs {
    int na, ncols;
    const T *Ainv;
    const double *ra;
    double *xa;
    int64_t la;
    int na_max, nlanes;
};
// lanes of a block step (blockIdx.y): one rank -- ra, xa la doubles apart; several ranks -- ra is the all-gathered array
// [rank][lane][na_max] (na_max > 0), column c of the global aggregate level = entry c % na_max of rank c / na_max
template <typename T>
__device__ __forceinline__ void dense_gemv_body(int bx, const double *flag, int64_t ld, const GemvArgs<T> &g) {
    if (flag) flag += (int64_t)blockIdx.y * ld;
    if (flag && flag[0] != 0.0) return;
    double *xa = g.xa + (int64_t)blockIdx.y * g.la;
    const double *ra = g.na_max == 0 ? g.ra + (int64_t)blockIdx.y * g.la : g.ra;
    const int na = g.na, ncols = g.ncols;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = bx * 4 + wid;
    if (row >= na) return;
    const T *__restrict__ Ar = g.Ainv + (size_t)row * ncols;
    double s = 0.0;
    if (g.na_max > 0 && g.nlanes > 1) {   // (strided gather of the lane's entries; set-up sizes: a few thousand columns)
        for (int j = lane; j < ncols; j += 64)
            s += (double)Ar[j] * ra[((int64_t)(j / g.na_max) * g.nlanes + blockIdx.y) * g.na_max + j % g.na_max];
    } else {
        // eight loads of the row in flight per lane (same summation order as the plain loop: the products are added one after the
        // other); a wave per row gives only ~1.3 waves per SIMD at 1300 rows, so the loads of one wave must overlap themselves
        int j = lane;
        for (; j + 7 * 64 < ncols; j += 8 * 64) {
            double av[8], rv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = (double)Ar[j + 64 * u];
                rv[u] = ra[j + 64 * u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += av[u] * rv[u];
        }
        for (; j < ncols; j += 64) s += (double)Ar[j] * ra[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) xa[row] = s;
}
// xa = Ainv ra (dense, row-major), one wave per row.  T = float for the global aggregate level of several ranks: the
// rows of the inverse are the largest per-iteration read there (2.4k x 19.6k at 8 x 10^4 elements), a preconditioner
// needs no more than single precision, and the symmetric matrix is rounded entry by entry, so it stays symmetric
// across the ranks that hold its rows; sums in double.
template <typename T>
__global__ __launch_bounds__(NT) void k_dense_gemv(const double *flag, int na, int ncols, const T *__restrict__ Ainv,
                                                   const double *__restrict__ ra, double *__restrict__ xa, int64_t ld = 0, int64_t la = 0,
                                                   int na_max = 0, int nlanes = 1) {
    const GemvArgs<T> g = {na, ncols, Ainv, ra, xa, la, na_max, nlanes};
    dense_gemv_body<T>((int)blockIdx.x, flag, ld, g);
}

// ---- merged launches: the coarse-grid chain rides in the launches of the fine level --------------------------------------
// The vertex gather, the aggregate restriction and the dense solve are tiny (5 - 8 us each, one after the other); the three
// fine-level launches they interleave with are independent of them.  One launch = the blocks of the fine kernel followed by the
// blocks of a coarse kernel (block-uniform branch on blockIdx.x): the coarse work runs beside the fine work on otherwise idle
// CUs and its three launches -- and their start-up latency on the stream -- disappear.
// (a) pairs-only gather-scatter of the exchange array W + vertex gather
__device__ __forceinline__ void pairs_body(int64_t bx, const int *__restrict__ idx, int64_t npairs, double *__restrict__ w,
                                           const double *__restrict__ gate, int64_t ldw, int64_t ldg) {
    if (gate && gate[(int64_t)blockIdx.y * ldg] != 0.0) return;
    w += (int64_t)blockIdx.y * ldw;
    const int64_t t = bx * (int64_t)NT + threadIdx.x;
    const int64_t np2 = npairs >> 1;
    if (t < np2) {
        const int4 q = reinterpret_cast<const int4 *>(idx)[t];
        if (q.z == q.x + 1 && q.w == q.y + 1 && !((q.x | q.y) & 1)) {
            const double2 a = *reinterpret_cast<const double2 *>(w + q.x), b = *reinterpret_cast<const double2 *>(w + q.y);
            double2 sv;
            sv.x = a.x + b.x;
            sv.y = a.y + b.y;
            *reinterpret_cast<double2 *>(w + q.x) = sv;
            *reinterpret_cast<double2 *>(w + q.y) = sv;
        } else {
            const double s0 = w[q.x] + w[q.y], s1 = w[q.z] + w[q.w];
            w[q.x] = s0;
            w[q.y] = s0;
            w[q.z] = s1;
            w[q.w] = s1;
        }
        return;
    }
    const int64_t g = 2 * np2 + (t - np2);
    if (g >= npairs) return;
    const int2 ab = reinterpret_cast<const int2 *>(idx)[g];
    const double sv = w[ab.x] + w[ab.y];
    w[ab.x] = sv;
    w[ab.y] = sv;
}
__global__ __launch_bounds__(NT) void k_pairs_gather(int nb_pairs, const int *__restrict__ idx, int64_t npairs, double *__restrict__ w,
                                                     int64_t ldw, const double *__restrict__ flag, int64_t ld, GatherArgs g) {
    if ((int)blockIdx.x < nb_pairs)
        pairs_body(blockIdx.x, idx, npairs, w, flag, ldw, ld);
    else
        q1_gather_body((int)blockIdx.x - nb_pairs, flag, ld, g);
}
// (c) the second pairs-only gather-scatter + the dense aggregate solve
template <typename T>
__global__ __launch_bounds__(NT) void k_pairs_gemv(int nb_pairs, const int *__restrict__ idx, int64_t npairs, double *__restrict__ w,
                                                   int64_t ldw, const double *__restrict__ flag, int64_t ld, GemvArgs<T> g) {
    if ((int)blockIdx.x < nb_pairs)
        pairs_body(blockIdx.x, idx, npairs, w, flag, ldw, ld);
    else
        dense_gemv_body<T>((int)blockIdx.x - nb_pairs, flag, ld, g);
}

__global__ void k_to_float(int64_t n, const double *__restrict__ a, float *__restrict__ b) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) b[i] = (float)a[i];
}

// ---- dense SPD inverse on the device (set-up): in-place Gauss-Jordan without pivoting, two launches per pivot ----
// rk = row k / pivot, ck = column k (before the step); bad[0] is raised when a pivot is not positive
__global__ __launch_bounds__(NT) void k_gj_prep(int n, int k, const double *__restrict__ A, double *__restrict__ rk,
                                                double *__restrict__ ck, int *__restrict__ bad) {
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j >= n) return;
    const double p = A[(size_t)k * n + k];
    if (j == 0 && !(p > 0.0)) bad[0] = 1;
    rk[j] = A[(size_t)k * n + j] / p;
    ck[j] = A[(size_t)j * n + k];
}
__global__ __launch_bounds__(NT) void k_gj_update(int n, int k, double *__restrict__ A, const double *__restrict__ rk,
                                                  const double *__restrict__ ck) {
    const int j = blockIdx.x * NT + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    const double p = ck[k];   // the pivot
    double v;
    if (i == k)
        v = j == k ? 1.0 / p : rk[j];
    else
        v = j == k ? -ck[i] / p : A[(size_t)i * n + j] - ck[i] * rk[j];
    A[(size_t)i * n + j] = v;
}

// ---- host helpers -------------------------------------------------------------------------------------
// symmetric eigen-decomposition by cyclic Jacobi rotations (n <= 16): A = V diag(w) V^T
void jacobi_eig(int n, std::vector<double> &A, std::vector<double> &V, std::vector<double> &w) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

// generalised symmetric problem A s = lam B s, B SPD: S^T B S = I, S^T A S = diag(lam). S row-major [point][mode].
int gen_eig(int n, const std::vector<double> &A, const std::vector<double> &B, std::vector<double> &S,
            std::vector<double> &lam) {
    std::vector<double> L((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = B[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0.0)) return 1;
        L[(size_t)j * n + j] = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double s = B[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / L[(size_t)j * n + j];
        }
    }
    // Linv (lower)
    std::vector<double> Li((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c) {
        Li[(size_t)c * n + c] = 1.0 / L[(size_t)c * n + c];
        for (int i = c + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += L[(size_t)i * n + k] * Li[(size_t)k * n + c];
            Li[(size_t)i * n + c] = -s / L[(size_t)i * n + i];
        }
    }
    std::vector<double> T((size_t)n * n, 0.0), Cm((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += Li[(size_t)i * n + k] * A[(size_t)k * n + j];
            T[(size_t)i * n + j] = s;
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += T[(size_t)i * n + k] * Li[(size_t)j * n + k];
            Cm[(size_t)i * n + j] = s;
        }
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) Cm[(size_t)i * n + j] = Cm[(size_t)j * n + i] = 0.5 * (Cm[(size_t)i * n + j] + Cm[(size_t)j * n + i]);
    std::vector<double> V;
    jacobi_eig(n, Cm, V, lam);
    S.assign((size_t)n * n, 0.0);   // S = Li^T V
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += Li[(size_t)k * n + i] * V[(size_t)k * n + j];
            S[(size_t)i * n + j] = s;
        }
    return 0;
}

// dense SPD inverse by Cholesky (in place, row-major); returns non-zero if not positive definite
int spd_inverse(int n, std::vector<double> &A) {
    std::vector<double> L((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0.0)) return 1;
        L[(size_t)j * n + j] = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / L[(size_t)j * n + j];
        }
    }
    std::vector<double> Li((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c) {
        Li[(size_t)c * n + c] = 1.0 / L[(size_t)c * n + c];
        for (int i = c + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += L[(size_t)i * n + k] * Li[(size_t)k * n + c];
            Li[(size_t)i * n + c] = -s / L[(size_t)i * n + i];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int k = i; k < n; ++k) s += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
            A[(size_t)i * n + j] = A[(size_t)j * n + i] = s;
        }
    return 0;
}

template <typename T>
int up(const std::vector<T> &v, T **d) {
    NLG_HIP(hipMalloc(d, sizeof(T) * std::max<size_t>(v.size(), 1)));
    if (!v.empty()) NLG_HIP(hipMemcpy(*d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

namespace nlg {

int pprec_setup(nlg_mesh *m, const nlg_mesh_desc *d) {
    if (getenv("NLG_COARSE_SCALE")) {
        const double cs = atof(getenv("NLG_COARSE_SCALE"));
        NLG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_coarse_scale), &cs, sizeof(double)));
    }
    nlg_pprec &P = m->pprec;
    nlg_ctx *ctx = m->ctx;
    hipStream_t st = ctx->stream;
    const int dim = m->dim, n = m->n, n2 = m->n2, np1 = m->np1, np2 = m->np2;
    const int64_t E = m->E;
    const nlg_ops1d &o = m->ops;
    // ---- host copies of what the set-up needs
    std::vector<double> vmult((size_t)m->lvn), binv((size_t)m->lvn);
    NLG_HIP(hipMemcpyAsync(vmult.data(), m->d_vmult, sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToHost, st));
    NLG_HIP(hipMemcpyAsync(binv.data(), m->d_binvm1, sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToHost, st));
    NLG_HIP(hipStreamSynchronize(st));
    const double *X[3] = {d->xm1, d->ym1, d->zm1};
    const double *msk[3] = {d->v1mask, d->v2mask, d->v3mask};
    // ---- 1. element-wise fast diagonalisation
    std::vector<double> Dh((size_t)n2 * n), Ih((size_t)n2 * n);
    for (int k = 0; k < n2; ++k)
        for (int j = 0; j < n; ++j) {
            Dh[(size_t)k * n + j] = o.w2[k] * o.D12[(size_t)k * n + j];
            Ih[(size_t)k * n + j] = o.w2[k] * o.I12[(size_t)k * n + j];
        }
    std::vector<double> hS((size_t)E * 3 * n2 * n2, 0.0), hden((size_t)E * np2, 0.0);
    std::vector<double> lam3((size_t)3 * n2);
    std::vector<double> Lall((size_t)E * 3, 0.0), endfac((size_t)E * 6, 0.0);   // edge lengths; end-point factors [e][d][side]
    const int mid = n / 2;
    double denmax = 0.0;
    for (int64_t e = 0; e < E; ++e) {
        for (int dd = 0; dd < dim; ++dd) {
            // face centres in direction dd
            double c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
            int cnt = 0;
            for (int p = 0; p < np1; ++p) {
                const int ijk[3] = {p % n, (p / n) % n, p / (n * n)};
                if (ijk[dd] == 0) {
                    for (int c = 0; c < dim; ++c) c0[c] += X[c][e * np1 + p];
                    ++cnt;
                } else if (ijk[dd] == n - 1) {
                    for (int c = 0; c < dim; ++c) c1[c] += X[c][e * np1 + p];
                }
            }
            double l = 0.0;
            for (int c = 0; c < dim; ++c) l += (c1[c] - c0[c]) * (c1[c] - c0[c]) / ((double)cnt * cnt);
            l = std::sqrt(l);
            std::vector<double> bi(n);
            for (int i = 0; i < n; ++i) bi[i] = 1.0 / (0.5 * l * o.w1[i]);
            for (int side = 0; side < 2; ++side) {
                int ijk[3] = {mid, mid, dim == 3 ? mid : 0};
                ijk[dd] = side == 0 ? 0 : n - 1;
                const int64_t q = e * np1 + ijk[0] + n * (ijk[1] + n * ijk[2]);
                const int k = side == 0 ? 0 : n - 1;
                bi[k] = (msk[dd][q] == 0.0) ? 0.0 : bi[k] * vmult[q];
                endfac[(size_t)e * 6 + dd * 2 + side] = (msk[dd][q] == 0.0) ? 0.0 : vmult[q];
            }
            Lall[(size_t)e * 3 + dd] = l;
            std::vector<double> A((size_t)n2 * n2), B((size_t)n2 * n2), S, lam;
            for (int a = 0; a < n2; ++a)
                for (int b = 0; b < n2; ++b) {
                    double sa = 0.0, sb = 0.0;
                    for (int j = 0; j < n; ++j) {
                        sa += Dh[(size_t)a * n + j] * bi[j] * Dh[(size_t)b * n + j];
                        sb += Ih[(size_t)a * n + j] * bi[j] * Ih[(size_t)b * n + j];
                    }
                    A[(size_t)a * n2 + b] = sa;
                    B[(size_t)a * n2 + b] = 0.25 * l * l * sb;
                }
            NLG_CHECK(gen_eig(n2, A, B, S, lam) == 0, "pprec_setup: 1-D mass matrix not positive definite (element %lld)", (long long)e);
            for (int q = 0; q < n2 * n2; ++q) hS[((size_t)e * 3 + dd) * n2 * n2 + q] = S[q];
            for (int a = 0; a < n2; ++a) lam3[(size_t)dd * n2 + a] = std::max(lam[a], 0.0);
        }
        for (int q = 0; q < np2; ++q) {
            const int a = q % n2, b = (q / n2) % n2, c = q / (n2 * n2);
            double den = lam3[a] + lam3[n2 + b] + (dim == 3 ? lam3[2 * n2 + c] : 0.0);
            hden[(size_t)e * np2 + q] = den;
            denmax = std::max(denmax, den);
        }
    }
    for (auto &v : hden) v = (v > 1e-12 * denmax) ? 1.0 / v : 0.0;
    NLG_TRY(up(hS, &P.d_S));
    NLG_TRY(up(hden, &P.d_invden));

    // ---- 1b. overlapping variant: extended 1-D operators from the line left neighbour | element | right neighbour
    // (3-D: exchange array in the face-grouped layout; 2-D: natural layout)
    if (((dim == 3 && m->gs.d_indices_fg && n <= 12 && n != 11) || (dim == 2 && n <= 8)) && m->gs.npairs > 0) {
        // normal edge length of the face neighbours, through the gather-scatter: every element puts its own
        // normal length on the interior points of its faces, the sum minus the own value is the neighbour's
        std::vector<double> hw((size_t)m->lvn, 0.0);
        auto face_node = [&](int dd, int side, int u, int v) {
            int ijk[3] = {0, 0, 0};
            ijk[dd] = side == 0 ? 0 : n - 1;
            ijk[(dd + 1) % dim] = u;
            if (dim == 3) ijk[(dd + 2) % 3] = v;
            return ijk[0] + n * (ijk[1] + n * ijk[2]);
        };
        for (int64_t e = 0; e < E; ++e)
            for (int dd = 0; dd < dim; ++dd)
                for (int side = 0; side < 2; ++side)
                    for (int v = 1; v < (dim == 3 ? n - 1 : 2); ++v)
                        for (int u = 1; u < n - 1; ++u) hw[(size_t)e * np1 + face_node(dd, side, u, v)] = Lall[(size_t)e * 3 + dd];
        double *dw = sem_scratch1(m, 0);
        NLG_CHECK(dw, "pprec_setup: scratch allocation failed");
        NLG_HIP(hipMemcpyAsync(dw, hw.data(), sizeof(double) * (size_t)m->lvn, hipMemcpyHostToDevice, st));
        {
            // halo on: a face neighbour on another rank counts like a local one (its ghost layer travels with the
            // halo exchange of the array W, pprec_fine)
            double *f1[1] = {dw};
            NLG_TRY(sem_gs(m, f1, 1));
        }
        NLG_HIP(hipMemcpyAsync(hw.data(), dw, sizeof(double) * (size_t)m->lvn, hipMemcpyDeviceToHost, st));
        NLG_HIP(hipStreamSynchronize(st));
        const int mx = n;   // extended 1-D size
        std::vector<double> hSx((size_t)E * 3 * mx * mx, 0.0), hlx((size_t)E * 3 * mx, 0.0);
        std::vector<double> Bl((size_t)3 * n), bq((size_t)3 * n), Ae((size_t)mx * mx), Be((size_t)mx * mx), Sx, lx;
        std::vector<double> Dl((size_t)mx * 3 * n), Il((size_t)mx * 3 * n);   // rows: the mx extended pressure points
        // symmetric weighting D M D, D = diag(count^-1/2), count = number of extended subdomains that contain the point
        // (1 + one per face neighbour whose ghost layer it is).  Prototype (docs/prototypes/precond_proto5.py, 6^3 elements,
        // lx1 = 6): unweighted 21 / 29 iterations with the exact / approximate coarse solve, weighted 14 / 18;
        // without overlap 29.
        std::vector<double> hwq((size_t)E * np2, 1.0);
        double dmax = 0.0;
        for (int64_t e = 0; e < E; ++e) {
            for (int dd = 0; dd < dim; ++dd) {
                const double lm = Lall[(size_t)e * 3 + dd];
                double ln[2];
                for (int side = 0; side < 2; ++side) {
                    const double sum = hw[(size_t)e * np1 + face_node(dd, side, mid, dim == 3 ? mid : 0)];
                    ln[side] = sum - lm > 1e-10 * lm ? sum - lm : 0.0;
                }
                const double len[3] = {ln[0], lm, ln[1]};
                for (int side = 0; side < 2; ++side)
                    if (ln[side] > 0.0)
                        for (int q = 0; q < np2; ++q) {
                            const int idx3[3] = {q % n2, (q / n2) % n2, q / (n2 * n2)};
                            if (idx3[dd] == (side == 0 ? 0 : n2 - 1)) hwq[(size_t)e * np2 + q] += 1.0;
                        }
                const int nvl = 3 * n - 2;
                std::fill(Bl.begin(), Bl.end(), 0.0);
                std::fill(Dl.begin(), Dl.end(), 0.0);
                std::fill(Il.begin(), Il.end(), 0.0);
                for (int q = 0; q < 3; ++q) {
                    if (len[q] == 0.0) continue;
                    const int ov = q * (n - 1);
                    for (int i = 0; i < n; ++i) Bl[ov + i] += 0.5 * len[q] * o.w1[i];
                    // extended rows owned by element q of the line: left -> row 0 (its last pressure point),
                    // middle -> rows 1..n2, right -> row mx-1 (its first pressure point)
                    const int r0 = q == 0 ? 0 : (q == 1 ? 1 : mx - 1);
                    const int k0 = q == 0 ? n2 - 1 : 0, nk = q == 1 ? n2 : 1;
                    for (int k = 0; k < nk; ++k)
                        for (int i = 0; i < n; ++i) {
                            Dl[(size_t)(r0 + k) * 3 * n + ov + i] = Dh[(size_t)(k0 + k) * n + i];
                            Il[(size_t)(r0 + k) * 3 * n + ov + i] = 0.5 * len[q] * Ih[(size_t)(k0 + k) * n + i];
                        }
                }
                for (int i = 0; i < nvl; ++i) bq[i] = Bl[i] > 0.0 ? 1.0 / Bl[i] : 0.0;
                // end points: with a neighbour, the far end of the neighbour is lumped as an interior interface (1/2);
                // without one, the own end point follows the rule of the non-overlapping operator (wall -> 0)
                if (len[0] > 0.0) bq[0] *= 0.5; else bq[n - 1] *= endfac[(size_t)e * 6 + dd * 2 + 0];
                if (len[2] > 0.0) bq[nvl - 1] *= 0.5; else bq[2 * (n - 1)] *= endfac[(size_t)e * 6 + dd * 2 + 1];
                for (int a = 0; a < mx; ++a)
                    for (int b = 0; b < mx; ++b) {
                        double sa = 0.0, sb = 0.0;
                        for (int i = 0; i < nvl; ++i) {
                            sa += Dl[(size_t)a * 3 * n + i] * bq[i] * Dl[(size_t)b * 3 * n + i];
                            sb += Il[(size_t)a * 3 * n + i] * bq[i] * Il[(size_t)b * 3 * n + i];
                        }
                        Ae[(size_t)a * mx + b] = sa;
                        Be[(size_t)a * mx + b] = sb;
                    }
                for (int side = 0; side < 2; ++side)
                    if (ln[side] == 0.0) {   // no neighbour: the ghost point is decoupled (its right-hand side is zero)
                        const int g = side == 0 ? 0 : mx - 1;
                        for (int b = 0; b < mx; ++b) Ae[(size_t)g * mx + b] = Ae[(size_t)b * mx + g] = Be[(size_t)g * mx + b] = Be[(size_t)b * mx + g] = 0.0;
                        Ae[(size_t)g * mx + g] = 1.0;
                        Be[(size_t)g * mx + g] = 1.0;
                    }
                NLG_CHECK(gen_eig(mx, Ae, Be, Sx, lx) == 0, "pprec_setup: extended 1-D mass matrix not positive definite (element %lld)", (long long)e);
                for (int q = 0; q < mx * mx; ++q) hSx[((size_t)e * 3 + dd) * mx * mx + q] = Sx[q];
                for (int a = 0; a < mx; ++a) hlx[((size_t)e * 3 + dd) * mx + a] = std::max(lx[a], 0.0);
            }
            double mxl = 0.0;
            for (int dd = 0; dd < dim; ++dd) {
                double mm = 0.0;
                for (int a = 0; a < mx; ++a) mm = std::max(mm, hlx[((size_t)e * 3 + dd) * mx + a]);
                mxl += mm;
            }
            dmax = std::max(dmax, mxl);
        }
        for (auto &v : hwq) v = 1.0 / std::sqrt(v);
        NLG_TRY(up(hwq, &P.d_wq));
        P.thrx = 1e-12 * dmax;
        NLG_TRY(up(hSx, &P.d_Sx));
        if (dim == 3) {
            // per-point constants of the extended n^3 grid, the same for every element: number of boundary directions,
            // slot in the face-grouped exchange array, index of the (clamped) pressure point, and which neighbours of
            // an interior point are ghost layers.  Computed per point in the kernel they were ~2/3 of its instructions.
            std::vector<int> tab((size_t)n * n * n);
            for (int c = 0; c < n; ++c)
                for (int b = 0; b < n; ++b)
                    for (int a = 0; a < n; ++a) {
                        const int nb = (a == 0 || a == n - 1) + (b == 0 || b == n - 1) + (c == 0 || c == n - 1);
                        const int a2 = std::min(std::max(a, 1), n - 2) - 1, b2 = std::min(std::max(b, 1), n - 2) - 1,
                                  c2 = std::min(std::max(c, 1), n - 2) - 1;
                        const int q2 = a2 + n2 * (b2 + n2 * c2);
                        const int fl = (a == 1) | ((a == n - 2) << 1) | ((b == 1) << 2) | ((b == n - 2) << 3) | ((c == 1) << 4) | ((c == n - 2) << 5);
                        tab[(size_t)a + n * (b + n * c)] = nb | (fg_slot(n, a, b, c) << 2) | (q2 << 13) | (fl << 23);
                    }
            NLG_TRY(up(tab, &P.d_exttab));
            // face slots of the exchange array W next to every pressure point (k_q1_restrict_local3s)
            {
                const int n2 = m->n2, N = n2 + 2;
                std::vector<int> ws((size_t)n2 * n2 * n2 * 3, -1);
                for (int q = 0; q < n2 * n2 * n2; ++q) {
                    const int a = q % n2, b = (q / n2) % n2, c = q / (n2 * n2);
                    if (a == 0) ws[3 * q + 0] = fg_slot(N, 0, b + 1, c + 1);
                    if (a == n2 - 1) ws[3 * q + 0] = fg_slot(N, N - 1, b + 1, c + 1);
                    if (b == 0) ws[3 * q + 1] = fg_slot(N, a + 1, 0, c + 1);
                    if (b == n2 - 1) ws[3 * q + 1] = fg_slot(N, a + 1, N - 1, c + 1);
                    if (c == 0) ws[3 * q + 2] = fg_slot(N, a + 1, b + 1, 0);
                    if (c == n2 - 1) ws[3 * q + 2] = fg_slot(N, a + 1, b + 1, N - 1);
                }
                NLG_TRY(up(ws, &P.d_wslot));
            }
        }
        NLG_TRY(up(hlx, &P.d_lamx));
        NLG_HIP(hipMalloc(&P.d_W, sizeof(double) * (size_t)m->lvs));
        NLG_HIP(hipMemsetAsync(P.d_W, 0, sizeof(double) * (size_t)m->lvs, st));
        P.overlap = true;
    }
    // ---- 2. coarse space: element vertices, trilinear hats at the GL points
    const int NC = 1 << dim;
    P.ncorner = NC;
    for (int k = 0; k < n2; ++k) P.hat1[k] = 0.5 * (1.0 + o.z2[k]);
    Hat hat;
    for (int k = 0; k < 12; ++k) hat.h1[k] = k < n2 ? P.hat1[k] : 0.0;
    std::vector<int> vg((size_t)E * NC);
    int nvert = 0;
    {
        const int64_t *glo = d->glo_num;
        std::vector<int64_t> lab((size_t)E * NC);
        for (int64_t e = 0; e < E; ++e)
            for (int c = 0; c < NC; ++c) {
                const int i = (c & 1) ? n - 1 : 0, j = (c & 2) ? n - 1 : 0, k = (c & 4) ? n - 1 : 0;
                lab[(size_t)e * NC + c] = glo[e * np1 + i + n * (j + n * k)];
            }
        std::vector<int64_t> u(lab);
        std::sort(u.begin(), u.end());
        u.erase(std::unique(u.begin(), u.end()), u.end());
        nvert = (int)u.size();
        for (size_t q = 0; q < lab.size(); ++q) vg[q] = (int)(std::lower_bound(u.begin(), u.end(), lab[q]) - u.begin());
    }
    P.nvert = nvert;
    // vertex -> incident (element, corner) entries
    std::vector<int> vp((size_t)nvert + 1, 0), vi((size_t)E * NC);
    for (size_t q = 0; q < vg.size(); ++q) vp[vg[q] + 1]++;
    for (int v = 0; v < nvert; ++v) vp[v + 1] += vp[v];
    {
        std::vector<int> pos(vp.begin(), vp.end() - 1);
        for (size_t q = 0; q < vg.size(); ++q) vi[pos[vg[q]]++] = (int)q;
    }
    // element adjacency (conforming hexahedra that share any node share a vertex)
    std::vector<std::vector<int>> adj((size_t)E);
    for (int64_t e = 0; e < E; ++e) {
        auto &a = adj[e];
        for (int c = 0; c < NC; ++c) {
            const int v = vg[(size_t)e * NC + c];
            for (int q = vp[v]; q < vp[v + 1]; ++q) a.push_back(vi[q] / NC);
        }
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    }
    // greedy colouring: two elements of one colour have no common neighbour
    std::vector<int> colour((size_t)E, -1);
    int ncol = 0;
    {
        std::vector<int> mark;
        for (int64_t e = 0; e < E; ++e) {
            mark.assign((size_t)ncol + 1, 0);
            for (int e1 : adj[e])
                for (int e2 : adj[e1])
                    if (colour[e2] >= 0) mark[colour[e2]] = 1;
            int cidx = 0;
            while (cidx < ncol && mark[cidx]) ++cidx;
            colour[e] = cidx;
            if (cidx == ncol) ++ncol;
        }
    }
    // ---- 3. A_c = R_1^T E R_1 by probing E on the device (rank-local: without the halo exchange)
    std::vector<std::unordered_map<int, double>> rows((size_t)nvert);
    {
        int *d_colour = nullptr;
        double *d_t8 = nullptr;
        NLG_TRY(up(colour, &d_colour));
        NLG_HIP(hipMalloc(&d_t8, sizeof(double) * (size_t)E * NC));
        double *pp = sem_scratch2(m, 0), *ep = sem_scratch2(m, 1);
        NLG_CHECK(pp && ep, "pprec_setup: scratch allocation failed");
        const bool halo_was = m->halo.active;
        m->halo.active = false;
        std::vector<double> t8((size_t)E * NC);
        std::vector<int> owner((size_t)E);
        const unsigned gp = (unsigned)((m->lpn + NT - 1) / NT), ge = (unsigned)((E + 3) / 4);
        int rc = 0;
        for (int col = 0; col < ncol && rc == 0; ++col) {
            std::fill(owner.begin(), owner.end(), -1);
            for (int64_t e = 0; e < E; ++e)
                if (colour[e] == col)
                    for (int e1 : adj[e]) owner[e1] = (int)e;
            for (int c = 0; c < NC && rc == 0; ++c) {
                if (dim == 3) {
                    NLG_LAUNCH(k_q1_probe<3>, dim3(gp), dim3(NT), 0, st, E, n2, hat, d_colour, col, c, pp);
                } else {
                    NLG_LAUNCH(k_q1_probe<2>, dim3(gp), dim3(NT), 0, st, E, n2, hat, d_colour, col, c, pp);
                }
                rc = sem_cdabdtp(m, pp, ep);
                if (rc) break;
                if (dim == 3) {
                    NLG_LAUNCH(k_q1_restrict_local<3>, dim3(ge), dim3(NT), 0, st, (const double *)nullptr, E, n2, hat, ep, d_t8, (double *)nullptr, (const double *)nullptr, nlg_pcg_upd{});
                } else {
                    NLG_LAUNCH(k_q1_restrict_local<2>, dim3(ge), dim3(NT), 0, st, (const double *)nullptr, E, n2, hat, ep, d_t8, (double *)nullptr, (const double *)nullptr, nlg_pcg_upd{});
                }
                if (hipMemcpyAsync(t8.data(), d_t8, sizeof(double) * t8.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess) {
                    set_error("pprec_setup: device error while probing the coarse operator");
                    rc = 1;
                    break;
                }
                for (int64_t e1 = 0; e1 < E; ++e1) {
                    const int e0 = owner[e1];
                    if (e0 < 0) continue;
                    const int vcol = vg[(size_t)e0 * NC + c];
                    for (int c1 = 0; c1 < NC; ++c1) {
                        const double val = t8[(size_t)e1 * NC + c1];
                        if (val != 0.0) rows[vg[(size_t)e1 * NC + c1]][vcol] += val;
                    }
                }
            }
        }
        m->halo.active = halo_was;
        hipFree(d_colour);
        hipFree(d_t8);
        if (rc) return rc;
    }
    // symmetrise (the probing is symmetric up to rounding) and build the CSR
    for (int u = 0; u < nvert; ++u)
        for (auto &kv : rows[u])
            if (kv.first > u) {
                auto it = rows[kv.first].find(u);
                const double other = it == rows[kv.first].end() ? kv.second : it->second;
                const double avg = 0.5 * (kv.second + other);
                kv.second = avg;
                rows[kv.first][u] = avg;
            }
    std::vector<int> rp((size_t)nvert + 1, 0), ci;
    std::vector<double> av, dinv((size_t)nvert, 0.0);
    for (int u = 0; u < nvert; ++u) {
        std::vector<std::pair<int, double>> r(rows[u].begin(), rows[u].end());
        std::sort(r.begin(), r.end());
        for (auto &kv : r) {
            ci.push_back(kv.first);
            av.push_back(kv.second);
            if (kv.first == u) dinv[u] = kv.second > 0 ? 1.0 / kv.second : 0.0;
        }
        rp[u + 1] = (int)ci.size();
    }
    // ---- 4. aggregates of vertices (greedy over the vertices that share an element) and the dense inverse on them
    std::vector<int> agg((size_t)nvert, -1);
    std::vector<char> viface((size_t)nvert, 0);   // vertex on a rank boundary
    int na = 0;
    int exact_max = 2048;
    if (const char *ev = getenv("NLG_COARSE_EXACT_MAX")) exact_max = atoi(ev);
    if (nvert <= exact_max) {
        for (int v = 0; v < nvert; ++v) agg[v] = v;
        na = nvert;
    } else {
        auto near = [&](int v, std::vector<int> &out) {
            out.clear();
            for (int q = vp[v]; q < vp[v + 1]; ++q) {
                const int e = vi[q] / NC;
                for (int c = 0; c < NC; ++c) out.push_back(vg[(size_t)e * NC + c]);
            }
            std::sort(out.begin(), out.end());
            out.erase(std::unique(out.begin(), out.end()), out.end());
        };
        std::vector<int> nb;
        // several ranks: the hats of the vertices on a rank boundary are cut there, and the two halves of one hat are
        // coupled as strongly as the operator penalises a jump, which inflates the diagonal the vertex-level Jacobi term
        // divides by.  Such vertices therefore become aggregates of their own (mode 1): the aggregate level is global,
        // knows the coupling between the halves and solves them exactly; their Jacobi term is dropped.
        // 3 x 16^3 elements, pressure iterations per time step: plain 2x2x2 aggregates 41.75, planar 2x2 patches of
        // boundary vertices 21 (mode 2), the same without their Jacobi term 16.5 (mode 3), singletons 12.75 (mode 1);
        // one rank 12.5.
        int iface_mode = 1;
        if (const char *ev = getenv("NLG_IFACE_AGG")) iface_mode = atoi(ev);
        if (!(ctx->distributed() && ctx->nranks > 1)) iface_mode = 0;
        if (iface_mode) {
            for (int idx : m->halo.h_cidx) {
                const int64_t e = idx / np1;
                const int pt = idx % np1;
                for (int c = 0; c < NC; ++c) {
                    const int i = (c & 1) ? n - 1 : 0, j = (c & 2) ? n - 1 : 0, k = (c & 4) ? n - 1 : 0;
                    if (pt == i + n * (j + n * k)) viface[vg[(size_t)e * NC + c]] = 1;
                }
            }
            for (int64_t e = 0; e < E; ++e) {
                int cnt = 0;
                for (int c = 0; c < NC; ++c) {
                    const int v = vg[(size_t)e * NC + c];
                    if (viface[v] && agg[v] < 0) ++cnt;
                }
                if (iface_mode == 1 ? cnt < 1 : cnt < NC / 2) continue;
                for (int c = 0; c < NC; ++c) {
                    const int v = vg[(size_t)e * NC + c];
                    if (viface[v] && agg[v] < 0) agg[v] = iface_mode == 1 ? na++ : na;
                }
                if (iface_mode != 1) ++na;
            }
        }
        // an element whose corners are all free becomes an aggregate (2 x 2 x 2 vertices on structured meshes): the
        // aggregate-level operator stays small enough for a dense inverse (nvert / 8 rows) and, unlike the
        // vertex-plus-all-neighbours aggregates used first (27 vertices), accurate enough for the overlapping local
        // solves (12^3 elements: 30.5 iterations with the 27-vertex aggregates, 21.75 with the exact coarse solve)
        for (int64_t e = 0; e < E; ++e) {
            bool free_all = true;
            for (int c = 0; c < NC; ++c)
                if (agg[vg[(size_t)e * NC + c]] >= 0) free_all = false;
            if (!free_all) continue;
            for (int c = 0; c < NC; ++c) agg[vg[(size_t)e * NC + c]] = na;
            ++na;
        }
        if (iface_mode)   // the far-side corners of the boundary elements
            for (int64_t e = 0; e < E; ++e) {
                int cnt = 0;
                for (int c = 0; c < NC; ++c)
                    if (agg[vg[(size_t)e * NC + c]] < 0) ++cnt;
                if (cnt < NC / 2) continue;
                for (int c = 0; c < NC; ++c)
                    if (agg[vg[(size_t)e * NC + c]] < 0) agg[vg[(size_t)e * NC + c]] = na;
                ++na;
            }
        for (int v = 0; v < nvert; ++v) {
            if (agg[v] >= 0) continue;
            near(v, nb);
            int best = -1;
            double bv = -1.0;
            for (int pass = 0; pass < 2 && best < 0; ++pass)   // first among the vertices of the same kind
                for (int w : nb) {
                    if (w == v || agg[w] < 0 || (pass == 0 && viface[w] != viface[v])) continue;
                    auto it = rows[v].find(w);
                    const double cv = it == rows[v].end() ? 0.0 : std::fabs(it->second);
                    if (cv > bv) {
                        bv = cv;
                        best = agg[w];
                    }
                }
            agg[v] = best >= 0 ? best : na++;
        }
        if (iface_mode == 1 || iface_mode == 3)
            for (int v = 0; v < nvert; ++v)
                if (viface[v]) dinv[v] = 0.0;
    }
    std::vector<double> Acc((size_t)na * na, 0.0);
    for (int u = 0; u < nvert; ++u)
        for (int q = rp[u]; q < rp[u + 1]; ++q) Acc[(size_t)agg[u] * na + agg[ci[q]]] += av[q];
    for (int i = 0; i < na; ++i)
        for (int j = i + 1; j < na; ++j) Acc[(size_t)i * na + j] = Acc[(size_t)j * na + i] = 0.5 * (Acc[(size_t)i * na + j] + Acc[(size_t)j * na + i]);
    const bool multi = ctx->distributed() && ctx->nranks > 1;
    if (!m->has_outflow && !multi) {
        // constant null space (the hats sum to one): shift it so that the inverse acts as the pseudo-inverse on
        // mean-free data.  (Several ranks: the shift is applied to the gathered global operator below.)
        double tr = 0.0;
        for (int i = 0; i < na; ++i) tr += Acc[(size_t)i * na + i];
        const double alpha = tr / na / na;   // the null vector of the aggregated operator is the vector of ones
        for (int i = 0; i < na; ++i)
            for (int j = 0; j < na; ++j) Acc[(size_t)i * na + j] += alpha;
    }
    std::vector<int> ap((size_t)na + 1, 0), am((size_t)nvert);
    for (int v = 0; v < nvert; ++v) ap[agg[v] + 1]++;
    for (int a = 0; a < na; ++a) ap[a + 1] += ap[a];
    {
        std::vector<int> pos(ap.begin(), ap.end() - 1);
        for (int v = 0; v < nvert; ++v) am[pos[agg[v]]++] = v;
    }
    P.na = na;
    NLG_TRY(up(vg, &P.d_vg));
    NLG_TRY(up(vp, &P.d_v2e_p));
    NLG_TRY(up(vi, &P.d_v2e_i));
    NLG_TRY(up(dinv, &P.d_dinv));
    NLG_TRY(up(agg, &P.d_agg));
    NLG_TRY(up(ap, &P.d_ap));
    NLG_TRY(up(am, &P.d_am));
    NLG_HIP(hipMalloc(&P.d_tq, sizeof(double) * (size_t)E * NC));
    for (double **v : {&P.d_rc, &P.d_x}) NLG_HIP(hipMalloc(v, sizeof(double) * (size_t)std::max(nvert, 1)));
    NLG_HIP(hipMalloc(&P.d_xa, sizeof(double) * (size_t)std::max(na, 1)));

    auto gj_inverse = [&](double *dA, int nn) -> int {
        double *rk = nullptr, *ck = nullptr;
        int *bad = nullptr, hbad = 0;
        NLG_HIP(hipMalloc(&rk, sizeof(double) * (size_t)nn));
        NLG_HIP(hipMalloc(&ck, sizeof(double) * (size_t)nn));
        NLG_HIP(hipMalloc(&bad, sizeof(int)));
        NLG_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
        const unsigned gx = (unsigned)((nn + NT - 1) / NT);
        for (int k = 0; k < nn; ++k) {
            NLG_LAUNCH(k_gj_prep, dim3(gx), dim3(NT), 0, st, nn, k, (const double *)dA, rk, ck, bad);
            NLG_LAUNCH(k_gj_update, dim3(gx, (unsigned)nn), dim3(NT), 0, st, nn, k, dA, (const double *)rk, (const double *)ck);
        }
        NLG_HIP(hipGetLastError());
        NLG_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
        NLG_HIP(hipStreamSynchronize(st));
        hipFree(rk);
        hipFree(ck);
        hipFree(bad);
        NLG_CHECK(hbad == 0, "pprec_setup: aggregate operator is not positive definite");
        return 0;
    };

    // large operators (several ranks): Cholesky factorisation and inverse from rocSOLVER — set-up only, a plain dense
    // library call; the two-launches-per-pivot Gauss-Jordan above streams the matrix once per pivot (18k unknowns: 25 s)
    auto spd_inverse_dev = [&](double *dA, int nn) -> int {
        int lib_min = 4096;
        if (const char *ev = getenv("NLG_LIB_INVERSE_MIN")) lib_min = atoi(ev);
        if (nn < lib_min) return gj_inverse(dA, nn);
        rocblas_handle hb = nullptr;
        NLG_CHECK(rocblas_create_handle(&hb) == rocblas_status_success, "pprec_setup: rocblas_create_handle failed");
        rocblas_set_stream(hb, st);
        int *dinfo = nullptr, hinfo[2] = {0, 0};
        NLG_HIP(hipMalloc(&dinfo, 2 * sizeof(int)));
        const rocblas_status s1 = rocsolver_dpotrf(hb, rocblas_fill_lower, nn, dA, nn, dinfo);
        const rocblas_status s2 = s1 == rocblas_status_success ? rocsolver_dpotri(hb, rocblas_fill_lower, nn, dA, nn, dinfo + 1) : s1;
        NLG_HIP(hipMemcpyAsync(hinfo, dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        NLG_HIP(hipStreamSynchronize(st));
        hipFree(dinfo);
        rocblas_destroy_handle(hb);
        NLG_CHECK(s1 == rocblas_status_success && s2 == rocblas_status_success, "pprec_setup: rocSOLVER potrf/potri failed (%d, %d)", (int)s1, (int)s2);
        NLG_CHECK(hinfo[0] == 0 && hinfo[1] == 0, "pprec_setup: aggregate operator is not positive definite (potrf info %d)", hinfo[0]);
        // column-major lower triangle = row-major upper triangle: mirror it
        NLG_LAUNCH(k_mirror_upper, dim3((unsigned)((nn + 255) / 256), (unsigned)nn), dim3(256), 0, st, nn, dA);
        NLG_HIP(hipGetLastError());
        return 0;
    };

    if (!multi) {
        NLG_TRY(up(Acc, &P.d_Ainv));
        NLG_TRY(spd_inverse_dev(P.d_Ainv, na));   // (library Cholesky above NLG_LIB_INVERSE_MIN unknowns, else Gauss-Jordan)
        P.na_max = P.ncols = na;
        NLG_HIP(hipMalloc(&P.d_ra, sizeof(double) * (size_t)std::max(na, 1)));
        P.ready = true;
        return 0;
    }

    // ---- 5. several ranks: ONE aggregate level over all ranks.  The vertex hats are cut at the rank boundaries (the
    // pressure space is discontinuous, so the cut functions are admissible), hence the diagonal block of a rank is the
    // halo-free operator assembled above and only the aggregates that touch an element with a shared node couple to
    // other ranks.  Those columns are probed with the global operator (halo on), one aggregate of one rank at a time;
    // the rows are gathered, the (12k x 12k at 8 x 10^4 elements) operator is inverted redundantly on every rank
    // and each rank keeps its own rows of the inverse.  Per application: one all-gather of na_max doubles.
    // (Rank-local coarse solves — the first version — took 45 pressure iterations on 2 x 512 elements against 14
    // on one rank: the block-Jacobi coarse solve over-corrects the nearly constant-per-rank modes.)
    {
        const int nr = ctx->nranks, me = ctx->rank;
        NLG_CHECK(st == ctx->stream, "pprec_setup: internal (stream)");
        auto gather_host = [&](const std::vector<double> &in, std::vector<double> &out) -> int {
            double *di = nullptr, *d_o = nullptr;
            const size_t c = in.size();
            NLG_HIP(hipMalloc(&di, sizeof(double) * c));
            NLG_HIP(hipMalloc(&d_o, sizeof(double) * c * nr));
            NLG_HIP(hipMemcpyAsync(di, in.data(), sizeof(double) * c, hipMemcpyHostToDevice, st));
            NLG_TRY(allgather_f64(ctx, di, d_o, (int64_t)c));
            out.resize(c * nr);
            NLG_HIP(hipMemcpyAsync(out.data(), d_o, sizeof(double) * c * nr, hipMemcpyDeviceToHost, st));
            NLG_HIP(hipStreamSynchronize(st));
            hipFree(di);
            hipFree(d_o);
            return 0;
        };
        // aggregates whose support holds an element with a node shared with another rank
        std::vector<char> velem((size_t)E, 0), aflag((size_t)na, 0);
        for (int idx : m->halo.h_cidx) velem[idx / np1] = 1;
        for (int64_t e = 0; e < E; ++e)
            if (velem[e])
                for (int c = 0; c < NC; ++c) aflag[agg[vg[(size_t)e * NC + c]]] = 1;
        std::vector<int> ifa;
        for (int a = 0; a < na; ++a)
            if (aflag[a]) ifa.push_back(a);
        double tr = 0.0;
        for (int i = 0; i < na; ++i) tr += Acc[(size_t)i * na + i];
        std::vector<double> info;
        NLG_TRY(gather_host({(double)na, (double)ifa.size(), tr}, info));
        std::vector<int> na_all(nr), ni_all(nr);
        int na_max = 0, ni_max = 0;
        double tr_all = 0.0, nreal = 0.0;
        for (int q = 0; q < nr; ++q) {
            na_all[q] = (int)info[3 * q];
            ni_all[q] = (int)info[3 * q + 1];
            tr_all += info[3 * q + 2];
            nreal += na_all[q];
            na_max = std::max(na_max, na_all[q]);
            ni_max = std::max(ni_max, ni_all[q]);
        }
        const int64_t ntot = (int64_t)nr * na_max;
        NLG_CHECK(ntot <= 60000, "pprec_setup: global aggregate level of %lld unknowns is too large for the dense inverse", (long long)ntot);
        std::vector<double> ifa_pad((size_t)std::max(ni_max, 1), -1.0), ifa_all;
        for (size_t i = 0; i < ifa.size(); ++i) ifa_pad[i] = ifa[i];
        NLG_TRY(gather_host(ifa_pad, ifa_all));
        // this rank's rows: diagonal block from the halo-free assembly, the rest by probing
        double *d_rows = nullptr, *d_full = nullptr;
        {
            std::vector<double> rowsH((size_t)na_max * ntot, 0.0);
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < na; ++j) rowsH[(size_t)i * ntot + (size_t)me * na_max + j] = Acc[(size_t)i * na + j];
            NLG_TRY(up(rowsH, &d_rows));
        }
        NLG_HIP(hipMalloc(&P.d_ra, sizeof(double) * (size_t)na_max));
        NLG_HIP(hipMemsetAsync(P.d_ra, 0, sizeof(double) * (size_t)na_max, st));
        NLG_HIP(hipMalloc(&P.d_rag, sizeof(double) * (size_t)ntot));
        double *pp = sem_scratch2(m, 0), *ep = sem_scratch2(m, 1);
        NLG_CHECK(pp && ep, "pprec_setup: scratch allocation failed");
        const unsigned gp = (unsigned)((m->lpn + NT - 1) / NT), ge = (unsigned)((E + 3) / 4);
        const size_t stride_i = (size_t)std::max(ni_max, 1);
        for (int r = 0; r < nr; ++r)
            for (int i = 0; i < ni_all[r]; ++i) {
                const int a_src = (int)ifa_all[(size_t)r * stride_i + i];
                const int a = me == r ? a_src : -1;
                if (dim == 3) {
                    NLG_LAUNCH(k_agg_probe<3>, dim3(gp), dim3(NT), 0, st, E, n2, hat, (const int *)P.d_vg, (const int *)P.d_agg, a, pp);
                } else {
                    NLG_LAUNCH(k_agg_probe<2>, dim3(gp), dim3(NT), 0, st, E, n2, hat, (const int *)P.d_vg, (const int *)P.d_agg, a, pp);
                }
                NLG_TRY(sem_cdabdtp(m, pp, ep));
                if (me == r) continue;   // own block: already there
                if (dim == 3) {
                    NLG_LAUNCH(k_q1_restrict_local<3>, dim3(ge), dim3(NT), 0, st, (const double *)nullptr, E, n2, hat, ep, P.d_tq, (double *)nullptr, (const double *)nullptr, nlg_pcg_upd{});
                } else {
                    NLG_LAUNCH(k_q1_restrict_local<2>, dim3(ge), dim3(NT), 0, st, (const double *)nullptr, E, n2, hat, ep, P.d_tq, (double *)nullptr, (const double *)nullptr, nlg_pcg_upd{});
                }
                NLG_LAUNCH(k_q1_gather, dim3((nvert + NT - 1) / NT), dim3(NT), 0, st, (const double *)nullptr, nvert, P.d_v2e_p, P.d_v2e_i, P.d_tq, P.d_rc, P.d_dinv, 0.0, P.d_x);
                NLG_LAUNCH(k_agg_restrict, dim3((na + 3) / 4), dim3(NT), 0, st, (const double *)nullptr, na, P.d_ap, P.d_am, P.d_rc, P.d_ra);
                NLG_LAUNCH(k_store_col, dim3((na + 255) / 256), dim3(256), 0, st, na, (const double *)P.d_ra, d_rows, ntot, (int64_t)r * na_max + a_src);
            }
        NLG_HIP(hipGetLastError());
        NLG_HIP(hipMalloc(&d_full, sizeof(double) * (size_t)ntot * ntot));
        NLG_TRY(allgather_f64(ctx, d_rows, d_full, (int64_t)na_max * ntot));
        if (getenv("NLG_PPREC_DEBUG")) {
            std::vector<double> F((size_t)ntot * ntot);
            NLG_HIP(hipMemcpyAsync(F.data(), d_full, sizeof(double) * F.size(), hipMemcpyDeviceToHost, st));
            NLG_HIP(hipStreamSynchronize(st));
            double amax = 0.0, asym = 0.0, rowsum = 0.0, offb = 0.0;
            for (int64_t i = 0; i < ntot; ++i) {
                double rs = 0.0;
                for (int64_t j = 0; j < ntot; ++j) {
                    amax = std::max(amax, std::fabs(F[i * ntot + j]));
                    asym = std::max(asym, std::fabs(F[i * ntot + j] - F[j * ntot + i]));
                    rs += F[i * ntot + j];
                    if (i / na_max != j / na_max) offb = std::max(offb, std::fabs(F[i * ntot + j]));
                }
                rowsum = std::max(rowsum, std::fabs(rs));
            }
            fprintf(stderr, "[pprec rank %d] ntot %lld na %d ni %d  max|A| %.3e  asym %.3e  max|row sum| %.3e  max off-block %.3e\n", me,
                    (long long)ntot, na, (int)ifa.size(), amax, asym, rowsum, offb);
        }
        int *d_na_of = nullptr;
        NLG_TRY(up(na_all, &d_na_of));
        const double alpha = m->has_outflow ? 0.0 : tr_all / nreal / nreal;
        NLG_LAUNCH(k_glob_fix, dim3((unsigned)((ntot + 255) / 256), (unsigned)ntot), dim3(256), 0, st, ntot, na_max, (const int *)d_na_of, alpha, d_full);
        NLG_TRY(spd_inverse_dev(d_full, (int)ntot));
        const int64_t nrow_el = (int64_t)std::max(na, 1) * ntot;
        NLG_HIP(hipMalloc(&P.d_Ainv32, sizeof(float) * (size_t)nrow_el));
        NLG_LAUNCH(k_to_float, dim3((unsigned)((nrow_el + 255) / 256)), dim3(256), 0, st, (int64_t)na * ntot,
                           (const double *)(d_full + (size_t)me * na_max * ntot), P.d_Ainv32);
        NLG_HIP(hipGetLastError());
        NLG_HIP(hipStreamSynchronize(st));
        hipFree(d_full);
        hipFree(d_rows);
        hipFree(d_na_of);
        P.na_max = na_max;
        P.ncols = (int)ntot;
    }
    P.ready = true;
    return 0;
}

// Scratch of the preconditioner for `nl` lanes of a block step: nl copies of the exchange array W, the element-corner
// residuals, the vertex-level and the aggregate-level vectors at constant strides (the kernels reach lane v through blockIdx.y).
// One copy exists after pprec_setup; more are made on first use.
int pprec_reserve_lanes(nlg_mesh *m, int nl) {
    nlg_pprec &P = m->pprec;
    NLG_CHECK(P.ready, "pprec: preconditioner not set up");
    NLG_CHECK(nl >= 1 && nl <= kMaxLanes, "pprec: %d lanes", nl);
    if (nl <= P.lanes_cap) return 0;
    hipStream_t st = m->ctx->stream;
    NLG_HIP(hipStreamSynchronize(st));
    const int NC = 1 << m->dim;
    auto regrow = [&](double **p, int64_t len1, int64_t *stride, bool exact = false) -> int {
        if (!*p) {
            *stride = 0;
            return 0;
        }
        const int64_t sd = exact ? std::max<int64_t>(len1, 1) : round_up(std::max<int64_t>(len1, 1), kAlign);
        double *q = nullptr;
        NLG_HIP(hipMalloc(&q, sizeof(double) * (size_t)(sd * nl)));
        NLG_HIP(hipMemsetAsync(q, 0, sizeof(double) * (size_t)(sd * nl), st));
        NLG_HIP(hipMemcpyAsync(q, *p, sizeof(double) * (size_t)len1, hipMemcpyDeviceToDevice, st));
        NLG_HIP(hipStreamSynchronize(st));
        NLG_HIP(hipFree(*p));
        *p = q;
        *stride = sd;
        return 0;
    };
    NLG_TRY(regrow(&P.d_W, m->lvs, &P.lW));
    NLG_TRY(regrow(&P.d_tq, m->E * NC, &P.lt));
    int64_t lv2 = 0, la2 = 0;
    NLG_TRY(regrow(&P.d_rc, P.nvert, &P.lv));
    NLG_TRY(regrow(&P.d_x, P.nvert, &lv2));
    const bool glob = P.ncols != P.na;                         // several ranks: d_ra holds na_max entries (zero-padded)
    NLG_TRY(regrow(&P.d_ra, glob ? P.na_max : P.na, &P.la, glob));   // several ranks: the lanes contiguous, [lane][na_max]
    NLG_TRY(regrow(&P.d_xa, P.na, &la2));
    if (glob) {
        // all-gathered aggregate residuals [rank][lane][na_max]: the lanes of a rank stay together in one all-gather
        NLG_HIP(hipFree(P.d_rag));
        NLG_HIP(hipMalloc(&P.d_rag, sizeof(double) * (size_t)P.ncols * nl));
    }
    NLG_CHECK(lv2 == P.lv && (la2 == P.la || glob), "pprec: inconsistent lane strides");
    P.la_x = la2;
    P.lanes_cap = nl;
    return 0;
}

// Coarse part of M^-1 r on `st`: xc[v] = omega dinv[v] (R_1^T r)[v] and P.d_xa = aggregate-level solve; pprec_fine
// adds the two while prolonging.  nl > 1: the lanes of a block step in the same launches (r, flag and the fused PCG update of
// lane v sit v * ld doubles behind the given pointers).
int pprec_coarse(nlg_mesh *m, hipStream_t st, const double *flag, const double *r, const double **xc, bool overlap,
                 const nlg_pcg_upd *upd, int nl, int64_t ld) {
    const nlg_pcg_upd uu = upd ? *upd : nlg_pcg_upd{};
    double *rw = const_cast<double *>(r);   // written only when the PCG update rides along
    nlg_pprec &P = m->pprec;
    NLG_CHECK(P.ready, "pprec: preconditioner not set up");
    NLG_TRY(pprec_reserve_lanes(m, nl));
    const int64_t E = m->E;
    const int nv = P.nvert;
    const double om = P.na == nv ? 0.0 : 0.7;   // exact coarse solve when every vertex is its own aggregate
    Hat hat;
    for (int k = 0; k < 12; ++k) hat.h1[k] = P.hat1[k];
    NLG_CHECK(!overlap || P.overlap, "pprec: the overlapping variant is not set up for this mesh");
    double *Wp = overlap ? P.d_W : (double *)nullptr;
    const dim3 gq((unsigned)((E + 3) / 4), (unsigned)nl);
    if (m->dim == 3) {
        if (P.d_wslot && m->n2 == 6)
            NLG_LAUNCH(k_q1_restrict_local3s<6>, gq, dim3(NT), 0, st, flag, E, hat, rw, P.d_tq, Wp, (const double *)P.d_wq, (const int *)P.d_wslot, uu, ld, P.lt, P.lW);
        else if (P.d_wslot && m->n2 == 8)
            NLG_LAUNCH(k_q1_restrict_local3s<8>, gq, dim3(NT), 0, st, flag, E, hat, rw, P.d_tq, Wp, (const double *)P.d_wq, (const int *)P.d_wslot, uu, ld, P.lt, P.lW);
        else if (P.d_wslot && m->n2 == 10)
            NLG_LAUNCH(k_q1_restrict_local3s<10>, gq, dim3(NT), 0, st, flag, E, hat, rw, P.d_tq, Wp, (const double *)P.d_wq, (const int *)P.d_wslot, uu, ld, P.lt, P.lW);
        else
            NLG_LAUNCH(k_q1_restrict_local<3>, gq, dim3(NT), 0, st, flag, E, m->n2, hat, rw, P.d_tq, Wp, (const double *)P.d_wq, uu, ld, P.lt, P.lW);
    } else {
        NLG_LAUNCH(k_q1_restrict_local<2>, gq, dim3(NT), 0, st, flag, E, m->n2, hat, rw, P.d_tq, Wp, (const double *)P.d_wq, uu, ld, P.lt, P.lW);
    }
    // 3-D overlapping variant: the rest of the coarse chain rides in the launches of the fine level (pprec_fine, merged launches)
    static const bool hfuse = !(getenv("NLG_HFUSE") && atoi(getenv("NLG_HFUSE")) == 0);
    P.coarse_pending = hfuse && overlap && m->gs.npairs > 0 && (m->dim == 2 ? nl == 1 : m->gs.d_indices_fg != nullptr);
    if (P.coarse_pending) {
        NLG_HIP(hipGetLastError());
        *xc = P.d_x;
        return 0;
    }
    NLG_LAUNCH(k_q1_gather, dim3((nv + NT - 1) / NT, nl), dim3(NT), 0, st, flag, nv, P.d_v2e_p, P.d_v2e_i, P.d_tq, P.d_rc, P.d_dinv, om, P.d_x, ld, P.lt, P.lv);
    NLG_LAUNCH(k_agg_restrict, dim3((P.na + 3) / 4, nl), dim3(NT), 0, st, flag, P.na, P.d_ap, P.d_am, P.d_rc, P.d_ra, ld, P.lv, P.la);
    const double *ra = P.d_ra;
    int na_max = 0;
    if (P.ncols != P.na) {   // several ranks: the aggregate level is global; ONE all-gather carries the lanes of every rank
        NLG_CHECK(st == m->ctx->stream, "pprec: the global aggregate level runs on the context's stream");
        NLG_TRY(allgather_f64(m->ctx, P.d_ra, P.d_rag, (int64_t)P.na_max * nl));
        ra = P.d_rag;
        na_max = nl > 1 ? P.na_max : 0;   // (one lane: the gathered array is the plain column vector)
    }
    const dim3 gg((unsigned)((P.na + 3) / 4), (unsigned)nl);
    if (P.d_Ainv32)
        NLG_LAUNCH(k_dense_gemv<float>, gg, dim3(NT), 0, st, flag, P.na, P.ncols, (const float *)P.d_Ainv32, ra, P.d_xa, ld, P.ncols != P.na ? P.la_x : P.la, na_max, nl);
    else
        NLG_LAUNCH(k_dense_gemv<double>, gg, dim3(NT), 0, st, flag, P.na, P.ncols, (const double *)P.d_Ainv, ra, P.d_xa, ld, P.ncols != P.na ? P.la_x : P.la, na_max, nl);
    NLG_HIP(hipGetLastError());
    *xc = P.d_x;   // the Jacobi term; pprec_fine adds xa[agg[v]] while prolonging
    return 0;
}

// ghost layers of face neighbours on other ranks: the copies of a face on a rank boundary are summed by the halo
// exchange exactly as the pairs kernel sums the two local copies of an interior face (edge and corner slots of W
// are never written and travel as zeros)
static int overlap_halo(nlg_mesh *m, hipStream_t st, bool face_grouped, int nl) {
    ++g_collectives;
    if (!m->halo.active) return 0;
    NLG_CHECK(st == m->ctx->stream, "pprec: the overlap exchange across ranks runs on the context's stream");
    double *f1[1] = {m->pprec.d_W};
    return halo_exchange(m, f1, 1, face_grouped, nl, m->pprec.lW);
}

// Fine part: z = sum_e R_e^T Etilde_e^-1 R_e r (+ R_1 xc when xc is given), launched on `st`.
int pprec_fine(nlg_mesh *m, hipStream_t st, const double *flag, const double *r, const double *xc, double *z,
               double *rz_part, bool overlap, int nl, int64_t ld) {
    nlg_pprec &P = m->pprec;
    NLG_CHECK(P.ready, "pprec: preconditioner not set up");
    NLG_TRY(pprec_reserve_lanes(m, nl));
    const int64_t E = m->E;
    Hat hat;
    for (int k = 0; k < 12; ++k) hat.h1[k] = P.hat1[k];
    const int *vg = P.d_vg;
    if (nl > 1 && !(overlap && m->dim == 3)) {
        // kernels without the lane dimension (2-D, the variant without overlap): lane by lane, on the lane's copies of the scratch
        NLG_CHECK(xc == nullptr || xc == P.d_x, "pprec_fine: foreign coarse vector with several lanes");
        nlg_pprec keep = P;
        int rc = 0;
        for (int v = 0; v < nl && rc == 0; ++v) {
            P.d_W = keep.d_W ? keep.d_W + v * keep.lW : nullptr;
            P.d_xa = keep.d_xa + v * (keep.ncols != keep.na ? keep.la_x : keep.la);
            rc = pprec_fine(m, st, flag ? flag + v * ld : nullptr, r + v * ld, xc ? keep.d_x + v * keep.lv : nullptr, z + v * ld,
                            rz_part ? rz_part + v * ld : nullptr, overlap, 1, 0);
        }
        P.d_W = keep.d_W, P.d_xa = keep.d_xa;
        return rc;
    }
    const int64_t la_x = P.ncols != P.na ? P.la_x : P.la;
    if (overlap && m->dim == 2) {
        NLG_CHECK(P.overlap, "pprec: the overlapping variant is not set up for this mesh");
        const unsigned gb = (unsigned)((E + 3) / 4);
        // merged launches as in 3-D (below): the vertex gather rides with the first pairs-only gather-scatter (natural layout here),
        // the aggregate restriction with the local solves, the dense solve with the second gather-scatter
        const bool fused = P.coarse_pending;
        P.coarse_pending = false;
        const int nbp = (int)((m->gs.npairs + NT - 1) / NT);
        const int nv = P.nvert;
        const bool glob = P.ncols != P.na;
        AggArgs ag = {0, nullptr, nullptr, nullptr, nullptr, 0, 0};
        if (fused) {
            const GatherArgs gg = {nv, P.d_v2e_p, P.d_v2e_i, P.d_tq, P.d_rc, P.d_dinv, P.na == nv ? 0.0 : 0.7, P.d_x, P.lt, P.lv};
            NLG_LAUNCH(k_pairs_gather, dim3((unsigned)(nbp + (nv + NT - 1) / NT)), dim3(NT), 0, st, nbp, (const int *)m->gs.d_indices, m->gs.npairs, P.d_W,
                       (int64_t)0, flag, (int64_t)0, gg);
            ag = AggArgs{P.na, P.d_ap, P.d_am, P.d_rc, P.d_ra, P.lv, P.la};
        } else {
            NLG_TRY(sem_gs_pairs(m, P.d_W, flag));
        }
        NLG_TRY(overlap_halo(m, st, false, 1));
        const unsigned nb_agg = fused ? (unsigned)((P.na + 3) / 4) : 0u;
#define FX2_CASE(N_)                                                                                                  \
    case N_:                                                                                                          \
        NLG_LAUNCH((k_fdm_ext2<N_>), dim3(gb + nb_agg), dim3(NT), 0, st, flag, E, P.d_Sx, P.d_lamx, P.thrx, r, P.d_wq, P.d_W, z, (int)gb, ag); \
        break;
        switch (m->n) {
            FX2_CASE(4) FX2_CASE(5) FX2_CASE(6) FX2_CASE(7) FX2_CASE(8)
            default: set_error("pprec: overlapping variant built for lx1 = 4..8, got %d", m->n); return 1;
        }
#undef FX2_CASE
        if (fused) {
            const double *ra = P.d_ra;
            if (glob) {
                NLG_CHECK(st == m->ctx->stream, "pprec: the global aggregate level runs on the context's stream");
                NLG_TRY(allgather_f64(m->ctx, P.d_ra, P.d_rag, (int64_t)P.na_max));
                ra = P.d_rag;
            }
            const dim3 gc((unsigned)(nbp + (P.na + 3) / 4));
            if (P.d_Ainv32) {
                const GemvArgs<float> gv = {P.na, P.ncols, (const float *)P.d_Ainv32, ra, P.d_xa, 0, 0, 1};
                NLG_LAUNCH(k_pairs_gemv<float>, gc, dim3(NT), 0, st, nbp, (const int *)m->gs.d_indices, m->gs.npairs, P.d_W, (int64_t)0, flag, (int64_t)0, gv);
            } else {
                const GemvArgs<double> gv = {P.na, P.ncols, (const double *)P.d_Ainv, ra, P.d_xa, 0, 0, 1};
                NLG_LAUNCH(k_pairs_gemv<double>, gc, dim3(NT), 0, st, nbp, (const int *)m->gs.d_indices, m->gs.npairs, P.d_W, (int64_t)0, flag, (int64_t)0, gv);
            }
        } else {
            NLG_TRY(sem_gs_pairs(m, P.d_W, flag));
        }
        NLG_TRY(overlap_halo(m, st, false, 1));
        const unsigned gf = (unsigned)((E * m->np2 + NT - 1) / NT);
#define FF2_CASE(N_)                                                                                                  \
    case N_:                                                                                                          \
        NLG_LAUNCH((k_sch_finish2<N_>), dim3(gf), dim3(NT), 0, st, flag, E, P.d_W, r, P.d_wq, xc, P.d_xa, P.d_agg, vg, hat, z, rz_part); \
        break;
        switch (m->n) {
            FF2_CASE(4) FF2_CASE(5) FF2_CASE(6) FF2_CASE(7) FF2_CASE(8)
            default: break;
        }
#undef FF2_CASE
        NLG_HIP(hipGetLastError());
        return 0;
    }
    if (overlap) {
        // pprec_coarse has packed the adjacent layers into P.d_W (same stream)
        NLG_CHECK(P.overlap && m->dim == 3, "pprec: the overlapping variant is not set up for this mesh");
        const dim3 gb((unsigned)((E + 3) / 4), (unsigned)nl);
        const bool fused = P.coarse_pending;   // set by pprec_coarse: the coarse chain is still to run
        P.coarse_pending = false;
        const int nbp = (int)((m->gs.npairs + NT - 1) / NT);
        const int nv = P.nvert;
        const bool glob = P.ncols != P.na;
        AggArgs ag = {0, nullptr, nullptr, nullptr, nullptr, 0, 0};
        int nb_agg = 0;
        if (fused) {
            // (a) pairs-only gather-scatter of W + vertex gather of the element-corner residuals
            const GatherArgs gg = {nv, P.d_v2e_p, P.d_v2e_i, P.d_tq, P.d_rc, P.d_dinv, P.na == nv ? 0.0 : 0.7, P.d_x, P.lt, P.lv};
            NLG_LAUNCH(k_pairs_gather, dim3((unsigned)(nbp + (nv + NT - 1) / NT), (unsigned)nl), dim3(NT), 0, st, nbp, (const int *)m->gs.d_indices_fg,
                       m->gs.npairs, P.d_W, P.lW, flag, ld, gg);
            ag = AggArgs{P.na, P.d_ap, P.d_am, P.d_rc, P.d_ra, P.lv, P.la};
        } else {
            NLG_TRY(sem_gs_pairs_fg(m, P.d_W, flag, nl, P.lW, ld));
        }
        NLG_TRY(overlap_halo(m, st, true, nl));
#define FX_CASE(N_)                                                                                                   \
    case N_: {                                                                                                        \
        constexpr int WPE_ = (N_ * N_ + 63) / 64;   /* one column per lane: 1 wave up to lx1 = 8, 2 at 9 / 10, 3 at 12 */ \
        nb_agg = fused ? (P.na + WPE_ - 1) / WPE_ : 0;   /* (b) + aggregate restriction: one wave per aggregate */     \
        NLG_LAUNCH((k_fdm_ext<N_, 1, WPE_>), dim3((unsigned)(E + nb_agg), (unsigned)nl), dim3(64 * WPE_), 0, st, flag, E, P.d_Sx, P.d_lamx, P.thrx, r, P.d_wq, P.d_W, z, (const int *)P.d_exttab, ld, P.lW, (int)E, ag); \
    } break;
        // lx1 = 8: the six transforms on the matrix pipe (k_fdm_ext_mfma8); NLG_FDM_MFMA=0 selects the vector-pipe kernel (A/B runs)
        static const bool fdm_mfma = !(getenv("NLG_FDM_MFMA") && atoi(getenv("NLG_FDM_MFMA")) == 0);
        if (m->n == 8 && fdm_mfma) {
            nb_agg = fused ? P.na : 0;
            NLG_LAUNCH(k_fdm_ext_mfma8, dim3((unsigned)(E + nb_agg), (unsigned)nl), dim3(64), 0, st, flag, E, P.d_Sx, P.d_lamx, P.thrx, r, P.d_wq, P.d_W, z,
                       (const int *)P.d_exttab, ld, P.lW, (int)E, ag);
        } else
        switch (m->n) {
            FX_CASE(4) FX_CASE(5) FX_CASE(6) FX_CASE(7) FX_CASE(8) FX_CASE(9) FX_CASE(10) FX_CASE(12)
            default: set_error("pprec: overlapping variant built for lx1 = 4..10 and 12, got %d", m->n); return 1;
        }
#undef FX_CASE
        if (fused) {
            const double *ra = P.d_ra;
            int na_max = 0;
            if (glob) {   // several ranks: the aggregate level is global; ONE all-gather carries the lanes of every rank
                NLG_CHECK(st == m->ctx->stream, "pprec: the global aggregate level runs on the context's stream");
                NLG_TRY(allgather_f64(m->ctx, P.d_ra, P.d_rag, (int64_t)P.na_max * nl));
                ra = P.d_rag;
                na_max = nl > 1 ? P.na_max : 0;
            }
            // (c) second pairs-only gather-scatter of W + dense aggregate solve
            const dim3 gc((unsigned)(nbp + (P.na + 3) / 4), (unsigned)nl);
            if (P.d_Ainv32) {
                const GemvArgs<float> gv = {P.na, P.ncols, (const float *)P.d_Ainv32, ra, P.d_xa, glob ? P.la_x : P.la, na_max, nl};
                NLG_LAUNCH(k_pairs_gemv<float>, gc, dim3(NT), 0, st, nbp, (const int *)m->gs.d_indices_fg, m->gs.npairs, P.d_W, P.lW, flag, ld, gv);
            } else {
                const GemvArgs<double> gv = {P.na, P.ncols, (const double *)P.d_Ainv, ra, P.d_xa, glob ? P.la_x : P.la, na_max, nl};
                NLG_LAUNCH(k_pairs_gemv<double>, gc, dim3(NT), 0, st, nbp, (const int *)m->gs.d_indices_fg, m->gs.npairs, P.d_W, P.lW, flag, ld, gv);
            }
        } else {
            NLG_TRY(sem_gs_pairs_fg(m, P.d_W, flag, nl, P.lW, ld));
        }
        NLG_TRY(overlap_halo(m, st, true, nl));
#define FF_CASE(N_)                                                                                                   \
    case N_:                                                                                                          \
        NLG_LAUNCH((k_sch_finish<N_>), gb, dim3(NT), 0, st, flag, E, P.d_W, r, P.d_wq, xc, P.d_xa, P.d_agg, vg, hat, z, rz_part, (const int *)P.d_wslot, ld, P.lW, P.lv, la_x); \
        break;
        switch (m->n) {
            FF_CASE(4) FF_CASE(5) FF_CASE(6) FF_CASE(7) FF_CASE(8) FF_CASE(9) FF_CASE(10) FF_CASE(12)
            default: break;
        }
#undef FF_CASE
        NLG_HIP(hipGetLastError());
        return 0;
    }
#define FDM_CASE(N_)                                                                                                  \
    if (m->dim == 3)                                                                                                  \
        NLG_LAUNCH((k_fdm<N_ - 2, 3>), dim3((unsigned)((E + 3) / 4)), dim3(NT), 0, st, flag, E, P.d_S, P.d_invden, r, xc, P.d_xa, P.d_agg, vg, hat, z, rz_part); \
    else                                                                                                              \
        NLG_LAUNCH((k_fdm<N_ - 2, 2>), dim3((unsigned)((E + 3) / 4)), dim3(NT), 0, st, flag, E, P.d_S, P.d_invden, r, xc, P.d_xa, P.d_agg, vg, hat, z, rz_part);
    switch (m->n) {
        case 4: FDM_CASE(4); break;
        case 5: FDM_CASE(5); break;
        case 6: FDM_CASE(6); break;
        case 7: FDM_CASE(7); break;
        case 8: FDM_CASE(8); break;
        case 9: FDM_CASE(9); break;
        case 10: FDM_CASE(10); break;
        case 12: FDM_CASE(12); break;
        default: set_error("pprec: unsupported lx1 = %d", m->n); return 1;
    }
#undef FDM_CASE
    NLG_HIP(hipGetLastError());
    return 0;
}

void pprec_free(nlg_mesh *m) {
    nlg_pprec &P = m->pprec;
    double *dp[] = {P.d_S, P.d_invden, P.d_dinv, P.d_Ainv, P.d_rc, P.d_x, P.d_ra, P.d_xa, P.d_rag, P.d_tq, P.d_Sx, P.d_lamx, P.d_W, P.d_wq};
    for (double *p : dp)
        if (p) hipFree(p);
    int *ip[] = {P.d_agg, P.d_ap, P.d_am, P.d_vg, P.d_v2e_p, P.d_v2e_i, P.d_exttab, P.d_wslot};
    for (int *p : ip)
        if (p) hipFree(p);
    if (P.d_Ainv32) hipFree(P.d_Ainv32);
    P = nlg_pprec();
}

}  // namespace nlg
