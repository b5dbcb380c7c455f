/* ORACLE-side CPU port (test / measurement infrastructure, never linked into the product): C + OpenMP restatement of
 * the operators whose unit times make up bench.py's `cpu_baseline` (SURVEY.md 8d "CPU baseline beside it"): the
 * element-loop Helmholtz operator, the consistent Poisson operator, the dealiased linearised convection, the
 * gather-scatter and the per-vector dot / axpby of the reference's vector type.  Same formulas as oracle/sem.py
 * (checked against it in tests/test_cpu_oracle.py); structure follows the reference's element loops
 * (/root/reference/src/linops/neklab_linops.f90:268-426, src/vectors/real_vectors.f90:125-233).  3-D only. */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

int nl_threads(void) { return omp_get_max_threads(); }
void nl_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

/* out[c][b][a'] = sum_a M[a'][a] in[c][b][a]  (x fastest), M is mo x mi row-major */
static void apx(const double *M, int mo, int mi, const double *in, double *out, int ny, int nz) {
    for (int q = 0; q < ny * nz; ++q)
        for (int o = 0; o < mo; ++o) {
            double s = 0.0;
            for (int i = 0; i < mi; ++i) s += M[o * mi + i] * in[q * mi + i];
            out[q * mo + o] = s;
        }
}
static void apy(const double *M, int mo, int mi, const double *in, double *out, int nx, int nz) {
    for (int k = 0; k < nz; ++k)
        for (int o = 0; o < mo; ++o)
            for (int a = 0; a < nx; ++a) {
                double s = 0.0;
                for (int j = 0; j < mi; ++j) s += M[o * mi + j] * in[a + nx * (j + mi * k)];
                out[a + nx * (o + mo * k)] = s;
            }
}
static void apz(const double *M, int mo, int mi, const double *in, double *out, int nx, int ny) {
    for (int o = 0; o < mo; ++o)
        for (int q = 0; q < nx * ny; ++q) {
            double s = 0.0;
            for (int k = 0; k < mi; ++k) s += M[o * mi + k] * in[q + nx * ny * k];
            out[q + nx * ny * o] = s;
        }
}
static void transpose(const double *M, int r, int c, double *T) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) T[j * r + i] = M[i * c + j];
}
/* three-direction tensor product: out = (Mz x My x Mx) in, matrices mo x mi */
static void tens3(const double *Mx, const double *My, const double *Mz, int mo, int mi, const double *in, double *out, double *t1, double *t2) {
    apx(Mx, mo, mi, in, t1, mi, mi);
    apy(My, mo, mi, t1, t2, mo, mi);
    apz(Mz, mo, mi, t2, out, mo, mo);
}

/* w = h1 D^T G D u + h2 B u, element-local, one field */
void nl_axhelm(long E, int n, const double *D, const double *const *G, const double *bm1, const double *u, double *w, double h1, double h2) {
    const int np = n * n * n;
    double *Dt = malloc(sizeof(double) * n * n);
    transpose(D, n, n, Dt);
#pragma omp parallel
    {
        double *ur = malloc(sizeof(double) * np * 6), *us = ur + np, *ut = us + np, *a = ut + np, *b = a + np, *c = b + np;
#pragma omp for schedule(static)
        for (long e = 0; e < E; ++e) {
            const double *ue = u + e * np;
            apx(D, n, n, ue, ur, n, n);
            apy(D, n, n, ue, us, n, n);
            apz(D, n, n, ue, ut, n, n);
            for (int q = 0; q < np; ++q) {
                const long g = e * np + q;
                const double r = ur[q], s = us[q], t = ut[q];
                a[q] = G[0][g] * r + G[1][g] * s + G[2][g] * t;
                b[q] = G[1][g] * r + G[3][g] * s + G[4][g] * t;
                c[q] = G[2][g] * r + G[4][g] * s + G[5][g] * t;
            }
            apx(Dt, n, n, a, ur, n, n);
            apy(Dt, n, n, b, us, n, n);
            apz(Dt, n, n, c, ut, n, n);
            for (int q = 0; q < np; ++q) w[e * np + q] = h1 * (ur[q] + us[q] + ut[q]) + h2 * bm1[e * np + q] * ue[q];
        }
        free(ur);
    }
    free(Dt);
}

/* QQ^T in place: groups of local copies (CSR) */
void nl_gs(long ngroups, const long *off, const long *idx, double *f) {
#pragma omp parallel for schedule(static)
    for (long g = 0; g < ngroups; ++g) {
        double s = 0.0;
        for (long q = off[g]; q < off[g + 1]; ++q) s += f[idx[q]];
        for (long q = off[g]; q < off[g + 1]; ++q) f[idx[q]] = s;
    }
}

/* w_i = sum_j T_j^T (rst2w[j][i] o p),  T_j = D12 along j, I12 otherwise (n2 x n matrices) */
void nl_opgradt(long E, int n, int n2, const double *I12, const double *D12, const double *const *g, const double *p, double *const *w) {
    const int np1 = n * n * n, np2 = n2 * n2 * n2;
    double *It = malloc(sizeof(double) * n * n2 * 2), *Dt = It + n * n2;
    transpose(I12, n2, n, It);
    transpose(D12, n2, n, Dt);
#pragma omp parallel
    {
        double *q = malloc(sizeof(double) * (np2 + 3 * np1)), *t1 = q + np2, *t2 = t1 + np1, *o = t2 + np1;
#pragma omp for schedule(static)
        for (long e = 0; e < E; ++e)
            for (int i = 0; i < 3; ++i) {
                double *wi = w[i] + e * np1;
                for (int r = 0; r < np1; ++r) wi[r] = 0.0;
                for (int j = 0; j < 3; ++j) {
                    const double *gj = g[j * 3 + i] + e * np2;
                    for (int r = 0; r < np2; ++r) q[r] = gj[r] * p[e * np2 + r];
                    tens3(j == 0 ? Dt : It, j == 1 ? Dt : It, j == 2 ? Dt : It, n, n2, q, o, t1, t2);
                    for (int r = 0; r < np1; ++r) wi[r] += o[r];
                }
            }
        free(q);
    }
    free(It);
}

/* out = sum_i sum_j rst2w[j][i] o (T_j u_i) */
void nl_opdiv(long E, int n, int n2, const double *I12, const double *D12, const double *const *g, const double *const *u, double *out) {
    const int np1 = n * n * n, np2 = n2 * n2 * n2;
#pragma omp parallel
    {
        double *t1 = malloc(sizeof(double) * 3 * np1), *t2 = t1 + np1, *o = t2 + np1;
#pragma omp for schedule(static)
        for (long e = 0; e < E; ++e) {
            double *oe = out + e * np2;
            for (int r = 0; r < np2; ++r) oe[r] = 0.0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    tens3(j == 0 ? D12 : I12, j == 1 ? D12 : I12, j == 2 ? D12 : I12, n2, n, u[i] + e * np1, o, t1, t2);
                    const double *gj = g[j * 3 + i] + e * np2;
                    for (int r = 0; r < np2; ++r) oe[r] += gj[r] * o[r];
                }
        }
        free(t1);
    }
}

/* weak dealiased linearised convection (direct): out_i = J^T [ sum_j Ur_j du_i/dr_j + ur_j dU_i/dr_j ],
 * Ur_j = sum_m rstdw[j][m] Uf_m, ur_j likewise (oracle/sem.py lns_conv_weak) */
void nl_conv(long E, int n, int nd, const double *Jd, const double *DJd, const double *const *rd, const double *const *U, const double *const *u,
             double *const *out) {
    const int np1 = n * n * n, npd = nd * nd * nd;
    double *Jt = malloc(sizeof(double) * n * nd);
    transpose(Jd, nd, n, Jt);
#pragma omp parallel
    {
        double *buf = malloc(sizeof(double) * npd * 30);
        double *Uf[3], *uf[3], *dU[3][3], *du[3][3], *t1 = buf + 24 * npd, *t2 = t1 + npd, *acc = t2 + npd, *Urj = acc + npd, *urj = Urj + npd;
        for (int m = 0; m < 3; ++m) {
            Uf[m] = buf + m * npd;
            uf[m] = buf + (3 + m) * npd;
            for (int j = 0; j < 3; ++j) {
                dU[m][j] = buf + (6 + 3 * m + j) * npd;
                du[m][j] = buf + (15 + 3 * m + j) * npd;
            }
        }
#pragma omp for schedule(static)
        for (long e = 0; e < E; ++e) {
            for (int m = 0; m < 3; ++m) {
                tens3(Jd, Jd, Jd, nd, n, U[m] + e * np1, Uf[m], t1, t2);
                tens3(Jd, Jd, Jd, nd, n, u[m] + e * np1, uf[m], t1, t2);
                for (int j = 0; j < 3; ++j) {
                    tens3(j == 0 ? DJd : Jd, j == 1 ? DJd : Jd, j == 2 ? DJd : Jd, nd, n, U[m] + e * np1, dU[m][j], t1, t2);
                    tens3(j == 0 ? DJd : Jd, j == 1 ? DJd : Jd, j == 2 ? DJd : Jd, nd, n, u[m] + e * np1, du[m][j], t1, t2);
                }
            }
            for (int i = 0; i < 3; ++i) {
                for (int q = 0; q < npd; ++q) acc[q] = 0.0;
                for (int j = 0; j < 3; ++j) {
                    for (int q = 0; q < npd; ++q) {
                        const long g = e * npd + q;
                        Urj[q] = rd[j * 3 + 0][g] * Uf[0][q] + rd[j * 3 + 1][g] * Uf[1][q] + rd[j * 3 + 2][g] * Uf[2][q];
                        urj[q] = rd[j * 3 + 0][g] * uf[0][q] + rd[j * 3 + 1][g] * uf[1][q] + rd[j * 3 + 2][g] * uf[2][q];
                    }
                    for (int q = 0; q < npd; ++q) acc[q] += Urj[q] * du[i][j][q] + urj[q] * dU[i][j][q];
                }
                /* project back: out = (Jt x Jt x Jt) acc, matrices n x nd */
                apx(Jt, n, nd, acc, t1, nd, nd);
                apy(Jt, n, nd, t1, t2, n, nd);
                apz(Jt, n, nd, t2, out[i] + e * np1, n, n);
            }
        }
        free(buf);
    }
    free(Jt);
}

/* reference vector primitives: mass-weighted dot of one component (glsc3), two-sweep axpby */
double nl_glsc3(long n, const double *a, const double *b, const double *w) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (long i = 0; i < n; ++i) s += a[i] * b[i] * w[i];
    return s;
}
void nl_axpby(long n, double alpha, const double *x, double beta, double *y) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) y[i] *= beta;   /* cmult sweep */
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) y[i] += alpha * x[i];   /* add2s2 sweep */
}
/* vector work of one PCG iteration on one field: x += a p ; r -= a w ; z = m r ; (r,z) ; (p, w) ; p = z + b p */
double nl_cgvec(long n, double *x, double *r, double *z, double *p, const double *w, const double *m, const double *wt) {
    double rz = 0.0, pw = 0.0;
#pragma omp parallel for reduction(+ : rz, pw) schedule(static)
    for (long i = 0; i < n; ++i) {
        x[i] += 0.1 * p[i];
        r[i] -= 0.1 * w[i];
        z[i] = m[i] * r[i];
        rz += r[i] * z[i] * wt[i];
        pw += p[i] * w[i] * wt[i];
    }
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) p[i] = z[i] + 0.5 * p[i];
    return rz + pw;
}

/* ---- one real time step: the two PCG solvers of oracle/lns.py (pcg_helm, pcg_E) on the operators above, every vector pass an
 * OpenMP loop, so that bench.py's cpu_baseline can TIME a whole time step instead of composing unit times (oracle/cpu_step.py drives
 * them; tests/test_cpu_oracle.py checks the step against ExptA.advance).  The loop bodies follow the numpy twin line by line. ---- */

/* y = (acc ? y : 0) + scale o sum_j c[j] x[j]   (scale may be null) */
void nl_lincomb(long n, int k, const double *const *x, const double *c, const double *scale, double *y, int acc) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += c[j] * x[j][i];
        if (scale) s *= scale[i];
        y[i] = acc ? y[i] + s : s;
    }
}
/* y = m o (a + b - c)  (the residual of the tentative velocity: mask (rhs + grad^T p - H u)); any of b, c may be null */
void nl_residual(long n, const double *m, const double *a, const double *b, const double *c, double *y) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) y[i] = m[i] * (a[i] + (b ? b[i] : 0.0) - (c ? c[i] : 0.0));
}
/* y = a + s * (m o b)  (m may be null) */
void nl_add_scaled(long n, const double *a, double s, const double *m, const double *b, double *y) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) y[i] = a[i] + s * (m ? m[i] * b[i] : b[i]);
}
double nl_sum(long n, const double *a) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (long i = 0; i < n; ++i) s += a[i];
    return s;
}
void nl_shift(long n, double *a, double s) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) a[i] -= s;
}

#define NL_FLOOR2 1e-28
/* Jacobi-PCG for the three velocity components at once (oracle/lns.py pcg_helm): r holds b on entry; returns the iteration count */
int nl_pcg_helm(long E, int n, const double *D, const double *const *G, const double *bm1, long ngroups, const long *off, const long *idx,
                const double *const *mask, const double *minv, const double *vmult, const double *wnorm, double nu, double h2,
                double *const *r, double *const *x, double *const *z, double *const *p, double *const *w, double tol2, int maxit, int fixed) {
    const long N = E * n * n * n;
    double rz = 0.0;
    for (int c = 0; c < 3; ++c) {
        double *xc = x[c], *rc = r[c], *zc = z[c], *pc = p[c];
        const double *mc = mask[c];
        double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
        for (long i = 0; i < N; ++i) {
            xc[i] = 0.0;
            zc[i] = mc[i] * minv[i] * rc[i];
            pc[i] = zc[i];
            s += rc[i] * zc[i] * vmult[i];
        }
        rz += s;
    }
    int it = 0;
    const int lim = fixed > 0 ? fixed : maxit;
    double rn20 = -1.0;
    while (it < lim) {
        double rn2 = 0.0;
        for (int c = 0; c < 3; ++c) rn2 += nl_glsc3(N, r[c], r[c], wnorm);
        if (rn20 < 0.0) rn20 = rn2;
        if (rn2 <= NL_FLOOR2 * rn20) break;
        if (fixed <= 0 && rn2 < tol2) break;
        double pw = 0.0;
        for (int c = 0; c < 3; ++c) {
            nl_axhelm(E, n, D, G, bm1, p[c], w[c], nu, h2);
            nl_gs(ngroups, off, idx, w[c]);
            double *wc = w[c];
            const double *mc = mask[c], *pc = p[c];
            double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
            for (long i = 0; i < N; ++i) {
                wc[i] *= mc[i];
                s += pc[i] * wc[i] * vmult[i];
            }
            pw += s;
        }
        const double alpha = rz / pw;
        double rzn = 0.0;
        for (int c = 0; c < 3; ++c) {
            double *xc = x[c], *rc = r[c], *zc = z[c];
            const double *pc = p[c], *wc = w[c], *mc = mask[c];
            double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
            for (long i = 0; i < N; ++i) {
                xc[i] += alpha * pc[i];
                rc[i] -= alpha * wc[i];
                zc[i] = mc[i] * minv[i] * rc[i];
                s += rc[i] * zc[i] * vmult[i];
            }
            rzn += s;
        }
        const double beta = rzn / rz;
        rz = rzn;
        for (int c = 0; c < 3; ++c) {
            double *pc = p[c];
            const double *zc = z[c];
#pragma omp parallel for schedule(static)
            for (long i = 0; i < N; ++i) pc[i] = zc[i] + beta * pc[i];
        }
        ++it;
    }
    return it;
}

/* Jacobi-PCG for the consistent Poisson operator on the mean-free subspace (oracle/lns.py pcg_E): r holds b on entry; u[3] is
 * velocity-mesh scratch for E p = D (mask binvm1 QQ^T D^T p).  tol2 = (ptol / scale)^2; proj: remove the means (no outflow). */
int nl_pcg_E(long E, int n, int n2, const double *I12, const double *D12, const double *const *rst2w, long ngroups, const long *off, const long *idx,
             const double *const *mbinv, const double *minv, const double *bm2, double volvm2, int proj, double *r, double *x, double *z, double *p,
             double *w, double *const *u, double tol2, int maxit, int fixed) {
    const long N2 = E * n2 * n2 * n2, N1 = E * n * n * n;
    if (proj) nl_shift(N2, r, nl_sum(N2, r) / (double)N2);
    double rz = 0.0;
    {
        double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
        for (long i = 0; i < N2; ++i) {
            x[i] = 0.0;
            z[i] = minv[i] * r[i];
            p[i] = z[i];
            s += r[i] * z[i];
        }
        rz = s;
    }
    if (proj) nl_shift(N2, p, nl_sum(N2, p) / (double)N2);
    int it = 0;
    const int lim = fixed > 0 ? fixed : maxit;
    double rn20 = -1.0;
    while (it < lim) {
        double rn2 = 0.0;
#pragma omp parallel for reduction(+ : rn2) schedule(static)
        for (long i = 0; i < N2; ++i) rn2 += r[i] * r[i] / bm2[i];
        rn2 /= volvm2;
        if (rn20 < 0.0) rn20 = rn2;
        if (rn2 <= NL_FLOOR2 * rn20) break;
        if (fixed <= 0 && rn2 < tol2) break;
        nl_opgradt(E, n, n2, I12, D12, rst2w, p, u);
        for (int c = 0; c < 3; ++c) {
            nl_gs(ngroups, off, idx, u[c]);
            double *uc = u[c];
            const double *mb = mbinv[c];
#pragma omp parallel for schedule(static)
            for (long i = 0; i < N1; ++i) uc[i] *= mb[i];
        }
        nl_opdiv(E, n, n2, I12, D12, rst2w, (const double *const *)u, w);
        double pw = 0.0;
#pragma omp parallel for reduction(+ : pw) schedule(static)
        for (long i = 0; i < N2; ++i) pw += p[i] * w[i];
        const double alpha = rz / pw;
        const double wm = proj ? nl_sum(N2, w) / (double)N2 : 0.0;
        double rzn = 0.0;
#pragma omp parallel for reduction(+ : rzn) schedule(static)
        for (long i = 0; i < N2; ++i) {
            x[i] += alpha * p[i];
            r[i] -= alpha * (w[i] - wm);
            z[i] = minv[i] * r[i];
            rzn += r[i] * z[i];
        }
        const double beta = rzn / rz;
        rz = rzn;
        const double zm = proj ? nl_sum(N2, z) / (double)N2 : 0.0;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < N2; ++i) p[i] = (z[i] - zm) + beta * p[i];
        ++it;
    }
    return it;
}
