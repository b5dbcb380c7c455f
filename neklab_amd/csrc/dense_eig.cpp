// Small dense real nonsymmetric eigen-solver (host): Householder reduction to Hessenberg form followed
// by the implicit double-shift QR algorithm with eigenvector back-substitution (the classical
// EISPACK orthes/ortran/hqr2 sequence, Wilkinson & Reinsch 1971).  This stands in for the LAPACK call
// that LightKrylov makes through stdlib `eig` inside `eigs` (call site
// /root/reference/src/neklab_analysis.f90:80-81); no LAPACK is available in this image.
//
// Convention of the result = LAPACK dgeev: eigenvalue j is wr[j] + i*wi[j]; for a complex pair
// (wi[j] > 0, wi[j+1] < 0) columns j and j+1 of vr hold the real and imaginary part of the
// eigenvector of the eigenvalue with positive imaginary part.  Every eigenvector has unit 2-norm.
#include <algorithm>
#include <cmath>
#include <vector>

#include "neklab_gpu.h"

namespace {

struct Work {
    int n;
    std::vector<double> H, V, d, e, ort;
    double &h(int i, int j) { return H[(size_t)i * n + j]; }
    double &v(int i, int j) { return V[(size_t)i * n + j]; }
};

void cdiv(double xr, double xi, double yr, double yi, double &cr, double &ci) {
    double r, d;
    if (std::fabs(yr) > std::fabs(yi)) {
        r = yi / yr;
        d = yr + r * yi;
        cr = (xr + r * xi) / d;
        ci = (xi - r * xr) / d;
    } else {
        r = yr / yi;
        d = yi + r * yr;
        cr = (r * xr + xi) / d;
        ci = (r * xi - xr) / d;
    }
}

void orthes(Work &w) {
    const int n = w.n, low = 0, high = n - 1;
    for (int m = low + 1; m <= high - 1; ++m) {
        double scale = 0.0;
        for (int i = m; i <= high; ++i) scale += std::fabs(w.h(i, m - 1));
        if (scale != 0.0) {
            double hh = 0.0;
            for (int i = high; i >= m; --i) {
                w.ort[i] = w.h(i, m - 1) / scale;
                hh += w.ort[i] * w.ort[i];
            }
            double g = std::sqrt(hh);
            if (w.ort[m] > 0) g = -g;
            hh -= w.ort[m] * g;
            w.ort[m] -= g;
            for (int j = m; j < n; ++j) {
                double f = 0.0;
                for (int i = high; i >= m; --i) f += w.ort[i] * w.h(i, j);
                f /= hh;
                for (int i = m; i <= high; ++i) w.h(i, j) -= f * w.ort[i];
            }
            for (int i = 0; i <= high; ++i) {
                double f = 0.0;
                for (int j = high; j >= m; --j) f += w.ort[j] * w.h(i, j);
                f /= hh;
                for (int j = m; j <= high; ++j) w.h(i, j) -= f * w.ort[j];
            }
            w.ort[m] *= scale;
            w.h(m, m - 1) = scale * g;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) w.v(i, j) = (i == j) ? 1.0 : 0.0;
    for (int m = high - 1; m >= low + 1; --m) {
        if (w.h(m, m - 1) != 0.0) {
            for (int i = m + 1; i <= high; ++i) w.ort[i] = w.h(i, m - 1);
            for (int j = m; j <= high; ++j) {
                double g = 0.0;
                for (int i = m; i <= high; ++i) g += w.ort[i] * w.v(i, j);
                g = (g / w.ort[m]) / w.h(m, m - 1);
                for (int i = m; i <= high; ++i) w.v(i, j) += g * w.ort[i];
            }
        }
    }
}

int hqr2(Work &w) {
    const int nn = w.n;
    int n = nn - 1;
    const int low = 0, high = nn - 1;
    const double eps = std::pow(2.0, -52.0);
    double exshift = 0.0;
    double p = 0, q = 0, r = 0, s = 0, z = 0, t, ww, x, y;
    double norm = 0.0;
    for (int i = 0; i < nn; ++i)
        for (int j = std::max(i - 1, 0); j < nn; ++j) norm += std::fabs(w.h(i, j));
    int iter = 0, total_iter = 0;
    while (n >= low) {
        int l = n;
        while (l > low) {
            s = std::fabs(w.h(l - 1, l - 1)) + std::fabs(w.h(l, l));
            if (s == 0.0) s = norm;
            if (std::fabs(w.h(l, l - 1)) < eps * s) break;
            --l;
        }
        if (l == n) {
            w.h(n, n) += exshift;
            w.d[n] = w.h(n, n);
            w.e[n] = 0.0;
            --n;
            iter = 0;
        } else if (l == n - 1) {
            ww = w.h(n, n - 1) * w.h(n - 1, n);
            p = (w.h(n - 1, n - 1) - w.h(n, n)) / 2.0;
            q = p * p + ww;
            z = std::sqrt(std::fabs(q));
            w.h(n, n) += exshift;
            w.h(n - 1, n - 1) += exshift;
            x = w.h(n, n);
            if (q >= 0) {
                z = (p >= 0) ? p + z : p - z;
                w.d[n - 1] = x + z;
                w.d[n] = w.d[n - 1];
                if (z != 0.0) w.d[n] = x - ww / z;
                w.e[n - 1] = 0.0;
                w.e[n] = 0.0;
                x = w.h(n, n - 1);
                s = std::fabs(x) + std::fabs(z);
                p = x / s;
                q = z / s;
                r = std::sqrt(p * p + q * q);
                p /= r;
                q /= r;
                for (int j = n - 1; j < nn; ++j) {
                    z = w.h(n - 1, j);
                    w.h(n - 1, j) = q * z + p * w.h(n, j);
                    w.h(n, j) = q * w.h(n, j) - p * z;
                }
                for (int i = 0; i <= n; ++i) {
                    z = w.h(i, n - 1);
                    w.h(i, n - 1) = q * z + p * w.h(i, n);
                    w.h(i, n) = q * w.h(i, n) - p * z;
                }
                for (int i = low; i <= high; ++i) {
                    z = w.v(i, n - 1);
                    w.v(i, n - 1) = q * z + p * w.v(i, n);
                    w.v(i, n) = q * w.v(i, n) - p * z;
                }
            } else {
                w.d[n - 1] = x + p;
                w.d[n] = x + p;
                w.e[n - 1] = z;
                w.e[n] = -z;
            }
            n -= 2;
            iter = 0;
        } else {
            x = w.h(n, n);
            y = 0.0;
            ww = 0.0;
            if (l < n) {
                y = w.h(n - 1, n - 1);
                ww = w.h(n, n - 1) * w.h(n - 1, n);
            }
            if (iter == 10) {
                exshift += x;
                for (int i = low; i <= n; ++i) w.h(i, i) -= x;
                s = std::fabs(w.h(n, n - 1)) + std::fabs(w.h(n - 1, n - 2));
                x = y = 0.75 * s;
                ww = -0.4375 * s * s;
            }
            if (iter == 30) {
                s = (y - x) / 2.0;
                s = s * s + ww;
                if (s > 0) {
                    s = std::sqrt(s);
                    if (y < x) s = -s;
                    s = x - ww / ((y - x) / 2.0 + s);
                    for (int i = low; i <= n; ++i) w.h(i, i) -= s;
                    exshift += s;
                    x = y = ww = 0.964;
                }
            }
            ++iter;
            if (++total_iter > 60 * nn + 1000) return 1;   // no convergence
            int m = n - 2;
            while (m >= l) {
                z = w.h(m, m);
                r = x - z;
                s = y - z;
                p = (r * s - ww) / w.h(m + 1, m) + w.h(m, m + 1);
                q = w.h(m + 1, m + 1) - z - r - s;
                r = w.h(m + 2, m + 1);
                s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                p /= s;
                q /= s;
                r /= s;
                if (m == l) break;
                if (std::fabs(w.h(m, m - 1)) * (std::fabs(q) + std::fabs(r)) <
                    eps * (std::fabs(p) * (std::fabs(w.h(m - 1, m - 1)) + std::fabs(z) + std::fabs(w.h(m + 1, m + 1)))))
                    break;
                --m;
            }
            for (int i = m + 2; i <= n; ++i) {
                w.h(i, i - 2) = 0.0;
                if (i > m + 2) w.h(i, i - 3) = 0.0;
            }
            for (int k = m; k <= n - 1; ++k) {
                const bool notlast = (k != n - 1);
                if (k != m) {
                    p = w.h(k, k - 1);
                    q = w.h(k + 1, k - 1);
                    r = notlast ? w.h(k + 2, k - 1) : 0.0;
                    x = std::fabs(p) + std::fabs(q) + std::fabs(r);
                    if (x == 0.0) continue;
                    p /= x;
                    q /= x;
                    r /= x;
                }
                s = std::sqrt(p * p + q * q + r * r);
                if (p < 0) s = -s;
                if (s != 0) {
                    if (k != m)
                        w.h(k, k - 1) = -s * x;
                    else if (l != m)
                        w.h(k, k - 1) = -w.h(k, k - 1);
                    p += s;
                    x = p / s;
                    y = q / s;
                    z = r / s;
                    q /= p;
                    r /= p;
                    for (int j = k; j < nn; ++j) {
                        p = w.h(k, j) + q * w.h(k + 1, j);
                        if (notlast) {
                            p += r * w.h(k + 2, j);
                            w.h(k + 2, j) -= p * z;
                        }
                        w.h(k, j) -= p * x;
                        w.h(k + 1, j) -= p * y;
                    }
                    for (int i = 0; i <= std::min(n, k + 3); ++i) {
                        p = x * w.h(i, k) + y * w.h(i, k + 1);
                        if (notlast) {
                            p += z * w.h(i, k + 2);
                            w.h(i, k + 2) -= p * r;
                        }
                        w.h(i, k) -= p;
                        w.h(i, k + 1) -= p * q;
                    }
                    for (int i = low; i <= high; ++i) {
                        p = x * w.v(i, k) + y * w.v(i, k + 1);
                        if (notlast) {
                            p += z * w.v(i, k + 2);
                            w.v(i, k + 2) -= p * r;
                        }
                        w.v(i, k) -= p;
                        w.v(i, k + 1) -= p * q;
                    }
                }
            }
        }
    }
    if (norm == 0.0) return 0;
    // back-substitution for the vectors of the quasi-triangular form
    for (n = nn - 1; n >= 0; --n) {
        p = w.d[n];
        q = w.e[n];
        if (q == 0) {
            int l = n;
            w.h(n, n) = 1.0;
            for (int i = n - 1; i >= 0; --i) {
                ww = w.h(i, i) - p;
                r = 0.0;
                for (int j = l; j <= n; ++j) r += w.h(i, j) * w.h(j, n);
                if (w.e[i] < 0.0) {
                    z = ww;
                    s = r;
                } else {
                    l = i;
                    if (w.e[i] == 0.0) {
                        w.h(i, n) = (ww != 0.0) ? -r / ww : -r / (eps * norm);
                    } else {
                        x = w.h(i, i + 1);
                        y = w.h(i + 1, i);
                        q = (w.d[i] - p) * (w.d[i] - p) + w.e[i] * w.e[i];
                        t = (x * s - z * r) / q;
                        w.h(i, n) = t;
                        if (std::fabs(x) > std::fabs(z))
                            w.h(i + 1, n) = (-r - ww * t) / x;
                        else
                            w.h(i + 1, n) = (-s - y * t) / z;
                    }
                    t = std::fabs(w.h(i, n));
                    if ((eps * t) * t > 1)
                        for (int j = i; j <= n; ++j) w.h(j, n) /= t;
                }
            }
        } else if (q < 0) {
            int l = n - 1;
            if (std::fabs(w.h(n, n - 1)) > std::fabs(w.h(n - 1, n))) {
                w.h(n - 1, n - 1) = q / w.h(n, n - 1);
                w.h(n - 1, n) = -(w.h(n, n) - p) / w.h(n, n - 1);
            } else {
                double cr, ci;
                cdiv(0.0, -w.h(n - 1, n), w.h(n - 1, n - 1) - p, q, cr, ci);
                w.h(n - 1, n - 1) = cr;
                w.h(n - 1, n) = ci;
            }
            w.h(n, n - 1) = 0.0;
            w.h(n, n) = 1.0;
            for (int i = n - 2; i >= 0; --i) {
                double ra = 0.0, sa = 0.0, vr, vi, cr, ci;
                for (int j = l; j <= n; ++j) {
                    ra += w.h(i, j) * w.h(j, n - 1);
                    sa += w.h(i, j) * w.h(j, n);
                }
                ww = w.h(i, i) - p;
                if (w.e[i] < 0.0) {
                    z = ww;
                    r = ra;
                    s = sa;
                } else {
                    l = i;
                    if (w.e[i] == 0) {
                        cdiv(-ra, -sa, ww, q, cr, ci);
                        w.h(i, n - 1) = cr;
                        w.h(i, n) = ci;
                    } else {
                        x = w.h(i, i + 1);
                        y = w.h(i + 1, i);
                        vr = (w.d[i] - p) * (w.d[i] - p) + w.e[i] * w.e[i] - q * q;
                        vi = (w.d[i] - p) * 2.0 * q;
                        if (vr == 0.0 && vi == 0.0)
                            vr = eps * norm * (std::fabs(ww) + std::fabs(q) + std::fabs(x) + std::fabs(y) + std::fabs(z));
                        cdiv(x * r - z * ra + q * sa, x * s - z * sa - q * ra, vr, vi, cr, ci);
                        w.h(i, n - 1) = cr;
                        w.h(i, n) = ci;
                        if (std::fabs(x) > (std::fabs(z) + std::fabs(q))) {
                            w.h(i + 1, n - 1) = (-ra - ww * w.h(i, n - 1) + q * w.h(i, n)) / x;
                            w.h(i + 1, n) = (-sa - ww * w.h(i, n) - q * w.h(i, n - 1)) / x;
                        } else {
                            cdiv(-r - y * w.h(i, n - 1), -s - y * w.h(i, n), z, q, cr, ci);
                            w.h(i + 1, n - 1) = cr;
                            w.h(i + 1, n) = ci;
                        }
                    }
                    t = std::max(std::fabs(w.h(i, n - 1)), std::fabs(w.h(i, n)));
                    if ((eps * t) * t > 1)
                        for (int j = i; j <= n; ++j) {
                            w.h(j, n - 1) /= t;
                            w.h(j, n) /= t;
                        }
                }
            }
        }
    }
    // back-transformation to the eigenvectors of the original matrix
    for (int j = nn - 1; j >= low; --j) {
        for (int i = low; i <= high; ++i) {
            z = 0.0;
            for (int k = low; k <= std::min(j, high); ++k) z += w.v(i, k) * w.h(k, j);
            w.v(i, j) = z;
        }
    }
    return 0;
}

}  // namespace

extern "C" int nlg_dense_eig(int n, const double *A, int lda, double *wr, double *wi, double *vr, int ldvr) {
    if (n <= 0 || !A || !wr || !wi || !vr || lda < n || ldvr < n) return 1;
    Work w;
    w.n = n;
    w.H.assign((size_t)n * n, 0.0);
    w.V.assign((size_t)n * n, 0.0);
    w.d.assign(n, 0.0);
    w.e.assign(n, 0.0);
    w.ort.assign(n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) w.h(i, j) = A[(size_t)j * lda + i];
    orthes(w);
    if (hqr2(w)) return 2;
    for (int j = 0; j < n; ++j) {
        wr[j] = w.d[j];
        wi[j] = w.e[j];
    }
    // normalise and store column-major
    int j = 0;
    while (j < n) {
        if (w.e[j] > 0.0 && j + 1 < n) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += w.v(i, j) * w.v(i, j) + w.v(i, j + 1) * w.v(i, j + 1);
            s = s > 0 ? 1.0 / std::sqrt(s) : 1.0;
            for (int i = 0; i < n; ++i) {
                vr[(size_t)j * ldvr + i] = w.v(i, j) * s;
                vr[(size_t)(j + 1) * ldvr + i] = w.v(i, j + 1) * s;
            }
            j += 2;
        } else {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += w.v(i, j) * w.v(i, j);
            s = s > 0 ? 1.0 / std::sqrt(s) : 1.0;
            for (int i = 0; i < n; ++i) vr[(size_t)j * ldvr + i] = w.v(i, j) * s;
            j += 1;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Symmetric tridiagonal eigen-decomposition (implicit QL, EISPACK tql2): d[n] diagonal, e[n] sub-diagonal in
// e[1..n-1] (e[0] unused).  On return d holds the eigenvalues in ascending order, Z (row-major n x n) the
// eigenvectors as columns.  Used for the singular values of the Lanczos bidiagonal matrix in nlg_svds.
extern "C" int nlg_symtridiag_eig(int n, double *d, double *e_in, double *Z) {
    if (n <= 0 || !d || !e_in || !Z) return 1;
    std::vector<double> e(e_in, e_in + n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Z[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::pow(2.0, -52.0);
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) {
            if (std::fabs(e[m]) <= eps * tst1) break;
            ++m;
        }
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) return 2;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c;
                const double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = Z[(size_t)k * n + i + 1];
                        Z[(size_t)k * n + i + 1] = s * Z[(size_t)k * n + i] + c * h;
                        Z[(size_t)k * n + i] = c * Z[(size_t)k * n + i] - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    // sort ascending
    for (int i = 0; i < n - 1; ++i) {
        int k = i;
        double p = d[i];
        for (int j = i + 1; j < n; ++j)
            if (d[j] < p) {
                k = j;
                p = d[j];
            }
        if (k != i) {
            d[k] = d[i];
            d[i] = p;
            for (int j = 0; j < n; ++j) std::swap(Z[(size_t)j * n + i], Z[(size_t)j * n + k]);
        }
    }
    return 0;
}
