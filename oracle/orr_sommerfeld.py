"""ORACLE-side independent known answer: temporal Orr-Sommerfeld eigenvalues of plane Poiseuille flow U = 1 - y^2 by
Chebyshev collocation (Trefethen, Spectral Methods in MATLAB, program 40 -- clamped boundary conditions through the
(1 - y^2) factor).  Test infrastructure only.  Orszag (1971): Re = 10000, alpha = 1: c = 0.23752649 + 0.00373967 i."""
import numpy as np
import scipy.linalg as sl


def cheb(N):
    x = np.cos(np.pi * np.arange(N + 1) / N)
    c = np.hstack([2, np.ones(N - 1), 2]) * (-1) ** np.arange(N + 1)
    X = np.tile(x, (N + 1, 1)).T
    dX = X - X.T
    D = np.outer(c, 1 / c) / (dX + np.eye(N + 1))
    D -= np.diag(D.sum(1))
    return D, x


def orr_sommerfeld(Re, alpha, N=128):
    """Phase speeds c (perturbation ~ exp(i alpha (x - c t))), sorted by decreasing growth rate Im c."""
    D, y = cheb(N)
    D2 = D @ D
    D4 = D2 @ D2
    I = np.eye(N + 1)
    U = 1 - y ** 2
    S = np.diag(np.hstack([0, 1 / (1 - y[1:-1] ** 2), 0]))
    D4c = (np.diag(1 - y ** 2) @ D4 - 8 * np.diag(y) @ D2 @ D - 12 * D2) @ S
    D2i, D4i, Ii = D2[1:-1, 1:-1], D4c[1:-1, 1:-1], I[1:-1, 1:-1]
    Ui, Uppi = np.diag(U[1:-1]), np.diag(-2 * np.ones(N - 1))
    # (U - c)(D^2 - a^2) phi - U'' phi = (D^2 - a^2)^2 phi / (i a Re)
    A = Ui @ (D2i - alpha ** 2 * Ii) - Uppi - (D4i - 2 * alpha ** 2 * D2i + alpha ** 4 * Ii) / (1j * alpha * Re)
    B = D2i - alpha ** 2 * Ii
    c = sl.eigvals(A, B)
    c = c[np.isfinite(c) & (np.abs(c) < 10.0) & (c.imag < 1.0)]   # drop the spurious modes of the boundary rows
    return c[np.argsort(-c.imag)]
