"""ORACLE (test infrastructure only) -- Arnoldi / Krylov-Schur eigensolver over the abstract
vector API, restated in numpy.  Never imported by the product.

PARITY UNPINNED at the level of LightKrylov's internals (restart policy, selection); the converged eigenvalues do
not depend on them and are pinned end to end.  The algorithm is LightKrylov's `eigs` (nekStab/LightKrylov @ main, un-pinned,
/root/reference/LightKrylov_setup.sh:55-57), absent from `/root/reference`.  Its call site and the
post-processing ARE in the tree and are followed here:
/root/reference/src/neklab_analysis.f90:77-93 (`zero_basis`, `eigs(exptA, eigvecs, eigvals, residuals,
info, x0=, kdim=, transpose=, write_intermediate=)`, `eigvals = log(eigvals)/tau`).
The algorithm is restated from the published methods: Arnoldi with classical Gram-Schmidt plus one
re-orthogonalisation pass (CGS2, Giraud et al. 2005), Ritz residual estimate
`|h_{k+1,k}| |e_k^T y|` (Saad 2011), thick restart on the wanted Ritz subspace (Krylov-Schur,
Stewart 2001, in its orthonormal-basis-of-the-invariant-subspace form).

The structure deliberately mirrors how neklab + LightKrylov drive the vectors: one `dot` per basis
vector, one `axpby` per basis vector.
"""
from __future__ import annotations

import numpy as np


def innerprod(X, w):
    return np.array([x.dot(w) for x in X])


def cgs2_step(X, w):
    """w <- (I - X X^T B)^2 w ; returns accumulated coefficients."""
    h = innerprod(X, w)
    for hj, x in zip(h, X):
        w.axpby(-hj, x, 1.0)
    h2 = innerprod(X, w)
    for hj, x in zip(h2, X):
        w.axpby(-hj, x, 1.0)
    return h + h2


def arnoldi_step(matvec, V, H, k):
    """Extend A V_k = V_{k+1} H by one column (k is 0-based: uses V[k], creates V[k+1])."""
    w = matvec(V[k])
    h = cgs2_step(V[: k + 1], w)
    beta = w.norm()
    H[: k + 1, k] = h
    H[k + 1, k] = beta
    if beta > 0:
        w.scal(1.0 / beta)
    V[k + 1] = w
    return beta


def block_cgs2(V, k, s):
    """Block classical Gram-Schmidt with re-orthogonalisation of the s vectors V[k:k+s] against V[:k], then a Cholesky QR
    (twice) among them; returns the (k+s, s) coefficient matrix [projection coefficients; R] with
    W_old = V[:k] coef[:k] + W_new R.  The twin of nlg_basis_block_cgs2 (block Arnoldi; LightKrylov's block variants are
    not in the reference tree -- restated from the published method: Stathopoulos & Wu 2002 for CholQR2)."""
    W = V[k:k + s]
    coef = np.zeros((k + s, s))
    for _ in range(2):
        if k == 0:
            break
        h = np.array([[V[j].dot(W[v]) for v in range(s)] for j in range(k)])     # all projections from the same W
        for v in range(s):
            W[v].axpby(-1.0, lincomb(V, h[:, v], k), 1.0)
        coef[:k] += h
    R = np.eye(s)
    dep = [False] * s          # deflated (numerically dependent) columns, as in nlg_basis_block_cgs2: zero vector, zero diagonal of R
    hsq = np.sum(coef[:k] ** 2, axis=0)     # what the projections removed: |w|^2 = |w - V h|^2 + |h|^2 for an orthonormal basis
    for rnd in range(2):
        G = np.array([[W[a].dot(W[b]) for b in range(s)] for a in range(s)])
        gin = np.diag(G).copy()             # squared norms at entry to THIS round (the pivots are tested against these)
        if rnd == 0:
            for i in range(s):              # a column in span(V) to rounding: CGS2 leaves about eps |w| of it
                if not (gin[i] > 1e-24 * (gin[i] + hsq[i])):
                    dep[i] = True
        L = np.zeros((s, s))
        for i in range(s):
            for j in range(i + 1):
                a = G[i, j] - np.dot(L[i, :j], L[j, :j])
                if i == j:
                    if dep[i] or not (a > 1e-14 * gin[i]) or not (gin[i] > 0.0):
                        dep[i] = True
                    else:
                        L[i, i] = np.sqrt(a)
                else:
                    L[i, j] = 0.0 if dep[j] else a / L[j, j]
        Rr = L.T
        Linv = L.copy()
        for i in range(s):
            if dep[i]:
                Linv[i, i] = 1.0
        T = np.linalg.inv(Linv.T)
        for c in range(s):
            if dep[c]:
                T[:, c] = 0.0
        Wn = []
        for b in range(s):
            w = lincomb(W, T[:, b], s)
            Wn.append(w)
        for b in range(s):
            W[b] = Wn[b]
        R = Rr @ R
    coef[k:] = R
    for v in range(s):
        V[k + v] = W[v]
    return coef


def block_arnoldi_step(matvec, V, H, k, s):
    """V[k:k+s] -> V[k+s:k+2s]; H[:k+2s, k:k+s] written."""
    for v in range(s):
        V[k + s + v] = matvec(V[k + v])
    H[:k + 2 * s, k:k + s] = block_cgs2(V, k + s, s)


def lincomb(V, c, k):
    """sum_i c[i] V[i], i < k.  With the consistent treatment of the restart history (oracle/vectors.py
    CONSISTENT_RST) the combination also carries the combined history slots of the basis vectors."""
    w = V[0].copy()
    w.zero()
    if getattr(type(w), "CONSISTENT_RST", False):
        w.nrst = max(getattr(V[i], "nrst", 0) for i in range(k))
    for i in range(k):
        w.axpby(c[i], V[i], 1.0)
    return w


def ritz(H, k):
    """Eigen-decomposition of the leading k x k block; residuals from row k (0-based)."""
    lam, Y = np.linalg.eig(H[:k, :k])
    res = np.abs(H[k, :k] @ Y)
    order = np.argsort(-np.abs(lam), kind="stable")
    return lam[order], Y[:, order], res[order]


def select_wanted(lam, nkeep_min):
    """Keep Ritz values with modulus above the median, at least nkeep_min, conjugate pairs together.
    `lam` sorted by decreasing modulus."""
    k = len(lam)
    med = np.median(np.abs(lam))
    p = max(int(np.sum(np.abs(lam) > med)), nkeep_min)
    p = min(p, k - 1)
    # do not split a conjugate pair
    if p < k and abs(lam[p - 1].imag) > 0 and p >= 1:
        if np.isclose(lam[p - 1], np.conj(lam[p]), rtol=1e-10, atol=1e-14):
            p += 1
    return min(p, k - 1)


def real_basis(Y, lam, p):
    """Real orthonormal basis Q (k x p) of span of the first p Ritz vectors."""
    cols = []
    j = 0
    while j < p:
        if abs(lam[j].imag) > 0 and j + 1 < p:
            cols.append(Y[:, j].real)
            cols.append(Y[:, j].imag)
            j += 2
        else:
            cols.append(Y[:, j].real)
            j += 1
    M = np.stack(cols, axis=1)
    Q, _ = np.linalg.qr(M)
    return Q


def eigs(matvec, x0, nev, kdim, tol=None, max_restarts=50, new_vector=None, log=None):
    """Leading-modulus eigenpairs of the operator behind `matvec`.

    Returns (eigvals[nev], eigvecs list (LAPACK real convention: a conjugate pair occupies two
    consecutive vectors Re, Im), residuals[nev], info) with info = number of matvecs.
    """
    if tol is None:
        tol = np.sqrt(10.0 ** -15)
    V = [None] * (kdim + 1)
    H = np.zeros((kdim + 1, kdim))
    v0 = x0.copy()
    v0.scal(1.0 / v0.norm())
    V[0] = v0
    kstart, nmv = 0, 0
    lam = Y = res = None
    for _ in range(max_restarts + 1):
        k = kstart
        done = False
        while k < kdim:
            arnoldi_step(matvec, V, H, k)
            nmv += 1
            k += 1
            lam, Y, res = ritz(H, k)
            conv = int(np.sum(res < tol))
            if log is not None:
                log(nmv, lam, res, tol)
            if conv >= nev:
                done = True
                break
        if done or k < kdim:
            break
        # thick restart (Krylov-Schur): compress onto the wanted Ritz subspace
        p = select_wanted(lam, nev)
        Q = real_basis(Y, lam, p)
        p = Q.shape[1]
        Vnew = []
        for j in range(p):
            Vnew.append(lincomb(V, Q[:, j], k))
        S = Q.T @ H[:k, :k] @ Q
        b = H[k, :k] @ Q
        vk = V[k]
        H[:, :] = 0.0
        H[:p, :p] = S
        H[p, :p] = b
        for j in range(p):
            V[j] = Vnew[j]
        V[p] = vk
        kstart = p
    kf = k
    nev_out = min(nev, kf)
    vecs = []
    j = 0
    while j < nev_out:
        def comb(c):
            return lincomb(V, c, kf)
        if abs(lam[j].imag) > 0:
            y = Y[:, j] if lam[j].imag > 0 else np.conj(Y[:, j])
            vecs.append(comb(y.real))
            if j + 1 < nev_out:
                vecs.append(comb(y.imag))
            j += 2
        else:
            vecs.append(comb(Y[:, j].real))
            j += 1
    return lam[:nev_out], vecs[:nev_out], res[:nev_out], nmv


def svds(matvec, rmatvec, u0, nsv, kdim, tol=None):
    """Leading singular triplets by Golub-Kahan-Lanczos bidiagonalisation with full re-orthogonalisation -- the
    LightKrylov `svds` call of /root/reference/src/neklab_analysis.f90:136, restated (see csrc/krylov.hip nlg_svds).
    Returns (S[nsv], U list, V list, residuals[nsv], number of operator applications)."""
    if tol is None:
        tol = np.sqrt(10.0 ** -15)
    U = [None] * (kdim + 1)
    V = [None] * kdim
    alpha, beta = np.zeros(kdim), np.zeros(kdim + 1)
    u = u0.copy()
    u.scal(1.0 / u.norm())
    U[0] = u
    k = nmv = 0
    sig = Q = res = None
    while k < kdim:
        v = rmatvec(U[k])
        if k > 0:
            cgs2_step(V[:k], v)
        alpha[k] = v.norm()
        v.scal(1.0 / alpha[k])
        V[k] = v
        w = matvec(V[k])
        cgs2_step(U[: k + 1], w)
        beta[k + 1] = w.norm()
        w.scal(1.0 / beta[k + 1])
        U[k + 1] = w
        nmv += 2
        k += 1
        B = np.diag(alpha[:k]) + np.diag(beta[1:k], -1)
        P, sig, Qt = np.linalg.svd(B)
        Q = Qt.T
        res = np.abs(beta[k] * Q[k - 1, :])
        if int(np.sum(res < tol)) >= nsv:
            break
    nout = min(nsv, k)
    Us, Vs = [], []
    B = np.diag(alpha[:k]) + np.diag(beta[1:k], -1)
    for i in range(nout):
        Vs.append(lincomb(V, Q[:, i], k))
        Us.append(lincomb(U, (B @ Q[:, i]) / sig[i], k))
    return sig[:nout], Us, Vs, res[:nout], nmv


# ---------------------------------------------------------------------------------------------------------------------
# Newton-Krylov fixed-point solver: twin of neklab_amd/host.py gmres / newton_fixed_point_iteration.
# LightKrylov's `newton` and `gmres_rdp` (call site /root/reference/src/neklab_analysis.f90:186-194) are absent from the
# reference tree: restated from the published algorithms, PARITY UNPINNED at their internals; the tolerance schedulers
# follow /root/reference/src/systems/neklab_systems.f90:229-335.
def nek_dynamic_tol(tol_old, target, rnorm):
    maxtol, mintol = 1.0e-4, 10.0 * 10.0 ** -12
    target = min(max(target, mintol), maxtol)
    tol = max(0.1 * rnorm, target)
    if tol < 10.0 * target:
        tol = target
    return min(tol, maxtol)


def nek_constant_tol(tol_old, target, rnorm):
    return max(target, 10.0 * 10.0 ** -12)


def gmres(matvec, b, atol, kdim=30, maxiter=10, shift=-1.0, replay_history=False):
    """Restarted GMRES(kdim) for (A + shift I) x = b from a zero guess; Arnoldi on A (same Krylov space).
    replay_history=False: new Krylov vectors lose their restart history (see neklab_amd/host.py gmres)."""
    x = b.copy()
    x.zero()
    r = b.copy()
    nmv = 0
    res = r.norm()
    for _ in range(maxiter):
        beta = res
        if beta <= atol:
            break
        V = [None] * (kdim + 1)
        v0 = r.copy()
        v0.scal(1.0 / beta)
        V[0] = v0
        H = np.zeros((kdim + 1, kdim))
        R = np.zeros((kdim + 1, kdim))
        cs, sn = np.zeros(kdim), np.zeros(kdim)
        g = np.zeros(kdim + 1)
        g[0] = beta
        k = 0
        while k < kdim:
            arnoldi_step(matvec, V, H, k)
            nmv += 1
            if not replay_history:
                V[k + 1].clear_rst_fields()
            h = H[: k + 2, k].copy()
            h[k] += shift
            for i in range(k):
                t = cs[i] * h[i] + sn[i] * h[i + 1]
                h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1]
                h[i] = t
            d = np.hypot(h[k], h[k + 1])
            cs[k], sn[k] = (1.0, 0.0) if d == 0.0 else (h[k] / d, h[k + 1] / d)
            h[k], h[k + 1] = d, 0.0
            R[: k + 1, k] = h[: k + 1]
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            k += 1
            res = abs(g[k])
            if res <= atol:
                break
        y = np.linalg.solve(np.triu(R[:k, :k]), g[:k])
        x.axpby(1.0, lincomb(V, y, k), 1.0)
        if res <= atol:
            break
        Ax = matvec(x)
        nmv += 1
        r = b.copy()
        r.axpby(-1.0, Ax, 1.0)
        r.axpby(-shift, x, 1.0)
        res = r.norm()
    return x, res, nmv


def newton(nonlinear_map, jacobian_for, set_tolerance, X, tol, tol_mode=1, maxiter=40, kdim=30, log=None,
           replay_history=False):
    """X updated in place.  nonlinear_map(X) -> F(X); jacobian_for(X) -> matvec of exp(tau J(X)); set_tolerance(tol)."""
    sched = nek_constant_tol if tol_mode == 1 else nek_dynamic_tol
    final = sched(0.0, tol, 0.0)
    cur, rnorm = 0.0, 1.0
    residuals, nmv_total, converged = [], 0, False
    for it in range(maxiter + 1):
        new = sched(cur, tol, rnorm)
        if new != cur:
            cur = new
            set_tolerance(cur)
        r = nonlinear_map(X)
        rnorm = r.norm()
        residuals.append(rnorm)
        if log is not None:
            log("newton %2d  |F(X)| = %.6e  solver tol %.3e" % (it, rnorm, cur))
        if rnorm < tol and cur <= final:
            converged = True
            break
        if it == maxiter:
            break
        mv = jacobian_for(X)
        r.scal(-1.0)
        dx, res, nmv = gmres(mv, r, atol=sched(cur, tol, rnorm), kdim=kdim, replay_history=replay_history)
        nmv_total += nmv
        X.axpby(1.0, dx, 1.0)
    return {"converged": converged, "iterations": len(residuals) - 1, "residuals": residuals, "gmres_matvecs": nmv_total}


# ---------------------------------------------------------------------------------------------------------------------
# Resolvent by time stepping: twin of neklab_amd/host.py resolvent_linop (/root/reference/src/linops/resolvent.f90:17-44).
def resolvent_apply(A, f_re, f_im, omega, adjoint=False):
    """A: oracle ExptA built with tau = 2 pi / |omega| (1 for omega = 0).  Returns (x_re, x_im, b, gmres matvecs)."""
    tau = A.cfg.tau
    b = A.integrate_forced(None, f_re, f_im, omega, adjoint)
    rhs = b.copy()
    rhs.scal(-1.0)
    mv = (lambda v: A.matvec(v, adjoint=True)) if adjoint else A.matvec
    x, res, nmv = gmres(mv, rhs, atol=max(1.0e-6 * b.norm(), 1.0e-12), kdim=64)
    x.clear_rst_fields()
    A.set_tau(tau / 4.0)
    y = A.integrate_forced(x, f_re, f_im, omega, adjoint)
    A.set_tau(tau)
    return x, y, b, nmv
