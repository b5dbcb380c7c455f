"""ORACLE (test infrastructure only) -- numpy restatement of the spectral-element pieces of the
neklab hot path.  Never imported by the product (`neklab_amd/`); only `tests/`, `bench.py`'s
cpu_baseline leg and `__graft_entry__.smoke()` may use it.

PARITY PINNED AGAINST REFERENCE DATA (not against reference code): the arithmetic of this layer lives in
Nek5000 (un-vendored, un-pinned, `/root/reference/Nek5000_setup.sh:56-58`), absent from `/root/reference`, so it
is restated from the published algorithm (Deville, Fischer & Mund 2002; Fischer 1997).  The restatement is
checked against the Nek5000-generated base flow the reference ships (tests/test_cpu_reference_data.py: discrete
divergence 6e-12, steady momentum residual 2e-7 with these operators) -- see DESIGN.md section 2.  What the
reference tree itself pins is *which* operators are composed and how:

* Laplacian as `grad -> metric -> grad^T` with `rxm1..tzm1`, `jacmi`
  (reference: src/linops/neklab_linops.f90:332-366 `lap_1D`),
* linearised convective terms `(u.grad)Ub + (Ub.grad)u`, adjoint variant
  (reference: src/linops/neklab_linops.f90:268-313 `compute_LNS_conv`), dealiased on `lxd = 3*lx1/2`
  Gauss points (examples/cylinder/stability/direct/SIZE: lx1=6, lxd=9),
* pressure on the `lx2 = lx1-2` Gauss mesh (same SIZE file), gradient/divergence pair
  (reference: src/linops/neklab_linops.f90:368-380 `compute_LNS_gradp`),
* direct-stiffness summation / multiplicity (reference: src/vectors/real_vectors.f90:100-104
  `opdssum`, `vmult`, `dsavg`), Dirichlet masks (`bcdirvc`, :105),
* mass matrix `bm1` as the inner-product weight (reference: src/vectors/real_vectors.f90:217-224).

Layout: every field is `(E, n**dim)` with `ix` fastest (reference: real_vectors.f90:69).
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------------------
# 1-D quadrature / interpolation building blocks
# --------------------------------------------------------------------------------------
def _legendre(N, x):
    p0 = np.ones_like(x)
    if N == 0:
        return p0, np.zeros_like(x)
    p1 = x.copy()
    for k in range(2, N + 1):
        p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
    return p1, p0  # P_N, P_{N-1}


def gll(n):
    """GLL nodes and weights, n points."""
    N = n - 1
    x = -np.cos(np.pi * np.arange(n) / N)
    for _ in range(100):
        pN, pNm1 = _legendre(N, x)
        f = N * (pNm1 - x * pN)
        df = -N * (N + 1) * pN
        dx = f / df
        dx[0] = dx[-1] = 0.0
        x = x - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    x[0], x[-1] = -1.0, 1.0
    x = 0.5 * (x - x[::-1])
    pN, _ = _legendre(N, x)
    w = 2.0 / (N * (N + 1) * pN ** 2)
    return x, w


def gl(n):
    """Gauss-Legendre nodes and weights, n points."""
    k = np.arange(1, n + 1)
    x = -np.cos(np.pi * (k - 0.25) / (n + 0.5))
    for _ in range(100):
        pn, pnm1 = _legendre(n, x)
        dpn = n * (x * pn - pnm1) / (x * x - 1.0)
        dx = pn / dpn
        x = x - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    x = 0.5 * (x - x[::-1])
    pn, pnm1 = _legendre(n, x)
    dpn = n * (x * pn - pnm1) / (x * x - 1.0)
    w = 2.0 / ((1.0 - x * x) * dpn ** 2)
    return x, w


def bary_weights(x):
    n = len(x)
    w = np.ones(n)
    for j in range(n):
        for k in range(n):
            if k != j:
                w[j] /= (x[j] - x[k])
    return w


def interp_matrix(xfrom, xto):
    """I[k, j] = l_j(xto_k), Lagrange basis on xfrom (barycentric form)."""
    bw = bary_weights(xfrom)
    M = np.zeros((len(xto), len(xfrom)))
    for k, xk in enumerate(xto):
        d = xk - xfrom
        hit = np.where(np.abs(d) < 1e-15)[0]
        if len(hit):
            M[k, hit[0]] = 1.0
        else:
            t = bw / d
            M[k] = t / t.sum()
    return M


def deriv_matrix(x):
    """D[i, j] = l_j'(x_i) on nodes x."""
    n = len(x)
    bw = bary_weights(x)
    D = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if i != j:
                D[i, j] = (bw[j] / bw[i]) / (x[i] - x[j])
        D[i, i] = -np.sum(D[i, :])
    return D


def deriv_interp_matrix(xfrom, xto):
    """DJ[k, j] = l_j'(xto_k): derivative of the Lagrange basis on xfrom evaluated at xto."""
    return interp_matrix(xfrom, xto) @ deriv_matrix(xfrom)


# --------------------------------------------------------------------------------------
# tensor helpers; fields are (E, n**dim) -> (E, [nz,] ny, nx)
# --------------------------------------------------------------------------------------
def _ap(M, u, axis):
    """Apply 1-D operator M (m x n) along tensor axis 'x' (-1), 'y' (-2) or 'z' (-3)."""
    ax = {"x": -1, "y": -2, "z": -3}[axis]
    v = np.tensordot(u, M, axes=([ax], [1]))      # contracted axis goes last
    return np.moveaxis(v, -1, ax)


class SEM:
    """Discrete SEM operators on one mesh (oracle).  `mesh` carries the C-ABI input arrays."""

    def __init__(self, mesh, lxd=None):
        self.dim = dim = mesh.dim
        self.n = n = mesh.n
        self.E = E = mesh.E
        self.n2 = n2 = n - 2
        self.nd = nd = int(lxd) if lxd is not None else (3 * n) // 2
        self.axes = ("x", "y", "z")[:dim]
        self.shape1 = (E,) + (n,) * dim
        self.shape2 = (E,) + (n2,) * dim
        self.shaped = (E,) + (nd,) * dim
        self.lvn = E * n ** dim
        self.lpn = E * n2 ** dim
        self.z1, self.w1 = gll(n)
        self.z2, self.w2 = gl(n2)
        self.zd, self.wd = gl(nd)
        self.D = deriv_matrix(self.z1)
        self.I12 = interp_matrix(self.z1, self.z2)
        self.D12 = deriv_interp_matrix(self.z1, self.z2)
        self.Jd = interp_matrix(self.z1, self.zd)
        self.DJd = deriv_interp_matrix(self.z1, self.zd)
        self.glo = mesh.glo_num.reshape(-1)
        self.has_outflow = bool(mesh.has_outflow)

        X = [c.reshape(self.shape1) for c in ([mesh.x, mesh.y] + ([mesh.z] if dim == 3 else []))]
        self.X = X
        # dx_i/dr_j
        dX = [[_ap(self.D, X[i], self.axes[j]) for j in range(dim)] for i in range(dim)]
        if dim == 2:
            xr, xs = dX[0]
            yr, ys = dX[1]
            jac = xr * ys - xs * yr
            # rst[j][i] = J * d r_j / d x_i
            rst = [[ys, -xs], [-yr, xr]]
            w3 = self.w1[:, None] * self.w1[None, :]
            w32 = self.w2[:, None] * self.w2[None, :]
            w3d = self.wd[:, None] * self.wd[None, :]
        else:
            xr, xs, xt = dX[0]
            yr, ys, yt = dX[1]
            zr, zs, zt = dX[2]
            rst = [
                [ys * zt - yt * zs, xt * zs - xs * zt, xs * yt - xt * ys],
                [yt * zr - yr * zt, xr * zt - xt * zr, xt * yr - xr * yt],
                [yr * zs - ys * zr, xs * zr - xr * zs, xr * ys - xs * yr],
            ]
            jac = xr * rst[0][0] + xs * rst[1][0] + xt * rst[2][0]
            w3 = self.w1[:, None, None] * self.w1[None, :, None] * self.w1[None, None, :]
            w32 = self.w2[:, None, None] * self.w2[None, :, None] * self.w2[None, None, :]
            w3d = self.wd[:, None, None] * self.wd[None, :, None] * self.wd[None, None, :]
        if np.any(jac <= 0):
            raise ValueError("non-positive Jacobian")
        self.jac = jac
        self.rst = rst                       # Nek: rxm1, rym1, ... (J-scaled)
        self.w3 = w3
        self.bm1 = w3[None] * jac            # Nek: bm1
        self.G = [[None] * dim for _ in range(dim)]
        for i in range(dim):
            for j in range(i, dim):
                g = sum(rst[i][m] * rst[j][m] for m in range(dim)) * w3[None] / jac
                self.G[i][j] = g
                self.G[j][i] = g
        # pressure-mesh metrics: interpolated J-scaled metrics times GL weights
        self.rst2w = [[self.to_mesh2(rst[j][i]) * w32[None] for i in range(dim)] for j in range(dim)]
        self.bm2 = self.to_mesh2(jac) * w32[None]
        # dealiasing-mesh metrics (Nek: set_dealias_rx)
        self.rstdw = [[self.to_fine(rst[j][i]) * w3d[None] for i in range(dim)] for j in range(dim)]
        # multiplicity and assembled inverse mass
        self.mult = self.gs(np.ones(self.shape1))           # Nek: 1/vmult
        self.vmult = 1.0 / self.mult
        self.binvm1 = 1.0 / self.gs(self.bm1)               # Nek: binvm1 (assembled)
        self.mask = [m.reshape(self.shape1).copy() for m in mesh.mask]
        self.tmask = mesh.tmask.reshape(self.shape1).copy()
        self.volvm1 = float(np.sum(self.bm1))
        self.volvm2 = float(np.sum(self.bm2))
        # reference-element spacing for the CFL estimate (Nek: getdr)
        dr = np.empty(n)
        dr[0] = self.z1[1] - self.z1[0]
        dr[-1] = self.z1[-1] - self.z1[-2]
        dr[1:-1] = 0.5 * (self.z1[2:] - self.z1[:-2])
        self.rdr = 1.0 / dr

    # ---- shape helpers ---------------------------------------------------------------
    def f1(self, u):
        return np.asarray(u, dtype=np.float64).reshape(self.shape1)

    def f2(self, p):
        return np.asarray(p, dtype=np.float64).reshape(self.shape2)

    # ---- mesh transfers --------------------------------------------------------------
    def to_mesh2(self, u):
        for a in self.axes:
            u = _ap(self.I12, u, a)
        return u

    def to_fine(self, u):
        for a in self.axes:
            u = _ap(self.Jd, u, a)
        return u

    def from_fine_T(self, uf):
        for a in self.axes:
            uf = _ap(self.Jd.T, uf, a)
        return uf

    # ---- gather-scatter (direct stiffness summation) ---------------------------------
    def gs(self, u):
        flat = np.asarray(u).reshape(-1)
        glob = np.bincount(self.glo, weights=flat, minlength=int(self.glo.max()) + 1)
        return glob[self.glo].reshape(np.shape(u))

    def dsavg(self, u):
        """Nek dsavg: average over the copies of each shared dof."""
        return self.gs(u) * self.vmult

    # ---- inner product weights -------------------------------------------------------
    def glsc3(self, a, b):
        """Nek glsc3(a, b, bm1): sum_i a_i b_i bm1_i over LOCAL dofs (shared dofs counted per copy)."""
        return float(np.sum(self.f1(a) * self.f1(b) * self.bm1))

    # ---- element-local operators -----------------------------------------------------
    def grad_rst(self, u):
        return [_ap(self.D, u, a) for a in self.axes]

    def gradm1(self, u):
        """Physical gradient on the velocity mesh (Nek gradm1 without mass weighting, jacmi applied)."""
        ur = self.grad_rst(self.f1(u))
        return [sum(self.rst[j][i] * ur[j] for j in range(self.dim)) / self.jac for i in range(self.dim)]

    def axhelm_local(self, u, h1, h2):
        """w = h1 * D^T G D u + h2 * B u, element-local (no assembly, no mask)."""
        u = self.f1(u)
        ur = self.grad_rst(u)
        w = h2 * self.bm1 * u
        for i in range(self.dim):
            t = sum(self.G[i][j] * ur[j] for j in range(self.dim))
            w = w + h1 * _ap(self.D.T, t, self.axes[i])
        return w

    def helm_diag_local(self, h1, h2):
        """Exact diagonal of the local Helmholtz operator (cross metric terms included)."""
        D = self.D
        d2 = D * D                              # d2[l, i] = D[l, i]^2
        dd = np.diag(D)
        dg = h2 * self.bm1
        for i, a in enumerate(self.axes):
            dg = dg + h1 * _ap(d2.T, self.G[i][i], a)
        dim = self.dim
        if dim == 2:
            dx = dd[None, :]
            dy = dd[:, None]
            dg = dg + h1 * 2.0 * self.G[0][1] * (dx * dy)[None]
        else:
            dx = dd[None, None, :]
            dy = dd[None, :, None]
            dz = dd[:, None, None]
            dg = dg + h1 * 2.0 * (self.G[0][1] * (dx * dy)[None] + self.G[0][2] * (dx * dz)[None]
                                  + self.G[1][2] * (dy * dz)[None])
        return dg

    def _d12(self, u, j):
        """d u / d r_j evaluated on the pressure (GL) mesh."""
        for m, a in enumerate(self.axes):
            u = _ap(self.D12 if m == j else self.I12, u, a)
        return u

    def _d12T(self, p, j):
        for m, a in enumerate(self.axes):
            p = _ap((self.D12 if m == j else self.I12).T, p, a)
        return p

    def opdiv(self, u):
        """B2-weighted divergence on the pressure mesh: sum_i D_i u_i (Nek opdiv / multd)."""
        out = 0.0
        for i in range(self.dim):
            ui = self.f1(u[i])
            for j in range(self.dim):
                out = out + self.rst2w[j][i] * self._d12(ui, j)
        return out

    def opgradt(self, p):
        """Transpose of opdiv: (D_i^T p) on the velocity mesh, element-local (Nek opgradt / cdtp)."""
        p = self.f2(p)
        out = []
        for i in range(self.dim):
            acc = 0.0
            for j in range(self.dim):
                acc = acc + self._d12T(self.rst2w[j][i] * p, j)
            out.append(acc)
        return out

    def opbinv(self, w):
        """mask * binvm1 * dssum(w) per component (Nek opbinv with h2inv = 1)."""
        return [self.mask[i] * self.binvm1 * self.gs(w[i]) for i in range(self.dim)]

    def cdabdtp(self, p):
        """Consistent Poisson operator E p = D (mask B^-1 QQ^T) D^T p (Nek cdabdtp, h2inv = 1)."""
        return self.opdiv(self.opbinv(self.opgradt(p)))

    def e_diag(self):
        """Exact diagonal of E (used for the Jacobi preconditioner of the pressure solve)."""
        # (D_i^T e_k)_q = sum_j g_ji,k M_j(k,q),  M_j(k,q) = prod_m mat_{j,m}[k_m, q_m]
        # diag_k = sum_i sum_q c_i,q (D_i^T e_k)_q^2,  c_i = mask_i * binvm1
        dim = self.dim
        out = np.zeros(self.shape2)
        for i in range(dim):
            c = self.mask[i] * self.binvm1
            for j in range(dim):
                for jj in range(dim):
                    s = c
                    for m, a in enumerate(self.axes):
                        Mj = self.D12 if m == j else self.I12
                        Mk = self.D12 if m == jj else self.I12
                        s = _ap(Mj * Mk, s, a)
                    out += self.rst2w[j][i] * self.rst2w[jj][i] * s
        return out

    # ---- dealiased convection --------------------------------------------------------
    def fine_grad_rst(self, u):
        """(d/dr_j) of the coarse field u evaluated on the fine Gauss mesh."""
        out = []
        for j in range(self.dim):
            v = u
            for m, a in enumerate(self.axes):
                v = _ap(self.DJd if m == j else self.Jd, v, a)
            out.append(v)
        return out

    def conv_weak(self, c, u):
        """Weak dealiased convection  J^T W_d [(c . grad) u]  (Nek convop * bm1 == convect_new)."""
        cf = [self.to_fine(self.f1(ci)) for ci in c]
        ur = self.fine_grad_rst(self.f1(u))
        acc = 0.0
        for j in range(self.dim):
            cr = sum(self.rstdw[j][i] * cf[i] for i in range(self.dim))
            acc = acc + cr * ur[j]
        return self.from_fine_T(acc)

    def scalar_times_grad_weak(self, theta, Theta):
        """Weak dealiased  J^T W_d [theta dTheta/dx_i], i = 1..dim  (the temperature term of the adjoint momentum
        equation; the transpose of the u . grad Theta term of conv_weak)."""
        tf = self.to_fine(self.f1(theta))
        dT = self.fine_grad_rst(self.f1(Theta))
        return [self.from_fine_T(tf * sum(self.rstdw[j][i] * dT[j] for j in range(self.dim))) for i in range(self.dim)]

    def lns_conv_weak(self, U, u, adjoint=False):
        """Weak linearised convective term (reference: neklab_linops.f90:268-313).

        direct : N_i = (U.grad) u_i + (u.grad) U_i
        adjoint: N_i = -(U.grad) u_i + sum_m u_m dU_m/dx_i
        Evaluated on the fine mesh and projected back (B-weighted, not assembled).
        """
        dim = self.dim
        Uf = [self.to_fine(self.f1(a)) for a in U]
        uf = [self.to_fine(self.f1(a)) for a in u]
        dU = [self.fine_grad_rst(self.f1(a)) for a in U]    # dU[i][j] = d U_i / d r_j
        du = [self.fine_grad_rst(self.f1(a)) for a in u]
        Ur = [sum(self.rstdw[j][m] * Uf[m] for m in range(dim)) for j in range(dim)]
        out = []
        if not adjoint:
            ur = [sum(self.rstdw[j][m] * uf[m] for m in range(dim)) for j in range(dim)]
            for i in range(dim):
                acc = sum(Ur[j] * du[i][j] + ur[j] * dU[i][j] for j in range(dim))
                out.append(self.from_fine_T(acc))
        else:
            for i in range(dim):
                acc = -sum(Ur[j] * du[i][j] for j in range(dim))
                # sum_m u_m dU_m/dx_i = sum_m u_m sum_j (J dr_j/dx_i W) dU_m/dr_j
                for m in range(dim):
                    acc = acc + uf[m] * sum(self.rstdw[j][i] * dU[m][j] for j in range(dim))
                out.append(self.from_fine_T(acc))
        return out

    # ---- CFL (Nek compute_cfl) -------------------------------------------------------
    def compute_cfl(self, U, dt):
        dim = self.dim
        U = [self.f1(a) for a in U]
        tot = 0.0
        rd = self.rdr
        for j in range(dim):
            ur = sum(self.rst[j][i] * U[i] for i in range(dim)) / self.jac
            shape = [1] * (dim + 1)
            shape[dim - j] = self.n          # axis x is last
            tot = tot + np.abs(ur * rd.reshape(shape))
        return float(dt * np.max(tot))

    # ---- misc ------------------------------------------------------------------------
    def ortho(self, p):
        """Remove the mean of a pressure-mesh vector (Nek ortho), only without outflow."""
        if self.has_outflow:
            return p
        return p - np.sum(p) / p.size
