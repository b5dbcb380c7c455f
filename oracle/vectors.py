"""ORACLE (test infrastructure only) -- numpy restatement of neklab's `nek_dvector` vector space.

Follows, procedure by procedure, /root/reference/src/vectors/real_vectors.f90 and the type
declaration at /root/reference/src/vectors/neklab_vectors.f90:26-50.  The Nek5000 primitives it
calls (`glsc3`, `add2s2`, `cmult`, `copy`, `opdssum`, `dsavg`, `bcdirvc`) are not in the reference
tree; they are restated from their published meaning (PARITY UNPINNED at that level, see
oracle/sem.py header).  Never imported by the product.
"""
from __future__ import annotations

import numpy as np

MASK64 = (1 << 64) - 1


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Stateless 64-bit mixer (Steele/Lea/Flood splitmix64 finaliser), vectorised on uint64."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(MASK64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def u01(seed: int, counter: np.ndarray) -> np.ndarray:
    """Counter-based uniform [0,1): replaces the compiler RNG of real_vectors.f90:71 so that any
    element partition reproduces the same global field."""
    with np.errstate(over="ignore"):
        c = counter.astype(np.uint64)
        x = splitmix64(c ^ splitmix64(np.uint64(seed & MASK64) + np.zeros_like(c)))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def mth_rand(ix, iy, iz, ieg, xl, fcoeff, if3d):
    """reference: src/vectors/neklab_vectors.f90:305-314 (1-based ix, iy, iz, ieg)."""
    r = fcoeff[0] * (ieg + xl[0] * np.sin(xl[1])) + fcoeff[1] * ix * iy + fcoeff[2] * ix
    if if3d:
        r = fcoeff[0] * (ieg + xl[2] * np.sin(r)) + fcoeff[1] * iz * ix + fcoeff[2] * iz
    r = 1.0e3 * np.sin(r)
    r = 1.0e3 * np.sin(r)
    return np.cos(r)


class NekDVector:
    """State container, reference: neklab_vectors.f90:26-36.

    v[i]   : velocity components (dim arrays of (E, n**dim))
    pr     : pressure on the lx2 mesh
    theta  : list of scalar fields (ifto / ifpsco active ones only)
    *_rst  : lorder-1 restart-history copies of every field, `nrst` of them valid
    """

    def __init__(self, sem, nscal=0, lorder=3):
        self.sem = sem
        self.dim = sem.dim
        self.nscal = nscal
        self.lorder = lorder
        self.v = [np.zeros(sem.shape1) for _ in range(sem.dim)]
        self.pr = np.zeros(sem.shape2)
        self.theta = [np.zeros(sem.shape1) for _ in range(nscal)]
        self.v_rst = [[np.zeros(sem.shape1) for _ in range(sem.dim)] for _ in range(lorder - 1)]
        self.pr_rst = [np.zeros(sem.shape2) for _ in range(lorder - 1)]
        self.theta_rst = [[np.zeros(sem.shape1) for _ in range(nscal)] for _ in range(lorder - 1)]
        self.nrst = 0

    # -- helpers
    def main_fields(self):
        return self.v + [self.pr] + self.theta

    def rst_fields(self, irst):
        return self.v_rst[irst] + [self.pr_rst[irst]] + self.theta_rst[irst]

    def copy(self):
        out = NekDVector(self.sem, self.nscal, self.lorder)
        for a, b in zip(out.main_fields(), self.main_fields()):
            a[...] = b
        for r in range(self.lorder - 1):
            for a, b in zip(out.rst_fields(r), self.rst_fields(r)):
                a[...] = b
        out.nrst = self.nrst
        return out

    # -- reference: real_vectors.f90:37-50
    def zero(self):
        for a in self.main_fields():
            a[...] = 0.0
        for r in range(self.lorder - 1):
            for a in self.rst_fields(r):
                a[...] = 0.0
        self.nrst = 0

    # -- reference: real_vectors.f90:125-160 (pressure and the nrst valid history slots are scaled)
    def scal(self, alpha):
        for a in self.main_fields():
            a *= alpha
        for r in range(self.nrst):
            for a in self.rst_fields(r):
                a *= alpha

    # -- reference: real_vectors.f90:162-206
    # Default treatment of the restart history in axpby.  A literal reading of real_vectors.f90:188-192 adds
    # alpha * vec's MAIN field to every history slot (CONSISTENT_RST = False).  With that reading the reference's
    # own known answer is NOT reproduced (cylinder Re = 50: |mu_1| = 1.0194 at dt = 0.01, 1.0177 at dt = 0.005,
    # an O(dt) pollution), whereas combining vec's history slots gives 1.01578, dt-independent, against the
    # published 1.0156 +- 1e-4 (test/neklabTests.py:43-45; scripts/cyl_sens.py).  The default therefore is the
    # variant that reproduces the reference's published output; the literal one stays available.
    CONSISTENT_RST = True

    def axpby(self, alpha, vec, beta, consistent_rst=None):
        """self <- alpha*vec + beta*self on all fields and on the nrst valid history slots of self."""
        if consistent_rst is None:
            consistent_rst = NekDVector.CONSISTENT_RST
        self.scal(beta)
        for a, b in zip(self.main_fields(), vec.main_fields()):
            a += alpha * b
        for r in range(self.nrst):
            src = vec.rst_fields(r) if consistent_rst else vec.main_fields()
            for a, b in zip(self.rst_fields(r), src):
                a += alpha * b

    # -- reference: real_vectors.f90:208-233 (velocity + active scalars, weight bm1, no pressure)
    def dot(self, vec):
        s = 0.0
        for a, b in zip(self.v, vec.v):
            s += self.sem.glsc3(a, b)
        for a, b in zip(self.theta, vec.theta):
            s += self.sem.glsc3(a, b)
        return s

    def norm(self):
        return float(np.sqrt(self.dot(self)))

    # -- reference: real_vectors.f90:235-247
    def get_size(self):
        return (self.dim + self.nscal) * self.sem.lvn + self.sem.lpn

    # -- reference: real_vectors.f90:249-346
    def save_rst(self, vec_rst, irst):
        """irst is 1-based like the reference."""
        if irst >= self.lorder:
            raise ValueError("Cannot save rst fields %d for temporal order %d" % (irst, self.lorder))
        for a, b in zip(self.rst_fields(irst - 1), vec_rst.main_fields()):
            a[...] = b
        self.nrst = max(self.nrst, irst)

    def get_rst(self, vec_rst, irst):
        for a, b in zip(vec_rst.main_fields(), self.rst_fields(irst - 1)):
            a[...] = b

    def has_rst_fields(self):
        return self.nrst > 0

    def clear_rst_fields(self):
        self.nrst = 0

    # -- reference: real_vectors.f90:52-123
    def raw_noise(self, seed, elem_gid, field_id):
        """mth_rand noise for one field, with the counter-based RNG standing in for random_number."""
        sem = self.sem
        n, dim, E = sem.n, sem.dim, sem.E
        npts = n ** dim
        ijk = np.arange(npts)
        ix = (ijk % n) + 1
        iy = ((ijk // n) % n) + 1
        iz = ((ijk // (n * n)) % n) + 1 if dim == 3 else np.ones_like(ijk)
        ieg = (np.asarray(elem_gid, dtype=np.int64) + 1)[:, None]
        gpt = (ieg - 1) * npts + ijk[None, :]
        base = (gpt.astype(np.uint64) * np.uint64(8) + np.uint64(field_id)) * np.uint64(4)
        fc = [u01(seed, base + np.uint64(c)) * 1.0e4 for c in range(3)]
        xl = [sem.X[d].reshape(E, npts) for d in range(dim)]
        r = mth_rand(ix[None, :], iy[None, :], iz[None, :], ieg, xl, fc, dim == 3)
        return r.reshape(sem.shape1)

    def rand(self, ifnorm=False, seed=0, elem_gid=None, raw=None):
        """`raw` (list of per-field noise arrays) may be injected so that the continuity / mask /
        normalisation part can be compared exactly with another implementation."""
        sem = self.sem
        if elem_gid is None:
            elem_gid = np.arange(sem.E)
        nf = self.dim + self.nscal
        if raw is None:
            raw = [self.raw_noise(seed, elem_gid, f) for f in range(nf)]
        for i in range(self.dim):
            self.v[i] += raw[i]
        for m in range(self.nscal):
            self.theta[m] += raw[self.dim + m]
        # opdssum, opcolv(vmult), dsavg, bcdirvc
        for i in range(self.dim):
            a = sem.gs(self.v[i]) * sem.vmult
            a = sem.dsavg(a)
            self.v[i][...] = a * sem.mask[i]
        for m in range(self.nscal):
            a = sem.dsavg(self.theta[m])
            self.theta[m][...] = a * sem.tmask
        if ifnorm:
            self.scal(1.0 / self.norm())
        self.nrst = 0
