"""Synthetic spectral-element box meshes (input generator, not part of the hot path).

Produces exactly the arrays a Nek5000 host would hand across the C-ABI
(`include/neklab_gpu.h: nlg_mesh_create`): GLL point coordinates `xm1, ym1, zm1`
in Nek5000's element-major layout `ijke = ix + n*(iy + n*(iz + n*e))`
(reference: src/vectors/real_vectors.f90:69), the global vertex numbering that
Nek5000 keeps in `glo_num` (consumed by its gather-scatter), and the
Dirichlet masks `v1mask, v2mask, v3mask` (reference: real_vectors.f90:105
`bcdirvc(..., v1mask, v2mask, v3mask)`).

SURVEY.md §8(d) prescribes the synthetic benchmark input: structured `Ex*Ey*Ez` box,
smooth sinusoidal deformation (5 % of the element size) so that all metric factors are
non-trivial, walls Dirichlet-masked, optional periodic directions.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Sequence

import numpy as np


def gll_points(n: int) -> np.ndarray:
    """Gauss-Lobatto-Legendre nodes on [-1, 1] (n points). Newton on (1-x^2) P'_{n-1}."""
    if n < 2:
        raise ValueError("need n >= 2")
    N = n - 1
    x = -np.cos(np.pi * np.arange(n) / N)
    for _ in range(100):
        # Legendre recurrence up to degree N
        p0 = np.ones_like(x)
        p1 = x.copy()
        for k in range(2, N + 1):
            p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
        # p1 = P_N, p0 = P_{N-1};  f = (1-x^2) P_N' = N (P_{N-1} - x P_N)
        f = N * (p0 - x * p1)
        # f' = -N (N+1) P_N
        df = -N * (N + 1) * p1
        dx = f / df
        dx[0] = 0.0
        dx[-1] = 0.0
        x = x - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    x[0], x[-1] = -1.0, 1.0
    x = 0.5 * (x - x[::-1])  # enforce antisymmetry
    return x


@dataclass
class BoxMesh:
    """Host-side description of one (possibly partitioned) spectral-element mesh."""

    dim: int
    n: int                      # GLL points per direction (lx1)
    nel: tuple                  # elements per direction
    x: np.ndarray               # (E, n**dim) float64
    y: np.ndarray
    z: np.ndarray | None
    glo_num: np.ndarray         # (E, n**dim) int64, global (assembled) dof label, 0-based
    mask: list                  # [v1mask, v2mask, (v3mask)] each (E, n**dim) float64 in {0,1}
    tmask: np.ndarray           # scalar (temperature) Dirichlet mask
    periodic: tuple = ()
    lengths: tuple = ()
    has_outflow: bool = False   # if False the pressure has a constant null space
    elem_gid: np.ndarray | None = None   # (E,) global element id (Nek: lglel), 0-based
    extra: dict = field(default_factory=dict)

    @property
    def E(self) -> int:
        return self.x.shape[0]

    @property
    def npts(self) -> int:
        return self.n ** self.dim

    def take(self, elems: np.ndarray) -> "BoxMesh":
        """Sub-mesh holding the listed elements (element-wise partition)."""
        elems = np.asarray(elems)
        return BoxMesh(
            dim=self.dim, n=self.n, nel=self.nel,
            x=np.ascontiguousarray(self.x[elems]), y=np.ascontiguousarray(self.y[elems]),
            z=None if self.z is None else np.ascontiguousarray(self.z[elems]),
            glo_num=np.ascontiguousarray(self.glo_num[elems]),
            mask=[np.ascontiguousarray(m[elems]) for m in self.mask],
            tmask=np.ascontiguousarray(self.tmask[elems]),
            periodic=self.periodic, lengths=self.lengths, has_outflow=self.has_outflow,
            elem_gid=(np.arange(self.E)[elems] if self.elem_gid is None else self.elem_gid[elems]),
        )


def box_mesh(nel: Sequence[int], n: int, lengths: Sequence[float] | None = None,
             periodic: Sequence[bool] | None = None, deform: float = 0.05,
             origin: Sequence[float] | None = None, outflow_xmax: bool = False,
             last_range: Sequence[int] | None = None, ranges: Sequence[Sequence[int]] | None = None) -> BoxMesh:
    """Structured, smoothly deformed box of `prod(nel)` elements with `n` GLL points/direction.

    Elements are numbered lexicographically (x fastest) so that contiguous element blocks are
    spatially compact slabs.  Non-periodic boundaries are no-slip walls (all velocity masks 0)
    except `outflow_xmax`, which leaves the x-max face natural (Nek 'O').
    `last_range=(k0, k1)` generates only the element layers k0 <= k < k1 of the LAST direction, with the
    labels, element ids, coordinates and deformation of the full box: the element block one rank owns
    under the contiguous block distribution, without ever building the global mesh.
    `ranges=((i0, i1), (j0, j1), (k0, k1))` generates the sub-box of elements i0 <= i < i1, ... in every direction in the same way:
    one part of a recursive-coordinate-bisection partition (`rcb_boxes`).
    """
    dim = len(nel)
    assert dim in (2, 3)
    nel = tuple(int(e) for e in nel)
    lengths = tuple(float(l) for l in (lengths if lengths is not None else [float(e) for e in nel]))
    periodic = tuple(bool(p) for p in (periodic if periodic is not None else [False] * dim))
    origin = tuple(float(o) for o in (origin if origin is not None else [0.0] * dim))
    xi = gll_points(n)
    N = n - 1
    if ranges is None:
        k0, k1 = (0, nel[-1]) if last_range is None else (int(last_range[0]), int(last_range[1]))
        ranges = tuple((0, nel[d]) for d in range(dim - 1)) + ((k0, k1),)
    else:
        assert last_range is None and len(ranges) == dim
    ranges = tuple((int(a), int(b)) for a, b in ranges)
    for d in range(dim):
        assert 0 <= ranges[d][0] < ranges[d][1] <= nel[d], ranges
    nloc = tuple(b - a for a, b in ranges)
    E = int(np.prod(nloc))

    # 1-D global grid indices and undeformed coordinates per direction
    gidx, coord, ngrid = [], [], []
    for d in range(dim):
        h = lengths[d] / nel[d]
        e = np.arange(ranges[d][0], ranges[d][1])
        gi = e[:, None] * N + np.arange(n)[None, :]              # (nel_d, n)
        c = origin[d] + h * (e[:, None] + 0.5 * (xi[None, :] + 1.0))
        ng = nel[d] * N + (0 if periodic[d] else 1)
        if periodic[d]:
            gi = gi % ng
        gidx.append(gi)
        coord.append(c)
        ngrid.append(ng)

    # element lexicographic numbering, x fastest; point numbering ix fastest
    if dim == 2:
        ey, ex = np.meshgrid(np.arange(nloc[1]), np.arange(nloc[0]), indexing="ij")
        ex, ey = ex.ravel(), ey.ravel()
        X = np.broadcast_to(coord[0][ex][:, None, :], (E, n, n))
        Y = np.broadcast_to(coord[1][ey][:, :, None], (E, n, n))
        GI = np.broadcast_to(gidx[0][ex][:, None, :], (E, n, n))
        GJ = np.broadcast_to(gidx[1][ey][:, :, None], (E, n, n))
        glo = GI.astype(np.int64) + ngrid[0] * GJ.astype(np.int64)
        coords0 = [np.array(X, dtype=np.float64), np.array(Y, dtype=np.float64)]
        G = [GI, GJ]
    else:
        ez, ey, ex = np.meshgrid(np.arange(nloc[2]), np.arange(nloc[1]), np.arange(nloc[0]), indexing="ij")
        ex, ey, ez = ex.ravel(), ey.ravel(), ez.ravel()
        X = np.broadcast_to(coord[0][ex][:, None, None, :], (E, n, n, n))
        Y = np.broadcast_to(coord[1][ey][:, None, :, None], (E, n, n, n))
        Z = np.broadcast_to(coord[2][ez][:, :, None, None], (E, n, n, n))
        GI = np.broadcast_to(gidx[0][ex][:, None, None, :], (E, n, n, n))
        GJ = np.broadcast_to(gidx[1][ey][:, None, :, None], (E, n, n, n))
        GK = np.broadcast_to(gidx[2][ez][:, :, None, None], (E, n, n, n))
        glo = GI.astype(np.int64) + ngrid[0] * (GJ.astype(np.int64) + ngrid[1] * GK.astype(np.int64))
        coords0 = [np.array(X, dtype=np.float64), np.array(Y, dtype=np.float64), np.array(Z, dtype=np.float64)]
        G = [GI, GJ, GK]

    # Dirichlet masks: zero on non-periodic domain faces
    wall = np.zeros(glo.shape, dtype=bool)
    for d in range(dim):
        if periodic[d]:
            continue
        lo = G[d] == 0
        hi = G[d] == ngrid[d] - 1
        if d == 0 and outflow_xmax:
            wall |= lo
        else:
            wall |= lo | hi
    m = np.where(wall, 0.0, 1.0).reshape(E, -1)

    # smooth deformation, periodic and vanishing on the domain boundary
    coords = [c.copy() for c in coords0]
    if deform != 0.0:
        ph = [2.0 * np.pi * (coords0[d] - origin[d]) / lengths[d] for d in range(dim)]
        s = np.ones_like(coords0[0])
        for d in range(dim):
            s = s * np.sin(ph[d])
        for d in range(dim):
            h = lengths[d] / nel[d]
            # different phase pattern per direction so that cross metric terms appear
            sd = s * (1.0 + 0.5 * np.cos(ph[(d + 1) % dim]))
            coords[d] = coords0[d] + deform * h * sd
    flat = [np.ascontiguousarray(c.reshape(E, -1)) for c in coords]
    mesh = BoxMesh(
        dim=dim, n=n, nel=nel, x=flat[0], y=flat[1], z=(flat[2] if dim == 3 else None),
        glo_num=np.ascontiguousarray(glo.reshape(E, -1)),
        mask=[m.copy() for _ in range(dim)], tmask=m.copy(),
        periodic=periodic, lengths=lengths, has_outflow=bool(outflow_xmax),
        elem_gid=_global_element_ids(nel, ranges),
    )
    return mesh


def _global_element_ids(nel, ranges) -> np.ndarray:
    """Lexicographic (x fastest) ids in the full box of the elements of a sub-box, in the sub-box's own lexicographic order."""
    dim = len(nel)
    ax = [np.arange(a, b, dtype=np.int64) for a, b in ranges]
    if dim == 2:
        return (ax[0][None, :] + nel[0] * ax[1][:, None]).ravel()
    return (ax[0][None, None, :] + nel[0] * (ax[1][None, :, None] + nel[1] * ax[2][:, None, None])).ravel()


def rcb_boxes(nel: Sequence[int], nparts: int) -> list:
    """Recursive coordinate bisection of a structured box of elements into `nparts` sub-boxes (what genmap's
    bisection gives on a box): cut the longest direction (in elements) where the element count splits in the ratio of the part
    counts floor(n/2) : ceil(n/2), recurse.  Returns one ((i0, i1), (j0, j1)[, (k0, k1)]) per part, in bisection order."""
    def rec(box, n):
        if n == 1:
            return [box]
        d = max(range(len(box)), key=lambda q: (box[q][1] - box[q][0], -q))
        a, b = box[d]
        nlo = n // 2
        cut = a + int(round((b - a) * nlo / n))
        cut = min(max(cut, a + 1), b - 1)
        if b - a < 2:
            raise ValueError("rcb_boxes: more parts than elements along every direction")
        lo = tuple((a, cut) if q == d else box[q] for q in range(len(box)))
        hi = tuple((cut, b) if q == d else box[q] for q in range(len(box)))
        return rec(lo, nlo) + rec(hi, n - nlo)
    return rec(tuple((0, int(e)) for e in nel), int(nparts))


def partition_elements(E: int, nranks: int) -> list:
    """Contiguous element blocks per rank (Nek5000-style block distribution)."""
    base, rem = divmod(E, nranks)
    out, start = [], 0
    for r in range(nranks):
        cnt = base + (1 if r < rem else 0)
        out.append(np.arange(start, start + cnt))
        start += cnt
    return out
