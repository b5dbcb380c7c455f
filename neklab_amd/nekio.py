"""Readers / writer for the two Nek5000 binary formats the reference's cases ship:
`.f%05d` field files ("#std" header, as loaded by `load_fld` in examples/cylinder/stability/direct/1cyl.usr:15)
and `.re2` meshes (header "#v002": fp64 records; only the boundary-condition section is needed here because the
field file already carries the GLL coordinates of the curved elements); `write_fld` produces the same "#std" layout
that Nek5000's `outpost` writes for `outpost_dnek` (src/neklab_utils.f90:305-333).  Pure data plumbing, no arithmetic."""
from __future__ import annotations

import numpy as np


def read_fld(path):
    """-> dict(n, nel, elmap (global element ids, 1-based), x, y, [z], ux, uy, [uz], p) ; arrays (nel, n**dim)."""
    raw = open(path, "rb").read()
    hdr = raw[:132].decode().split()
    if hdr[0] != "#std":
        raise ValueError("not a Nek5000 field file: %r" % raw[:16])
    wd, nx, ny, nz, nel = int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4]), int(hdr[5])
    tag = np.frombuffer(raw[132:136], dtype=np.float32)[0]
    if abs(tag - 6.54321) > 1e-5:
        raise ValueError("byte-swapped field file (endian tag %r)" % tag)
    fields = hdr[11]
    dim = 3 if nz > 1 else 2
    off = 136
    elmap = np.frombuffer(raw[off: off + 4 * nel], dtype=np.int32).copy()
    off += 4 * nel
    dt = np.float64 if wd == 8 else np.float32
    npt = nx * ny * nz

    def rd(nc):
        nonlocal off
        a = np.frombuffer(raw[off: off + wd * nel * nc * npt], dtype=dt).reshape(nel, nc, npt).astype(np.float64)
        off += wd * nel * nc * npt
        return a

    out = {"n": nx, "nel": nel, "dim": dim, "elmap": elmap, "time": float(hdr[7])}
    if "X" in fields:
        X = rd(dim)
        out.update(x=X[:, 0], y=X[:, 1])
        if dim == 3:
            out["z"] = X[:, 2]
    if "U" in fields:
        U = rd(dim)
        out.update(ux=U[:, 0], uy=U[:, 1])
        if dim == 3:
            out["uz"] = U[:, 2]
    if "P" in fields:
        out["p"] = rd(1)[:, 0]
    if "T" in fields:
        out["t"] = rd(1)[:, 0]
    return out


def write_fld(path, n, dim, coords=None, vel=None, p=None, t=None, time=0.0, istep=0, elmap=None, nelgt=None,
              fid=0, nfiles=1, wdsize=8):
    """Write one Nek5000 field file ("#std" header as written by mfo_write_hdr, element map, then the X / U / P / T
    groups element by element, component by component; 3-D files end with the float32 min/max metadata records).

    coords, vel: sequences of `dim` arrays (nel, n**dim); p, t: arrays (nel, n**dim) ON THE VELOCITY MESH (Nek5000
    interpolates a Pn-Pn-2 pressure to mesh 1 before writing; see `host.outpost_dnek`)."""
    groups, code = [], ""
    for tag, g in (("X", coords), ("U", vel)):
        if g is not None:
            a = np.stack([np.asarray(c, dtype=np.float64).reshape(-1, n ** dim) for c in g], axis=1)
            if a.shape[1] != dim:
                raise ValueError("write_fld: %s needs %d components" % (tag, dim))
            groups.append(a)
            code += tag
    for tag, g in (("P", p), ("T", t)):
        if g is not None:
            groups.append(np.asarray(g, dtype=np.float64).reshape(-1, 1, n ** dim))
            code += tag
    if not groups:
        raise ValueError("write_fld: nothing to write")
    nel = groups[0].shape[0]
    if any(g.shape[0] != nel for g in groups):
        raise ValueError("write_fld: inconsistent element counts")
    elmap = np.arange(1, nel + 1, dtype=np.int32) if elmap is None else np.asarray(elmap, dtype=np.int32)
    nz = n if dim == 3 else 1
    mant = "%20.13E" % time            # Fortran e20.13 prints 0.ddddE+xx, C prints d.dddE+xx: renormalise
    m, ex = ("%.12E" % time).split("E")
    digits = m.replace("-", "").replace(".", "")
    mant = "%s0.%sE%+03d" % ("-" if time < 0 else "", digits, int(ex) + 1 if float(time) != 0.0 else 0)
    hdr = "#std %1d %2d %2d %2d %10d %10d %20s %9d %6d %6d %-10s%15s %s" % (
        wdsize, n, n, nz, nel, nel if nelgt is None else nelgt, mant, istep, fid, nfiles, code, "1.0000000E+00", "F")
    dt = np.float64 if wdsize == 8 else np.float32
    with open(path, "wb") as f:
        f.write(hdr.ljust(132).encode())
        f.write(np.float32(6.54321).tobytes())
        f.write(elmap.tobytes())
        for g in groups:
            f.write(np.ascontiguousarray(g, dtype=dt).tobytes())
        if dim == 3:
            for g in groups:
                mm = np.stack([g.min(axis=2), g.max(axis=2)], axis=2)      # (nel, nc, 2)
                f.write(np.ascontiguousarray(mm, dtype=np.float32).tobytes())
    return path


def read_re2_bcs(path):
    """-> (nel, dim, list of (global element id 1-based, face 1-based in preprocessor order, tag str)) for the
    first (velocity) boundary-condition section of a "#v002" .re2 file."""
    raw = open(path, "rb").read()
    hdr = raw[:80].decode().split()
    if hdr[0] not in ("#v002", "#v003"):   # v003 = v002 layout, velocity BC section only (Nek5000 reader_re2.f)
        raise ValueError("unsupported .re2 version %r" % hdr[0])
    nel, dim = int(hdr[1]), int(hdr[2])
    if abs(np.frombuffer(raw[80:84], dtype=np.float32)[0] - 6.54321) > 1e-5:
        raise ValueError("byte-swapped .re2")
    off = 84
    nvert = 4 if dim == 2 else 8
    off += 8 * (1 + dim * nvert) * nel                      # group + vertex coordinates
    ncurve = int(np.frombuffer(raw[off: off + 8], dtype=np.float64)[0])
    off += 8 + 64 * ncurve
    nbc = int(np.frombuffer(raw[off: off + 8], dtype=np.float64)[0])
    off += 8
    bcs = []
    for _ in range(nbc):
        rec = raw[off: off + 64]
        off += 64
        v = np.frombuffer(rec[:16], dtype=np.float64)
        bcs.append((int(v[0]), int(v[1]), rec[56:64].decode().strip()))
    return nel, dim, bcs


def face_nodes(n, dim, face):
    """Local point indices of a face in Nek5000's PREPROCESSOR numbering (2-D: 1 = s-, 2 = r+, 3 = s+, 4 = r-;
    3-D: 1 = s-, 2 = r+, 3 = s+, 4 = r-, 5 = t-, 6 = t+)."""
    idx = np.arange(n ** dim).reshape((n,) * dim)           # [k,] j, i
    if dim == 2:
        return {1: idx[0, :], 2: idx[:, n - 1], 3: idx[n - 1, :], 4: idx[:, 0]}[face].ravel()
    return {1: idx[:, 0, :], 2: idx[:, :, n - 1], 3: idx[:, n - 1, :], 4: idx[:, :, 0], 5: idx[0], 6: idx[n - 1]}[face].ravel()


# ---------------------------------------------------------------------------------------------------------------------
# .re2 geometry and .ma2 connectivity: what Nek5000 builds from the mesh files before the first step
# (`genxyz` for the GLL coordinates incl. curved sides, `get_vert` / `set_vert` for the global numbering).  The reference's
# cases ship both files (examples/cylinder/stability/direct/1cyl.re2, 1cyl.ma2).
def read_re2(path):
    """-> dict(nel, dim, xc, yc, [zc] (nel, 2**dim) vertex coordinates in the preprocessor's vertex order, curves = list of
    (element 1-based, edge 1-based, params[5], type), bcs = list of (element, face, tag) of the velocity section)."""
    raw = open(path, "rb").read()
    hdr = raw[:80].decode().split()
    if hdr[0] not in ("#v002", "#v003"):   # v003 = v002 layout, velocity BC section only (Nek5000 reader_re2.f)
        raise ValueError("unsupported .re2 version %r" % hdr[0])
    nel, dim = int(hdr[1]), int(hdr[2])
    if abs(np.frombuffer(raw[80:84], dtype=np.float32)[0] - 6.54321) > 1e-5:
        raise ValueError("byte-swapped .re2")
    off = 84
    nvert = 2 ** dim
    rec = np.frombuffer(raw[off: off + 8 * (1 + dim * nvert) * nel], dtype=np.float64).reshape(nel, 1 + dim * nvert)
    off += 8 * (1 + dim * nvert) * nel
    out = {"nel": nel, "dim": dim}
    if dim == 2:
        out["xc"], out["yc"] = rec[:, 1:5].copy(), rec[:, 5:9].copy()
    else:       # x1-4, y1-4, z1-4, x5-8, y5-8, z5-8
        out["xc"] = np.concatenate([rec[:, 1:5], rec[:, 13:17]], axis=1)
        out["yc"] = np.concatenate([rec[:, 5:9], rec[:, 17:21]], axis=1)
        out["zc"] = np.concatenate([rec[:, 9:13], rec[:, 21:25]], axis=1)
    ncurve = int(np.frombuffer(raw[off: off + 8], dtype=np.float64)[0])
    off += 8
    curves = []
    for _ in range(ncurve):
        r = raw[off: off + 64]
        off += 64
        v = np.frombuffer(r[:56], dtype=np.float64)
        curves.append((int(v[0]), int(v[1]), v[2:7].copy(), chr(r[56])))   # ccurve is character*1: genbox leaves the other seven bytes unset
    out["curves"] = curves
    nbc = int(np.frombuffer(raw[off: off + 8], dtype=np.float64)[0])
    off += 8
    bcs = []
    for _ in range(nbc):
        r = raw[off: off + 64]
        off += 64
        v = np.frombuffer(r[:16], dtype=np.float64)
        bcs.append((int(v[0]), int(v[1]), r[56:64].decode().strip()))
    out["bcs"] = bcs
    # further fields (temperature, passive scalars): same record layout, one section per field
    out["bcs_fields"] = [bcs]
    while off + 8 <= len(raw):
        nbc = int(np.frombuffer(raw[off: off + 8], dtype=np.float64)[0])
        off += 8
        sec = []
        for _ in range(nbc):
            r = raw[off: off + 64]
            off += 64
            v = np.frombuffer(r[:16], dtype=np.float64)
            sec.append((int(v[0]), int(v[1]), r[56:64].decode(errors="replace").strip()))
        out["bcs_fields"].append(sec)
    return out


def re2_gll_coords(xc, yc, curves, n, zc=None):
    """GLL coordinates (nel, n**dim) of every element as Nek5000's `genxyz` builds them: the (bi/tri)linear map of the
    vertices plus, for every curved edge, the difference between the curve and the straight edge blended linearly into
    the element (Gordon-Hall).  Curve types: 'C' circular arc of signed radius params[0] through the two end points of
    the edge (Nek5000 `arcsrf`: centre on the perpendicular bisector, GLL points equidistant in angle), 2-D, or on the
    edges 1-8 of an extruded 3-D element.  Vertex order = preprocessor (counter-clockwise) order."""
    from .mesh import gll_points
    xc, yc = np.asarray(xc, dtype=np.float64), np.asarray(yc, dtype=np.float64)
    nel = xc.shape[0]
    dim = 2 if zc is None else 3
    z = gll_points(n)
    h0, h1 = 0.5 * (1.0 - z), 0.5 * (1.0 + z)
    # 2-D shape functions of the four vertices 1:(-,-) 2:(+,-) 3:(+,+) 4:(-,+), arrays [j, i]
    S = [np.outer(h0, h0), np.outer(h0, h1), np.outer(h1, h1), np.outer(h1, h0)]
    if dim == 2:
        X = sum(S[q][None] * xc[:, q, None, None] for q in range(4))
        Y = sum(S[q][None] * yc[:, q, None, None] for q in range(4))
        Z = None
    else:
        zc = np.asarray(zc, dtype=np.float64)
        lo = [sum(S[q][None] * c[:, q, None, None] for q in range(4)) for c in (xc, yc, zc)]
        hi = [sum(S[q][None] * c[:, 4 + q, None, None] for q in range(4)) for c in (xc, yc, zc)]
        X, Y, Z = [lo[d][:, None] * h0[None, :, None, None] + hi[d][:, None] * h1[None, :, None, None] for d in range(3)]
    # 'm' (midside node) edges: the element is the biquadratic image of its 3 x 3 Lagrangian nodes (Nek5000 `xyzquad`): vertices,
    # edge midpoints (the curve record's point on an 'm' edge, the mean of the end points otherwise), centre node by
    # transfinite blending of the edges (Nek5000 `gh_face_extend`), interpolated from (-1, 0, 1) to the GLL points
    mids = {}
    for ie, isid, par, typ in curves:
        if typ == "m":
            if dim == 3:
                raise NotImplementedError("midside-node edges of 3-D elements")
            mids.setdefault(ie - 1, {})[isid] = (par[0], par[1])
    if mids:
        q0, q1, q2 = 0.5 * z * (z - 1.0), 1.0 - z * z, 0.5 * z * (z + 1.0)      # quadratic Lagrange basis on (-1, 0, 1)
        Q = np.stack([q0, q1, q2], axis=0)                                        # [node, gll point]
        for e, em in mids.items():
            for c, C in ((xc, X), (yc, Y)):
                k = 0 if c is xc else 1
                g = np.zeros((3, 3))                                              # [j, i]
                g[0, 0], g[0, 2], g[2, 2], g[2, 0] = c[e, 0], c[e, 1], c[e, 2], c[e, 3]
                ends = {1: (0, 1), 2: (1, 2), 3: (2, 3), 4: (3, 0)}
                pos = {1: (0, 1), 2: (1, 2), 3: (2, 1), 4: (1, 0)}
                for sd in (1, 2, 3, 4):
                    a, b = ends[sd]
                    g[pos[sd]] = em[sd][k] if sd in em else 0.5 * (c[e, a] + c[e, b])
                g[1, 1] = 0.5 * (g[0, 1] + g[1, 2] + g[2, 1] + g[1, 0]) - 0.25 * (g[0, 0] + g[0, 2] + g[2, 2] + g[2, 0])
                C[e] = Q.T @ g @ Q
    for ie, isid, par, typ in curves:
        e = ie - 1
        if typ == "m":
            continue
        if typ != "C":
            raise NotImplementedError("curve type %r (only circular arcs 'C' are built)" % typ)
        if isid > 8:
            raise NotImplementedError("circular arc on a vertical edge of a 3-D element")
        lvl = 0 if isid <= 4 else 4                       # 3-D: edges 5-8 lie in the top face
        s4 = (isid - 1) % 4 + 1
        a, b = lvl + s4 - 1, lvl + s4 % 4
        p1x, p1y, p2x, p2y = xc[e, a], yc[e, a], xc[e, b], yc[e, b]
        radius = par[0]
        gap = np.hypot(p1x - p2x, p1y - p2y)
        if abs(2.0 * radius) <= gap * 1.00001:
            raise ValueError("re2: radius %g too small for the edge of element %d" % (radius, ie))
        xs, ys = p2y - p1y, p1x - p2x
        xys = np.hypot(xs, ys)
        dth = abs(np.arcsin(0.5 * gap / radius))
        p12x, p12y = 0.5 * (p1x + p2x), 0.5 * (p1y + p2y)
        xcen = p12x - xs / xys * radius * np.cos(dth)
        ycen = p12y - ys / xys * radius * np.cos(dth)
        th0 = np.arctan2(p12y - ycen, p12x - xcen)
        r = z * (-1.0 if radius < 0.0 else 1.0)
        xcr = xcen + abs(radius) * np.cos(th0 + r * dth) - (h0 * p1x + h1 * p2x)
        ycr = ycen + abs(radius) * np.sin(th0 + r * dth) - (h0 * p1y + h1 * p2y)
        if s4 > 2:                                        # edges 3 and 4 run against the local axis
            xcr, ycr = xcr[::-1], ycr[::-1]
        blend = {1: np.outer(h0, np.ones(n)), 2: np.outer(np.ones(n), h1), 3: np.outer(h1, np.ones(n)), 4: np.outer(np.ones(n), h0)}[s4]
        dx = blend * (xcr[None, :] if s4 in (1, 3) else xcr[:, None])
        dy = blend * (ycr[None, :] if s4 in (1, 3) else ycr[:, None])
        if dim == 2:
            X[e] += dx
            Y[e] += dy
        else:
            hz = h0 if lvl == 0 else h1
            X[e] += hz[:, None, None] * dx[None]
            Y[e] += hz[:, None, None] * dy[None]
    out = [X.reshape(nel, -1), Y.reshape(nel, -1)]
    if dim == 3:
        out.append(Z.reshape(nel, -1))
    return out


def read_ma2(path):
    """-> dict(nel, nvert_global, depth, nrank, pmap (nel,) leaf of the recursive-bisection tree per element, vert
    (nel, 2**dim) global vertex ids, 1-based, in Nek5000's SYMMETRIC vertex order (x fastest))."""
    raw = open(path, "rb").read()
    hdr = raw[:132].decode().split()
    if hdr[0] != "#v001":
        raise ValueError("unsupported .ma2 version %r" % hdr[0])
    nel, nactive, depth, d2, npts, nrank, noutflow = (int(v) for v in hdr[1:8])
    if abs(np.frombuffer(raw[132:136], dtype=np.float32)[0] - 6.54321) > 1e-5:
        raise ValueError("byte-swapped .ma2")
    a = np.frombuffer(raw[136:], dtype=np.int32)
    nv = npts // nel
    a = a[: nel * (nv + 1)].reshape(nel, nv + 1)
    return {"nel": nel, "nvert_global": nrank if False else int(a[:, 1:].max()), "depth": depth, "nrank": d2, "pmap": a[:, 0].copy(),
            "vert": a[:, 1:].copy()}


def glo_num_from_vertices(vert, n, dim):
    """Global (assembled) numbering of the GLL points from the global vertex ids of a conforming mesh (Nek5000 `set_vert`):
    vertices by their id, edge-interior points by (the two end vertices, position counted from the smaller id), face-
    interior points (3-D) by (the face's smallest vertex, its two neighbours on the face ordered by id, the two positions),
    element interiors are private.  `vert` is (nel, 2**dim) in the symmetric order (x fastest).  Returns (nel, n**dim)
    int64 labels, 0-based, dense."""
    vert = np.asarray(vert, dtype=np.int64)
    nel = vert.shape[0]
    keys = {}
    lab = np.empty((nel, n ** dim), dtype=np.int64)

    def label(key):
        v = keys.get(key)
        if v is None:
            v = keys[key] = len(keys)
        return v

    rng = range(n)
    for e in range(nel):
        vv = vert[e]

        def vid(ci, cj, ck=0):
            return int(vv[ci + 2 * cj + 4 * ck])

        for p in range(n ** dim):
            i, j, k = p % n, (p // n) % n, (p // (n * n)) if dim == 3 else 0
            idx = (i, j, k)[:dim]
            onb = [q == 0 or q == n - 1 for q in idx]
            nb = sum(onb)
            if nb == dim:                                   # vertex
                c = [int(q == n - 1) for q in idx] + [0] * (3 - dim)
                key = ("v", vid(*c))
            elif nb == dim - 1:                             # edge interior: one free direction
                d = onb.index(False)
                c0 = [int(q == n - 1) for q in idx] + [0] * (3 - dim)
                c1 = list(c0)
                c0[d], c1[d] = 0, 1
                a, b = vid(*c0), vid(*c1)
                pos = idx[d]
                key = ("e", a, b, pos) if a < b else ("e", b, a, n - 1 - pos)
            elif nb == 1:                                   # face interior (3-D): two free directions
                d = onb.index(True)
                f = [q for q in range(3) if q != d]
                side = int(idx[d] == n - 1)
                corners = {}
                for s0 in (0, 1):
                    for s1 in (0, 1):
                        c = [0, 0, 0]
                        c[d], c[f[0]], c[f[1]] = side, s0, s1
                        corners[(s0, s1)] = vid(*c)
                (a0, a1), A = min(corners.items(), key=lambda kv: kv[1])
                B, pb = corners[(1 - a0, a1)], (idx[f[0]] if a0 == 0 else n - 1 - idx[f[0]])
                C, pc = corners[(a0, 1 - a1)], (idx[f[1]] if a1 == 0 else n - 1 - idx[f[1]])
                key = ("f", A, B, pb, C, pc) if B < C else ("f", A, C, pc, B, pb)
            else:
                key = ("i", e, p)
            lab[e, p] = label(key)
        _ = rng
    return lab


def partition_from_ma2(pmap, nranks):
    """Element -> rank from the recursive-bisection keys of a .ma2 file: elements ordered by key (ties by element number),
    then dealt out in contiguous, equally sized pieces -- a rank's elements are one subtree (or a run of neighbouring
    subtrees) of genmap's bisection, which is what keeps its surface small."""
    pmap = np.asarray(pmap)
    order = np.argsort(pmap, kind="stable")
    nel = len(pmap)
    part = np.empty(nel, dtype=np.int64)
    base, rem = divmod(nel, nranks)
    start = 0
    for r in range(nranks):
        cnt = base + (1 if r < rem else 0)
        part[order[start: start + cnt]] = r
        start += cnt
    return part
