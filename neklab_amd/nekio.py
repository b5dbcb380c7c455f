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
    if hdr[0] != "#v002":
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
