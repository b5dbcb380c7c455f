"""Shared loader for the reference-data fixture (tests/golden/reference_cyl_baseflow.npz)."""
import os

import numpy as np

from neklab_amd.mesh import BoxMesh

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cylinder_mesh():
    """(xc, yc, curves, vert, pmap) of the reference's 1cyl.re2 / 1cyl.ma2 (tests/golden/reference_cyl_mesh.npz)."""
    d = np.load(os.path.join(HERE, "golden", "reference_cyl_mesh.npz"))
    curves = [(int(e), int(s), p, "C") for e, s, p in zip(d["curve_elem"], d["curve_edge"], d["curve_par"])]
    return d["xc"], d["yc"], curves, d["vert"], d["pmap"]


def load_cylinder(with_bcs=False, dirichlet_tags=("v", "W"), geometry="fld"):
    """with_bcs=False: connectivity from coincident coordinates only, no masks (operator-level checks).
    with_bcs=True : the boundary conditions of 1cyl.re2 applied: 'v' / 'W' faces Dirichlet-masked, 'O' natural,
    'P' faces (y = -16 <-> y = +16) identified in the global numbering."""
    from neklab_amd.nekio import face_nodes
    d = np.load(os.path.join(HERE, "golden", "reference_cyl_baseflow.npz"))
    n = int(d["n"])
    x, y = d["x"].copy(), d["y"].copy()
    if geometry == "re2":
        # GLL coordinates rebuilt in double precision from the vertices and circular arcs of 1cyl.re2 (Nek5000's genxyz)
        # instead of the field file's, which carry float32 precision only (79 % of them are exactly float32 numbers)
        from neklab_amd.nekio import re2_gll_coords
        xc, yc, curves, vert, _ = load_cylinder_mesh()
        X, Y = re2_gll_coords(xc, yc, curves, n)
        x, y = X[d["elmap"] - 1], Y[d["elmap"] - 1]
    E = x.shape[0]
    yk = y.copy()
    if with_bcs:
        yk = np.where(np.abs(y - 16.0) < 1e-9, -16.0, y)      # periodic identification
    # global numbering from coincident coordinates (the field file carries no connectivity)
    key = np.round(np.stack([x.ravel(), yk.ravel()], 1) / 1e-7).astype(np.int64)
    _, glo = np.unique(key, axis=0, return_inverse=True)
    if geometry == "re2":
        # ... with that geometry the connectivity comes from the global vertex ids of 1cyl.ma2 as in Nek5000 (periodic
        # faces are already identified there); it is the same partition of the points as the coordinate-based one
        from neklab_amd.nekio import glo_num_from_vertices
        glo = glo_num_from_vertices(vert, n, 2)[d["elmap"] - 1]
    ones = np.ones((E, n * n))
    mask = [ones.copy(), ones.copy()]
    if with_bcs:
        pos = {int(g): k for k, g in enumerate(d["elmap"])}    # global element id -> position in the field file
        for ge, fc, tag in zip(d["bc_elem"], d["bc_face"], d["bc_tag"]):
            if str(tag) in dirichlet_tags:
                nodes = face_nodes(n, 2, int(fc))
                for m in mask:
                    m[pos[int(ge)], nodes] = 0.0
    hm = BoxMesh(dim=2, n=n, nel=(E, 1), x=x, y=y, z=None, glo_num=glo.reshape(E, n * n).astype(np.int64),
                 mask=mask, tmask=ones.copy(), has_outflow="O" not in dirichlet_tags, elem_gid=np.arange(E, dtype=np.int64))
    rr = np.hypot(x, y)
    interior = (x > -16 + 1e-6) & (x < 50 - 1e-6) & (np.abs(y) < 16 - 1e-6) & (rr > 0.5 + 1e-6)
    return hm, d["ux"].copy(), d["uy"].copy(), d["p"].copy(), float(d["re"]), int(d["lxd"]), interior


def _glo_from_coords(x, y, tol=1e-7):
    key = np.round(np.stack([x.ravel(), y.ravel()], 1) / tol).astype(np.int64)
    _, glo = np.unique(key, axis=0, return_inverse=True)
    return glo.reshape(x.shape).astype(np.int64)


def load_bfs(with_bcs=False):
    """The base flow of the reference's transient-growth case (tests/golden/reference_bfs_baseflow.npz; rounded backward-facing
    step, E = 2760, lx1 = 6, Re = 600).  with_bcs: the boundary ids of bfs.re2 with the tags bfs.usr:112-115 gives them --
    'W' and 'v' faces Dirichlet for every component, 'SYM' faces (the horizontal lines y = 20 and y = 1, x < -2) Dirichlet
    for the normal component uy only; no outflow face, so the pressure level is free.
    -> (mesh, ux, uy, p, re, lxd, interior mask)"""
    from neklab_amd.nekio import face_nodes
    d = np.load(os.path.join(HERE, "golden", "reference_bfs_baseflow.npz"))
    n = int(d["n"])
    x, y = d["x"].copy(), d["y"].copy()
    E = x.shape[0]
    glo = _glo_from_coords(x, y)
    ones = np.ones((E, n * n))
    mask = [ones.copy(), ones.copy()]
    onb = np.zeros((E, n * n))
    pos = {int(g): k for k, g in enumerate(d["elmap"])}
    tag_of = {int(i): str(t) for i, t in zip(d["id_list"], d["id_tag"])}
    for ge, fc, bid in zip(d["bc_elem"], d["bc_face"], d["bc_id"]):
        nodes = face_nodes(n, 2, int(fc))
        e = pos[int(ge)]
        onb[e, nodes] = 1.0
        if with_bcs:
            tag = tag_of[int(bid)]
            if tag in ("W", "v"):
                mask[0][e, nodes] = 0.0
                mask[1][e, nodes] = 0.0
            elif tag == "SYM":
                assert np.ptp(y[e, nodes]) < 1e-9          # horizontal: the normal component is uy
                mask[1][e, nodes] = 0.0
            else:
                raise ValueError(tag)
    # a mask (and "lies on the boundary") is a property of the global dof: it holds for every copy of the point
    def spread(arr, op, init):
        acc = np.full(int(glo.max()) + 1, init)
        op.at(acc, glo.ravel(), arr.ravel())
        return acc[glo]
    mask = [spread(m, np.minimum, 1.0) for m in mask]
    onb = spread(onb, np.maximum, 0.0)
    hm = BoxMesh(dim=2, n=n, nel=(E, 1), x=x, y=y, z=None, glo_num=glo, mask=mask, tmask=ones.copy(), has_outflow=False,
                 elem_gid=np.arange(E, dtype=np.int64))
    return hm, d["ux"].astype(np.float64), d["uy"].astype(np.float64), d["p"].copy(), float(d["re"]), int(d["lxd"]), onb < 0.5


def load_rayben():
    """The field file the reference ships for its Rayleigh-Benard case (tests/golden/reference_rayben_baseflow.npz; 10 x 4 box,
    lx1 = 10, periodic in x, walls at y = 0, 1; fields X U P T).  -> (mesh, ux, uy, p, t, lxd, prandtl, rayleigh)"""
    d = np.load(os.path.join(HERE, "golden", "reference_rayben_baseflow.npz"))
    n = int(d["n"])
    x, y = d["x"].copy(), d["y"].copy()
    E = x.shape[0]
    xk = np.where(np.abs(x - x.max()) < 1e-9, x.min(), x)      # 'P' faces: x = 0 <-> x = 2.0158
    glo = _glo_from_coords(xk, y)
    wall = (np.abs(y) < 1e-9) | (np.abs(y - 1.0) < 1e-9)       # 'W' / 't' faces of rayBen.re2 (rayBen.box)
    m = np.where(wall, 0.0, 1.0)
    hm = BoxMesh(dim=2, n=n, nel=(E, 1), x=x, y=y, z=None, glo_num=glo, mask=[m.copy(), m.copy()], tmask=m.copy(), has_outflow=False,
                 elem_gid=np.arange(E, dtype=np.int64))
    return hm, d["ux"].copy(), d["uy"].copy(), d["p"].copy(), d["t"].copy(), int(d["lxd"]), float(d["prandtl"]), float(d["rayleigh"])


def load_cylinder_re40_guess():
    """The initial guess of the reference's Newton-Krylov example (examples/cylinder/newton/Re40_fixed_point/BF.fld, fp32, on the mesh
    of the stability case) and the numbers read off the convergence plot shipped with it (tests/golden/reference_cyl_re40_guess.npz)."""
    d = np.load(os.path.join(HERE, "golden", "reference_cyl_re40_guess.npz"))
    return {k: (d[k].astype(np.float64) if d[k].dtype == np.float32 else d[k]) for k in d.files}


def load_tsyphon():
    """The mesh of the reference's temperature-coupled Newton example (examples/thermosyphon/baseflow/tsyphon.re2: the annulus
    1 <= r <= 2, 8 x 32 elements with circular-arc sides, lx1 = 8; velocity 'W' / temperature 't' on both walls, periodic in the angle --
    on the closed ring the periodic faces coincide, so the connectivity follows from the coordinates), with the case parameters and the
    Newton residuals read off the plot shipped with the case.  -> (mesh, dict of parameters)"""
    from neklab_amd.nekio import re2_gll_coords
    d = np.load(os.path.join(HERE, "golden", "reference_tsyphon_mesh.npz"))
    n = int(d["n"])
    curves = [(int(e), int(s), p, "C") for e, s, p in zip(d["curve_elem"], d["curve_edge"], d["curve_par"])]
    x, y = re2_gll_coords(d["xc"], d["yc"], curves, n)
    E = x.shape[0]
    glo = _glo_from_coords(x, y, tol=1e-9)
    r = np.hypot(x, y)
    wall = (np.abs(r - 1.0) < 1e-9) | (np.abs(r - 2.0) < 1e-9)
    assert int(wall.sum()) == 64 * n                      # the 64 'W' / 't' faces of the file
    m = np.where(wall, 0.0, 1.0)
    hm = BoxMesh(dim=2, n=n, nel=(E, 1), x=x, y=y, z=None, glo_num=glo, mask=[m.copy(), m.copy()], tmask=m.copy(), has_outflow=False,
                 elem_gid=np.arange(E, dtype=np.int64))
    return hm, {k: d[k] for k in d.files}
