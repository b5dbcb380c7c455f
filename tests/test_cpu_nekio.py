"""Nek5000 field-file writer: header layout, round trip through the reader, 2-D and 3-D."""
import numpy as np
import pytest

from neklab_amd import nekio

# header of examples/cylinder/stability/direct/BF_1cyl0.f00001 (data: the 132 header bytes of the reference's own fixture)
REF_HEADER = ("#std 8  6  6  1       1996       1996  0.1000000000000E+01       101      0      1 XUP"
              "         1.0000000E+00 F").ljust(132)


def test_header_matches_reference_fixture(tmp_path):
    n, dim, nel = 6, 2, 1996
    z = np.zeros((nel, n * n))
    p = nekio.write_fld(str(tmp_path / "a.f00001"), n, dim, coords=[z, z], vel=[z, z], p=z, time=1.0, istep=101)
    raw = open(p, "rb").read()
    assert raw[:132].decode() == REF_HEADER
    assert len(raw) == 2882360           # size of the reference fixture: same sections, same widths
    assert abs(np.frombuffer(raw[132:136], dtype=np.float32)[0] - 6.54321) < 1e-6


@pytest.mark.parametrize("dim", [2, 3])
def test_round_trip(tmp_path, dim):
    rng = np.random.default_rng(0)
    n, nel = 5, 7
    f = [rng.standard_normal((nel, n ** dim)) for _ in range(2 * dim + 2)]
    elmap = rng.permutation(nel).astype(np.int32) + 1
    path = nekio.write_fld(str(tmp_path / "b.f00003"), n, dim, coords=f[:dim], vel=f[dim:2 * dim], p=f[2 * dim],
                           t=f[2 * dim + 1], time=-12.5e-3, istep=3, elmap=elmap)
    d = nekio.read_fld(path)
    assert (d["n"], d["nel"], d["dim"]) == (n, nel, dim)
    assert d["time"] == -12.5e-3
    assert np.array_equal(d["elmap"], elmap)
    names = (["x", "y", "z"][:dim], ["ux", "uy", "uz"][:dim])
    for k, nm in enumerate(names[0]):
        assert np.array_equal(d[nm], f[k])
    for k, nm in enumerate(names[1]):
        assert np.array_equal(d[nm], f[dim + k])
    assert np.array_equal(d["p"], f[2 * dim]) and np.array_equal(d["t"], f[2 * dim + 1])
    if dim == 3:    # min/max metadata: float32 pairs per element and component after the data
        raw = open(path, "rb").read()
        nmeta = 4 * 2 * nel * (2 * dim + 2)
        meta = np.frombuffer(raw[-nmeta:], dtype=np.float32).reshape(-1, 2)
        assert np.all(meta[:, 0] <= meta[:, 1])
        assert np.isclose(meta[0, 0], f[0][0].min(), rtol=1e-6)


def test_velocity_only_and_errors(tmp_path):
    z = np.ones((3, 16))
    d = nekio.read_fld(nekio.write_fld(str(tmp_path / "c.f00001"), 4, 2, vel=[z, 2 * z]))
    assert "x" not in d and np.array_equal(d["uy"], 2 * z)
    with pytest.raises(ValueError):
        nekio.write_fld(str(tmp_path / "d.f00001"), 4, 2)
    with pytest.raises(ValueError):
        nekio.write_fld(str(tmp_path / "e.f00001"), 4, 2, vel=[z])


# ---- .re2 geometry (curved sides) and .ma2 connectivity, on the reference's own mesh (tests/golden/reference_cyl_mesh.npz) ----
def test_re2_geometry_rebuilds_the_reference_coordinates():
    """GLL coordinates rebuilt from the vertices and the 80 circular-arc records of 1cyl.re2 against the coordinates in
    the reference's base-flow file.  The file's coordinates carry float32 precision only (79 % of them are exactly
    float32-representable doubles: the field file was written from single-precision mesh data), so agreement is at the
    float32 level -- 1.9e-6 at |x| <= 50 -- not 1e-12; what pins the arc construction is the curved elements agreeing as
    well as the straight ones, and the points of the cylinder surface lying on r = 0.5 to rounding."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refdata import HERE, load_cylinder_mesh
    xc, yc, curves, vert, pmap = load_cylinder_mesh()
    d = np.load(os.path.join(HERE, "golden", "reference_cyl_baseflow.npz"))
    n = int(d["n"])
    X, Y = nekio.re2_gll_coords(xc, yc, curves, n)
    em = d["elmap"] - 1
    assert np.mean(d["x"].astype(np.float32).astype(np.float64) == d["x"]) > 0.75          # float32-quantised file coordinates
    assert np.abs(X[em] - d["x"]).max() < 4e-6 and np.abs(Y[em] - d["y"]).max() < 4e-6
    cur = np.zeros(len(xc), dtype=bool)
    cur[[c[0] - 1 for c in curves]] = True
    assert np.abs(X[em] - d["x"])[cur[em]].max() < 4e-7                                   # |x| < 1 there: float32 eps is smaller
    # the arcs: edge points of the radius -0.5 sides lie on the cylinder |r| = 0.5
    for ie, isid, par, typ in curves:
        if abs(abs(par[0]) - 0.5) < 1e-12:
            nodes = nekio.face_nodes(n, 2, isid)
            r = np.hypot(X[ie - 1, nodes], Y[ie - 1, nodes])
            assert np.max(np.abs(r - 0.5)) < 5e-8, (ie, isid, r)                           # vertices are float32 numbers
            mid = nodes[1:-1]
            # the arc itself (through the two float32 vertices) is exact: all points at one distance from the fitted centre
            assert np.ptp(np.hypot(X[ie - 1, nodes], Y[ie - 1, nodes])) < 5e-8
    # positive Jacobian everywhere (cross product of the element's bilinear tangents at the corners is enough here)
    Xe, Ye = X.reshape(-1, n, n), Y.reshape(-1, n, n)
    dxr, dyr = np.gradient(Xe, axis=2), np.gradient(Ye, axis=2)
    dxs, dys = np.gradient(Xe, axis=1), np.gradient(Ye, axis=1)
    assert np.all(dxr * dys - dxs * dyr > 0)


def test_ma2_connectivity_equals_coincident_coordinates():
    """Global numbering of the GLL points built from the global vertex ids of 1cyl.ma2 (Nek5000 set_vert) against the
    numbering from coincident coordinates that round 1 used: the same partition of the 71 856 points into 50 089 dofs,
    the periodic faces y = -16 / y = +16 included (genmap has already identified their vertices)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refdata import load_cylinder, load_cylinder_mesh
    xc, yc, curves, vert, pmap = load_cylinder_mesh()
    hm = load_cylinder(with_bcs=True)[0]
    hm2 = load_cylinder(with_bcs=True, geometry="re2")[0]
    a, b = hm.glo_num.ravel(), hm2.glo_num.ravel()
    assert len(np.unique(a)) == len(np.unique(b)) == 50089
    assert len(set(zip(a.tolist(), b.tolist()))) == 50089          # a bijection between the two label sets
    assert vert.shape == (1996, 4) and vert.min() == 1 and vert.max() == 2033
    # the bisection keys partition the mesh into balanced, contiguous pieces for power-of-two rank counts
    for nr in (2, 4, 8):
        part = nekio.partition_from_ma2(pmap, nr)
        cnt = np.bincount(part, minlength=nr)
        assert cnt.min() >= 1996 // nr - 1 and cnt.max() <= 1996 // nr + 2


def test_glo_num_from_vertices_3d():
    from neklab_amd.mesh import box_mesh
    n = 4
    hm = box_mesh((3, 2, 2), n, periodic=(True, False, False), deform=0.0)
    G = hm.glo_num.reshape(hm.E, n, n, n)
    vert = np.stack([G[:, ck * (n - 1), cj * (n - 1), ci * (n - 1)] for ck in (0, 1) for cj in (0, 1) for ci in (0, 1)], 1) + 1
    g3 = nekio.glo_num_from_vertices(vert, n, 3)
    assert len(set(zip(g3.ravel().tolist(), hm.glo_num.ravel().tolist()))) == len(np.unique(hm.glo_num)) == len(np.unique(g3))


def test_re2_reader_on_the_reference_files_if_present():
    import os
    D = "/root/reference/examples/cylinder/stability/direct/"
    if not os.path.exists(D + "1cyl.re2"):
        pytest.skip("reference tree not present (GPU box)")
    r, m = nekio.read_re2(D + "1cyl.re2"), nekio.read_ma2(D + "1cyl.ma2")
    assert (r["nel"], r["dim"], len(r["curves"]), len(r["bcs"])) == (1996, 2, 80, 208)
    assert {c[3] for c in r["curves"]} == {"C"} and {b[2] for b in r["bcs"]} == {"v", "W", "O", "P"}
    assert (m["nel"], m["depth"], m["nrank"]) == (1996, 10, 1024) and m["vert"].shape == (1996, 4)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from refdata import load_cylinder_mesh
    xc, yc, curves, vert, pmap = load_cylinder_mesh()
    assert np.array_equal(xc, r["xc"]) and np.array_equal(vert, m["vert"]) and np.array_equal(pmap, m["pmap"])


def test_pressure_from_mesh1_is_exact_for_polynomials():
    """host.pressure_from_mesh1 (what Nek5000's restart does with the mesh-1 pressure of a field file): tensor interpolation
    GLL -> GL inside every element, exact for polynomials of degree lx1 - 1 in the reference coordinates."""
    from types import SimpleNamespace
    from neklab_amd import host
    from neklab_amd.mesh import gll_points
    n, dim, E = 6, 2, 3
    z1 = gll_points(n)
    z2 = np.polynomial.legendre.leggauss(n - 2)[0]
    f = lambda r, s, e: (1.0 + e) * (r ** 5 - 0.3 * r ** 2 * s ** 3 + s ** 4 - 0.7)
    p1 = np.stack([f(z1[None, :], z1[:, None], e) for e in range(E)])          # [e][j][i]
    ref = np.stack([f(z2[None, :], z2[:, None], e) for e in range(E)])
    mesh = SimpleNamespace(host=SimpleNamespace(n=n, dim=dim, E=E))
    out = host.pressure_from_mesh1(mesh, p1).reshape(E, n - 2, n - 2)
    assert np.max(np.abs(out - ref)) < 1e-13
