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
