"""Snapshot I/O (no GPU): round trips, and that the writer for the BE-parity mode reproduces the structure of the
reference's own VTU files (checked against the decoded fixture)."""
import os

import numpy as np

from pfhubbenchmarks_amd import io as pio


def test_vti_roundtrip(tmp_path):
    f = np.random.default_rng(0).standard_normal((3, 5, 8))
    p = str(tmp_path / "a" / "c000000.vti")
    pio.write_vti(p, f, h=0.5)
    got = pio.read_vtu_pointdata(p)["c"].reshape(f.shape)
    np.testing.assert_array_equal(got, f)
    pio.write_pvd(str(tmp_path / "a" / "c.pvd"), [0.1, 0.3], [p, p])
    assert 'timestep="0.1"' in open(str(tmp_path / "a" / "c.pvd")).read()


def test_crossed_vtu_matches_reference_layout(tmp_path, golden_dir):
    d = np.load(os.path.join(golden_dir, "bm1_fields.npz"))
    pts, tri = pio.crossed_mesh(100, 200.0)
    np.testing.assert_array_equal(pts[:8], d["points_head"][:, :2])
    np.testing.assert_array_equal(pts[10201:10205], d["points_centre_head"][:, :2])
    np.testing.assert_array_equal(tri[:8], d["conn_head"])
    p = str(tmp_path / "conc000000.vtu")
    pio.write_vtu_crossed(p, d["c"][0], 100, 200.0, name="f_13-0")
    got = pio.read_vtu_pointdata(p)["f_13-0"]
    np.testing.assert_allclose(got, d["c"][0], rtol=1e-15, atol=0)


def test_plot_stats(tmp_path, golden_dir):
    pio.plot_stats(os.path.join(golden_dir, "bench1_out.csv"), str(tmp_path / "bench1"))
    assert os.path.getsize(str(tmp_path / "bench1_E.png")) > 1000
    assert os.path.getsize(str(tmp_path / "bench1_C.png")) > 1000
