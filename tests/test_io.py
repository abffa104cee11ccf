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


def test_field_store_and_process_bench1_roundtrip(tmp_path):
    """the per-step dump (counterpart of bench1.py:116-119,190-191) read back by the process_bench1 counterpart
    (process_bench1.py:8-43): names, times and values survive; the re-emitted PVD series carries the stats.csv times"""
    from pfhubbenchmarks_amd import postprocess
    from pfhubbenchmarks_amd.drivers import write_csv
    d = tmp_path / "bench1"
    rng = np.random.default_rng(3)
    fields = [rng.standard_normal((5, 7)) for _ in range(3)]
    times = [0.1, 0.3, 0.7]
    with pio.FieldStore(str(d / "conc.npz"), "w") as st:
        st.write_mesh("grid", h=2.0, shape=(5, 7), L=12.0)
        for f, t in zip(fields, times):
            st.write(f, "c", t)
    write_csv(str(d / "stats.csv"), [[t, 1.0 - t, 2.0] for t in times])
    mesh, tt, cs, stats = postprocess.process_bench1(str(d))
    assert mesh["kind"] == "grid" and mesh["h"] == 2.0 and stats.shape == (3, 3)
    np.testing.assert_array_equal(tt, times)
    for a, b in zip(cs, fields):
        np.testing.assert_array_equal(a, b)
    files = postprocess.write_series(str(d), mesh, tt, cs)
    assert [os.path.basename(f) for f in files] == ["c000000.vti", "c000001.vti", "c000002.vti"]
    np.testing.assert_array_equal(pio.read_vtu_pointdata(files[2])["c"].reshape(5, 7), fields[2])
    pvd = open(str(d / "c.pvd")).read()
    assert 'timestep="0.7"' in pvd and 'file="c000002.vti"' in pvd
    # a store that is shorter than stats.csv, or saved at other times, is refused
    write_csv(str(d / "stats.csv"), [[t, 0.0, 0.0] for t in times + [1.5]])
    import pytest
    with pytest.raises(ValueError):
        postprocess.process_bench1(str(d))


def test_field_store_crossed_mesh_series(tmp_path, golden_dir):
    from pfhubbenchmarks_amd import postprocess
    from pfhubbenchmarks_amd.drivers import write_csv
    g = np.load(os.path.join(golden_dir, "bm1_fields.npz"))
    d = tmp_path / "bench1"
    with pio.FieldStore(str(d / "conc.npz"), "w") as st:
        st.write_mesh("crossed", N=100, L=200.0)
        for i in range(2):
            st.write(g["c"][i], "c", g["times"][i])
    write_csv(str(d / "stats.csv"), [[g["times"][i], 0.0, 0.0] for i in range(2)])
    mesh, tt, cs, _ = postprocess.process_bench1(str(d))
    files = postprocess.write_series(str(d), mesh, tt, cs)
    assert files[1].endswith("c000001.vtu")
    np.testing.assert_allclose(pio.read_vtu_pointdata(files[1])["c"], g["c"][1], rtol=1e-15, atol=0)
