"""Field snapshot output and post-processing -- the downstream side of the hot path.

  write_vti / write_pvd     ParaView-readable snapshots of grid fields (the reference writes PVD/VTU per accepted step:
                            dolfin/bench6.py:141-153,227-229; HDF5 in bench1.py:116-119,190-191)
  write_vtu_crossed         nodal field of the BE-parity mode on the reference's own 'crossed' triangulation, same
                            point / cell order and PointData layout as the reference's results/bench1/conc00000N.vtu
  read_vtu_pointdata        reader for both (and for the reference's ASCII / zlib-binary files)
  plot_stats                Python port of stats.jl:17-44 (free energy and normalised solute vs log time)
  FieldStore                per-step field dump, the counterpart of the reference's HDF5File("results/bench1/conc.h5")
                            (bench1.py:116-119: outfile.write(mesh, "mesh"); :190-191: outfile.write(c, "c", t)) with the
                            same dataset names ("c/vector_<i>"), read back by postprocess.process_bench1 like
                            dolfin/process_bench1.py:9-32 reads the HDF5 file.  Container: .npz (h5py is not in this image).
"""
from __future__ import annotations

import base64
import os
import re
import zlib

import numpy as np


def _b64_block(arr):
    """VTK 'binary' DataArray payload, uncompressed: UInt32 byte count header + raw data, base64 together."""
    raw = np.ascontiguousarray(arr).tobytes()
    return base64.b64encode(np.uint32(len(raw)).tobytes() + raw).decode()


def write_vti(path, field, h=1.0, name="c"):
    """2-D (ny, nx) or 3-D (nz, ny, nx) float64 grid field -> VTK ImageData (.vti), point data."""
    f = np.asarray(field, dtype=np.float64)
    if f.ndim == 2:
        f = f[None]
    nz, ny, nx = f.shape
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ext = "0 %d 0 %d 0 %d" % (nx - 1, ny - 1, nz - 1)
    with open(path, "w") as o:
        o.write('<?xml version="1.0"?>\n<VTKFile type="ImageData" version="0.1" byte_order="LittleEndian">\n')
        o.write('<ImageData WholeExtent="%s" Origin="0 0 0" Spacing="%r %r %r">\n<Piece Extent="%s">\n'
                % (ext, h, h, h, ext))
        o.write('<PointData Scalars="%s">\n<DataArray type="Float64" Name="%s" format="binary">\n' % (name, name))
        o.write(_b64_block(f))
        o.write('\n</DataArray>\n</PointData>\n</Piece>\n</ImageData>\n</VTKFile>\n')


def write_pvd(path, times, files):
    """collection file, same shape as the reference's results/bench1/conc.pvd"""
    with open(path, "w") as o:
        o.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1">\n  <Collection>\n')
        for t, f in zip(times, files):
            o.write('    <DataSet timestep="%r" part="0" file="%s" />\n' % (float(t), os.path.basename(f)))
        o.write('  </Collection>\n</VTKFile>\n')


def crossed_mesh(N, L):
    """points (corners i + (N+1) j, then centres) and triangles in the reference's VTU order"""
    h = L / N
    n1 = N + 1
    jj, ii = np.meshgrid(np.arange(n1), np.arange(n1), indexing="ij")
    cj, ci = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    pts = np.concatenate([np.stack([ii.ravel() * h, jj.ravel() * h], 1),
                          np.stack([(ci.ravel() + 0.5) * h, (cj.ravel() + 0.5) * h], 1)])
    sw = (ci + n1 * cj).ravel()
    se, nw, ne = sw + 1, sw + n1, sw + n1 + 1
    ct = n1 * n1 + (ci + N * cj).ravel()
    tri = np.stack([np.stack([sw, se, ct], 1), np.stack([sw, nw, ct], 1), np.stack([se, ne, ct], 1),
                    np.stack([nw, ne, ct], 1)], 1).reshape(-1, 3)
    return pts, tri


def write_vtu_crossed(path, nodal, N=100, L=200.0, name="f"):
    """nodal: (N+1)^2 + N^2 values in the reference's node order -> UnstructuredGrid of 4 N^2 triangles (ASCII)."""
    pts, tri = crossed_mesh(N, L)
    v = np.asarray(nodal, dtype=np.float64).ravel()
    assert v.size == pts.shape[0]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as o:
        o.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid"  version="0.1"  >\n<UnstructuredGrid>\n')
        o.write('<Piece  NumberOfPoints="%d" NumberOfCells="%d">\n<Points>\n' % (pts.shape[0], tri.shape[0]))
        o.write('<DataArray  type="Float64"  NumberOfComponents="3"  format="ascii">')
        o.write("  ".join("%.16g %.16g 0" % (x, y) for x, y in pts))
        o.write('</DataArray>\n</Points>\n<Cells>\n<DataArray  type="UInt32"  Name="connectivity"  format="ascii">')
        o.write("  ".join("%d %d %d" % tuple(t) for t in tri))
        o.write('</DataArray>\n<DataArray  type="UInt32"  Name="offsets"  format="ascii">')
        o.write(" ".join(str(3 * (i + 1)) for i in range(tri.shape[0])))
        o.write('</DataArray>\n<DataArray  type="UInt8"  Name="types"  format="ascii">')
        o.write(" ".join("5" for _ in range(tri.shape[0])))
        o.write('</DataArray>\n</Cells>\n<PointData  Scalars="%s"> \n' % name)
        o.write('<DataArray  type="Float64"  Name="%s"  format="ascii">' % name)
        o.write("  ".join("%.16e" % x for x in v))
        o.write('</DataArray> \n</PointData> \n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n')


_DT = {"Float64": np.float64, "UInt32": np.uint32, "UInt8": np.uint8, "Int32": np.int32, "Float32": np.float32}


def read_vtu_pointdata(path):
    """{name: array} of the PointData arrays of a .vtu/.vti written by this module or by the reference (legacy VTK
    XML: ascii, uncompressed base64, or vtkZLibDataCompressor base64)."""
    text = open(path).read()
    compressed = "vtkZLibDataCompressor" in text[:500]
    pd = re.search(r"<PointData[^>]*>(.*?)</PointData>", text, flags=re.S)
    out = {}
    for m in re.finditer(r"<DataArray\s+([^>]*)>(.*?)</DataArray>", pd.group(1), flags=re.S):
        attrs = dict(re.findall(r'(\w+)="([^"]*)"', m.group(1)))
        dt, body = _DT[attrs["type"]], m.group(2).strip()
        if attrs.get("format") == "binary" and compressed:
            nblocks = int(np.frombuffer(base64.b64decode(body[:16])[:4], dtype=np.uint32)[0])
            hbytes = 4 * (3 + nblocks)
            hchars = 4 * ((hbytes + 2) // 3)
            head = np.frombuffer(base64.b64decode(body[:hchars])[:hbytes], dtype=np.uint32)
            payload, off, buf = base64.b64decode(body[hchars:]), 0, bytearray()
            for cs in head[3:]:
                buf += zlib.decompress(payload[off:off + int(cs)])
                off += int(cs)
            out[attrs.get("Name", "f")] = np.frombuffer(bytes(buf), dtype=dt)
        elif attrs.get("format") == "binary":
            raw = base64.b64decode(body)
            out[attrs.get("Name", "f")] = np.frombuffer(raw[4:], dtype=dt)
        else:
            out[attrs.get("Name", "f")] = np.array(body.split(), dtype=dt)
    return out


def plot_stats(csv_path, out_prefix):
    """stats.jl:17-44 for one benchmark: <out_prefix>_E.png (free energy vs log t), <out_prefix>_C.png (solute
    normalised by its first value, ylim (0, 1.01))."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    d = np.loadtxt(csv_path, delimiter=",", skiprows=1)
    for col, suffix, title, ylim in ((1, "_E", "Total Free Energy vs Time", None),
                                     (2, "_C", "Total Solute vs Time (Normalized)", (0, 1.01))):
        fig, ax = plt.subplots()
        y = d[:, col] if col == 1 else d[:, col] / d[0, col]
        ax.semilogx(d[:, 0], y, color="black", linewidth=2.0)
        ax.set_title(title)
        if ylim:
            ax.set_ylim(*ylim)
        fig.savefig(out_prefix + suffix + ".png")
        plt.close(fig)


class FieldStore:
    """Write-once / read-many container of per-step fields.

    writer:  st = FieldStore(path, "w"); st.write_mesh(kind="grid", h=..., shape=...) ; st.write(c, "c", t) per accepted
             step (dataset "c/vector_<i>", time in "c/time_<i>" -- DOLFIN's HDF5 naming, bench1.py:190-191); st.close()
    reader:  st = FieldStore(path, "r"); st.mesh() -> dict; st.count("c"); st.read("c", i); st.time("c", i)
    The file is a plain numpy .npz (one zip member per dataset), written when the writer closes."""

    def __init__(self, path, mode="r"):
        if mode not in ("r", "w"):
            raise ValueError("FieldStore mode must be 'r' or 'w'")
        self.path, self.mode = path, mode
        self._data, self._count = {}, {}
        if mode == "r":
            with np.load(path, allow_pickle=False) as z:
                self._data = {k: z[k] for k in z.files}
            for k in self._data:
                m = re.match(r"(.+)/vector_(\d+)$", k)
                if m:
                    self._count[m.group(1)] = max(self._count.get(m.group(1), 0), int(m.group(2)) + 1)

    # -- writer
    def write_mesh(self, kind, **meta):
        """kind = "grid" (uniform lattice: h, shape) or "crossed" (the reference's triangulation: N, L)"""
        assert self.mode == "w"
        self._data["mesh/kind"] = np.array(kind)
        for k, v in meta.items():
            self._data["mesh/" + k] = np.asarray(v)

    def write(self, field, name, t):
        assert self.mode == "w"
        i = self._count.get(name, 0)
        self._data["%s/vector_%d" % (name, i)] = np.array(field, dtype=np.float64, copy=True)
        self._data["%s/time_%d" % (name, i)] = np.float64(t)
        self._count[name] = i + 1

    def close(self):
        if self.mode == "w" and self._data is not None:
            os.makedirs(os.path.dirname(os.path.abspath(self.path)), exist_ok=True)
            with open(self.path, "wb") as fh:       # a file object: np.savez would append ".npz" to a bare name
                np.savez(fh, **self._data)
        self._data = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- reader
    def mesh(self):
        return {k[5:]: (v.item() if v.ndim == 0 else v) for k, v in self._data.items() if k.startswith("mesh/")}

    def count(self, name):
        return self._count.get(name, 0)

    def read(self, name, i):
        return self._data["%s/vector_%d" % (name, i)]

    def time(self, name, i):
        return float(self._data["%s/time_%d" % (name, i)])
