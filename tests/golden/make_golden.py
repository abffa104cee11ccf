#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's committed RESULT files.

Runs only in the build container (needs /root/reference); the GPU box only sees the generated data.
What is copied is *data* (result CSVs, decoded point-data of VTU snapshots), never reference source:

  results/bench1_out.csv, results/bench6_out.csv      -> verbatim (time,total_free_energy,total_solute)
  results/bench1/conc00000{0..5}.vtu  PointData f_13-0     -> bm1_fields.npz  (c at the 20 201 mesh vertices)
  results/bench6/{conc,phi}00000{0..5}.vtu  f_3357-{0,2}   -> bm6_fields.npz  (c, phi)
  results/bench{1,6}/conc.pvd                              -> frame times inside the npz files
  results/bench2_out.csv, results/bench3_out.csv      -> verbatim (BM2: ..., total_solute; BM3: ..., solid_fraction)
  results/bench2/{conc,eta1..eta4}00000{0..3}.vtu          -> bm2_fields.npz  (c and the four order parameters, 4 frames)
  results/bench3: only the .pvd index files are committed there (no field data to decode)

VTU layout (VTK XML UnstructuredGrid): bench1 files are zlib-compressed base64 ("binary"), header
UInt32 [nblocks, blocksize, last_blocksize, csize_1..csize_nblocks]; bench6 files are ASCII.
Also stores the first triangles' connectivity so the oracle's mesh numbering can be checked.
"""
import base64
import os
import re
import shutil
import sys
import zlib

import numpy as np

REF = os.environ.get("PF_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _data_arrays(text):
    """Yield (attrs dict, body string) for every <DataArray ...>body</DataArray>."""
    for m in re.finditer(r"<DataArray\s+([^>]*)>(.*?)</DataArray>", text, flags=re.S):
        attrs = dict(re.findall(r'(\w+)="([^"]*)"', m.group(1)))
        yield attrs, m.group(2)


_DT = {"Float64": np.float64, "UInt32": np.uint32, "UInt8": np.uint8, "Int32": np.int32, "Float32": np.float32}


def _decode_binary(body, dtype):
    body = body.strip()
    # header: first the 3 fixed words to learn nblocks
    first = base64.b64decode(body[:16])  # 12 bytes -> 16 chars
    nblocks = int(np.frombuffer(first[:4], dtype=np.uint32)[0])
    hbytes = 4 * (3 + nblocks)
    hchars = 4 * ((hbytes + 2) // 3)
    header = np.frombuffer(base64.b64decode(body[:hchars])[:hbytes], dtype=np.uint32)
    csizes = header[3:]
    payload = base64.b64decode(body[hchars:])
    out = bytearray()
    off = 0
    for cs in csizes:
        out += zlib.decompress(payload[off:off + int(cs)])
        off += int(cs)
    return np.frombuffer(bytes(out), dtype=dtype)


def read_vtu(path):
    with open(path, "r") as f:
        text = f.read()
    compressed = "vtkZLibDataCompressor" in text[:400]
    arrays = {}
    order = []
    for attrs, body in _data_arrays(text):
        dt = _DT[attrs["type"]]
        if attrs.get("format") == "binary":
            assert compressed
            a = _decode_binary(body, dt)
        else:
            a = np.array(body.split(), dtype=dt)
        name = attrs.get("Name", "Points" if not order else None)
        if name is None:
            name = "arr%d" % len(order)
        arrays[name] = a
        order.append(name)
    return arrays


def pvd_times(path):
    with open(path) as f:
        return np.array([float(t) for t in re.findall(r'timestep="([^"]+)"', f.read())])


def main():
    res = os.path.join(REF, "results")
    for name in ("bench1_out.csv", "bench6_out.csv", "bench2_out.csv", "bench3_out.csv"):
        shutil.copyfile(os.path.join(res, name), os.path.join(OUT, name))
        os.chmod(os.path.join(OUT, name), 0o644)

    # ---- BM1 fields
    t1 = pvd_times(os.path.join(res, "bench1", "conc.pvd"))
    frames = []
    for i in range(6):
        a = read_vtu(os.path.join(res, "bench1", "conc%06d.vtu" % i))
        pd = [k for k in a if k.startswith("f_")]
        assert len(pd) == 1, pd
        frames.append(a[pd[0]].astype(np.float64))
        if i == 0:
            pts = a["Points"].reshape(-1, 3)
            conn = a["connectivity"].reshape(-1, 3)
    frames = np.stack(frames)
    assert frames.shape == (6, 20201)
    np.savez_compressed(os.path.join(OUT, "bm1_fields.npz"), times=t1[:6], c=frames,
                        points_head=pts[:8], points_centre_head=pts[10201:10205],
                        conn_head=conn[:8].astype(np.int64))

    # ---- BM6 fields (first 6 frames are the ones that coincide with the CSV's time grid)
    t6 = pvd_times(os.path.join(res, "bench6", "conc.pvd"))
    cs, ps = [], []
    for i in range(6):
        a = read_vtu(os.path.join(res, "bench6", "conc%06d.vtu" % i))
        pd = [k for k in a if k.startswith("f_")]
        assert len(pd) == 1, pd
        cs.append(a[pd[0]].astype(np.float64))
        b = read_vtu(os.path.join(res, "bench6", "phi%06d.vtu" % i))
        pd = [k for k in b if k.startswith("f_")]
        assert len(pd) == 1, pd
        ps.append(b[pd[0]].astype(np.float64))
    np.savez_compressed(os.path.join(OUT, "bm6_fields.npz"), times=t6[:6], c=np.stack(cs), phi=np.stack(ps))
    # ---- BM2 fields: c, eta1..eta4 of the first 4 frames (the frame times are the CSV's rows 0..3)
    t2 = pvd_times(os.path.join(res, "bench2", "conc.pvd"))
    fields = {}
    for stem, key in (("conc", "c"), ("eta1", "eta1"), ("eta2", "eta2"), ("eta3", "eta3"), ("eta4", "eta4")):
        fr = []
        for i in range(4):
            a = read_vtu(os.path.join(res, "bench2", "%s%06d.vtu" % (stem, i)))
            pd = [k for k in a if k.startswith("f_")]
            assert len(pd) == 1, pd
            fr.append(a[pd[0]].astype(np.float64))
        fields[key] = np.stack(fr)
        assert fields[key].shape == (4, 20201)
    np.savez_compressed(os.path.join(OUT, "bm2_fields.npz"), times=t2[:4], **fields)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    sys.exit(main())
