"""512^3 spectral step vs where the three half-spectrum arrays sit relative to each other (PFHIP_SPEC_PLACE) and vs the
kernel-form switches, all in ONE process (same clocks, same box).  Usage on the GPU box:
    python tools/spectral_place_ab.py sweep|forms|repeat [n=512]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

what = sys.argv[1] if len(sys.argv) > 1 else "sweep"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
BASE = {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_ROWK": "0", "PFHIP_FFT3D_CWY": "0", "PFHIP_SPEC_PLACE": "0,0",
        "PFHIP_FFT3D_PLANEPAD": "0", "PFHIP_FFT3D_ZBLOCK": "0", "PFHIP_FFT3D_CW": "0", "PFHIP_FFT3D_ZEARLY": "0"}


def run(env, steps=20, blocks=3):
    e = dict(BASE)
    e.update(env)
    os.environ.update(e)
    with PhaseFieldSolver(dim=3, n=N, h=1.0, scheme="spectral") as s:
        s.set_ic_bm1()
        s.step(1e-2, 60)      # pre-heat (~0.15 s)
        s.sync()
        best = []
        for _ in range(blocks):
            t0 = time.perf_counter()
            s.step(1e-2, steps)
            s.sync()
            best.append((time.perf_counter() - t0) / steps * 1e3)
        F = s.diagnostics()[0]
    return sorted(best)[len(best) // 2], F


if what == "sweep":
    offs = [0, 64, 128, 192, 256, 320, 384, 448]
    print("ms/step; rows: ghat offset KB, columns: scratch offset KB", offs, flush=True)
    for q in ("0", "1"):
        print("PFHIP_FFT3D_QUEUE=%s" % q, flush=True)
        for a in offs:
            row = []
            for b in offs:
                ms, _ = run({"PFHIP_SPEC_PLACE": "%d,%d" % (a, b), "PFHIP_FFT3D_QUEUE": q})
                row.append(ms)
            print("%4d  " % a + " ".join("%.3f" % v for v in row), flush=True)
elif what == "fine":
    offs = [0, 4, 8, 16, 24, 32, 40, 48, 56, 64, 96]
    print("ms/step, QUEUE=1; rows: ghat offset KB, columns: scratch offset KB", offs, flush=True)
    for a in offs:
        row = []
        for b in offs:
            ms, _ = run({"PFHIP_SPEC_PLACE": "%d,%d" % (a, b), "PFHIP_FFT3D_QUEUE": "1"})
            row.append(ms)
        print("%4d  " % a + " ".join("%.3f" % v for v in row), flush=True)
elif what == "layout":
    # every layout twice, interleaved (A B C ... A B C ...): a drift of the box shows up as a difference between the rounds
    forms = [{"PFHIP_FFT3D_PLANEPAD": "0"}, {"PFHIP_FFT3D_PLANEPAD": "1"}, {"PFHIP_FFT3D_PLANEPAD": "2"},
             {"PFHIP_FFT3D_PLANEPAD": "3"}, {"PFHIP_FFT3D_PLANEPAD": "5"}, {"PFHIP_FFT3D_PLANEPAD": "9"},
             {"PFHIP_FFT3D_PLANEPAD": "33"}, {"PFHIP_FFT3D_ZBLOCK": "4"}, {"PFHIP_FFT3D_ZBLOCK": "3"},
             {"PFHIP_FFT3D_PLANEPAD": "1", "PFHIP_FFT3D_CW": "8"}, {"PFHIP_FFT3D_PLANEPAD": "3", "PFHIP_FFT3D_CW": "8"},
             {"PFHIP_FFT3D_PLANEPAD": "1", "PFHIP_FFT3D_QUEUE": "1"}]
    res = [[] for _ in forms]
    for rnd in range(3):
        for i, env in enumerate(forms):
            res[i].append(run(env)[0])
    for env, r in zip(forms, res):
        print(" ".join("%.3f" % v for v in r), env, flush=True)
elif what == "base":
    # absolute placement: block start = multiple of 1 GiB + offset
    os.environ["PFHIP_SPEC_ALIGN_MB"] = "1024"
    offs = [0, 2, 8, 32, 64, 128, 192, 256, 320, 384, 448, 512, 576, 636, 640, 704, 768, 832, 896, 960, 1000]
    for o in offs:
        ms, F = run({"PFHIP_SPEC_BASE_MB": str(o)})
        print("%5d MiB  %.3f ms/step" % (o, ms), flush=True)
elif what == "align":
    # does the ABSOLUTE placement of the spectrum block matter?  the same form created again and again (the allocator hands
    # out alternating addresses), then with the block start rounded up to 2 MiB .. 1 GiB
    os.environ["PFHIP_SPECTRAL_VERBOSE"] = "1"
    forms = [{}, {}, {}, {}, {"PFHIP_SPEC_ALIGN_MB": "2"}, {"PFHIP_SPEC_ALIGN_MB": "2"}, {"PFHIP_SPEC_ALIGN_MB": "64"},
             {"PFHIP_SPEC_ALIGN_MB": "64"}, {"PFHIP_SPEC_ALIGN_MB": "1024"}, {"PFHIP_SPEC_ALIGN_MB": "1024"},
             {"PFHIP_FFT3D_QUEUE": "1"}, {"PFHIP_FFT3D_QUEUE": "1"}, {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_SPEC_ALIGN_MB": "1024"},
             {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_SPEC_ALIGN_MB": "1024"}, {"PFHIP_FFT3D_QUEUE": "2"}, {"PFHIP_FFT3D_QUEUE": "2"},
             {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_ROWK": "7"}, {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_ROWK": "7"},
             {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_ROWK": "8"}, {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_ROWK": "8"},
             {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_CW": "4"}, {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_CW": "4"}]
    for env in forms:
        e = dict({"PFHIP_SPEC_ALIGN_MB": "0"}, **env)
        ms, F = run(e)
        print("%.3f ms/step  F=%.9e  %s" % (ms, F, env), flush=True)
elif what == "forms3":
    os.environ["PFHIP_SPEC_PROBE"] = "4"
    forms = [{}, {"PFHIP_FFT3D_NT": "1"}, {"PFHIP_FFT3D_ROWK": "8"}, {"PFHIP_FFT3D_ROWK": "8", "PFHIP_FFT3D_NT": "1"},
             {"PFHIP_FFT3D_ROWK": "1"}, {"PFHIP_FFT3D_QWGS": "1"}]
    res = [[] for _ in forms]
    for rnd in range(3):
        for i, env in enumerate(forms):
            res[i].append(run(dict({"PFHIP_FFT3D_NT": "0", "PFHIP_FFT3D_QWGS": "0"}, **env))[0])
    for env, r in zip(forms, res):
        print(" ".join("%.3f" % v for v in r), env, flush=True)
elif what == "forms2":
    forms = [{}, {"PFHIP_FFT3D_CW": "4"}, {"PFHIP_FFT3D_ZEARLY": "1"}, {"PFHIP_FFT3D_ROWK": "2"}, {"PFHIP_FFT3D_ROWK": "1"},
             {"PFHIP_FFT3D_CWY": "4"}, {"PFHIP_FFT3D_QUEUE": "1"}]
    res = [[] for _ in forms]
    for rnd in range(3):
        for i, env in enumerate(forms):
            res[i].append(run(env)[0])
    for env, r in zip(forms, res):
        print(" ".join("%.3f" % v for v in r), env, flush=True)
elif what == "repeat":
    for env in ({}, {"PFHIP_FFT3D_QUEUE": "1"}, {"PFHIP_FFT3D_ROWK": "2"}, {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_ROWK": "2"}):
        print(env, " ".join("%.3f" % run(env)[0] for _ in range(5)), flush=True)
else:
    forms = [{}, {"PFHIP_FFT3D_QUEUE": "1"}, {"PFHIP_FFT3D_QUEUE": "2"}, {"PFHIP_FFT3D_ROWK": "1"}, {"PFHIP_FFT3D_ROWK": "2"},
             {"PFHIP_FFT3D_ROWK": "3"}, {"PFHIP_FFT3D_ROWK": "6"}, {"PFHIP_FFT3D_QUEUE": "1", "PFHIP_FFT3D_ROWK": "2"}]
    for env in forms:
        ms, F = run(env)
        print("%.3f ms/step  F=%.9e  %s" % (ms, F, env), flush=True)
