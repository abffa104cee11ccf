"""In-process A/B of environment-selected forms of the 512^3 spectral step (or BM6 FD + Poisson): one handle per form,
created, timed and destroyed in turn, the whole list run ROUNDS times (the allocator hands consecutive handles of one
process the same block, so the forms see the same placement; PFHIP_SPEC_PROBE=0 unless a form sets it).
Usage: python tools/spectral_env_ab.py "NAME:K=V;K=V" "NAME2:..." [--rounds 2] [--model bm1|bm6] [--check]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pfhubbenchmarks_amd.solver import PhaseFieldSolver

ap = argparse.ArgumentParser()
ap.add_argument("forms", nargs="+")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--model", default="bm1")
ap.add_argument("--n", default="512", help="points per axis: N or NX,NY,NZ")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--dummy-streams", type=int, default=0, help="create this many extra HIP streams before every handle (shifts the stream -> hardware queue assignment)")
ap.add_argument("--check", action="store_true", help="compare the field after 5 steps with the first form's")
a = ap.parse_args()
forms = []
for f in a.forms:
    name, _, kv = f.partition(":")
    forms.append((name, dict(x.split("=", 1) for x in kv.split(";") if x)))
allkeys = {k for _, e in forms for k in e}
ref = None
for rnd in range(a.rounds):
    for name, env in forms:
        for k in allkeys:
            os.environ.pop(k, None)
        os.environ.update(env)
        scheme = "spectral" if a.model == "bm1" else "fd"
        import torch
        dummies = [torch.cuda.Stream() for _ in range(a.dummy_streams)]
        nn = tuple(int(v) for v in a.n.split(","))
        nn = nn[0] if len(nn) == 1 else nn
        with PhaseFieldSolver(dim=3, n=nn, h=1.0, scheme=scheme, model=a.model) as s:
            (s.set_ic_bm1 if a.model == "bm1" else s.set_ic_bm6)()
            dt = 1e-2 if a.model == "bm1" else 5e-4
            diff = None
            if a.check and rnd == 0:
                s.step(dt, 5)
                c = s.get_c()
                if ref is None:
                    ref = c
                diff = float(np.abs(c - ref).max())
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.4:
                s.step(dt, 20)
                s.sync()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                s.step(dt, a.steps)
                s.sync()
                ts.append((time.perf_counter() - t0) / a.steps * 1e3)
            print("round %d  %-28s %.4f ms/step (min %.4f max %.4f)%s" % (
                rnd, name, sorted(ts)[2], min(ts), max(ts), "" if diff is None else "  max|c - c_first| %.3e" % diff), flush=True)
