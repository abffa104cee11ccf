"""What do the first launches after an idle period pay for?  ("What's weak" #1)

The fused 512^3 step runs ~0.45 ms for the first ~25 launches after the GPU has been idle and ~0.36 ms afterwards.
This tool separates the candidates with the launch timeline (pf_timing_samples), the in-kernel shader clock
(pfk_clock_probe) and the driver's own view (sysfs pp_dpm_* / hwmon, sampled by a host thread):

  A  cold start: 2 s idle -> 80 steps; per-launch durations + sysfs samples on the same time axis
  B  idle gaps:  steady state -> sleep g -> 40 steps, g = 0.001 .. 2 s   (how long does "warm" last?)
  C  2 s idle -> 0.3 s of pfk_stream_copy (another HBM-bound kernel) -> 40 steps   (chip state or kernel state?)
  D  2 s idle -> 0.3 s of fma-only kernel on every CU (no memory traffic) -> 40 steps   (shader clock or memory side?)
  E  2 s idle -> steps interleaved with 30 us clock probes   (shader clock during the ramp)
  F  the copy kernel itself from cold (does it ramp too?)

Usage on the GPU box:  python tools/ramp_probe.py [n=512] > gpurun_out/ramp_probe.log
"""
import ctypes as C
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pfhubbenchmarks_amd import lib as L
from pfhubbenchmarks_amd.solver import PhaseFieldSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
DT = 5e-4


def sysfs_sources():
    """readable clock / power files of card 0"""
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent",
                     "mem_busy_percent", "power_dpm_force_performance_level"):
            p = os.path.join(card, name)
            if os.access(p, os.R_OK):
                out[name] = p
        for hw in glob.glob(os.path.join(card, "hwmon/hwmon*")):
            for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "power1_cap"):
                p = os.path.join(hw, name)
                if os.access(p, os.R_OK):
                    out[name] = p
        if out:
            break
    return out


def read_src(path):
    try:
        with open(path) as f:
            s = f.read()
    except OSError as e:
        return "ERR %s" % e
    if "pp_dpm" in path and "force" not in path:      # "0: 132Mhz\n1: 2100Mhz *"  -> the starred level
        cur = [ln for ln in s.splitlines() if ln.rstrip().endswith("*")]
        return cur[0].strip() if cur else s.strip().replace("\n", " | ")
    return s.strip()


class Sampler(threading.Thread):
    def __init__(self, srcs, period=0.0):
        super().__init__(daemon=True)
        self.srcs, self.period, self.rows, self.stop = srcs, period, [], False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            self.rows.append((t, {k: read_src(p) for k, p in self.srcs.items()}))
            if self.period:
                time.sleep(self.period)


def fmt(a):
    return " ".join("%.3f" % v for v in a)


def steps_timeline(s, k):
    """k steps with per-launch events; returns (durations ms, starts ms) of exactly these launches"""
    s.timing(True)
    s.step(DT, k)
    s.sync()
    d, st = s.timing_samples()
    s.timing(False)
    return d, st


def main():
    lib = L.load()
    srcs = sysfs_sources()
    print("sysfs sources:", {k: v for k, v in srcs.items()})
    for k, p in srcs.items():
        print("  %s = %s" % (k, read_src(p)))
    full = {k: open(p).read().strip().replace("\n", " | ") for k, p in srcs.items() if "pp_dpm" in k}
    print("dpm tables:", full)
    dev = torch.device("cuda", 0)
    cells = n ** 3
    gap = cells + 8192
    blk = torch.ones(gap + cells, dtype=torch.float64, device=dev)
    src, dst = blk[:cells], blk[gap:]
    probe_out = torch.zeros(2 * 4096, dtype=torch.float64, device=dev)
    st0 = torch.cuda.current_stream()

    with PhaseFieldSolver(dim=3, n=n, h=1.0) as s:
        s.set_ic_bm1(0.5, 0.05)
        s.sync()
        sp = None  # handle's own stream (library-created); the copy / probe kernels go on torch's stream after a sync

        def copy_for(seconds):
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < seconds:
                for _ in range(20):
                    L.check(lib.pfk_stream_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), cells,
                                                C.c_void_p(st0.cuda_stream)))
                torch.cuda.synchronize()
                k += 20
            return k

        def busy_for(seconds):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < seconds:
                L.check(lib.pfk_clock_probe(C.c_void_p(probe_out.data_ptr()), 1024, 20000, 1, C.c_void_p(st0.cuda_stream)))
                torch.cuda.synchronize()
            return float(probe_out[0:2048:2].median().item())

        def clock_now(busy=0):
            L.check(lib.pfk_clock_probe(C.c_void_p(probe_out.data_ptr()), 8, 30, busy, C.c_void_p(st0.cuda_stream)))
            torch.cuda.synchronize()
            return float(probe_out[0:16:2].median().item())

        # ---- A: cold start with sysfs sampling
        print("\n== A: 2 s idle, then 80 steps (per-launch ms; start offset ms)")
        time.sleep(2.0)
        smp = Sampler(srcs)
        smp.start()
        time.sleep(0.02)
        t_host0 = time.perf_counter()
        d, st = steps_timeline(s, 80)
        t_host1 = time.perf_counter()
        time.sleep(0.02)
        smp.stop = True
        smp.join()
        print("dur :", fmt(d))
        print("start:", fmt(st - st[0]))
        print("host wall for 80 steps: %.2f ms" % ((t_host1 - t_host0) * 1e3))
        print("sysfs samples (ms relative to first launch call; only changes shown): %d samples" % len(smp.rows))
        last = None
        for t, row in smp.rows:
            if row != last:
                print("  %+8.2f ms  %s" % ((t - t_host0) * 1e3, row))
                last = row

        # steady-state reference
        d, _ = steps_timeline(s, 200)
        steady = float(np.median(d[100:]))
        print("\nsteady-state median: %.4f ms" % steady)

        # ---- B: idle gaps
        print("\n== B: steady state -> sleep g -> 40 steps: mean of launches 0-4 / 5-9 / 10-19 / 30-39 (ms)")
        for g in (0.0, 0.001, 0.003, 0.01, 0.03, 0.1, 0.3, 1.0, 2.0):
            steps_timeline(s, 100)
            time.sleep(g)
            d, _ = steps_timeline(s, 40)
            print("  gap %6.3f s: %.4f %.4f %.4f %.4f   first launch %.4f" % (
                g, d[0:5].mean(), d[5:10].mean(), d[10:20].mean(), d[30:40].mean(), d[0]))

        # ---- C: another HBM-bound kernel first
        print("\n== C: 2 s idle -> 0.3 s of pfk_stream_copy -> 40 steps")
        time.sleep(2.0)
        k = copy_for(0.3)
        d, _ = steps_timeline(s, 40)
        print("  (%d copies)  %.4f %.4f %.4f %.4f   first launch %.4f" % (
            k, d[0:5].mean(), d[5:10].mean(), d[10:20].mean(), d[30:40].mean(), d[0]))

        # ---- D: compute-only load first
        print("\n== D: 2 s idle -> 0.3 s of fma-only kernel on all CUs -> 40 steps")
        time.sleep(2.0)
        mhz = busy_for(0.3)
        d, _ = steps_timeline(s, 40)
        print("  (busy kernel clock %.0f MHz)  %.4f %.4f %.4f %.4f   first launch %.4f" % (
            mhz, d[0:5].mean(), d[5:10].mean(), d[10:20].mean(), d[30:40].mean(), d[0]))

        # ---- D2: tiny warm-up amounts: how much work does it take?
        print("\n== D2: 2 s idle -> W seconds of steps (untimed) -> 20 steps, W = 0.005 .. 0.3")
        for w in (0.005, 0.01, 0.02, 0.05, 0.1, 0.3):
            time.sleep(2.0)
            t0 = time.perf_counter()
            cnt = 0
            while time.perf_counter() - t0 < w:
                s.step(DT, 5)
                s.sync()
                cnt += 5
            d, _ = steps_timeline(s, 20)
            print("  preheat %.3f s (%d steps): next 20 launches mean %.4f  (first %.4f, last5 %.4f)" % (
                w, cnt, d.mean(), d[0], d[15:].mean()))

        # ---- E: shader clock during the ramp
        print("\n== E: 2 s idle -> [clock probe, 2 steps] x 30: idle-lane clock MHz and step ms")
        time.sleep(2.0)
        rows = []
        for i in range(30):
            mhz = clock_now(0)
            d, _ = steps_timeline(s, 2)
            rows.append((mhz, d.mean()))
        print("  " + "  ".join("%.0f/%.3f" % r for r in rows))
        print("   steady: clock %.0f MHz (sleeping probe), %.0f MHz (fma probe)" % (clock_now(0), clock_now(1)))

        # ---- F: does the copy kernel ramp too?
        print("\n== F: 2 s idle -> 60 copies, per-launch ms via events")
        time.sleep(2.0)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
        evs[0].record(st0)
        for i in range(60):
            L.check(lib.pfk_stream_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), cells,
                                        C.c_void_p(st0.cuda_stream)))
            evs[i + 1].record(st0)
        torch.cuda.synchronize()
        print("  " + fmt([evs[i].elapsed_time(evs[i + 1]) for i in range(60)]))


if __name__ == "__main__":
    main()
