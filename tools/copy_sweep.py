"""Device-copy ceiling sweep (1 GiB read + 1 GiB write, as one 512^3 CH step): pfk_stream_copy forms x workgroups per
CU, torch's copy_ and hipMemcpyDtoD for comparison.  Usage on the GPU box: python tools/copy_sweep.py [n_doubles]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pfhubbenchmarks_amd import lib as L


def timed(fn, reps=20, warm=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512 ** 3
    lib = L.load()
    src = torch.ones(n, dtype=torch.float64, device="cuda")
    dst = torch.empty_like(src)
    st = torch.cuda.current_stream()
    res = {}
    for mode in range(6):
        for wgs in ((2, 4, 8, 16, 32, 64) if mode < 5 else (1,)):
            lib.pfk_set_tuning(6, mode)
            lib.pfk_set_tuning(5, wgs)
            t = timed(lambda: L.check(lib.pfk_stream_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n,
                                                          C.c_void_p(st.cuda_stream))))
            res["pfk mode %d, %2d wg/cu" % (mode, wgs)] = 16.0 * n / t / 1e9
    res["torch copy_"] = 16.0 * n / timed(lambda: dst.copy_(src)) / 1e9
    res["torch read-only sum"] = 8.0 * n / timed(lambda: src.sum()) / 1e9
    res["torch write-only fill_"] = 8.0 * n / timed(lambda: dst.fill_(2.0)) / 1e9
    for k, v in res.items():
        print("%-28s %8.1f GB/s" % (k, v), flush=True)
    best = max(res.items(), key=lambda kv: kv[1] if kv[0].startswith("pfk") else 0.0)
    print(json.dumps({"n_doubles": n, "best": best[0], "best_GBps": best[1], "all": res}))


if __name__ == "__main__":
    main()
