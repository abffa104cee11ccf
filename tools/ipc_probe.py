"""Can two processes on ONE GPU share device memory and events (torch CUDA IPC)?  Feasibility probe for a peer-copy
halo transport (and for a 2-rank GPU test of the slab path on a 1-GPU box).  Usage: python tools/ipc_probe.py"""
import os
import sys
import time

import torch
import torch.multiprocessing as mp


def child(q_in, q_out):
    torch.cuda.set_device(0)
    try:
        t = q_in.get(timeout=60)                     # tensor rebuilt from the parent's IPC handle
        q_out.put(("tensor", float(t.sum().item()), t.data_ptr()))
        t.add_(1.0)                                  # write into the parent's memory
        torch.cuda.synchronize()
        q_out.put(("written", 0, 0))
        eh = q_in.get(timeout=60)
        ev = torch.cuda.Event.from_ipc_handle(torch.device("cuda", 0), eh)
        s = torch.cuda.current_stream()
        s.wait_event(ev)
        torch.cuda.synchronize()
        q_out.put(("event_ok", float(t[0].item()), 0))
    except Exception as e:                           # noqa: BLE001
        q_out.put(("error", repr(e), 0))


def main():
    mp.set_start_method("spawn", force=True)
    torch.cuda.set_device(0)
    q_in, q_out = mp.Queue(), mp.Queue()
    p = mp.Process(target=child, args=(q_in, q_out))
    p.start()
    t = torch.ones(1 << 20, dtype=torch.float64, device="cuda")
    q_in.put(t)
    print("child:", q_out.get(timeout=120))
    msg = q_out.get(timeout=120)
    print("child:", msg)
    torch.cuda.synchronize()
    print("parent sees", float(t[0].item()), "(2.0 expected if the child's write landed)")
    try:
        ev = torch.cuda.Event(enable_timing=False, interprocess=True)
        t.mul_(3.0)
        ev.record()
        q_in.put(ev.ipc_handle())
        print("child:", q_out.get(timeout=120), "(6.0 expected)")
    except Exception as e:                           # noqa: BLE001
        print("interprocess event failed:", repr(e))
        q_in.put(None)
    p.join(timeout=30)
    if p.is_alive():
        p.terminate()


if __name__ == "__main__":
    main()
