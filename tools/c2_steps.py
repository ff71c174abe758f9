#!/usr/bin/env python3
"""Why does `bench.py --steps 20 --warmup 5` read 4-5 % slower than `--steps 50`?  

One process, one box: the bench's own timed loop (fence, K back-to-back steps, fence) for several
(warmup, steps) pairs in the order given, then a per-launch series (every launch between its own
pair of events, in launch order, from a cold start after a sleep) and the host's enqueue time per
step.  usage: python tools/c2_steps.py [W:K ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import libvips_amd  # noqa: E402
from bench import lcg_image_device  # noqa: E402
from libvips_amd import Image, lib  # noqa: E402


def main():
    pairs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or \
        [(5, 20), (5, 50), (5, 200), (5, 20), (0, 20), (50, 20), (5, 20)]
    device = torch.device("cuda", 0)
    libvips_amd.init(0)
    stream = torch.cuda.Stream(device=device)
    lib.vips_hip_set_stream(stream.cuda_stream)
    n = 16384
    with torch.cuda.stream(stream):
        src = lcg_image_device(torch, n, n, 4, 12345, device)
    torch.cuda.synchronize()
    im = Image.new_from_tensor(src)

    def step():
        return im.reduce(8.0, 8.0, kernel="lanczos3")

    with torch.cuda.stream(stream):
        for w, k in pairs:
            for _ in range(w):
                step()
            e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(k):
                step()
            t_enq = time.perf_counter() - t0
            e1.record(stream)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            print("warmup %3d steps %3d: wall %.4f ms/step  events %.4f ms/step  host enqueue %.4f ms/step"
                  % (w, k, wall / k * 1e3, e0.elapsed_time(e1) / k, t_enq / k * 1e3), flush=True)
        # per-launch series from idle
        for idle in (0.0, 0.5):
            torch.cuda.synchronize()
            time.sleep(idle)
            evs = []
            for _ in range(60):
                a, b = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                a.record(stream)
                step()
                b.record(stream)
                evs.append((a, b))
            torch.cuda.synchronize()
            series = [a.elapsed_time(b) for a, b in evs]
            gaps = [evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(len(evs) - 1)]
            print("per-launch ms after %.1f s idle: %s" % (idle, " ".join("%.4f" % x for x in series)))
            print("  gaps between launches (event to event): %s" % " ".join("%.4f" % x for x in gaps[:20]))
        # back-to-back with the gates on (what `kernel_ms_isolated` is)
        lib.vips_hip_gate_reset()
        lib.vips_hip_gate_enable(1)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        lib.vips_hip_gate_enable(0)
        for kname, (launches, total) in libvips_amd.gate_report().items():
            print("gates: %s %d launches mean %.4f ms" % (kname, launches, total / launches))


if __name__ == "__main__":
    main()
