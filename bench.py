#!/usr/bin/env python
"""bench.py -- the BASELINE.json metric: Mpixels/s on the 16k x 16k Lanczos3 reduce.

One "step" = one pass of vips_reduce(8, 8, kernel=lanczos3) (BASELINE config 2:
16384x16384 uchar RGBA -> 2048x2048) over one synthetic image that is already
resident in HBM when the timed region starts.  Pixels counted are INPUT pixels of
the first op.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: independent images, one per rank (the batched path of north_star: images
partition across GPUs, no data-path collective) -> weak scaling; timing is
barrier + synchronize on both sides, MAX over ranks; value = all ranks' pixels / that.

Adds to the JSON line:
  roofline      dominant kernel: algorithmic bytes per launch / mean launch duration,
                measured with HIP events on the stream the kernel runs on
  cpu_baseline  the reference itself (oracle/_ref, scalar C paths, all host cores)
                timed on the same workload on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def lcg_image_device(torch, width, height, bands, seed, device):
    """The SURVEY.md 8(d) LCG byte stream generated on the device (jump-ahead form):
    s = s*1664525 + 1013904223 mod 2^32, byte = s >> 24.  Bit-identical to
    tests.helpers.lcg_bytes."""
    n = width * height * bands
    B = 1 << 16
    M = 0xFFFFFFFF
    ak = np.empty(B, dtype=np.uint64)
    ck = np.empty(B, dtype=np.uint64)
    aa, cc = 1, 0
    for k in range(B):
        aa = (aa * 1664525) & M
        cc = (cc * 1664525 + 1013904223) & M
        ak[k] = aa
        ck[k] = cc
    nblocks = (n + B - 1) // B
    seeds = np.empty(nblocks, dtype=np.uint64)
    s = seed & M
    aB, cB = int(ak[-1]), int(ck[-1])
    for i in range(nblocks):
        seeds[i] = s
        s = (aB * s + cB) & M
    d_ak = torch.from_numpy(ak.astype(np.int64)).to(device)
    d_ck = torch.from_numpy(ck.astype(np.int64)).to(device)
    d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(device)
    out = torch.empty(nblocks * B, dtype=torch.uint8, device=device)
    chunk = 1024  # blocks per pass: 64 Mi elements of int64 scratch
    for b0 in range(0, nblocks, chunk):
        sd = d_seeds[b0:b0 + chunk]
        # 32x32-bit products overflow int64 only above 2^63; mask keeps the low 32 bits
        v = (d_ak[None, :] * sd[:, None] + d_ck[None, :]) & M
        out[b0 * B:(b0 + sd.numel()) * B] = (v >> 24).to(torch.uint8).reshape(-1)
    return out[:n].reshape(height, width, bands).contiguous()


def traffic_for(kernel_name):
    """HBM bytes per launch for the dominant kernel, from the committed rocprofv3 PMC
    summary (profiles/traffic.json; PMC passes cannot run inside the timed bench)."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (IOError, ValueError):
        return None
    keys = [k for k in table if not k.startswith("_") and kernel_name.startswith(k)]
    if not keys:
        return None
    return table[max(keys, key=len)].get("traffic_bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=16384, help="image edge (16384 = BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    import libvips_amd
    from libvips_amd import Image, lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    libvips_amd.init(local_rank)
    dist = None
    # BENCH_FORCE_DIST=1 takes the RCCL path with a single rank too (how the N > 1 code is
    # smoke-tested on a one-GPU box)
    if world > 1 or (os.environ.get("BENCH_FORCE_DIST") and "RANK" in os.environ):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    # All library work goes to a torch-visible stream so torch.cuda.synchronize and
    # the barrier bracket exactly the kernels being timed.
    stream = torch.cuda.Stream(device=device)
    lib.vips_hip_set_stream(stream.cuda_stream)

    n = args.size
    shrink = 8.0
    with torch.cuda.stream(stream):
        src = lcg_image_device(torch, n, n, 4, 12345 + rank, device)
    torch.cuda.synchronize()
    im = Image.new_from_tensor(src)

    def step():
        return im.reduce(shrink, shrink, kernel="lanczos3")

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        out = None
        for _ in range(args.warmup):
            out = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        fence()
        elapsed = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    in_pixels = float(n) * n
    mpix_s = world * in_pixels * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel: HIP events around every launch, on its stream
    roofline = None
    with torch.cuda.stream(stream):
        lib.vips_hip_gate_reset()
        lib.vips_hip_gate_enable(1)
        gate_steps = max(3, min(args.steps, 20))
        for _ in range(gate_steps):
            out = step()
        torch.cuda.synchronize()
        lib.vips_hip_gate_enable(0)
        report = libvips_amd.gate_report()
        lib.vips_hip_gate_reset()
    oh = ow = int(n / shrink + 0.5)
    # algorithmic bytes per launch (read each input byte once, write each
    # output byte once), per kernel of the pipeline
    alg_bytes = {
        "reduce_fused_u8": n * n * 4 + oh * ow * 4,
        "reducev": n * n * 4 + oh * n * 4,
        "reduceh": oh * n * 4 + oh * ow * 4,
    }
    if report:
        name, (launches, total_ms) = max(report.items(), key=lambda kv: kv[1][1])
        mean_ms = total_ms / launches
        key = next((k for k in alg_bytes if name.startswith(k)), None)
        if key is not None:
            achieved = alg_bytes[key] / (mean_ms * 1e-3) / 1e9
            roofline = {
                "bound": "hbm",
                "kernel": name,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic_for(name),
                "kernel_ms": round(mean_ms, 4),
                "algorithmic_bytes": alg_bytes[key],
                "kernels": {k: {"launches": v[0], "mean_ms": round(v[1] / v[0], 4)} for k, v in report.items()},
            }
            # also against what this box's HBM delivers -- a device-to-device copy of
            # the same 1 GiB (read + write traffic), timed with events on the same stream
            try:
                with torch.cuda.stream(stream):
                    dst = torch.empty_like(src)
                    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                    dst.copy_(src)
                    best_copy = 1e9
                    for _ in range(5):
                        e0.record(stream)
                        dst.copy_(src)
                        e1.record(stream)
                        e1.synchronize()
                        best_copy = min(best_copy, e0.elapsed_time(e1))
                    del dst
                roofline["measured_copy_GBps"] = round(2 * src.numel() / (best_copy * 1e-3) / 1e9, 1)
                roofline["frac_of_measured_copy"] = round(achieved / roofline["measured_copy_GBps"], 4)
            except Exception:  # the reference rates are a courtesy, never a reason to fail the bench
                pass

    # ---- CPU baseline: the reference itself on this box's host cores (rank 0, N=1 only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests import helpers

        host = src.cpu().numpy()
        cores = os.cpu_count() or 1
        if helpers.have_ref():
            secs = helpers.Ref.time_chain("reduce:hshrink=8,vshrink=8,kernel=lanczos3", host, repeats=3,
                                          concurrency=cores)
            cpu_baseline = {
                "value": round(in_pixels / secs / 1e6, 1),
                "unit": "Mpixels/s",
                "cores": helpers.Ref.concurrency(),
                "kind": "reference",
                "sample": "full %dx%dx4 u8 image, vips_reduce(8,8,lanczos3) -> write_to_memory, "
                          "best of 3; libvips 8.19.0 scalar C path (no Highway/ORC)" % (n, n),
            }
        else:
            rows = 2048
            t1 = time.perf_counter()
            helpers.Port.reduce(host[:rows], shrink, shrink, "lanczos3")
            secs = time.perf_counter() - t1
            cpu_baseline = {
                "value": round(float(n) * rows / secs / 1e6, 1),
                "unit": "Mpixels/s",
                "cores": 1,
                "kind": "port",
                "sample": "top %d rows of the %dx%dx4 image, oracle/port, single thread" % (rows, n, n),
            }

    if rank == 0:
        line = {
            "metric": "Mpixels/s, vips_reduce Lanczos3 16384x16384 uchar RGBA -> 2048x2048 (input pixels)",
            "value": round(mpix_s, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (LCG bytes, seed 12345 + rank, generated on device)",
            "config": {
                "workload": "vips_reduce(hshrink=8, vshrink=8, kernel=lanczos3) %dx%dx4 u8 -> %dx%dx4, "
                            "BASELINE configs[1]" % (n, n, ow, oh),
                "images_per_step_per_gpu": 1,
                "partition": "one independent image per GPU, no data-path collective",
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
