#!/usr/bin/env python
"""bench.py -- the BASELINE.json metric (Mpixels/s on the 16k x 16k Lanczos3 reduce) and the
other BASELINE configurations as parity-checked measurements.

  python bench.py --gpus 1 --steps K --warmup W [--config c2|c3|c4|c5slab|c5]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... [--config c2|c4|c5]

--config c2 (default): one "step" = one pass of vips_reduce(8, 8, kernel=lanczos3) (BASELINE
configs[1]: 16384x16384 uchar RGBA -> 2048x2048) over one synthetic image that is already
resident in HBM when the timed region starts; pixels counted are INPUT pixels of the first op
(SURVEY.md 8(d)).  The JSON line also carries

  roofline      dominant kernel: algorithmic bytes per launch / mean launch duration, measured
                with HIP events on the stream the kernel runs on (the library's gates)
  parity        the timed steps' output compared with the compiled reference (oracle/_ref),
                after the timed region
  cpu_baseline  the reference itself (scalar C paths, all host cores) on the same workload
  configs       C3, C4 and the C5 slab, each run at BASELINE size on this GPU, each with ms per
                step, algorithmic bytes, fraction of the HBM roofline, per-kernel gate times,
                a parity check against the compiled reference and a bounded CPU baseline
                (rank 0, N = 1 only; --no-configs skips them)

--config c3 | c4 | c5slab | c5 make that configuration the line's own metric:
  c3      gaussblur(sigma 8) -> colourspace(sRGB -> Lab) on 32768x32768x3 float, one GPU
  c4      batched thumbnail pipeline resize(1/8) -> sharpen on 8192x8192x3 uchar images
          generated on the device; N > 1: images round-robin over ranks (libvips_amd.sharding
          .batch_indices), no data-path collective, weak scaling
  c5slab  conv 31x31 float mask on one GPU's share of the 65536x65536 ushort image
          (65536 x 8192 rows + 15-row halos)
  c5      the whole 65536x65536 image in row strips over the ranks: strip plan -> one halo
          exchange over RCCL (libvips_amd.sharding.exchange_halos) -> conv on the strip; strong
          scaling; every rank checks rows of its strip against the reference

N > 1 timing: barrier + synchronize on both sides, MAX over ranks, value = all ranks' pixels / that.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector / matrix peak


def fp64_stream_tflops():
    """What a bare stream of independent v_fma_f64 reaches on this part, from the committed probe
    (tools/valu_probe2.hip -> profiles/r04_valu_probe2.txt: cycles at 2.4 GHz per wave64
    instruction per SIMD, 8 waves per SIMD): 128 flops per instruction x 1024 SIMDs.  A second
    denominator beside the spec sheet's 78.6 for the two FP64-bound configurations; None when the
    probe's output is not there."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_valu_probe2.txt")
    try:
        for line in open(path):
            if line.startswith("v_fmac_f64 sgpr"):  # (the form the kernels issue: a scalar coefficient)
                cycles = float(line.split()[2])
                return round(128.0 / (cycles / 2.4e9) * 1024 / 1e12, 1)
    except (OSError, ValueError, IndexError):
        pass
    return None


def with_fp64_stream(entry):
    rate = fp64_stream_tflops()
    if rate and entry.get("tflops"):
        entry["fp64_stream_tflops"] = rate
        entry["frac_of_fp64_stream"] = round(entry["tflops"] / rate, 4)
    return entry
M32 = 0xFFFFFFFF
_LCG_TABLES = {}


def _lcg_tables(torch, device):
    key = str(device)
    if key not in _LCG_TABLES:
        B = 1 << 16
        ak = np.empty(B, dtype=np.uint64)
        ck = np.empty(B, dtype=np.uint64)
        aa, cc = 1, 0
        for k in range(B):
            aa = (aa * 1664525) & M32
            cc = (cc * 1664525 + 1013904223) & M32
            ak[k] = aa
            ck[k] = cc
        _LCG_TABLES[key] = (int(ak[-1]), int(ck[-1]), torch.from_numpy(ak.astype(np.int64)).to(device),
                            torch.from_numpy(ck.astype(np.int64)).to(device))
    return _LCG_TABLES[key]


def lcg_image_device(torch, width, height, bands, seed, device, out=None):
    """The SURVEY.md 8(d) LCG byte stream generated on the device (jump-ahead form):
    s = s*1664525 + 1013904223 mod 2^32, byte = s >> 24.  Bit-identical to
    tests.helpers.lcg_bytes."""
    n = width * height * bands
    B = 1 << 16
    aB, cB, d_ak, d_ck = _lcg_tables(torch, device)
    nblocks = (n + B - 1) // B
    seeds = np.empty(nblocks, dtype=np.uint64)
    s = seed & M32
    for i in range(nblocks):
        seeds[i] = s
        s = (aB * s + cB) & M32
    d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(device)
    flat = torch.empty(nblocks * B, dtype=torch.uint8, device=device)
    chunk = 1024  # blocks per pass: 64 Mi elements of int64 scratch
    for b0 in range(0, nblocks, chunk):
        sd = d_seeds[b0:b0 + chunk]
        # 32x32-bit products overflow int64 only above 2^63; mask keeps the low 32 bits
        v = (d_ak[None, :] * sd[:, None] + d_ck[None, :]) & M32
        flat[b0 * B:(b0 + sd.numel()) * B] = (v >> 24).to(torch.uint8).reshape(-1)
    res = flat[:n].reshape(height, width, bands)
    if out is not None:
        out.copy_(res)
        return out
    return res.contiguous()


def file_sha(path):
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except IOError:
        return None


_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_isa_sha(symbol_part, lib_path=None):
    """Hash of a kernel's MACHINE CODE as it sits in the built library: the bytes of the gfx950
    function whose mangled name contains `symbol_part`, read from the code objects bundled in
    libvipship.so (clang offload bundle -> ELF64 -> .symtab -> the function's bytes in .text).
    Branches are PC-relative, so the hash moves only when the kernel's instructions do: comments,
    renamed statements and edits to other kernels of the same file leave it alone (round 4 stamped
    the counters with a hash of the source text and a rename made them look stale)."""
    import struct

    if lib_path is None:
        lib_path = os.path.join(ROOT, "libvips_amd", "lib", "libvipship.so")
    try:
        data = open(lib_path, "rb").read()
    except IOError:
        return None
    found = []
    pos = 0
    while True:
        p = data.find(_BUNDLE_MAGIC, pos)
        if p < 0:
            break
        pos = p + len(_BUNDLE_MAGIC)
        (n,) = struct.unpack_from("<Q", data, p + 24)
        q = p + 32
        if n > 64:
            continue
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            triple = data[q:q + tl]
            q += tl
            if b"gfx950" not in triple:
                continue
            elf = data[p + off:p + off + size]
            if elf[:4] != b"\x7fELF":
                continue
            (shoff,) = struct.unpack_from("<Q", elf, 0x28)
            shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
            secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
            for sec in secs:
                if sec[1] != 2:  # SHT_SYMTAB
                    continue
                stroff = secs[sec[6]][4]
                for j in range(sec[5] // 24):
                    st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from(
                        "<IBBHQQ", elf, sec[4] + j * 24)
                    if (st_info & 0xF) != 2 or st_shndx >= len(secs) or not st_size:
                        continue
                    e = elf.index(b"\0", stroff + st_name)
                    if symbol_part.encode() in elf[stroff + st_name:e]:
                        text = secs[st_shndx]
                        fo = st_value - text[3] + text[4]
                        found.append((elf[stroff + st_name:e], elf[fo:fo + st_size]))
    if not found:
        return None
    h = hashlib.sha256()
    for name, code in sorted(found):
        h.update(name + b"\0" + code)
    return h.hexdigest()[:16]


def traffic_for(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/traffic.json; PMC passes cannot run inside the timed bench).  An entry is only
    used while the kernel it was measured on is unchanged: its `isa_sha` stamp is the hash of the
    kernel's machine code in the built library (kernel_isa_sha), not of its source text."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (IOError, ValueError):
        return None
    keys = [k for k in table if not k.startswith("_") and kernel_name.startswith(k)]
    if not keys:
        return None
    entry = table[max(keys, key=len)]
    sym = entry.get("symbol")
    if not sym or entry.get("isa_sha") != kernel_isa_sha(sym):
        return None  # stale: the kernel's instructions changed since the counters were collected
    return entry.get("traffic_bytes")


class Ctx(object):
    """Device, stream, process group."""

    def __init__(self, args):
        import torch

        import libvips_amd
        from libvips_amd import lib

        self.torch = torch
        self.lib = lib
        self.vh = libvips_amd
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            if self.world == 1 and args.gpus > 1:
                raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
            args.gpus = self.world
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        libvips_amd.init(self.local_rank)
        self.dist = None
        # BENCH_FORCE_DIST=1 takes the RCCL path with a single rank too (how the N > 1 code is
        # smoke-tested on a one-GPU box)
        if self.world > 1 or (os.environ.get("BENCH_FORCE_DIST") and "RANK" in os.environ):
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="nccl", device_id=self.device)
            assert dist.get_world_size() == self.world, "RCCL world size differs from WORLD_SIZE"
            self.dist = dist
        # All library work goes to a torch-visible stream so torch.cuda.synchronize and the
        # barrier bracket exactly the kernels being timed.
        self.stream = torch.cuda.Stream(device=self.device)
        lib.vips_hip_set_stream(self.stream.cuda_stream)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def settle(self, step, chunk=20, max_launches=1500, tol=0.01):
        """Run `step` until the GPU's clocks have settled: chunks of `chunk` launches, each between
        a pair of events, until two consecutive chunks agree within `tol` (or max_launches).
        On these boxes the first ~100-300 launches of a heavy kernel after a quiet spell run
        5-25 % slower (tools/c2_steps.py, profiles/r03_c2_ramp.txt: fast for ~8 launches, then a
        dip, then a slow recovery -- the power controller hunting), which is longer than the
        5-step warm-up the driver asks for.  Returns what it saw; the bench line carries it."""
        torch = self.torch
        means = []
        with torch.cuda.stream(self.stream):
            while len(means) * chunk < max_launches:
                e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                e0.record(self.stream)
                for _ in range(chunk):
                    step()
                e1.record(self.stream)
                e1.synchronize()
                means.append(e0.elapsed_time(e1) / chunk)
                if len(means) >= 3 and abs(means[-1] - means[-2]) <= tol * means[-1] and \
                        abs(means[-2] - means[-3]) <= tol * means[-2]:
                    break
        return {"launches": len(means) * chunk, "first_chunk_ms": round(means[0], 4),
                "slowest_chunk_ms": round(max(means), 4), "last_chunk_ms": round(means[-1], 4),
                "chunk": chunk}

    def timed(self, step, steps, warmup):
        """warmup untimed steps, then exactly `steps` between fences; MAX over ranks (s)."""
        torch = self.torch
        with torch.cuda.stream(self.stream):
            out = None
            for _ in range(warmup):
                out = step()
            e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            self.fence()
            t0 = time.perf_counter()
            e0.record(self.stream)  # HIP events on the kernels' own stream, around the timed region
            for _ in range(steps):
                out = step()
            e1.record(self.stream)
            self.fence()
            elapsed = time.perf_counter() - t0
            self.event_ms = e0.elapsed_time(e1)
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out

    def gates(self, step, n):
        """Per-kernel launch counts and mean durations (HIP events on the kernels' stream)."""
        with self.torch.cuda.stream(self.stream):
            self.lib.vips_hip_gate_reset()
            self.lib.vips_hip_gate_enable(1)
            for _ in range(n):
                step()
            self.torch.cuda.synchronize()
            self.lib.vips_hip_gate_enable(0)
            report = self.vh.gate_report()
            self.lib.vips_hip_gate_reset()
        return report

    def trim(self):
        self.torch.cuda.synchronize()
        self.torch.cuda.empty_cache()
        self.lib.vips_hip_pool_trim()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def kernels_of(report):
    return {k: {"launches": v[0], "mean_ms": round(v[1] / v[0], 4)} for k, v in report.items()}


def ref_or_none():
    from tests import helpers

    return helpers if helpers.have_ref() else None


def cpu_baselines(helpers, chain, host, pixels, interp=0, repeats=5, single_rows=None, what=""):
    """The compiled reference on this box's host cores: best of `repeats` with every core
    (VIPS_CONCURRENCY = cores) and with ONE thread; the one-thread run may take a
    bounded sample of the rows (`single_rows`) so that the bench stays within minutes."""
    cores = os.cpu_count() or 1
    secs = helpers.Ref.time_chain(chain, host, repeats=repeats, interpretation=interp, concurrency=cores)
    used = helpers.Ref.concurrency()
    sample1 = host if single_rows is None or single_rows >= host.shape[0] else np.ascontiguousarray(host[:single_rows])
    frac = float(sample1.shape[0]) / host.shape[0]
    secs1 = helpers.Ref.time_chain(chain, sample1, repeats=repeats, interpretation=interp, concurrency=1)
    helpers.Ref.lib().ref_init(cores)
    return {
        # `cores` = the worker count libvips ACTUALLY ran with (vips_concurrency_get() after
        # vips_concurrency_set(host cpus): libvips caps it at MAX_THREADS 1024, iofuncs/thread.c:164-223);
        # beside it what was asked for and the CPUs this process may run on
        "value": round(pixels / secs / 1e6, 1), "unit": "Mpixels/s", "cores": used, "kind": "reference",
        "threads_requested": cores, "cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores,
        "sample": "%s, best of %d; libvips 8.19.0 scalar C path (no Highway/ORC)" % (what, repeats),
        "single_core": {"value": round(pixels * frac / secs1 / 1e6, 1), "unit": "Mpixels/s", "cores": 1,
                        "sample": "%s, best of %d, VIPS_CONCURRENCY=1" %
                                  (what if frac == 1.0 else "the top %d rows of it" % sample1.shape[0], repeats)},
    }


def same_float(a, b):
    """Bit-exact, else the largest difference in units in the last place."""
    if a.shape != b.shape:
        return False, None
    ai = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    ulp = int(np.abs(ai - bi).max()) if ai.size else 0
    return ulp == 0, ulp


# ------------------------------------------------------------------------------------- C2

def run_c2(ctx, args):
    torch, lib, vh = ctx.torch, ctx.lib, ctx.vh
    from libvips_amd import Image

    n = args.size
    shrink = 8.0
    with torch.cuda.stream(ctx.stream):
        src = lcg_image_device(torch, n, n, 4, 12345 + ctx.rank, ctx.device)
    torch.cuda.synchronize()
    im = Image.new_from_tensor(src)

    def step():
        return im.reduce(shrink, shrink, kernel="lanczos3")

    # The line's value / ms_per_step / roofline.frac are THE DRIVER'S LITERAL REGION: W warm-up steps, then
    # exactly K timed steps, the first W + K launches of the process (the GPU has only run the image
    # generator before).  On these boxes the first few hundred launches after a quiet spell run a few
    # per cent slower (the power controller's clock ramp): the same region once more after Ctx.settle is
    # reported BESIDE it as `settled` / roofline.frac_settled -- the rate a pipeline of images sees --
    # never instead of it.
    elapsed, out = ctx.timed(step, args.steps, args.warmup)
    region_ms = ctx.event_ms / args.steps  # one launch per step: the kernel's average launch duration
    settled = None
    if not args.no_settle:
        info = ctx.settle(step)
        elapsed2, out = ctx.timed(step, args.steps, args.warmup)
        settled = {"ms_per_step": round(elapsed2 / args.steps * 1e3, 4),
                   "kernel_ms": round(ctx.event_ms / args.steps, 4),
                   "value": round(ctx.world * float(n) * n * args.steps / elapsed2 / 1e6, 1),
                   "what": "the same W + K region again after %d more launches (clocks settled)" % info["launches"],
                   "settle": info}
    in_pixels = float(n) * n
    mpix_s = ctx.world * in_pixels * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel: HIP events around every launch, on its stream
    roofline = None
    report = ctx.gates(step, max(3, min(args.steps, 20)))
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
        isolated_ms = total_ms / launches  # every launch alone between its own pair of events
        # the fused path is ONE kernel per step, so the events around the timed region give its
        # average launch duration over exactly the launches that were timed; a multi-kernel
        # fallback path is priced from its dominant kernel's gate time instead
        mean_ms = region_ms if len(report) == 1 else isolated_ms
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
                "kernel_ms_isolated": round(isolated_ms, 4),
                "algorithmic_bytes": alg_bytes[key],
                "kernels": kernels_of(report),
            }
            # also against what this box's HBM delivers to a kernel of the same shape --
            # a READ of the same 1 GiB (the reduce reads 64 bytes for every byte it writes; a copy,
            # half of whose traffic is writes, is the wrong yardstick): a sum over the image, timed
            # with events on the same stream
            try:
                with torch.cuda.stream(ctx.stream):
                    flat = src.view(-1).view(torch.int64)
                    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                    flat.sum()
                    best_read = 1e9
                    for _ in range(5):
                        e0.record(ctx.stream)
                        flat.sum()
                        e1.record(ctx.stream)
                        e1.synchronize()
                        best_read = min(best_read, e0.elapsed_time(e1))
                roofline["measured_read_GBps"] = round(src.numel() / (best_read * 1e-3) / 1e9, 1)
                roofline["frac_of_measured_read"] = round(achieved / roofline["measured_read_GBps"], 4)
            except Exception:  # the reference rates are a courtesy, never a reason to fail the bench
                pass

    if roofline and settled:
        # the same fraction once the clocks have settled (a side figure: `frac` is the driver's region)
        roofline["frac_settled"] = round(roofline["algorithmic_bytes"] / (settled["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if roofline:
        roofline["frac_cold"] = roofline["frac"]  # (rounds 3-5 reported the literal region under this name)

    # ---- parity of the timed steps' output + CPU baseline: the reference itself on this box's
    # host cores (rank 0, N = 1 only)
    parity = None
    cpu_baseline = None
    helpers = ref_or_none()
    if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
        from tests import helpers as th

        host = src.cpu().numpy()
        got = out.numpy()
        cores = os.cpu_count() or 1
        chain = "reduce:hshrink=8,vshrink=8,kernel=lanczos3"
        if helpers is not None:
            want = th.Ref.run_chain(chain, host)
            parity = {"against": "oracle/_ref (compiled reference), whole output",
                      "bit_exact": bool(got.shape == want.shape and np.array_equal(got, want)),
                      "checksum": th.checksum(got)}
            cpu_baseline = cpu_baselines(th, chain, host, in_pixels, single_rows=n // 4,
                                         what="full %dx%dx4 u8 image, vips_reduce(8,8,lanczos3) -> write_to_memory" % (n, n))
        else:
            rows = 2048
            t1 = time.perf_counter()
            want = th.Port.reduce(host[:rows], shrink, shrink, "lanczos3")
            secs = time.perf_counter() - t1
            # rows of the output whose taps all lie in the sample
            ok_rows = want.shape[0] - 4
            parity = {"against": "oracle/port, top %d output rows" % ok_rows,
                      "bit_exact": bool(np.array_equal(got[:ok_rows], want[:ok_rows])),
                      "checksum": th.checksum(got)}
            cpu_baseline = {
                "value": round(float(n) * rows / secs / 1e6, 1),
                "unit": "Mpixels/s",
                "cores": 1,
                "kind": "port",
                "sample": "top %d rows of the %dx%dx4 image, oracle/port, single thread" % (rows, n, n),
            }
        if parity and not parity["bit_exact"]:
            raise SystemExit("bench.py: C2 output differs from the oracle: %r" % (parity,))
    del im, src, out
    ctx.trim()

    return {
        "metric": "Mpixels/s, vips_reduce Lanczos3 16384x16384 uchar RGBA -> 2048x2048 (input pixels)",
        "value": round(mpix_s, 1),
        "unit": "Mpixels/s",
        "n_gpus": ctx.world,
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
        "settled": settled,
        "parity": parity,
        "cpu_baseline": cpu_baseline,
    }


# ------------------------------------------------------------------------------- module e2e

def run_module_e2e(ctx, kernel_ms=None):
    """SURVEY.md 8(d) "exclude H2D / D2H (report separately)": the C2 operation end to end the way a
    libvips user meets it -- `reduce_hip` from the loadable module on a HOST-resident 16384 x 16384
    RGBA image, evaluated through vips_image_write_to_memory() -- beside the built-in `reduce` on
    the same cores, with the two PCIe legs timed on their own.  Two module runs: the whole image in
    one piece (pageable upload) and, with a 512 MiB budget, the overlapped strip loop (pull into
    pinned memory / upload / kernels / download on two streams)."""
    import ctypes

    helpers = ref_or_none()
    if helpers is None or not helpers.have_module():
        return None
    torch, lib = ctx.torch, ctx.lib
    helpers.Ref.load_module()
    n = 16384
    with torch.cuda.stream(ctx.stream):
        src = lcg_image_device(torch, n, n, 4, 12345, ctx.device)
    torch.cuda.synchronize()
    host = src.cpu().numpy()
    nbytes = host.nbytes
    cores = os.cpu_count() or 1
    args = "hshrink=8,vshrink=8,kernel=lanczos3"
    want = helpers.Ref.run_chain("reduce:" + args, host)
    got = helpers.Ref.run_chain("reduce_hip:" + args, host)
    exact = bool(got.shape == want.shape and np.array_equal(got, want))
    t_whole = helpers.Ref.time_chain("reduce_hip:" + args, host, repeats=3, concurrency=cores)
    os.environ["VIPS_HIP_BUDGET"] = "512m"
    try:
        got_strips = helpers.Ref.run_chain("reduce_hip:" + args, host)
        t_strips = helpers.Ref.time_chain("reduce_hip:" + args, host, repeats=3, concurrency=cores)
    finally:
        del os.environ["VIPS_HIP_BUDGET"]
    exact = exact and bool(np.array_equal(got_strips, want))
    t_ref = helpers.Ref.time_chain("reduce:" + args, host, repeats=3, concurrency=cores)

    # the PCIe legs alone: the 1 GiB image up (pageable as libvips hands it over, and pinned), the
    # 16 MiB result down
    def best(fn, reps=3):
        t = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t = min(t, time.perf_counter() - t0)
        return t

    dev = lib.vips_hip_malloc(nbytes)
    pinned = lib.vips_hip_malloc_host(nbytes)
    ctypes.memmove(pinned, host.ctypes.data, nbytes)
    out_bytes = want.nbytes
    h2d_pageable = best(lambda: lib.vips_hip_memcpy_h2d(dev, host.ctypes.data, nbytes))
    h2d_pinned = best(lambda: lib.vips_hip_memcpy_h2d(dev, pinned, nbytes))
    d2h_pinned = best(lambda: lib.vips_hip_memcpy_d2h(pinned, dev, out_bytes))
    lib.vips_hip_free(dev)
    lib.vips_hip_free_host(pinned)
    del src
    ctx.trim()
    if not exact:
        raise SystemExit("bench.py: reduce_hip through the module differs from the built-in reduce")
    return {
        "name": "module_e2e",
        "workload": "reduce_hip (loadable module) vs reduce (built-in) on a host-resident %dx%dx4 u8 image, "
                    "vips_image_write_to_memory, %d host cores" % (n, n, cores),
        "ms_module_whole_image": round(t_whole * 1e3, 2),
        "ms_module_strips_512m": round(t_strips * 1e3, 2),
        "ms_builtin_reduce": round(t_ref * 1e3, 2),
        "speedup_vs_builtin": round(t_ref / min(t_whole, t_strips), 2),
        "h2d_ms_pageable": round(h2d_pageable * 1e3, 2),
        "h2d_GBps_pageable": round(nbytes / h2d_pageable / 1e9, 1),
        "h2d_ms_pinned": round(h2d_pinned * 1e3, 2),
        "h2d_GBps_pinned": round(nbytes / h2d_pinned / 1e9, 1),
        "d2h_ms_result": round(d2h_pinned * 1e3, 3),
        "kernel_ms": kernel_ms,
        "mpixels_per_s_e2e": round(float(n) * n / min(t_whole, t_strips) / 1e6, 1),
        "parity": {"against": "the built-in reduce in the same process, whole output, both module paths",
                   "bit_exact": exact},
        "note": "the device kernel is %s of the end-to-end time: a host-resident image is bound by PCIe"
                % ("%.1f %%" % (100.0 * kernel_ms / (min(t_whole, t_strips) * 1e3)) if kernel_ms else "a small part"),
    }


# ------------------------------------------------------------------------------------- C1

def run_c1(ctx, steps, warmup, verify=True, cpu=True, size=4096):
    """BASELINE configs[0], the device part: vips_thumbnail_image(width 512) of a size x size x 3
    uchar sRGB image resident in HBM (what `vipsthumbnail --size 512x512` runs after the load:
    vips_resize(1/8), here the one-kernel chain).  One step = one thumbnail."""
    torch = ctx.torch
    from libvips_amd import Image

    n = size
    with torch.cuda.stream(ctx.stream):
        src = lcg_image_device(torch, n, n, 3, 12345, ctx.device)
    torch.cuda.synchronize()
    im = Image.new_from_tensor(src, interpretation="srgb")

    def step():
        return im.thumbnail_image(512)

    elapsed, out = ctx.timed(step, steps, warmup)
    ms = elapsed / steps * 1e3
    report = ctx.gates(step, 4)
    t = out.width
    alg = n * n * 3 + t * out.height * 3
    entry = {
        "name": "c1",
        "workload": "vips_thumbnail_image(512) of %dx%dx3 u8 sRGB -> %dx%dx3 (the device part of BASELINE configs[0])"
                    % (n, n, t, out.height),
        "ms": round(ms, 4),
        "steps": steps,
        "mpixels_per_s": round(float(n) * n / (ms * 1e-3) / 1e6, 1),
        "algorithmic_bytes": alg,
        # one 50 MB image read again every step sits in the 256 MiB Infinity Cache and its kernel
        # is 133 short blocks: the step is bound by launch and memory LATENCY, not by HBM rate
        "bound": "latency",
        "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "frac_of": "8 TB/s HBM, for scale only (input is Infinity-Cache resident)",
        "dtype": "u8",
        "kernels": kernels_of(report),
    }
    helpers = ref_or_none()
    if helpers is not None and (verify or cpu):
        chain = "thumbnail_image:width=512"
        interp = helpers.INTERP["srgb"]
        host = src.cpu().numpy()
        if verify:
            want = helpers.Ref.run_chain(chain, host, interp)
            got = out.numpy()
            exact = got.shape == want.shape and bool(np.array_equal(got, want))
            entry["parity"] = {"against": "oracle/_ref (compiled reference), whole thumbnail", "bit_exact": exact}
            if not exact:
                raise SystemExit("bench.py: C1 thumbnail differs from the reference")
        if cpu:
            entry["cpu_baseline"] = cpu_baselines(helpers, chain, host, float(n) * n, interp,
                                                  what="the same %dx%dx3 image through vips_thumbnail_image" % (n, n))
    del im, src, out
    ctx.trim()
    return entry


# ------------------------------------------------------------------------------------- C3

def run_c3(ctx, steps, warmup, verify=True, cpu=True, size=32768):
    """BASELINE configs[2]: vips_gaussblur(sigma 8) + vips_colourspace(sRGB -> Lab) on
    size x size x 3 float.  One step = the whole pipeline over the image."""
    torch = ctx.torch
    from libvips_amd import Image

    n = size
    with torch.cuda.stream(ctx.stream):
        u8 = lcg_image_device(torch, n, n, 3, 12345, ctx.device)
        src = u8.float()
        del u8
        # $VIPS_BENCH_C3_INPUT=float: the whole entry on the float image proper (for a profiler: the
        # kernel has one name for both inputs); default: the integers 0..255, the float image beside it
        float_only = os.environ.get("VIPS_BENCH_C3_INPUT") == "float"
        if float_only:
            src.add_(0.25)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    im = Image.new_from_tensor(src, interpretation="srgb")

    def step():
        # vips_gaussblur + vips_colourspace as ONE call: on this image both blur passes and the
        # colour route run in one streaming kernel (vips_hip_gaussblur_colourspace)
        return im.gaussblur_colourspace(8.0, "lab")

    elapsed, out = ctx.timed(step, steps, warmup)
    ms = elapsed / steps * 1e3
    report = ctx.gates(step, 2)
    alg = 2 * n * n * 12
    # the reference sums both passes in double (convi.c:721-741): 2 x 29 taps per band element
    flops = 2.0 * 58 * n * n * 3
    entry = {
        "name": "c3",
        "workload": "vips_gaussblur(sigma=8) + vips_colourspace(sRGB->Lab) %dx%dx3 f32, BASELINE configs[2]" % (n, n),
        "ms": round(ms, 3),
        "steps": steps,
        "mpixels_per_s": round(float(n) * n / (ms * 1e-3) / 1e6, 1),
        "algorithmic_bytes": alg,
        # bound by FP64 instruction issue, not by HBM: both fractions
        "bound": "fp64",
        "tflops": round(flops / (ms * 1e-3) / 1e12, 2),
        "frac": round(flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
        "frac_fp64": round(flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
        "frac_hbm": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "float_mode": "the reference's own arithmetic (the fused blur + colour kernel has no other mode)",
        "dtype": "f32 (f64 sums)",
        "kernels": kernels_of(report),
    }
    helpers = ref_or_none()
    if helpers is not None and (verify or cpu):
        chain = "gaussblur:sigma=8;colourspace:space=lab"
        interp = helpers.INTERP["srgb"]
        halo = 14  # 29-tap mask
        worst = 0
        exact = True
        checked = []
        rows = 96
        # strips at the top edge, across the middle (several row segments of the fused kernel)
        # and at the bottom edge; full width (every column strip of the kernel)
        for r0 in ((0, n // 2 - rows // 2, n - rows) if verify else ()):
            i0, i1 = max(r0 - halo, 0), min(r0 + rows + halo, n)
            host = src[i0:i1].cpu().numpy()
            want = helpers.Ref.run_chain(chain, host, interp)[r0 - i0:r0 - i0 + rows]
            got = out.extract_area(0, r0, n, rows).numpy()
            ok, ulp = same_float(got, want)
            exact = exact and ok
            worst = max(worst, ulp if ulp is not None else 1 << 30)
            checked.append([r0, r0 + rows])
        if verify:
            entry["parity"] = {"against": "oracle/_ref (compiled reference)", "rows": checked,
                               "bit_exact": exact, "max_ulp": worst, "tolerance_ulp": 1}
            if worst > 1:
                raise SystemExit("bench.py: C3 output differs from the reference by %d ULP" % worst)
        if cpu:
            srows = 512
            host = src[n // 2:n // 2 + srows].cpu().numpy()
            entry["cpu_baseline"] = cpu_baselines(helpers, chain, host, float(n) * srows, interp, repeats=5, single_rows=64,
                                                  what="%d rows x %d x 3 f32 of the same image, whole pipeline" % (srows, n))
    # The input holds the integers 0 .. 255 (a float image cast from uchar), so the horizontal pass runs on
    # packed bytes (convsep_int_body.h; the test is per wave, the bits do not depend on it).  The same
    # pipeline on a float image proper -- every pixel + 0.25, both passes in double -- beside it:
    entry["input"] = "float pixels holding the integers 0..255 (LCG bytes cast to float): integer horizontal pass"
    if float_only:
        entry["input"] = "the LCG bytes + 0.25 as float (no window of integers): horizontal pass in double"
    elif ctx.world == 1:
        try:
            with torch.cuda.stream(ctx.stream):
                src.add_(0.25)
            torch.cuda.synchronize()
            k2 = max(2, min(steps, 5))
            # three warm-up steps and the EVENT time of the region (round 5's driver run had 62.9 ms here
            # against 14.3-14.4 in every other run: one warm-up step, wall clock -- whatever the host did
            # in that region was in the figure); the wall-clock figure stays beside it
            elapsed2, out2 = ctx.timed(step, k2, 3)
            entry["float_input"] = {"what": "the same image + 0.25 (no window of integers): horizontal pass in double",
                                    "ms": round(ctx.event_ms / k2, 3), "ms_wall": round(elapsed2 / k2 * 1e3, 3),
                                    "steps": k2, "warmup": 3}
            entry["ms_float_input"] = entry["float_input"]["ms"]
            del out2
        except Exception as exc:  # a side measurement: never lose the line over it
            entry["float_input"] = {"error": repr(exc)}
    del im, src, out
    ctx.trim()
    return with_fp64_stream(entry)


# ------------------------------------------------------------------------------------- C4

def run_c4(ctx, steps, warmup, images, verify=True, cpu=True, size=8192):
    """BASELINE configs[3]: batched thumbnail pipeline.  `images` images per rank of
    size x size x 3 uchar (sRGB) generated on the device, each through vips_resize(1/8)
    (= shrinkv 4, reducev 2, shrinkh 4, reduceh 2) -> vips_sharpen() -> sRGB uchar.  One
    step = the whole per-rank batch."""
    torch = ctx.torch
    from libvips_amd import Image, sharding

    n = size
    total = images * ctx.world
    mine = sharding.batch_indices(total, ctx.world, ctx.rank)  # round-robin over ranks
    store = torch.empty((len(mine), n, n, 3), dtype=torch.uint8, device=ctx.device)
    with torch.cuda.stream(ctx.stream):
        for k, idx in enumerate(mine):
            lcg_image_device(torch, n, n, 3, 12345 + idx, ctx.device, out=store[k])
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    ims = [Image.new_from_tensor(store[k], interpretation="srgb") for k in range(len(mine))]

    import ctypes

    from libvips_amd import _ffi
    from libvips_amd._ffi import lib
    from libvips_amd.image import KERNELS

    # The step drives the C ABI itself -- the batch entry point and the batch unref, two calls --
    # on arrays made once: what is timed is the library, as a C caller sees it (through the Python
    # mirror, wrapping 1024 results in objects and dropping the previous 1024 one ctypes call at a
    # time adds ~1.5 ms of interpreter time per step, measured against the mock runtime).
    count = len(ims)
    handles_in = (ctypes.c_void_p * count)(*[im._h.value for im in ims])
    handles_out = (ctypes.c_void_p * count)()
    lanczos3 = KERNELS["lanczos3"]

    def step():
        # the previous step's thumbnails go back to the pool; then one launch per 64 images for the
        # whole resize chain and one for the sharpen (resize_stream.hip, colour.hip sharpen_fused_u8)
        # (the QUEUED form: like the C2 launches of this file, a step is enqueued and the region ends
        # with a fence -- the host prepares batch k + 1 while batch k runs, as a thumbnail service would)
        lib.vips_hip_image_unref_many(handles_out, count)
        if lib.vips_hip_resize_sharpen_batch_queue(handles_in, count, handles_out, 0.125, lanczos3, 2.0, 0.5, 2.0,
                                                   10.0, 20.0, 0.0, 3.0, 8):
            _ffi.check(-1)

    elapsed, _ = ctx.timed(step, steps, warmup)
    ms = elapsed / steps * 1e3
    # every rank's own rate (events on its stream): a straggler shows as MIN well under MAX
    own = len(mine) * steps / (ctx.event_ms * 1e-3)
    rank_rates = {"min": own, "max": own}
    if ctx.dist is not None:
        lo = torch.tensor([own], dtype=torch.float64, device=ctx.device)
        hi = lo.clone()
        ctx.dist.all_reduce(lo, op=ctx.dist.ReduceOp.MIN)
        ctx.dist.all_reduce(hi, op=ctx.dist.ReduceOp.MAX)
        rank_rates = {"min": float(lo.item()), "max": float(hi.item())}
    # the kernels of one step, HIP events around each gate, as time per image of the batch
    report = {k: (v[0], v[1] / len(ims)) for k, v in ctx.gates(step, 1).items()}
    outs = [Image(h) if h else None for h in handles_out]  # (the objects now own the last step's results)
    t = n // 8
    alg_image = n * n * 3 + t * t * 3
    alg = alg_image * len(ims)
    entry = {
        "name": "c4",
        "workload": "%d x (vips_resize(1/8) + vips_sharpen on %dx%dx3 u8 sRGB -> %dx%dx3), BASELINE configs[3]"
                    % (total, n, n, t, t),
        "images_per_gpu": len(ims),
        "ms": round(ms, 3),
        "ms_per_image": round(ms / len(ims), 4),
        "steps": steps,
        "images_per_s": round(total / (ms * 1e-3), 1),
        "images_per_s_per_rank": {"min": round(rank_rates["min"], 1), "max": round(rank_rates["max"], 1)},
        "mpixels_per_s": round(float(n) * n * total / (ms * 1e-3) / 1e6, 1),
        "algorithmic_bytes": alg,
        "bound": "hbm",
        "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "dtype": "u8",
        "kernels_per_image": {k: {"gates_per_step": v[0], "ms_per_image": round(v[1], 4)} for k, v in report.items()},
    }
    helpers = ref_or_none()
    if helpers is not None and (verify or cpu):
        chain = "resize:scale=0.125;sharpen:"
        interp = helpers.INTERP["srgb"]
        if verify:
            checked = []
            exact = True
            # first, a pair either side of a 64-image launch boundary, then one thumbnail of every 64-image launch
            # (16 more at 1 024 images), middle, last
            picks = {0, 63, 64, len(ims) // 2, len(ims) - 1} | {64 * j + (7 * j) % 64 for j in range(len(ims) // 64)}
            for k in sorted(k for k in picks if 0 <= k < len(ims)):
                host = store[k].cpu().numpy()
                want = helpers.Ref.run_chain(chain, host, interp)
                got = outs[k].numpy()
                exact = exact and got.shape == want.shape and bool(np.array_equal(got, want))
                checked.append(int(mine[k]))
            entry["parity"] = {"against": "oracle/_ref (compiled reference), whole thumbnails",
                               "images": checked, "bit_exact": exact}
            if not exact:
                raise SystemExit("bench.py: C4 thumbnails differ from the reference")
        if cpu and ctx.rank == 0:
            host = store[0].cpu().numpy()
            entry["cpu_baseline"] = cpu_baselines(helpers, chain, host, float(n) * n, interp, single_rows=n // 4,
                                                  what="one %dx%dx3 image through the same pipeline" % (n, n))
    del ims, outs, store
    ctx.trim()
    return entry


# ------------------------------------------------------------------------------------- C5

def c5_mask(vh):
    return vh.gaussmat(5, 0.01, False, "float")  # 31 x 31 doubles


def c5_rows_device(torch, width, row0, rows, device):
    """Rows [row0, row0 + rows) of the 65536-wide ushort LCG image (seed 12345): the byte
    stream position of a row is known, so any strip is generated without the rest."""
    # jump the LCG ahead by row0 * width * 2 bytes
    skip = row0 * width * 2
    a, c, s = 1664525, 1013904223, 12345
    # (a, c)^skip by squaring
    ra, rc = 1, 0
    pa, pc = a, c
    k = skip
    while k:
        if k & 1:
            ra, rc = (pa * ra) & M32, (pa * rc + pc) & M32
        pa, pc = (pa * pa) & M32, (pa * pc + pc) & M32
        k >>= 1
    seed = (ra * s + rc) & M32
    img = lcg_image_device(torch, width, rows, 2, seed, device)
    return img.view(torch.uint16).reshape(rows, width, 1)


def run_c5slab(ctx, steps, warmup, verify=True, cpu=True, width=65536, rows=8192, im_height=65536):
    """One GPU's share of BASELINE configs[4]: vips_conv with the 31x31 float mask on rows
    [top, top + rows) of the width x im_height ushort image, its input window carrying the
    15-row halos a neighbour would send (libvips_amd.sharding)."""
    torch, vh = ctx.torch, ctx.vh
    from libvips_amd import sharding

    mask, scale = c5_mask(vh)
    plan = sharding.StripPlan(im_height, im_height, im_height // rows, sharding.conv_need(31, im_height))
    slab = (im_height // rows) // 2  # a slab from the middle: halos on both sides
    w0, w1 = plan.windows[slab]
    with torch.cuda.stream(ctx.stream):
        window = c5_rows_device(torch, width, w0, w1 - w0, ctx.device)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    def step():
        return sharding.conv_strip(window, w0, plan, slab, mask, scale=scale, precision="float")

    elapsed, out = ctx.timed(step, steps, warmup)
    ms = elapsed / steps * 1e3
    report = ctx.gates(step, 1)
    o0, o1 = plan.out_bounds[slab]
    alg = width * (o1 - o0) * (2 + 4)
    flops = 2.0 * 961 * width * (o1 - o0)
    entry = {
        "name": "c5slab",
        "workload": "vips_conv 31x31 float mask, rows %d..%d of the %dx%d u16 image (+15-row halos) -> f32: "
                    "one GPU's share of BASELINE configs[4]" % (o0, o1, width, im_height),
        "ms": round(ms, 3),
        "steps": steps,
        "mpixels_per_s": round(float(width) * (o1 - o0) / (ms * 1e-3) / 1e6, 1),
        "algorithmic_bytes": alg,
        "bound": "fp64",
        "tflops": round(flops / (ms * 1e-3) / 1e12, 2),
        "frac": round(flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
        "frac_hbm": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "dtype": "f64 sums -> f32",
        "kernels": kernels_of(report),
    }
    helpers = ref_or_none()
    if helpers is not None and (verify or cpu):
        if verify:
            worst, exact, checked = 0, True, []
            nrows = 8
            for r0 in (o0, (o0 + o1) // 2, o1 - nrows):
                host = window[r0 - 15 - w0:r0 + nrows + 15 - w0].cpu().numpy()
                want = helpers.Ref.run_mask("conv", host, mask, scale, 0.0, "precision=float")[15:15 + nrows]
                got = out[r0 - o0:r0 - o0 + nrows].cpu().numpy()
                ok, ulp = same_float(got, want)
                exact = exact and ok
                worst = max(worst, ulp if ulp is not None else 1 << 30)
                checked.append([r0, r0 + nrows])
            entry["parity"] = {"against": "oracle/_ref (compiled reference)", "rows": checked,
                               "bit_exact": exact, "max_ulp": worst, "tolerance_ulp": 1}
            if worst > 1:
                raise SystemExit("bench.py: C5 slab differs from the reference by %d ULP" % worst)
        if cpu:
            srows = 64
            host = window[:srows + 30].cpu().numpy()
            cores = os.cpu_count() or 1

            def best_of(rows, reps):
                sample = np.ascontiguousarray(host[:rows])
                best = 1e30
                for _ in range(reps):
                    t1 = time.perf_counter()
                    helpers.Ref.run_mask("conv", sample, mask, scale, 0.0, "precision=float")
                    best = min(best, time.perf_counter() - t1)
                return best

            helpers.Ref.lib().ref_init(cores)
            secs = best_of(srows + 30, 5)
            used = helpers.Ref.concurrency()
            helpers.Ref.lib().ref_init(1)
            rows1 = 34  # (4 output rows' worth of windows: one core takes ~1 s per 16 rows of 65536)
            secs1 = best_of(rows1, 3)
            helpers.Ref.lib().ref_init(cores)
            entry["cpu_baseline"] = {
                "value": round(float(width) * (srows + 30) / secs / 1e6, 1), "unit": "Mpixels/s",
                "cores": used, "kind": "reference",
                "sample": "%d rows x %d u16 of the same window, best of 5; libvips 8.19.0 scalar C path" % (srows + 30, width),
                "single_core": {"value": round(float(width) * rows1 / secs1 / 1e6, 1), "unit": "Mpixels/s", "cores": 1,
                                "sample": "%d rows of it, best of 3, VIPS_CONCURRENCY=1" % rows1}}
    del window, out
    ctx.trim()
    return with_fp64_stream(entry)


def run_c5(ctx, steps, warmup, verify=True, width=65536, im_height=65536):
    """BASELINE configs[4] across the ranks: the image is cut into one row strip per rank,
    every rank generates ITS strip, gets its 15-row halos from its neighbours in one exchange
    (RCCL send/recv) and convolves its strip; strong scaling.  The halo exchange is inside the
    timed step."""
    torch, vh = ctx.torch, ctx.vh
    from libvips_amd import sharding

    mask, scale = c5_mask(vh)
    plan = sharding.StripPlan(im_height, im_height, ctx.world, sharding.conv_need(31, im_height))
    s0, s1 = plan.in_bounds[ctx.rank]  # conv: output rows = input rows
    # the rank's strip lives in its persistent window (sharding.StripWindow): the exchange
    # receives the 15-row halos straight into the window's margins, nothing else moves
    sw = sharding.StripWindow(plan, ctx.rank, (width, 1), torch.uint16, ctx.device)
    with torch.cuda.stream(ctx.stream):
        sw.own.copy_(c5_rows_device(torch, width, s0, s1 - s0, ctx.device))
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    strip = sw.own
    # ... and so does its output strip (17 GB at world 1: not something to allocate per step)
    result = torch.empty((s1 - s0, width, 1), dtype=torch.float32, device=ctx.device)

    def step():
        if ctx.dist is not None:
            sw.exchange(ctx.dist)
        return sharding.conv_strip(sw.window, sw.top, plan, ctx.rank, mask, scale=scale, precision="float",
                                   out=result)

    elapsed, out = ctx.timed(step, steps, warmup)
    ms = elapsed / steps * 1e3
    parity = None
    helpers = ref_or_none()
    if verify and helpers is not None:
        # every rank checks rows at both ends of its strip (the rows that depend on the halos)
        worst, exact, checked = 0, True, []
        nrows = 4
        for r0 in (s0, s1 - nrows):
            i0, i1 = max(r0 - 15, 0), min(r0 + nrows + 15, im_height)
            host = c5_rows_device(torch, width, i0, i1 - i0, ctx.device).cpu().numpy()
            want = helpers.Ref.run_mask("conv", host, mask, scale, 0.0, "precision=float")[r0 - i0:r0 - i0 + nrows]
            got = out[r0 - s0:r0 - s0 + nrows].cpu().numpy()
            ok, ulp = same_float(got, want)
            exact = exact and ok
            worst = max(worst, ulp if ulp is not None else 1 << 30)
            checked.append([r0, r0 + nrows])
        flag = torch.tensor([worst], dtype=torch.int64, device=ctx.device)
        if ctx.dist is not None:
            ctx.dist.all_reduce(flag, op=ctx.dist.ReduceOp.MAX)
        worst = int(flag.item())
        parity = {"against": "oracle/_ref (compiled reference), first and last %d rows of every rank's strip" % nrows,
                  "rank0_rows": checked, "max_ulp_all_ranks": worst, "tolerance_ulp": 1}
        if worst > 1:
            raise SystemExit("bench.py: C5 strips differ from the reference by %d ULP" % worst)
    flops = 2.0 * 961 * width * im_height
    alg = width * im_height * 6
    line = {
        "metric": "Mpixels/s, vips_conv 31x31 float mask on %dx%d ushort in row strips over the GPUs" % (width, im_height),
        "value": round(float(width) * im_height / (ms * 1e-3) / 1e6, 1),
        "unit": "Mpixels/s",
        "n_gpus": ctx.world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(ms, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (LCG ushorts, seed 12345, every rank generates its strip on its device)",
        "config": {"workload": "vips_conv(31x31 gaussmat sigma 5, precision=float) %dx%dx1 u16 -> f32, "
                               "BASELINE configs[4]" % (width, im_height),
                   "partition": "%d row strips, one 15-row halo exchange per step over RCCL send/recv" % ctx.world},
        "roofline": {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 2), "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                     "traffic": None, "algorithmic_bytes": alg,
                     "note": "FP64 compute bound (961 taps per pixel); peak = the 78.6 TFLOP/s FP64 rate"},
        "parity": parity,
        "cpu_baseline": None,
    }
    del strip, out, result, sw
    ctx.trim()
    return line


def c5_entry(line):
    """The whole-image C5 line (world 1: the 65536 x 65536 image as ONE strip on this GPU) as a
    configs[] entry next to the slab."""
    roof = line["roofline"]
    return with_fp64_stream({
        "name": "c5",
        "workload": line["config"]["workload"] + " -- the WHOLE image on one GPU (25.8 GB in + out)",
        "ms": line["ms_per_step"],
        "steps": line["steps"],
        "mpixels_per_s": line["value"],
        "algorithmic_bytes": roof["algorithmic_bytes"],
        "bound": "fp64",
        "tflops": roof["achieved"],
        "frac": roof["frac"],
        "frac_hbm": round(roof["algorithmic_bytes"] / (line["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "dtype": "f64 sums -> f32",
        "parity": line["parity"],
    })


# ------------------------------------------------------------------------------------- ops

def ops_table():
    """The single calls of the path a libvips user makes one at a time (north_star's list), each
    on its own synthetic image: (name, input spec, device call, reference chain / mask call).
    Input spec = (edge, bands, dtype, interpretation, how it is made)."""
    k3 = np.array([[-1, -1, -1], [-1, 16, -1], [-1, -1, -1]], dtype=np.float64)  # the 3 x 3 sharpen mask of conv.c's doc
    b5 = np.array([1, 4, 6, 4, 1], dtype=np.float64)
    k5 = np.outer(b5, b5)  # 5 x 5 binomial, scale 256
    ops = []

    def chain(name, spec, call, chain_text, **kw):
        ops.append(dict(name=name, spec=spec, call=call, chain=chain_text, **kw))

    def masked(name, spec, call, nick, mask, scale, args, **kw):
        ops.append(dict(name=name, spec=spec, call=call, mask=(nick, mask, scale, args), **kw))

    rgb8 = (8192, 3, "u8", "srgb", "lcg")
    rgb16 = (8192, 3, "u16", "rgb16", "lcg")
    rgba8 = (8192, 4, "u8", "srgb", "lcg")
    rgba16 = (16384, 4, "u16", "rgb16", "lcg")
    rgbf = (8192, 3, "f32", "srgb", "lcg")
    chain("reducev_8", rgb8, lambda im: im.reducev(8), "reducev:vshrink=8,kernel=lanczos3")
    chain("reduceh_8", rgb8, lambda im: im.reduceh(8), "reduceh:hshrink=8,kernel=lanczos3")
    chain("reduce_rgb_8", rgb8, lambda im: im.reduce(8, 8), "reduce:hshrink=8,vshrink=8,kernel=lanczos3")
    chain("reduce_rgb_7.3", rgb8, lambda im: im.reduce(7.3, 7.3), "reduce:hshrink=7.3,vshrink=7.3,kernel=lanczos3")
    # the reference's common case: a size that does not divide the image (thumbnail.c:413, resize.c:207-228) --
    # the whole vips_resize chain (box shrinks, then fractional reduces) and vips_thumbnail_image
    chain("resize_rgb_to_1000", rgb8, lambda im: im.resize(1000.0 / 8192.0), "resize:scale=%r" % (1000.0 / 8192.0))
    chain("thumbnail_500", rgb8, lambda im: im.thumbnail_image(500), "thumbnail_image:width=500")
    chain("shrinkv_4", rgb8, lambda im: im.shrinkv(4), "shrinkv:vshrink=4")
    chain("shrinkh_4", rgb8, lambda im: im.shrinkh(4), "shrinkh:hshrink=4")
    masked("convi_3x3_u8", rgb8, lambda im: im.conv(k3, scale=8, precision="integer"), "conv", k3, 8.0, "precision=integer")
    masked("convi_5x5_u8", rgb8, lambda im: im.conv(k5, scale=256, precision="integer"), "conv", k5, 256.0, "precision=integer")
    masked("convi_3x3_u16", rgb16, lambda im: im.conv(k3, scale=8, precision="integer"), "conv", k3, 8.0, "precision=integer")
    masked("convi_5x5_u16", rgb16, lambda im: im.conv(k5, scale=256, precision="integer"), "conv", k5, 256.0, "precision=integer")
    chain("gaussblur_s2_u8", rgb8, lambda im: im.gaussblur(2.0), "gaussblur:sigma=2")
    chain("gaussblur_s8_u8", rgb8, lambda im: im.gaussblur(8.0), "gaussblur:sigma=8")
    chain("gaussblur_s2_u16", rgb16, lambda im: im.gaussblur(2.0), "gaussblur:sigma=2")
    chain("gaussblur_s8_u16", rgb16, lambda im: im.gaussblur(8.0), "gaussblur:sigma=8")
    chain("gaussblur_s2_f32", rgbf, lambda im: im.gaussblur(2.0, precision="float"), "gaussblur:sigma=2,precision=float",
          float_out=True)
    chain("colourspace_srgb_lab_u8", rgb8, lambda im: im.colourspace("lab"), "colourspace:space=lab", float_out=True)
    chain("colourspace_srgb_lab_f32", rgbf, lambda im: im.colourspace("lab"), "colourspace:space=lab", float_out=True)
    chain("colourspace_srgb_labs_u8", rgb8, lambda im: im.colourspace("labs"), "colourspace:space=labs")
    chain("colourspace_lab_srgb_f32", (8192, 3, "f32", "lab", "lab"), lambda im: im.colourspace("srgb"),
          "colourspace:space=srgb")
    chain("sharpen_u8", rgb8, lambda im: im.sharpen(), "sharpen:")
    chain("cast_u8_f32", rgb8, lambda im: im.cast("float"), "cast:format=float", float_out=True)
    chain("premultiply_u8", rgba8, lambda im: im.premultiply(), "premultiply:", float_out=True)
    chain("reduce_rgba16_8", rgba16, lambda im: im.reduce(8, 8), "reduce:hshrink=8,vshrink=8,kernel=lanczos3")
    chain("shrink_rgba16_4", rgba16, lambda im: im.shrink(4, 4), "shrink:hshrink=4,vshrink=4")
    # ---- round 6 (VERDICT r5 "missing 4"): the rows north_star names that the table never timed
    chain("shrinkv_rgba16_4", rgba16, lambda im: im.shrinkv(4), "shrinkv:vshrink=4")
    chain("shrinkh_rgba16_4", rgba16, lambda im: im.shrinkh(4), "shrinkh:hshrink=4")
    chain("reduce_f32_8", rgbf, lambda im: im.reduce(8, 8), "reduce:hshrink=8,vshrink=8,kernel=lanczos3", float_out=True)
    chain("shrink_f32_4", rgbf, lambda im: im.shrink(4, 4), "shrink:hshrink=4,vshrink=4", float_out=True)
    chain("unpremultiply_u8", rgba8, lambda im: im.unpremultiply(), "unpremultiply:", float_out=True)
    # vips_thumbnail_image of an RGBA image: premultiply -> resize -> unpremultiply -> cast (thumbnail.c:848-904)
    chain("thumbnail_rgba_500", rgba8, lambda im: im.thumbnail_image(500), "thumbnail_image:width=500")
    # the upsizing half of vips_resize: vips_affine + bicubic (resize.c:230-300, bicubic.cpp:482-600)
    chain("resize_bicubic_x2.5", (3276, 3, "u8", "srgb", "lcg"), lambda im: im.resize(2.5, kernel="cubic"),
          "resize:scale=2.5,kernel=cubic")
    chain("colourspace_lab_xyz_f32", (8192, 3, "f32", "lab", "lab"), lambda im: im.colourspace("xyz"),
          "colourspace:space=xyz", float_out=True)
    chain("colourspace_xyz_scrgb_f32", (8192, 3, "f32", "xyz", "xyz"), lambda im: im.colourspace("scrgb"),
          "colourspace:space=scrgb", float_out=True)
    masked("conva_5x5_u8", rgb8, lambda im: im.conv(k5, scale=256, precision="approximate"), "conv", k5, 256.0,
           "precision=approximate")
    # vips_sharpen on an image whose neighbouring pixels are near each other (a photograph's case; the LCG noise
    # above is the other end: every pixel outside the LUT's flat centre): bilinear x 8 of 1024 x 1024 of noise
    chain("sharpen_u8_smooth", (8192, 3, "u8", "srgb", "smooth"), lambda im: im.sharpen(), "sharpen:")
    chain("sharpen_u8_smooth_2k", (2000, 3, "u8", "srgb", "smooth"), lambda im: im.sharpen(), "sharpen:")
    return ops


def run_ops(ctx, steps, warmup, verify=True, cpu=True, only=None):
    """One entry per single call of the path: ms per call (HIP events around `steps` calls on the
    library's stream), algorithmic bytes (every input byte read once, every output byte written
    once), the fraction of the 8 TB/s HBM roofline, the kernels that ran (gates), the WHOLE output
    compared with the compiled reference, and the reference's own time on this box's cores."""
    torch = ctx.torch
    from libvips_amd import Image

    helpers = ref_or_none()
    dtypes = {"u8": torch.uint8, "u16": torch.uint16, "f32": torch.float32}
    cores = os.cpu_count() or 1
    entries = []
    made = {}
    for op in ops_table():
        if only and not any(s in op["name"] for s in only):
            continue
        edge, bands, dt, interp, how = op["spec"]
        key = op["spec"]
        if key not in made:
            made.clear()  # one input image alive at a time
            ctx.trim()
            with torch.cuda.stream(ctx.stream):
                if dt == "u16":
                    src = lcg_image_device(torch, edge, edge, 2 * bands, 12345, ctx.device).view(torch.uint16)
                    src = src.reshape(edge, edge, bands)
                elif how in ("lab", "xyz"):
                    u8 = lcg_image_device(torch, edge, edge, bands, 12345, ctx.device)
                    lab = Image.new_from_tensor(u8, interpretation="srgb").colourspace(how)
                    src = torch.from_numpy(lab.numpy()).to(ctx.device)
                    del lab, u8
                elif how == "smooth":
                    small = lcg_image_device(torch, (edge + 7) // 8, (edge + 7) // 8, bands, 12345, ctx.device)
                    up = torch.nn.functional.interpolate(small.permute(2, 0, 1)[None].float(), scale_factor=8,
                                                         mode="bilinear", align_corners=False)
                    src = up[0, :, :edge, :edge].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).contiguous()
                    del small, up
                else:
                    src = lcg_image_device(torch, edge, edge, bands, 12345, ctx.device)
                    if dt == "f32":
                        src = src.float()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            made[key] = src
        src = made[key]
        im = Image.new_from_tensor(src, interpretation=interp)
        call = op["call"]

        def step():
            return call(im)

        elapsed, out = ctx.timed(step, steps, warmup)
        ms = ctx.event_ms / steps
        report = ctx.gates(step, 2)
        out_bytes = out.width * out.height * out.bands * {"uchar": 1, "char": 1, "ushort": 2, "short": 2, "uint": 4,
                                                           "int": 4, "float": 4, "double": 8}.get(out.format, 4)
        alg = src.numel() * src.element_size() + out_bytes
        dominant = max(report.items(), key=lambda kv: kv[1][1])[0] if report else None
        entry = {
            "name": op["name"],
            "input": "%dx%dx%d %s" % (edge, edge, bands, dt),
            "output": "%dx%dx%d %s" % (out.width, out.height, out.bands, out.format),
            "ms": round(ms, 4),
            "steps": steps,
            "algorithmic_bytes": alg,
            "GBps": round(alg / (ms * 1e-3) / 1e9, 1),
            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "mpixels_per_s": round(float(edge) * edge / (ms * 1e-3) / 1e6, 1),
            "kernel": dominant,
            "kernels": kernels_of(report),
        }
        if helpers is not None and (verify or cpu):
            host = src.cpu().numpy()
            ri = helpers.INTERP[interp]
            t1 = time.perf_counter()
            if "chain" in op:
                want = helpers.Ref.run_chain(op["chain"], host, ri)
            else:
                nick, mask, scale, margs = op["mask"]
                want = helpers.Ref.run_mask(nick, host, mask, scale, 0.0, margs, ri)
            secs = time.perf_counter() - t1
            if verify:
                got = out.numpy()
                if op.get("float_out"):
                    ok, ulp = same_float(got, want)
                    entry["parity"] = {"against": "oracle/_ref, whole output", "bit_exact": bool(ok), "max_ulp": ulp,
                                       "tolerance_ulp": 1}
                    bad = ulp is None or ulp > 1
                else:
                    ok = got.shape == want.shape and got.dtype == want.dtype and bool(np.array_equal(got, want))
                    entry["parity"] = {"against": "oracle/_ref, whole output", "bit_exact": ok}
                    bad = not ok
                if bad:
                    raise SystemExit("bench.py: ops[%s] differs from the reference: %r" % (op["name"], entry["parity"]))
                del got
            if cpu:
                if "chain" in op:
                    secs = helpers.Ref.time_chain(op["chain"], host, repeats=2, interpretation=ri, concurrency=cores)
                entry["cpu_baseline"] = {"value": round(float(edge) * edge / secs / 1e6, 1), "unit": "Mpixels/s",
                                         "cores": helpers.Ref.concurrency(), "kind": "reference",
                                         "sample": "the whole image, %s" % ("best of 2" if "chain" in op else
                                                                              "one run incl. the copy out")}
            del host, want
        del im, out
        entries.append(entry)
    made.clear()
    ctx.trim()
    return entries


def entry_as_line(entry, ctx, steps, warmup, metric, scaling="weak"):
    """A configs[] entry promoted to the bench line (--config c3 / c4 / c5slab)."""
    bound = entry.get("bound", "hbm")
    ms = entry["ms"]
    if bound == "hbm":
        roof = {"bound": "hbm", "achieved": round(entry["algorithmic_bytes"] / (ms * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": entry["frac"], "traffic": None,
                "algorithmic_bytes": entry["algorithmic_bytes"],
                "kernels": entry.get("kernels") or entry.get("kernels_per_image")}
    else:
        roof = {"bound": "mfma", "achieved": entry["tflops"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": entry["frac"], "traffic": None, "algorithmic_bytes": entry["algorithmic_bytes"],
                "kernels": entry.get("kernels"),
                "note": "FP64 compute bound (961 taps per pixel); peak = the 78.6 TFLOP/s FP64 rate"}
    return {
        "metric": metric,
        "value": entry["mpixels_per_s"],
        "unit": "Mpixels/s",
        "n_gpus": ctx.world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": entry["dtype"],
        "data": "synthetic (LCG bytes, seed 12345 + image index, generated on device)",
        "config": {k: entry[k] for k in ("workload", "images_per_gpu", "ms_per_image", "images_per_s",
                                         "images_per_s_per_rank") if k in entry},
        "roofline": roof,
        "parity": entry.get("parity"),
        "cpu_baseline": entry.get("cpu_baseline"),
    }


def compact_summary(line):
    """The other BASELINE configs and the per-call table as SCALARS inside `roofline` (the driver's stored
    record keeps the scalar fields of `roofline`); nothing nested is added."""
    by = {e.get("name"): e for e in line.get("configs", []) if isinstance(e, dict)}
    others = {}
    c3, c4, c5, c5s, c1 = (by.get(k) for k in ("c3", "c4", "c5", "c5slab", "c1"))
    if c3:
        others.update({"c3_ms": c3["ms"], "c3_frac_fp64": c3["frac_fp64"], "c3_frac_hbm": c3["frac_hbm"]})
        if isinstance(c3.get("float_input"), dict) and "ms" in c3["float_input"]:
            others["c3_ms_float_input"] = c3["float_input"]["ms"]
        if "frac_of_fp64_stream" in c3:
            others["c3_frac_of_fp64_stream"] = c3["frac_of_fp64_stream"]
    if c4:
        others.update({"c4_ms": c4["ms"], "c4_frac": c4["frac"], "c4_ms_per_image": c4["ms_per_image"],
                       "c4_images": c4["images_per_gpu"]})
        for kname, kv in (c4.get("kernels_per_image") or {}).items():
            if isinstance(kv, dict) and "ms_per_image" in kv:
                others["c4_%s_ms_per_image" % kname] = kv["ms_per_image"]
    if c5:
        others.update({"c5_ms": c5["ms"], "c5_tflops": c5.get("tflops"), "c5_frac_fp64": c5.get("frac")})
    if c5s:
        others.update({"c5slab_ms": c5s["ms"], "c5slab_tflops": c5s.get("tflops")})
    if c1:
        others["c1_ms"] = c1["ms"]
    e2e = by.get("module_e2e")
    if e2e and "ms" in e2e:
        others["module_e2e_ms"] = e2e["ms"]
    ops = {e["name"]: e["frac"] for e in line.get("ops", []) if isinstance(e, dict) and "frac" in e}
    roof = line.get("roofline")
    if roof is not None:
        roof.update(others)
        for name, frac in ops.items():
            roof["op_%s_frac" % name] = frac


FINAL_LINE_LIMIT = 6000  # characters: the driver parses the LAST stdout line (round 5's 24.5 KB line was not parsed)
STANDARD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")


def _clip(v, n):
    """Strings cut to n characters; a float that is not finite becomes None (the line must parse strictly)."""
    if isinstance(v, float) and (v != v or v in (float("inf"), float("-inf"))):
        return None
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "\u2026"


def final_line(line, full_path=None):
    """The record the driver parses: the LAST line of stdout, small.  The standard keys, `config`, a FLAT
    `roofline` (scalars only: C2's own figures, the other configs' and every single call's fraction),
    `cpu_baseline`, `parity`, `frac_cold` -- and nothing else; the full table (`configs`, `ops`, `settled`,
    per-kernel gate times) goes to `full_path` and to an earlier stdout line.  Guaranteed to stay under
    FINAL_LINE_LIMIT characters: the per-call fractions are the first thing dropped, loudly."""
    out = {k: _clip(line[k], 200) for k in STANDARD_KEYS if k in line}
    if isinstance(out.get("config"), dict):
        out["config"] = {k: _clip(v, 160) for k, v in out["config"].items() if not isinstance(v, (dict, list))}
    roof = line.get("roofline")
    if isinstance(roof, dict):
        out["roofline"] = {k: _clip(v, 80) for k, v in roof.items() if not isinstance(v, (dict, list))}
    else:
        out["roofline"] = roof
    cpu = line.get("cpu_baseline")
    if isinstance(cpu, dict):
        flat = {k: _clip(v, 200) for k, v in cpu.items() if not isinstance(v, (dict, list))}
        single = cpu.get("single_core")
        if isinstance(single, dict) and "value" in single:
            flat["single_core_value"] = single["value"]
        out["cpu_baseline"] = flat
    else:
        out["cpu_baseline"] = cpu
    par = line.get("parity")
    entries = [e for e in line.get("configs", []) + line.get("ops", []) if isinstance(e, dict)]
    verdicts = {e.get("name"): (e.get("parity") or {}).get("bit_exact") for e in entries}
    if isinstance(par, dict) or verdicts:
        flat = {k: _clip(v, 120) for k, v in (par or {}).items() if not isinstance(v, (dict, list))}
        if verdicts:
            flat["entries_checked"] = sum(1 for v in verdicts.values() if v is not None)
            flat["entries_bit_exact"] = sum(1 for v in verdicts.values() if v is True)
            # (float entries within 1 ULP but not bit-exact, or anything not compared, are NAMED)
            flat["entries_not_bit_exact"] = ",".join(sorted(str(k) for k, v in verdicts.items() if v is False))
            flat["entries_unchecked"] = ",".join(sorted(str(k) for k, v in verdicts.items() if v is None))
        out["parity"] = flat
    else:
        out["parity"] = par
    if isinstance(roof, dict) and "frac_cold" in roof:
        out["frac_cold"] = roof["frac_cold"]
    if isinstance(line.get("settled"), dict):
        out["ms_per_step_settled"] = line["settled"].get("ms_per_step")
    if full_path:
        out["full"] = full_path
    if len(json.dumps(out)) > FINAL_LINE_LIMIT and isinstance(out.get("roofline"), dict):
        dropped = [k for k in out["roofline"] if k.startswith("op_")]
        for k in dropped:
            del out["roofline"][k]
        out["roofline"]["op_fracs_dropped"] = len(dropped)
    if len(json.dumps(out)) > FINAL_LINE_LIMIT:  # never print a line the driver cannot read
        out = {k: out[k] for k in STANDARD_KEYS if k in out}
        out["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} \
            if isinstance(roof, dict) else None
        out["cpu_baseline"] = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")} if isinstance(cpu, dict) else None
    return out


def write_full(line):
    """The whole table where it can be read later: gpurun_out/ (merged back from the GPU box) if it exists or
    can be made, else the working directory.  Returns the path relative to the repo, or None."""
    for d in (os.path.join(ROOT, "gpurun_out"), os.getcwd()):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_full.json")
            with open(path, "w") as fh:
                json.dump(line, fh, indent=1)
                fh.write("\n")
            return os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5slab", "c5", "ops"])
    ap.add_argument("--ops", default=None, help="ops: comma-separated substrings of the entries to run (default all)")
    ap.add_argument("--size", type=int, default=None, help="image edge (default: the BASELINE size of the config)")
    ap.add_argument("--images", type=int, default=None, help="c4: images per GPU (default 1024 at N=1 -- the whole "
                                                             "BASELINE batch, 206 GB, on one GPU -- and 1024 / N at N>1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-settle", action="store_true",
                    help="c2: time the first W + K launches of the process only (no clock-settling phase)")
    ap.add_argument("--no-configs", action="store_true", help="c2: leave the C3/C4/C5 entries out")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparisons with the reference")
    args = ap.parse_args()

    ctx = Ctx(args)
    verify = not args.no_verify
    cpu = not args.no_cpu_baseline and ctx.rank == 0 and ctx.world == 1
    line = None
    if args.config == "c2":
        args.size = args.size or 16384
        line = run_c2(ctx, args)
        if ctx.rank == 0 and ctx.world == 1 and not args.no_configs and args.size == 16384:
            k = max(2, min(args.steps, 5))
            line["configs"] = [
                run_c1(ctx, 20, 3, verify, cpu),
                run_c3(ctx, k, 2, verify, cpu),
                run_c4(ctx, max(1, min(args.steps, 3)), 1, args.images or 1024, verify, cpu),
                run_c5slab(ctx, max(2, min(args.steps, 4)), 2, verify, cpu),
                c5_entry(run_c5(ctx, 2, 1, verify)),
            ]
            line["ops"] = run_ops(ctx, 5, 2, verify, cpu)
            if verify and cpu:
                e2e = run_module_e2e(ctx, line["roofline"]["kernel_ms"] if line.get("roofline") else None)
                if e2e:
                    line["configs"].append(e2e)
            compact_summary(line)
    elif args.config == "c3":
        e = run_c3(ctx, args.steps, args.warmup, verify, cpu, args.size or 32768)
        line = entry_as_line(e, ctx, args.steps, args.warmup,
                             "Mpixels/s, vips_gaussblur(sigma 8) + sRGB->Lab on 32768x32768x3 float")
    elif args.config == "c4":
        images = args.images or max(1, 1024 // ctx.world)
        e = run_c4(ctx, args.steps, args.warmup, images, verify, cpu, args.size or 8192)
        line = entry_as_line(e, ctx, args.steps, args.warmup,
                             "Mpixels/s (input), batched thumbnail pipeline resize(1/8)+sharpen over 8192x8192x3 uchar images")
    elif args.config == "c5slab":
        size = args.size or 65536
        e = run_c5slab(ctx, args.steps, args.warmup, verify, cpu, width=size, rows=max(size // 8, 64), im_height=size)
        line = entry_as_line(e, ctx, args.steps, args.warmup,
                             "Mpixels/s, vips_conv 31x31 float mask on a 65536x8192 ushort slab (+halos)")
    elif args.config == "ops":
        ops = run_ops(ctx, args.steps, args.warmup, verify, cpu, only=args.ops.split(",") if args.ops else None)
        worst = min(ops, key=lambda e: e["frac"])
        line = {
            "metric": "fraction of the HBM roofline of the slowest single call of the path (table in `ops`)",
            "value": worst["frac"], "unit": "fraction of 8 TB/s", "n_gpus": ctx.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": worst["ms"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/u16/f32 per entry", "data": "synthetic (LCG bytes, seed 12345)",
            "config": {"workload": "every single call of the path on its own image, slowest: %s" % worst["name"]},
            "roofline": {"bound": "hbm", "achieved": worst["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": worst["frac"], "traffic": None, "kernel": worst["kernel"]},
            "cpu_baseline": worst.get("cpu_baseline"),
            "ops": ops,
        }
    else:
        size = args.size or 65536
        line = run_c5(ctx, args.steps, args.warmup, verify, width=size, im_height=size)
    if ctx.rank == 0:
        # the full table first (a file, and an earlier stdout line with a prefix no JSON reader takes for a
        # record), then the compact record as the LAST line: that is the one the driver parses
        full_path = write_full(line)
        print("bench_full " + json.dumps(line))
        sys.stdout.flush()
        print(json.dumps(final_line(line, full_path), allow_nan=False))
    ctx.close()


if __name__ == "__main__":
    main()
