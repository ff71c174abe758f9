"""Multi-GPU partition logic, covered on CPU with world_size-2/3 gloo process groups:
strip bounds, halo plans from the C ABI's need() functions, the point-to-point halo
exchange, and that a rank's window holds exactly the rows of the full image it needs
(so running the region op on it reproduces the single-GPU result rows -- checked with the
oracle port, which is what the GPU test then repeats through the HIP kernel)."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from libvips_amd import _ffi, sharding
from tests import helpers
from tests.helpers import PortCC


def test_batch_and_strip_bounds():
    assert sharding.batch_indices(10, 4, 1) == [1, 5, 9]
    assert sum(len(sharding.batch_indices(1024, 8, r)) for r in range(8)) == 1024
    for total in (1, 7, 8, 65536, 2049):
        for world in (1, 2, 3, 8):
            bounds = [sharding.strip_bounds(total, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == total
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in bounds]
            assert max(sizes) - min(sizes) <= 1


def test_plan_conv_halos():
    # BASELINE config 5 geometry: 31x31 mask, 65536 rows over 8 GPUs -> 15-row halos
    plan = sharding.StripPlan(65536, 65536, 8, sharding.conv_need(31, 65536))
    assert plan.windows[0] == (0, 8192 + 15)
    assert plan.windows[3] == (3 * 8192 - 15, 4 * 8192 + 15)
    assert plan.windows[7] == (7 * 8192 - 15, 65536)
    tr = plan.transfers()
    assert len(tr) == 14  # 7 neighbour pairs, both directions
    assert all(hi - lo == 15 for _, _, lo, hi in tr)
    assert all(abs(s - d) == 1 for s, d, _, _ in tr)


def test_plan_reducev_halos():
    # C2 split over 2 GPUs: x8 lanczos3 (49 taps) needs rows [8*top - 20, ...) clipped
    lib = _ffi.lib
    r = lib.vips_hip_reduce_new(5, 8.0, 16384, 2048, math.nan)
    try:
        plan = sharding.StripPlan(16384, 2048, 2, sharding.reducev_need(r))
        assert plan.out_bounds == [(0, 1024), (1024, 2048)]
        # reducev.cpp:539-542 in embedded rows, minus the 24-row embed, clipped to the image
        assert plan.windows[0] == (0, 8 * 1024 + 49 - 24)
        assert plan.windows[1] == (8 * 1024 - 24, 16384)
    finally:
        lib.vips_hip_reduce_free(r)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, width, bands, mask_h, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = helpers.lcg_image(width, height, bands, np.uint16, 91)
        plan = sharding.StripPlan(height, height, world, sharding.conv_need(mask_h, height))
        i0, i1 = plan.in_bounds[rank]
        strip = torch.from_numpy(np.ascontiguousarray(full[i0:i1]).view(np.int16))
        window, top = sharding.exchange_halos(strip, plan, rank, dist)
        np.save(os.path.join(out_dir, "w%d.npy" % rank), window.numpy().view(np.uint16))
        np.save(os.path.join(out_dir, "t%d.npy" % rank), np.array([top]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_gloo(world, tmp_path):
    height, width, bands, mask_h = 97, 40, 1, 31
    port = _free_port()
    mp.spawn(_worker, args=(world, port, height, width, bands, mask_h, str(tmp_path)), nprocs=world,
             join=True)
    full = helpers.lcg_image(width, height, bands, np.uint16, 91)
    plan = sharding.StripPlan(height, height, world, sharding.conv_need(mask_h, height))
    mask, scale = PortCC.gaussmat(5, 0.01, False, "float")
    assert mask.shape[0] == mask_h
    want = PortCC.conv(full, mask, scale, 0.0, "float")
    for rank in range(world):
        window = np.load(os.path.join(str(tmp_path), "w%d.npy" % rank))
        top = int(np.load(os.path.join(str(tmp_path), "t%d.npy" % rank))[0])
        w0, w1 = plan.windows[rank]
        assert top == w0 and window.shape[0] == w1 - w0
        # the window is exactly those rows of the full image
        assert np.array_equal(window, full[w0:w1])
        # and it is enough: a conv of the window whose own edges are NOT image edges gives the
        # strip's rows, except where the window edge is the image edge (clamp = same thing)
        o0, o1 = plan.out_bounds[rank]
        local = PortCC.conv(np.ascontiguousarray(window), mask, scale, 0.0, "float")
        lo, hi = o0 - w0, o1 - w0
        inner = slice(lo, hi)
        exp = want[o0:o1]
        got = local[inner]
        # rows whose taps would clamp at a window edge that is not an image edge are the
        # halo rows themselves, which lie outside [lo, hi): so the strips must agree
        assert np.array_equal(got.view(np.uint8), exp.view(np.uint8)), rank


@pytest.mark.gpu
@pytest.mark.parametrize("height,width,world", [(600, 400, 4), (602, 640, 3), (1031, 2100, 5)])
def test_conv_strips_on_gpu_match_whole_image(height, width, world):
    """Single GPU, world emulated sequentially: every rank's window -> vips_hip_conv_gen
    reproduces the rows of the whole-image conv (what the 8-GPU C5 run does per device).
    Widths >= 512 take the LDS row-streaming kernel (C5's own); strips whose height is not a
    multiple of its 4-row blocks sit in windows that end exactly at their last needed row
    (the last block must not read below the window)."""
    import libvips_amd
    from libvips_amd import Image

    libvips_amd.init(0)
    full = helpers.lcg_image(width, height, 1, np.uint16, 92)
    mask, scale = libvips_amd.gaussmat(5, 0.01, False, "float")
    whole = Image.new_from_array(full).conv(mask, scale=scale, precision="float").numpy()
    plan = sharding.StripPlan(height, height, world, sharding.conv_need(mask.shape[0], height))
    for rank in range(world):
        w0, w1 = plan.windows[rank]
        window = torch.from_numpy(np.ascontiguousarray(full[w0:w1]).view(np.int16)).cuda()
        out = sharding.conv_strip(window.view(torch.uint16), w0, plan, rank, mask, scale, 0.0, "float")
        o0, o1 = plan.out_bounds[rank]
        assert np.array_equal(out.cpu().numpy().view(np.uint8), whole[o0:o1].view(np.uint8))


@pytest.mark.gpu
def test_reduce_strips_on_gpu_match_whole_image():
    """BASELINE config 2 split into row strips (single GPU, world emulated sequentially): each
    rank's window (its rows plus the halo vips_hip_reducev_need reports) through the fused
    vips_hip_reduce_gen gives exactly its rows of the whole-image reduce."""
    import math

    import libvips_amd
    from libvips_amd import KERNELS, Image, lib
    from libvips_amd._ffi import check_handle

    libvips_amd.init(0)
    height, width = 2051, 1600
    full = helpers.lcg_image(width, height, 4, np.uint8, 93)
    whole = Image.new_from_array(full).reduce(8, 8, kernel="lanczos3").numpy()
    out_height = whole.shape[0]
    world = 4
    rv = check_handle(lib.vips_hip_reduce_new(KERNELS["lanczos3"], 8.0, height, out_height, math.nan))
    try:
        plan = sharding.StripPlan(height, out_height, world, sharding.reducev_need(rv))
        for rank in range(world):
            w0, w1 = plan.windows[rank]
            window = torch.from_numpy(np.ascontiguousarray(full[w0:w1])).cuda()
            out = sharding.reduce_strip(window, w0, plan, rank, width, 8.0, 8.0)
            o0, o1 = plan.out_bounds[rank]
            assert np.array_equal(out.cpu().numpy(), whole[o0:o1]), rank
    finally:
        lib.vips_hip_reduce_free(rv)


def _bench_c5_worker(rank, world, port, size, out_dir):
    """What bench.py --config c5 does per rank up to the kernel call, on CPU tensors over gloo:
    strip plan -> this rank generates ITS rows with the jump-ahead LCG -> one halo exchange."""
    import torch.distributed as dist

    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert dist.get_world_size() == world
        plan = sharding.StripPlan(size, size, world, sharding.conv_need(31, size))
        s0, s1 = plan.in_bounds[rank]
        strip = bench.c5_rows_device(torch, size, s0, s1 - s0, torch.device("cpu"))
        # as bench.run_c5: the strip lives in the rank's persistent window, every step's exchange
        # receives into the window's margins (two steps here: nothing is reallocated or rebuilt)
        sw = sharding.StripWindow(plan, rank, (size, 1), torch.int16, torch.device("cpu"))
        sw.own.copy_(strip.view(torch.int16))
        where = sw.window.data_ptr()
        for step in range(2):
            if step:
                for t in plan.transfers():  # scribble over the halos: the next exchange must refill them
                    if t[1] == rank:
                        sw.window[t[2] - sw.top:t[3] - sw.top] = -1
            window, w0 = sw.exchange(dist)
            assert window.data_ptr() == where and w0 == sw.top
        np.save(os.path.join(out_dir, "w%d.npy" % rank), window.numpy().view(np.uint16))
        np.save(os.path.join(out_dir, "t%d.npy" % rank), np.array([w0]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_c5_program_gloo(world, tmp_path):
    """bench.py --config c5 (BASELINE config 4... configs[4]: the 65536^2 conv in row strips with
    one RCCL halo exchange), its multi-rank part on CPU: every rank's window -- own strip from
    the jump-ahead generator plus the rows its neighbours sent -- is exactly those rows of the
    one-piece image, and a conv of the window gives the rank's rows of the whole-image conv."""
    size = 192
    port = _free_port()
    mp.spawn(_bench_c5_worker, args=(world, port, size, str(tmp_path)), nprocs=world, join=True)
    full = helpers.lcg_bytes(size * size * 2, 12345).view(np.uint16).reshape(size, size, 1)
    plan = sharding.StripPlan(size, size, world, sharding.conv_need(31, size))
    mask, scale = PortCC.gaussmat(5, 0.01, False, "float")
    want = PortCC.conv(full, mask, scale, 0.0, "float")
    for rank in range(world):
        window = np.load(os.path.join(str(tmp_path), "w%d.npy" % rank))
        w0, w1 = plan.windows[rank]
        assert int(np.load(os.path.join(str(tmp_path), "t%d.npy" % rank))[0]) == w0
        assert np.array_equal(window, full[w0:w1])
        o0, o1 = plan.out_bounds[rank]
        local = PortCC.conv(np.ascontiguousarray(window), mask, scale, 0.0, "float")
        assert np.array_equal(local[o0 - w0:o1 - w0].view(np.uint8), want[o0:o1].view(np.uint8)), rank


def test_bench_c4_partition():
    """bench.py --config c4 deals the batch round-robin (sharding.batch_indices): every image
    goes to exactly one rank, ranks differ by at most one image, seeds follow the image index."""
    for world in (1, 2, 3, 8):
        total = 128 * world + (3 if world == 3 else 0)
        seen = []
        for rank in range(world):
            mine = sharding.batch_indices(total, world, rank)
            assert all(i % world == rank for i in mine)
            seen += mine
        assert sorted(seen) == list(range(total))


@pytest.mark.gpu
@pytest.mark.parametrize("config,extra", [("c5", ["--size", "2048"]), ("c4", ["--size", "1024", "--images", "6"]),
                                          ("c2", ["--size", "2048", "--no-configs"])])
def test_bench_programs_under_torchrun(config, extra):
    """The commands the driver launches for N > 1 (python -m torch.distributed.run ... bench.py
    --gpus N --config ...), here with one rank and BENCH_FORCE_DIST=1 so that the RCCL process
    group, the barrier / all-reduce timing and (c5) the halo-exchange call are really taken."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(helpers.ROOT, "bench.py"), "--gpus", "1",
           "--steps", "2", "--warmup", "1", "--config", config, "--no-cpu-baseline"] + extra
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    if config == "c5":
        assert line["scaling"] == "strong" and line["parity"]["max_ulp_all_ranks"] <= 1
