"""Multi-GPU partitioning of the hot path.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPUs,
"gloo" in the CPU tests).  Two partitions, neither with a data-path collective:

* **batch** (BASELINE config 4): independent images are dealt round-robin to ranks,
  ``batch_indices``; ``bench.py --gpus N`` times this shape.
* **row strips of one image** (config 5, also 2/3): rank r owns a contiguous strip of output
  rows.  An op with a vertical footprint (conv: mask_height // 2 rows above, the rest below;
  reducev: ``vips_hip_reducev_need``) needs a few input rows owned by its neighbours: ONE
  nearest-neighbour exchange of halo rows (send/recv pairs, ~2 MB per side for C5) before
  the kernel, then every rank runs the ordinary region op on its window.  Image edges need
  no halo: the kernels clamp (vips_embed COPY semantics).

The arithmetic of who-needs-what is the C ABI's (``vips_hip_*_need``); this module only
moves rows.  Everything here works on CPU tensors too, which is how the gloo tests cover it.
"""
import ctypes

import numpy as np


def batch_indices(n_images, world, rank):
    """Images rank ``rank`` processes: round-robin, as libvips' thread pool deals tiles."""
    return list(range(rank, n_images, world))


def strip_bounds(total_rows, world, rank):
    """Contiguous, near-equal split of ``total_rows`` rows: [row0, row1)."""
    base, extra = divmod(total_rows, world)
    row0 = rank * base + min(rank, extra)
    return row0, row0 + base + (1 if rank < extra else 0)


def conv_need(mask_height, in_height):
    """need(out_top, out_height) -> (in_top, in_height) for a conv with this mask
    (convi.c:778-782: out rect grown by the mask, after the embed by mask/2), clipped."""
    above = mask_height // 2
    below = mask_height - 1 - above

    def need(out_top, out_height):
        lo = max(out_top - above, 0)
        hi = min(out_top + out_height + below, in_height)
        return lo, hi - lo

    return need


def reducev_need(reduce_handle):
    """need() for a VipsHipReduce built for the vertical axis (vips_hip_reducev_need)."""
    from ._ffi import lib

    def need(out_top, out_height):
        t0, tn = ctypes.c_int(), ctypes.c_int()
        lib.vips_hip_reducev_need(reduce_handle, out_top, out_height, ctypes.byref(t0), ctypes.byref(tn))
        return t0.value, tn.value

    return need


class StripPlan(object):
    """Who owns which input rows, and which rows each rank must fetch from which neighbour."""

    def __init__(self, in_height, out_height, world, need):
        self.world = world
        self.in_height = in_height
        self.out_height = out_height
        self.in_bounds = [strip_bounds(in_height, world, r) for r in range(world)]
        self.out_bounds = [strip_bounds(out_height, world, r) for r in range(world)]
        self.windows = []
        for r in range(world):
            o0, o1 = self.out_bounds[r]
            if o1 > o0:
                top, n = need(o0, o1 - o0)
            else:
                top, n = self.in_bounds[r][0], 0
            # a rank always keeps its own rows in the window
            i0, i1 = self.in_bounds[r]
            lo = min(top, i0) if n else i0
            hi = max(top + n, i1) if n else i1
            self.windows.append((lo, hi))

    def transfers(self):
        """[(src_rank, dst_rank, row0, row1)]: rows of src's strip that dst's window needs."""
        out = []
        for dst in range(self.world):
            w0, w1 = self.windows[dst]
            for src in range(self.world):
                if src == dst:
                    continue
                s0, s1 = self.in_bounds[src]
                lo, hi = max(w0, s0), min(w1, s1)
                if hi > lo:
                    out.append((src, dst, lo, hi))
        return out


class StripWindow(object):
    """A rank's PERSISTENT window: its own rows with room for the halo rows of its neighbours
    above and below, allocated once.  The rank keeps its strip IN the window (``own`` is a view of
    it: generate or load the strip there), and every exchange receives the neighbours' rows
    straight into the halo margins and sends straight from ``own`` -- no window is built per
    step and nothing is copied twice (the per-step window cost a copy of the
    whole strip, 1 GiB for BASELINE config 5 at 8 ranks)."""

    def __init__(self, plan, rank, row_shape, dtype, device):
        import torch

        self.plan, self.rank = plan, rank
        self.i0, self.i1 = plan.in_bounds[rank]
        self.top, self.bottom = plan.windows[rank]
        self.window = torch.empty((self.bottom - self.top,) + tuple(row_shape), dtype=dtype, device=device)
        self.own = self.window[self.i0 - self.top:self.i1 - self.top]
        self._transfers = [t for t in plan.transfers() if rank in (t[0], t[1])]

    def exchange(self, dist=None, group=None):
        """One batch of point-to-point sends / receives (RCCL ncclSend / ncclRecv pairs on GPUs):
        rows of ``own`` out of the window, halo rows into it.  Slices of whole rows are
        contiguous, so the transports read and write the window itself."""
        if dist is None:
            import torch.distributed as dist
        ops = []
        for src, dst, lo, hi in self._transfers:
            if src == self.rank:
                ops.append(dist.P2POp(dist.isend, self.window[lo - self.top:hi - self.top], dst, group))
            else:
                ops.append(dist.P2POp(dist.irecv, self.window[lo - self.top:hi - self.top], src, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return self.window, self.top


def exchange_halos(strip, plan, rank, dist=None, group=None):
    """Return this rank's window (its strip plus the halo rows the plan says it needs) for a
    strip held OUTSIDE a window: builds a StripWindow, copies the strip in and exchanges.  A
    program that steps repeatedly keeps a StripWindow instead (bench.py --config c5).

    ``strip``: tensor [rows, width, bands] holding rows plan.in_bounds[rank]."""
    i0, i1 = plan.in_bounds[rank]
    assert strip.shape[0] == i1 - i0
    sw = StripWindow(plan, rank, tuple(strip.shape[1:]), strip.dtype, strip.device)
    sw.own.copy_(strip)
    return sw.exchange(dist, group)


def _order_after_torch(tensor):
    """The library launches on its own (non-blocking) per-thread stream; ``exchange_halos``
    fills the window on torch's current stream (async copies, RCCL receives) and
    ``torch.empty`` may hand out memory the caching allocator last used there.  Unless the
    library was put on that very stream (``vips_hip_set_stream``), finish torch's queued work
    on the tensor's device before a library kernel touches it -- what
    ``Image.new_from_tensor`` does."""
    import torch

    from ._ffi import lib

    if not tensor.is_cuda:
        return
    cur = torch.cuda.current_stream(tensor.device)
    if (lib.vips_hip_get_stream() or 0) != cur.cuda_stream:
        cur.synchronize()


def conv_strip(window, window_top, plan, rank, mask, scale=1.0, offset=0.0, precision="float", out=None):
    """Run vips_hip_conv_gen on this rank's window: returns its strip of output rows
    (a torch CUDA tensor; ``out`` if one of the right shape and type is passed: a program that
    steps repeatedly keeps its output strip like it keeps its window).  ``window`` is what
    exchange_halos / StripWindow.exchange returned."""
    import torch

    from . import PRECISIONS
    from ._ffi import Region, check, check_handle, lib
    from .image import DTYPE_FORMATS, FORMAT_DTYPES

    m = np.ascontiguousarray(np.asarray(mask, dtype=np.float64))
    if m.ndim == 1:
        m = m[None, :]
    conv = check_handle(lib.vips_hip_conv_new(
        m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), m.shape[1], m.shape[0], float(scale),
        float(offset), PRECISIONS[precision]))
    try:
        rows, width, bands = window.shape
        in_fmt = DTYPE_FORMATS[np.dtype(str(window.dtype).replace("torch.", ""))]
        out_fmt = lib.vips_hip_conv_out_format(conv, in_fmt)
        o0, o1 = plan.out_bounds[rank]
        out_dtype = getattr(torch, np.dtype(FORMAT_DTYPES[out_fmt]).name)
        if out is None:
            out = torch.empty((o1 - o0, width, bands), device=window.device, dtype=out_dtype)
        assert tuple(out.shape) == (o1 - o0, width, bands) and out.dtype == out_dtype and out.is_contiguous()
        rin = Region(window.data_ptr(), 0, window_top, width, rows, width, plan.in_height, bands, in_fmt,
                     width * bands * window.element_size())
        rout = Region(out.data_ptr(), 0, o0, width, o1 - o0, width, plan.out_height, bands, out_fmt,
                      width * bands * out.element_size())
        _order_after_torch(window)
        check(lib.vips_hip_conv_gen(conv, ctypes.byref(rin), ctypes.byref(rout)))
        check(lib.vips_hip_synchronize())
        return out
    finally:
        lib.vips_hip_conv_free(conv)


def reduce_strip(window, window_top, plan, rank, in_width, hshrink, vshrink, kernel="lanczos3"):
    """Run the fused vips_hip_reduce_gen on this rank's window of a strip-partitioned image
    (BASELINE config 2 split across GPUs): returns its strip of output rows (a torch CUDA
    tensor).  ``plan`` was built with ``reducev_need`` of a reduce made for the WHOLE image
    (in_height -> out_height), so strips are bit-identical to the single-device result."""
    import math

    import torch

    from . import KERNELS
    from ._ffi import Region, check, check_handle, lib
    from .image import DTYPE_FORMATS

    rows, width, bands = window.shape
    assert width == in_width
    fmt = DTYPE_FORMATS[np.dtype(str(window.dtype).replace("torch.", ""))]
    out_width = int(in_width / hshrink + 0.5)  # VIPS_ROUND_UINT, reduceh.cpp:437
    rv = check_handle(lib.vips_hip_reduce_new(KERNELS[kernel], float(vshrink), plan.in_height, plan.out_height,
                                              math.nan))
    rh = check_handle(lib.vips_hip_reduce_new(KERNELS[kernel], float(hshrink), in_width, out_width, math.nan))
    try:
        o0, o1 = plan.out_bounds[rank]
        out = torch.empty((o1 - o0, out_width, bands), device=window.device, dtype=window.dtype)
        rin = Region(window.data_ptr(), 0, window_top, width, rows, width, plan.in_height, bands, fmt,
                     width * bands * window.element_size())
        rout = Region(out.data_ptr(), 0, o0, out_width, o1 - o0, out_width, plan.out_height, bands, fmt,
                      out_width * bands * out.element_size())
        _order_after_torch(window)
        r = lib.vips_hip_reduce_gen(rv, rh, ctypes.byref(rin), ctypes.byref(rout))
        if r == 1:  # not an even-integer RGBA uchar case: the two general passes
            mid = torch.empty((o1 - o0, width, bands), device=window.device, dtype=window.dtype)
            _order_after_torch(mid)
            rmid = Region(mid.data_ptr(), 0, o0, width, o1 - o0, width, plan.out_height, bands, fmt,
                          width * bands * mid.element_size())
            check(lib.vips_hip_reducev_gen(rv, ctypes.byref(rin), ctypes.byref(rmid)))
            check(lib.vips_hip_reduceh_gen(rh, ctypes.byref(rmid), ctypes.byref(rout)))
        else:
            check(r)
        check(lib.vips_hip_synchronize())
        return out
    finally:
        lib.vips_hip_reduce_free(rv)
        lib.vips_hip_reduce_free(rh)
