// uchar fast paths: the gfx950 counterparts of the reference's Highway kernels
// (resample/reduceh_hwy.cpp:79, reducev_hwy.cpp:94, shrinkh_hwy.cpp:68,
// shrinkv_hwy.cpp:90,133).  Each *_try returns 1 when it handled the call,
// 0 when the geometry is outside what the fast kernel covers (the caller then
// uses the general kernel) and -1 on a launch error.  Only the vertical passes have one:
// after them the image is small and resample.hip's row-batched horizontal kernels are
// launch-latency bound.
#pragma once

#include "resample.h"

namespace vh {

int reduceh_u8x3_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile);
int reducev_u8_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const ReducePos *pos, const short *table, int tile);
int shrinkv_u8_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out);
// the same on n rects of one geometry, one launch per 64
int shrinkv_u8_batch_try(int vshrink, const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n);

const ReducePos *reduce_device_positions(_VipsHipReduce *r, int start, int count, int tile);
// device-resident coefficient table of a reduce (short or double), created on first use
int reduce_tables(_VipsHipReduce *r, bool want_float, const void **table);

// vips_resize's tail (reducev -> shrinkh -> reduceh) on a whole uchar image, resize_tail.hip
// (n images of one geometry, one launch per 64)
int resize_tail_u8_try(_VipsHipReduce *rv, int hshrink, int shrunk_width, _VipsHipReduce *rh,
	const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n, int tile);

// vips_resize's whole downsizing chain (shrinkv -> reducev -> shrinkh -> reduceh) on n uchar images
// of one size in one streaming kernel, resize_stream.hip; 1 = handled, 0 = not its case
int resize_stream_u8_try(_VipsHipReduce *rv, int vshrink, _VipsHipReduce *rh, int hshrink, int shrunk_height,
	int shrunk_width, const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n, int tile);
// the same for any residual reduce, from a host-made schedule of the vertical pass, resize_streamg.hip
int resize_streamg_u8_try(_VipsHipReduce *rv, int vshrink, _VipsHipReduce *rh, int hshrink, int shrunk_height,
	int shrunk_width, const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n, int tile);
// the same from images (ops_resample.cpp); 0 = done, 1 = not its case, -1 = error
int resize_batch_u8(VipsHipImage *const *in, int n, VipsHipImage **out, double scale, int kernel, double gap);
// ops_resample.cpp: vips_premultiply(uchar) + vips_resize of an RGBA uchar image, the premultiply on the first kernel's
// loads; 0 done, 1 not covered, -1 error
int resize_premul_u8(VipsHipImage *in, VipsHipImage **out, double hscale, double vscale);
// ... followed by vips_sharpen, in ONE kernel (resize_sharpen.hip); the blur mask as convi's integers,
// sharpen.c's LUT as 65536 host ints; 0 = done, 1 = not its case, -1 = error
int resize_sharpen_batch_u8(VipsHipImage *const *in, int n, VipsHipImage **out, double scale, int kernel, double gap,
	const int *coef, int ncoef, int mask_scale, const int *lut);

// resample16.hip: the ushort streaming kernels (whole images); 1 = handled, 0 = not their case
int reducev16_stream_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile);
int reduceh16_stream_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, const ReducePos *pos,
	const short *table);
int shrinkv16_stream_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out);
// ... and the uchar vertical reduce with a coefficient row per output row, on the same schedule
int reducev8_stream_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile);
// ... as a banded matrix product on the matrix cores (reduce_band.hip)
int reducev_band_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile);
int reduceh_band_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile);
// vips_shrinkv(vs, ceil) + vips_reducev as one kernel (`in`: the image before the shrink; `r`: the reduce's plan)
int shrinkv_reducev_band_try(_VipsHipReduce *r, int vs, int mid_height, const VipsHipRegion *in, const VipsHipRegion *out, int tile,
	bool premul = false);
int shrinkh16_stream_try(int hshrink, const VipsHipRegion *in, const VipsHipRegion *out);
int shrinkbox16_try(int hshrink, int vshrink, const VipsHipRegion *in, const VipsHipRegion *out);
// reduceh_u8.hip: vips_reduceh on uchar with one coefficient row and first taps 4 or 8 pixels apart,
// packed bytes (whole rows); 1 = handled, 0 = not its case
int reduceh_u8p_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile);
// shrinkh_u8.hip: vips_shrinkh on uchar, packed bytes (whole rows); 1 = handled, 0 = not its case
int shrinkh_u8_stream_try(int hshrink, const VipsHipRegion *in, const VipsHipRegion *out);

} // namespace vh
