// Internal: colour / cast / band plumbing shared with the image-level operations.
#pragma once

#include "internal.h"

#include <vector>

struct _VipsHipReduce;

namespace vh {

// vips_cast of bands [in_first, in_first + take) of `in` into bands
// [out_first, ...) of `out`: vips_cast + vips_extract_band + vips_bandjoin in one pass
// (conversion/cast.c:120-330, extract.c, bandjoin.c).
int band_cast(const VipsHipRegion *in, int in_first, const VipsHipRegion *out, int out_first, int take);

// A fused chain of colour steps (VipsHipColourStep values).
int colour_route(const int *steps, int n_steps, double alpha_scale, const VipsHipRegion *in,
	const VipsHipRegion *out);

// vips_sharpen on a whole 3-band uchar sRGB image in one kernel (colour.hip); 1 = not its case
// (win: the part of the LUT that is not constant, as shorts on the device -- the kernel with every table in LDS;
// nullptr: the kernel that reads the whole LUT through global memory)
struct SharpenLutWindow {
	int lo, n, below, above; // lut[i] = below for i < lo, above for i >= lo + n
	int zero_lo, zero_hi;    // lut[d + 32768] == 0 for zero_lo <= d <= zero_hi, the run around d = 0 (empty: lo > hi)
	const short *lut_win;    // device: n entries from index lo
};
int sharpen_fused_u8(const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n_images,
	const int *to_steps, int n_to, const int *from_steps, int n_from, const int *coef, int n, int scale,
	const int *lut, const SharpenLutWindow *win);

// host copies of the 8-bit sRGB tables (256 floats, 257 ints) and the cube-root table (100000
// floats), as the device tables are made (LabQ2sRGB.c:130-160, XYZ2Lab.c:92-106)
void colour_tables_host(std::vector<float> &v2Y_8, std::vector<int> &Y2v_8, std::vector<float> &cbrt);

// the LDS-sized exact form of the cube-root table (cbrt_exact.h) on the calling thread's device;
// nullptr when this host's cbrtf does not fit the scheme
struct CbrtExact;
const CbrtExact *cbrt_exact_tables();
struct CbrtQuad;
const CbrtQuad *cbrt_quad_tables(); // (cbrt_quad.h)

// vips_resize's downsizing chain then vips_sharpen on n 3-band uchar sRGB images of one geometry in
// ONE streaming kernel (resize_sharpen.hip); arguments as resize_stream_u8_try + the blur mask as
// convi's integers and sharpen.c's LUT (host).  1 = handled, 0 = not its case, -1 = error
int resize_sharpen_stream_u8_try(_VipsHipReduce *rv, int vshrink, _VipsHipReduce *rh, int hshrink, int shrunk_height,
	int shrunk_width, const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n, int tile, const int *coef,
	int ncoef, int scale, const int *lut);

int premultiply_region(const VipsHipRegion *in, const VipsHipRegion *out, double max_alpha, int uchar,
	int inverse);

} // namespace vh
