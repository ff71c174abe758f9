// Internal: colour / cast / band plumbing shared with the image-level operations.
#pragma once

#include "internal.h"

#include <vector>

namespace vh {

// vips_cast of bands [in_first, in_first + take) of `in` into bands
// [out_first, ...) of `out`: vips_cast + vips_extract_band + vips_bandjoin in one pass
// (conversion/cast.c:120-330, extract.c, bandjoin.c).
int band_cast(const VipsHipRegion *in, int in_first, const VipsHipRegion *out, int out_first, int take);

// A fused chain of colour steps (VipsHipColourStep values).
int colour_route(const int *steps, int n_steps, double alpha_scale, const VipsHipRegion *in,
	const VipsHipRegion *out);

// vips_sharpen on a whole 3-band uchar sRGB image in one kernel (colour.hip); 1 = not its case
int sharpen_fused_u8(const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n_images,
	const int *to_steps, int n_to, const int *from_steps, int n_from, const int *coef, int n, int scale,
	const int *lut);

int premultiply_region(const VipsHipRegion *in, const VipsHipRegion *out, double max_alpha, int uchar,
	int inverse);

} // namespace vh
