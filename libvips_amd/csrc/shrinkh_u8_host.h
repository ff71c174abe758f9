// Host side of the packed-byte vips_shrinkh on uchar (shrinkh_u8_body.h): when it applies, the
// launch geometry.  Included by shrinkh_u8.hip (kernel launches) and by
// tests/emul/shrinkh_u8_emul.cpp (host fiber runs).
#pragma once

#include "reduce_u8.h"
#include "resample.h"
#include "shrinkh_u8_body.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace vh {

// defined by the including file; 0 on success
static int sh8_launch(int bands, int hs_template, const Sh8Args &a, int gx, int gy);

// 1 = handled, 0 = not this kernel's case (the caller takes shrinkh_general), -1 = error
int shrinkh_u8_stream_try(int hshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	if (getenv("VIPS_HIP_NO_SHRINKH_U8"))
		return 0;
	if (in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR || in->bands != out->bands ||
		in->bands < 1 || in->bands > 4)
		return 0;
	// a compiled-in factor, or boxes of whole 4-pixel groups
	if (hshrink < 2 || hshrink > 4096 || (hshrink > 8 && hshrink % 4))
		return 0;
	// whole rows (any range of them: the rows of the two regions correspond one to one)
	if (in->left || out->left || in->width != in->im_width || out->width != out->im_width ||
		out->top < in->top || out->top + out->height > in->top + in->height)
		return 0;
	// rows read and written as dwords
	if (((uintptr_t) in->data | (uintptr_t) in->stride | (uintptr_t) out->data | (uintptr_t) out->stride) & 3)
		return 0;
	if ((long long) in->width * in->bands >= (1LL << 31))
		return 0;
	Sh8Args a;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data + (long long) (out->top - in->top) * (long long) in->stride;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_width = in->im_width;
	a.out_width = out->width;
	a.height = out->height;
	a.hshrink = hshrink;
	a.mult = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) hshrink));
	a.quads = (out->width + 3) / 4;
	const int gx = (a.quads + SH8_NT - 1) / SH8_NT;
	// enough blocks for every CU to hold its eight waves per SIMD several times over
	const int pairs = (out->height + SH8_ROWS - 1) / SH8_ROWS;
	int gy = (8192 + gx - 1) / gx;
	gy = gy < 1 ? 1 : gy > pairs ? pairs : gy;
	Gate gate("shrinkh_u8_stream");
	const int rc = sh8_launch(in->bands, hshrink <= 8 ? hshrink : 0, a, gx, gy);
	return rc ? -1 : 1;
}

} // namespace vh
