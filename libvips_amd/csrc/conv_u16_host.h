// Host side of the streaming integer convolution on ushort (conv_u16_body.h): the coefficient pairs,
// the division constants, geometry and launch.  Included by conv_u16.hip (which defines
// cu16_launch() as a kernel launch) and by tests/emul/conv_u16_emul.cpp (host fiber runs).
#pragma once

#include "conv.h"
#include "conv_u16_body.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace vh {

// defined by the including file; 0 on success, 1 when there is no build for the case
static int cu16_launch(int bands, int mh, int h, const Cu16Args &a, int grid, size_t lds);

// vips_conv (precision integer) with a mask of 1, 3 or 5 rows and at most 5 columns on a ushort image.
// 1 = not this kernel's case (nothing launched), 0 = done, -1 = error.
int conv_u16_2d_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c)
{
	if (getenv("VIPS_HIP_NO_CONV_U16"))
		return 1;
	if (c->precision != VIPS_HIP_PRECISION_INTEGER || in->format != VIPS_HIP_FORMAT_USHORT || out->format != VIPS_HIP_FORMAT_USHORT)
		return 1;
	if (in->bands != out->bands || in->width != out->width || in->height != out->height)
		return 1;
	if (in->bands != 1 && in->bands != 3 && in->bands != 4)
		return 1;
	if (((uintptr_t) in->data | (uintptr_t) out->data | in->stride | out->stride) & 3)
		return 1;
	if ((long long) in->width * in->bands >= (1LL << 29) || in->width < 2 || in->height < 1)
		return 1;
	const int mw = c->mask_width, mh = c->mask_height;
	if (!(mw & 1) || !(mh & 1) || mw > 5 || mh > CU16_MAXMH || mw < 1 || mh < 1 || (mw == 1 && mh == 1))
		return 1;
	// the rounding of conv_u16_body.h: offset 0, a positive scale, the C path's rounding term, sums in 32 bits
	if (c->offset_i != 0 || c->scale_i < 1 || c->rounding != c->scale_i / 2)
		return 1;
	long long abs_sum = 0, sum = 0;
	for (int k = 0; k < c->nnz; k++) {
		if (c->coeffi[k] > 32767 || c->coeffi[k] < -32768)
			return 1;
		abs_sum += c->coeffi[k] < 0 ? -c->coeffi[k] : c->coeffi[k];
		sum += c->coeffi[k];
	}
	if (abs_sum * 65535 + c->rounding >= (1LL << 31))
		return 1;
	Cu16Args a;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.height = in->height;
	a.half = mw / 2;
	a.vhalf = mh / 2;
	a.acc0 = (int) (c->rounding + 32768 * sum);
	const unsigned int scale = (unsigned int) c->scale_i;
	if (scale >= 2) {
		int l = 0;
		while ((1ULL << l) < scale)
			l++;
		const unsigned long long m = ((1ULL << (31 + l)) + scale - 1) / scale;
		if (m >= (1ULL << 32))
			return 1;
		a.mult = (unsigned int) m;
		a.shift = l - 1;
		// the identity, where it could break: around every multiple of the scale up to 2^31 (sampled when
		// there are many); once per scale and process
		static std::mutex mutex;
		static std::map<unsigned int, bool> checked;
		std::lock_guard<std::mutex> lock(mutex);
		auto it = checked.find(scale);
		if (it == checked.end()) {
			bool ok = true;
			const unsigned long long top = (1ULL << 31) - 1;
			const unsigned long long step = top / scale > 200000 ? (top / scale) / 200000 : 1;
			for (unsigned long long k = 1; k * scale <= top && ok; k += step)
				for (unsigned long long x = k * scale - 1; x <= k * scale && x <= top; x++)
					if ((unsigned int) (((unsigned long long) (unsigned int) x * m) >> 32) >> a.shift != (unsigned int) (x / scale))
						ok = false;
			if ((unsigned int) ((top * m) >> 32) >> a.shift != (unsigned int) (top / scale))
				ok = false;
			it = checked.emplace(scale, ok).first;
		}
		if (!it->second)
			return 1;
	}
	// the dense mask, row by row (zero taps were squeezed out of the plan); output c, window dword j holds the
	// taps 2 j - 2 - c + half (low lane) and + 1 (high lane)
	std::vector<int> dense((size_t) mw * mh, 0);
	for (int k = 0; k < c->nnz; k++)
		dense[c->pos[k]] = c->coeffi[k];
	for (int i = 0; i < mh; i++)
		for (int cc = 0; cc < 2; cc++)
			for (int j = 0; j < 3; j++) {
				unsigned int w = 0;
				for (int h = 0; h < 2; h++) {
					const int tap = 2 * j + h - 2 - cc + a.half;
					const int v = tap >= 0 && tap < mw ? dense[(size_t) i * mw + tap] : 0;
					w |= (unsigned int) (unsigned short) (short) v << (16 * h);
				}
				a.cvec[(i * 2 + cc) * 3 + j] = w;
			}
	// blocks of 4 wave strips across, segments down: about one residency round of 8 blocks per CU
	a.wout = 2 * 62;
	const int block_cols = a.wout * (CU16_NT / 64);
	a.strips = (a.width + block_cols - 1) / block_cols;
	int want_segs = (256 * 16 + a.strips - 1) / a.strips;
	int seg_rows = (a.height + want_segs - 1) / want_segs;
	if (seg_rows < 6 * mh)
		seg_rows = 6 * mh;
	if (getenv("VIPS_HIP_CONV_U16_SEG"))
		seg_rows = atoi(getenv("VIPS_HIP_CONV_U16_SEG"));
	if (seg_rows < 1)
		seg_rows = 1;
	if (seg_rows > a.height)
		seg_rows = a.height;
	a.seg_rows = seg_rows;
	a.segs = (a.height + seg_rows - 1) / seg_rows;
	a.off_slot = 0;
	int *counter = (int *) vips_hip_malloc(sizeof(int));
	if (!counter)
		return -1;
	if (hipMemsetAsync(counter, 0, sizeof(int), stream()) != hipSuccess) {
		vips_hip_free(counter);
		return hip_failed(hipErrorUnknown, "hipMemsetAsync");
	}
	a.counter = counter;
	const int items = a.strips * a.segs;
	int grid = 256 * 8;
	if (grid > items)
		grid = items;
	Gate gate("conv_u16_2d");
	const int h = a.half < 1 ? 1 : a.half; // (a 1-wide mask: as 3 wide with zero taps)
	const int r = cu16_launch(in->bands, mh, h, a, grid, 16);
	vips_hip_free(counter);
	return r;
}

} // namespace vh
