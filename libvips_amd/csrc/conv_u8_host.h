// Host side of the packed-byte integer convolutions on uchar (conv_u8_body.h): the coefficient
// dwords, the rounding constants, geometry and launches.  Included by conv_u8.hip (which defines
// cu8_launch_*() as kernel launches) and by tests/emul/conv_u8_emul.cpp (host fiber runs).
#pragma once

#include "conv.h"
#include "conv_u8_body.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace vh {

// defined by the including file; 0 on success
static int cu8_launch_sep(int bands, int nd, bool regs, const Cu8Args &a, int grid, size_t lds);
static int cu8_launch_2d(int bands, int mh, const Cu8Args &a, int grid, size_t lds);

namespace {

// the mask row `c` (n taps, centre n / 2) shifted by k = 0..3 bytes and cut into nd dwords:
// dword j, byte bl of shift k holds tap 4 (j - pd) + bl - k + h (conv_u8_body.h)
void cu8_cvec(const int *c, int n, int pd, int nd, unsigned int *out)
{
	const int h = n / 2;
	for (int k = 0; k < 4; k++)
		for (int j = 0; j < nd; j++) {
			unsigned int w = 0;
			for (int bl = 0; bl < 4; bl++) {
				const int tap = 4 * (j - pd) + bl - k + h;
				const int v = tap >= 0 && tap < n ? c[tap] : 0;
				w |= (unsigned int) (unsigned char) (signed char) v << (8 * bl);
			}
			out[k * nd + j] = w;
		}
}

// what both kernels need of the images and the plan; false: not their case
bool cu8_common(const _VipsHipImage *in, const _VipsHipImage *out, const _VipsHipConv *c, double offset2, Cu8Args *a)
{
	if (getenv("VIPS_HIP_NO_CONV_U8"))
		return false;
	if (c->precision != VIPS_HIP_PRECISION_INTEGER || in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR)
		return false;
	if (in->bands != out->bands || in->width != out->width || in->height != out->height)
		return false;
	if (in->bands != 1 && in->bands != 3 && in->bands != 4)
		return false;
	if (((uintptr_t) in->data | (uintptr_t) out->data | in->stride | out->stride) & 3)
		return false;
	if ((long long) in->width * in->bands >= (1LL << 30) || in->width < 4 || in->height < 1)
		return false;
	// the rounding of conv_u8_body.h: offset 0, 1 <= scale <= 8000, |numerator| < 2^24
	if (c->offset_i != 0 || (int) rint(offset2) != 0 || c->scale_i < 1 || c->scale_i > 8000 || c->rounding != c->scale_i / 2)
		return false;
	long long abs_sum = 0, sum = 0;
	for (int k = 0; k < c->nnz; k++) {
		if (c->coeffi[k] < -127 || c->coeffi[k] > 127)
			return false;
		abs_sum += c->coeffi[k] < 0 ? -c->coeffi[k] : c->coeffi[k];
		sum += c->coeffi[k];
	}
	if (abs_sum * 255 + c->rounding >= (1LL << 24))
		return false;
	memset(a, 0, sizeof(*a));
	a->in = (const unsigned char *) in->data;
	a->out = (unsigned char *) out->data;
	a->in_stride = (long long) in->stride;
	a->out_stride = (long long) out->stride;
	a->width = in->width;
	a->height = in->height;
	a->acc0 = (int) (c->rounding + 128 * sum);
	a->rscale = 1.0f / (float) c->scale_i;
	a->bias = (float) (-0.5 + 1.0 / (2.0 * c->scale_i));
	return true;
}

// blocks of 4 wave strips across, segments of whole quads down, ~`per_cu` items per CU (dealt by an
// atomic counter)
void cu8_geometry(Cu8Args *a, int per_cu, int min_quads)
{
	a->wout = 4 * (64 - 2 * a->pd);
	const int block_cols = a->wout * (CU8_NT / 64);
	a->strips = (a->width + block_cols - 1) / block_cols;
	const int quads = (a->height + 3) >> 2;
	// one block = 4 waves; the chip holds 256 x per_cu of them: that many items, i.e. one residency
	// round (fewer, longer items leave the SIMDs short of waves: 576 items of 128 rows ran 2 x slower
	// than 1026 of 72), but a segment re-makes its halo rows: not shorter than min_quads
	int want_segs = (256 * per_cu + a->strips - 1) / a->strips;
	int seg_quads = (quads + want_segs - 1) / want_segs;
	if (seg_quads < min_quads)
		seg_quads = min_quads;
	if (getenv("VIPS_HIP_CONV_U8_SEG"))
		seg_quads = atoi(getenv("VIPS_HIP_CONV_U8_SEG"));
	if (seg_quads < 1)
		seg_quads = 1;
	if (seg_quads > quads)
		seg_quads = quads;
	a->seg_rows = 4 * seg_quads;
	a->segs = (quads + seg_quads - 1) / seg_quads;
}

int cu8_counter(Cu8Args *a)
{
	int *counter = (int *) vips_hip_malloc(sizeof(int));
	if (!counter)
		return -1;
	if (hipMemsetAsync(counter, 0, sizeof(int), stream()) != hipSuccess) {
		vips_hip_free(counter);
		return hip_failed(hipErrorUnknown, "hipMemsetAsync");
	}
	a->counter = counter;
	return 0;
}

} // namespace

// Both passes of vips_convsep / vips_gaussblur (precision integer) on a uchar image.
// 1 = not this kernel's case (nothing launched), 0 = done, -1 = error.
int conv_u8_sep_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c, double offset2)
{
	Cu8Args a;
	if (c->mask_height != 1 || c->nnz != c->mask_width || !(c->mask_width & 1) || c->mask_width > 33)
		return 1;
	if (!cu8_common(in, out, c, offset2, &a))
		return 1;
	const int n = c->mask_width;
	a.half = a.vhalf = n / 2;
	a.hq = (a.half + 3) / 4;
	a.pd = a.hq;
	const int nd = 2 * a.hq + 1;
	if (nd > CU8_MAXND || a.hq < 1)
		return 1;
	cu8_cvec(c->coeffi.data(), n, a.hq, nd, a.cvec);
	cu8_geometry(&a, 4, 8 * a.hq);
	// the lanes' private rings: in registers (3 quads always; the longer masks unless
	// $VIPS_HIP_CONV_U8_RING=lds), else in LDS
	const char *ring_env = getenv("VIPS_HIP_CONV_U8_RING");
	const bool regs = nd == 3 || !(ring_env && strcmp(ring_env, "lds") == 0);
	const size_t ring_bytes = regs ? 0 : (size_t) nd * in->bands * CU8_NT * 16;
	a.off_ring = 0;
	a.off_slot = (int) ring_bytes;
	const size_t lds = ring_bytes + 16;
	if (lds > 160 * 1024)
		return 1;
	if (cu8_counter(&a))
		return -1;
	const int items = a.strips * a.segs;
	int per_cu = lds > 20 * 1024 ? (int) ((160 * 1024) / lds) : 8;
	int grid = 256 * (per_cu < 1 ? 1 : per_cu);
	if (grid > items)
		grid = items;
	Gate gate("conv_u8_sep");
	const int r = cu8_launch_sep(in->bands, nd, regs, a, grid, lds);
	vips_hip_free(a.counter);
	return r;
}

// vips_conv (precision integer) with a mask of 3, 5 or 7 rows and at most 9 columns on a uchar image.
int conv_u8_2d_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c)
{
	Cu8Args a;
	const int mw = c->mask_width, mh = c->mask_height;
	if (!(mw & 1) || !(mh & 1) || mw > 9 || mh > CU8_MAXMH || mh < 3 || mw < 1)
		return 1;
	if (!cu8_common(in, out, c, 0.0, &a))
		return 1;
	a.half = mw / 2;
	a.vhalf = mh / 2;
	a.pd = 1; // one lane of halo either side: ND = 3 whatever the width
	a.hq = 1;
	// the dense mask, row by row (zero taps were squeezed out of the plan)
	std::vector<int> dense((size_t) mw * mh, 0);
	for (int k = 0; k < c->nnz; k++)
		dense[c->pos[k]] = c->coeffi[k];
	for (int i = 0; i < mh; i++)
		cu8_cvec(&dense[(size_t) i * mw], mw, 1, 3, a.cvec + i * 12);
	cu8_geometry(&a, 16, 6);
	a.off_ring = 0;
	a.off_slot = 0;
	const size_t lds = 16;
	if (cu8_counter(&a))
		return -1;
	const int items = a.strips * a.segs;
	int grid = 256 * 8;
	if (grid > items)
		grid = items;
	Gate gate("conv_u8_2d");
	const int r = cu8_launch_2d(in->bands, mh, a, grid, lds);
	vips_hip_free(a.counter);
	return r;
}

} // namespace vh
