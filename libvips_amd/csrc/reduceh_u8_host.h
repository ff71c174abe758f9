// Host side of the packed-byte vips_reduceh on uchar (reduceh_u8_body.h): when it applies, the
// coefficient dwords, the launch geometry.  Included by reduceh_u8.hip (kernel launches) and by
// tests/emul/reduceh_u8_emul.cpp (host fiber runs).
#pragma once

#include "reduce_u8.h"
#include "resample.h"
#include "reduceh_u8_body.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace vh {

// defined by the including file; 0 on success
static int rh8_launch(int bands, int step4, int nd, const Rh8Args &a, int gx, int gy, size_t lds);

// 1 = handled, 0 = not this kernel's case (the caller takes reduceh_u8_lds), -1 = error
int reduceh_u8p_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	if (getenv("VIPS_HIP_NO_REDUCEH_U8P"))
		return 0;
	if (in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR || in->bands != out->bands ||
		in->bands < 1 || in->bands > 4 || out->width < 1)
		return 0;
	// whole rows (any range of them)
	if (in->left || out->left || in->width != in->im_width || out->width != out->im_width ||
		out->top < in->top || out->top + out->height > in->top + in->height)
		return 0;
	if (((uintptr_t) in->data | (uintptr_t) in->stride | (uintptr_t) out->data | (uintptr_t) out->stride) & 3)
		return 0;
	if ((long long) in->width * in->bands >= (1LL << 31))
		return 0;
	// one coefficient row, first taps a constant multiple of 4 pixels apart
	std::vector<ReducePos> pos;
	reduce_positions(r, 0, out->width, tile, pos);
	const int phase = pos[0].phase;
	const int step = out->width > 1 ? pos[1].first - pos[0].first : 4;
	if (step != 4 && step != 8)
		return 0;
	for (int x = 0; x < out->width; x++)
		if (pos[x].phase != phase || pos[x].first != pos[0].first + x * step)
			return 0;
	const int n = r->n_point;
	const int f0 = pos[0].first;
	const int shift = ((f0 % 4) + 4) % 4;
	const int need = (shift + n + 3) / 4;
	static const int nds[] = { 3, 5, 7, 9, 13 };
	int nd = 0;
	for (int cand : nds)
		if (cand >= need) {
			nd = cand;
			break;
		}
	if (!nd)
		return 0;
	Rh8Args a;
	memset(&a, 0, sizeof(a));
	const short *c = r->matrixs.data() + (size_t) phase * n;
	long long csum = 0;
	for (int i = 0; i < n; i++) {
		if (c[i] > 16383 || c[i] < -16384)
			return 0;
		csum += c[i];
		// tap i sits at byte shift + i of the window
		const int at = shift + i;
		const unsigned int hi = (unsigned int) ((c[i] >> 7) & 0xff), lo = (unsigned int) (c[i] & 127);
		a.chi[at >> 2] |= hi << (8 * (at & 3));
		a.clo[at >> 2] |= lo << (8 * (at & 3));
	}
	a.in = (const unsigned char *) in->data + (long long) (out->top - in->top) * (long long) in->stride;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_width = in->im_width;
	a.out_width = out->width;
	a.height = out->height;
	a.f_al = f0 - shift;
	a.step = step;
	const int step4 = step / 4;
	a.plane_dw = (255 * step4 + nd + 3) / 4 * 4 + 4;
	a.kconst = (int) (128 * csum + 2048);
	const size_t lds = (size_t) RH8_ROWS * in->bands * a.plane_dw * 4;
	const int gx = (out->width + 4 * RH8_QUADS - 1) / (4 * RH8_QUADS);
	const int trips = (out->height + RH8_ROWS - 1) / RH8_ROWS;
	int gy = (4096 + gx - 1) / gx;
	gy = gy < 1 ? 1 : gy > trips ? trips : gy;
	Gate gate("reduceh_u8_packed");
	const int rc = rh8_launch(in->bands, step4, nd, a, gx, gy, lds);
	return rc ? -1 : 1;
}

} // namespace vh
