// Host side of conv_u8_mfma_body.h: the Toeplitz operands, rounding constants, geometry and the launch.
// Included by conv_u8_mfma.hip (which defines cm_launch() as a kernel launch) and by
// tests/emul/conv_u8_mfma_emul.cpp (host fiber runs).
#pragma once

#include "conv.h"
#include "conv_u8_mfma_body.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace vh {

// defined by the including file; 0 on success
static int cm_launch(int bands, const CmArgs &a, int grid, size_t lds);

namespace {

// a small integer as IEEE half bits (|v| <= 2048: exact)
unsigned int cm_half_bits(int v)
{
	if (v == 0)
		return 0;
	const unsigned int sign = v < 0 ? 0x8000u : 0u;
	unsigned int m = (unsigned int) (v < 0 ? -v : v);
	int e = 0;
	while ((m >> (e + 1)) != 0)
		e++; // m in [2^e, 2^(e+1))
	const unsigned int frac = (m << (10 - e)) & 0x3ffu; // e <= 11; for e = 11 only m = 2048 (frac 0) is exact
	return sign | ((unsigned int) (e + 15) << 10) | (e <= 10 ? frac : 0u);
}

// the Toeplitz operand of a mask, as conv_u8_mfma_body.h lays it out: [4][64][4] dwords.  Lane (m = l & 31, hf),
// k-step s, slot idx <-> position w = 16 s + 8 (idx >> 2) + 4 hf + (idx & 3) of the 64-wide window (columns
// in pass 1, mid rows in pass 2), tap w - m - (hp - half) of output m
void cm_tables(const int *c, int n, int hp, unsigned int *tz)
{
	const int half = n / 2;
	for (int s = 0; s < 4; s++)
		for (int l = 0; l < 64; l++) {
			const int m = l & 31, hf = l >> 5;
			for (int q = 0; q < 4; q++) {
				unsigned int w1 = 0;
				for (int e = 0; e < 2; e++) {
					const int idx = 2 * q + e;
					const int w = 16 * s + 8 * (idx >> 2) + 4 * hf + (idx & 3);
					const int k = w - m - (hp - half);
					w1 |= cm_half_bits(k >= 0 && k < n ? c[k] : 0) << (16 * e);
				}
				tz[(s * 64 + l) * 4 + q] = w1;
			}
		}
}

// the operands live on the device for the life of the process, one pair per (device, mask)
struct CmTables {
	unsigned int *tz;
};
std::mutex cm_mutex;
std::map<std::vector<int>, CmTables> cm_cache;

int cm_tables_device(const int *c, int n, int hp, CmTables *out)
{
	std::vector<int> key(c, c + n);
	key.push_back(hp);
	key.push_back(current_device());
	std::lock_guard<std::mutex> lock(cm_mutex);
	auto it = cm_cache.find(key);
	if (it == cm_cache.end()) {
		std::vector<unsigned int> host(4 * 64 * 4);
		cm_tables(c, n, hp, host.data());
		unsigned int *d = (unsigned int *) upload(host.data(), host.size() * sizeof(unsigned int));
		if (!d)
			return -1;
		CmTables t = { d };
		it = cm_cache.emplace(key, t).first;
	}
	*out = it->second;
	return 0;
}

} // namespace

// Both passes of vips_convsep / vips_gaussblur (precision integer) on a uchar image, on the matrix cores.
// 1 = not this kernel's case (nothing launched), 0 = done, -1 = error.
int conv_u8_mfma_sep_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c, double offset2)
{
	const char *env = getenv("VIPS_HIP_CONV_U8_MFMA");
	if ((env && atoi(env) == 0) || getenv("VIPS_HIP_NO_CONV_U8"))
		return 1;
	if (c->mask_height != 1 || c->nnz != c->mask_width || !(c->mask_width & 1) || c->mask_width > 33 || c->mask_width < 3)
		return 1;
	if (c->precision != VIPS_HIP_PRECISION_INTEGER || in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR)
		return 1;
	if (in->bands != out->bands || in->width != out->width || in->height != out->height)
		return 1;
	if (in->bands < 1 || in->bands > 4)
		return 1;
	if (((uintptr_t) in->data | (uintptr_t) out->data | in->stride | out->stride) & 3)
		return 1;
	// (the staging's lane offsets are 32-bit: 33 rows of the image; tiny images stay with the packed-byte kernel)
	if ((long long) in->stride * 34 >= (1LL << 31) || in->width < 32 || in->height < 8)
		return 1;
	// the rounding of conv_u8_body.h: offset 0, 1 <= scale <= 8000, numerators below 2^24
	if (c->offset_i != 0 || (int) rint(offset2) != 0 || c->scale_i < 1 || c->scale_i > 8000 || c->rounding != c->scale_i / 2)
		return 1;
	long long abs_sum = 0;
	for (int k = 0; k < c->nnz; k++) {
		if (c->coeffi[k] <= -2048 || c->coeffi[k] >= 2048) // an exact half
			return 1;
		abs_sum += c->coeffi[k] < 0 ? -c->coeffi[k] : c->coeffi[k];
	}
	if (abs_sum * 255 + c->rounding >= (1LL << 24))
		return 1;
	const int B = in->bands, n = c->mask_width;
	CmArgs a;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.height = in->height;
	a.half = n / 2;
	a.hp = (a.half + 3) & ~3;
	a.strips = (a.width + CM_BW - 1) / CM_BW;
	a.in_dw = (CM_BW + 2 * a.hp) * B / 4;
	a.in_pitch = a.in_dw + (a.in_dw & 1);
	if (!((a.in_pitch / 2) & 1))
		a.in_pitch += 2;
	a.in_buf = (CM_ROWS * a.in_pitch + 255) & ~255;
	a.out_pitch = 32 * B + 2;
	a.acc0 = ldexpf((float) c->rounding, -24);
	a.k1 = ldexpf(1.0f / (float) c->scale_i, 24);
	a.bias = (float) (-0.5 + 1.0 / (2.0 * c->scale_i));
	const size_t lds = (size_t) (2 * a.in_buf + CM_ROWS * a.out_pitch) * sizeof(unsigned int);
	// segments: one residency round of blocks (LDS allows 3 per CU), a segment re-makes one chunk of 32 rows
	{
		const int chunks = (a.height + CM_ROWS - 1) / CM_ROWS;
		int per_cu = (int) ((160 * 1024) / lds);
		per_cu = per_cu < 1 ? 1 : per_cu > 3 ? 3 : per_cu;
		const char *e = getenv("VIPS_HIP_CONV_MFMA_PER_CU");
		if (e && atoi(e) > 0)
			per_cu = atoi(e);
		int segs = (256 * per_cu) / a.strips;
		segs = segs < 1 ? 1 : segs > chunks ? chunks : segs;
		int seg_chunks = (chunks + segs - 1) / segs;
		e = getenv("VIPS_HIP_CONV_MFMA_SEG");
		if (e && atoi(e) > 0)
			seg_chunks = atoi(e);
		if (seg_chunks < 2 && chunks >= 2)
			seg_chunks = 2;
		a.seg_rows = CM_ROWS * seg_chunks;
		a.segs = (a.height + a.seg_rows - 1) / a.seg_rows;
	}
	CmTables tabs;
	if (cm_tables_device(c->coeffi.data(), n, a.hp, &tabs))
		return -1;
	a.tz = tabs.tz;
	Gate gate("conv_u8_mfma_sep");
	return cm_launch(B, a, a.strips * a.segs, lds);
}

} // namespace vh
