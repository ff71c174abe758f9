// Host side of conv_u8_mfma_body.h: the Toeplitz operands, rounding constants, geometry and the launch.
// Included by conv_u8_mfma.hip (which defines cm_launch() as a kernel launch) and by
// tests/emul/conv_u8_mfma_emul.cpp (host fiber runs).
#pragma once

#include "conv.h"
#include "conv_u8_mfma_body.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <vector>

namespace vh {

// defined by the including file; 0 on success
static int cm_launch(int bands, bool wide, bool twod, const CmArgs &a, int grid, size_t lds);
// ... the ushort form (separable masks): bands = samples per pixel
static int cm_launch16(int bands, bool wide, const CmArgs &a, int grid, size_t lds);

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
// in pass 1, mid rows in pass 2), tap w - m - (hp - half) of output m.  [lo, hi): the window positions inside
// the image -- a tap on a position outside is added to position lo / hi - 1 (the embed copies the edge column
// there).  false: a folded coefficient is not an exact half any more (|c| >= 2048)
bool cm_tables(const int *c, int n, int hp, int lo, int hi, unsigned int *tz)
{
	const int half = n / 2;
	std::vector<int> col(64 * 32, 0); // [w][m]
	for (int m = 0; m < 32; m++)
		for (int k = 0; k < n; k++) {
			int w = m + (hp - half) + k;
			w = w < lo ? lo : w >= hi ? hi - 1 : w;
			if (w >= 0 && w < 64)
				col[w * 32 + m] += c[k];
		}
	for (int v : col)
		if (v <= -2048 || v >= 2048)
			return false;
	for (int s = 0; s < 4; s++)
		for (int l = 0; l < 64; l++) {
			const int m = l & 31, hf = l >> 5;
			for (int q = 0; q < 4; q++) {
				unsigned int w1 = 0;
				for (int e = 0; e < 2; e++) {
					const int idx = 2 * q + e;
					const int w = 16 * s + 8 * (idx >> 2) + 4 * hf + (idx & 3);
					w1 |= cm_half_bits(col[w * 32 + m]) << (16 * e);
				}
				tz[(s * 64 + l) * 4 + q] = w1;
			}
		}
	return true;
}

// clip(floor((S + h) / scale), 0, 255) as cvt_pk_u8(fma(S 2^-24, k1, bias)): the constants, and every S whose
// quotient does not saturate walked once per scale (a negative S + h has a value below 0.5 and a quotient <= 0)
bool cm_rounding(int scale, int h, float *k1, float *bias)
{
	static std::mutex mutex;
	static std::map<std::pair<int, int>, std::pair<float, float>> good;
	std::lock_guard<std::mutex> lock(mutex);
	auto it = good.find(std::make_pair(scale, h));
	if (it == good.end()) {
		const float rs = 1.0f / (float) scale;
		const float b = (float) (-0.5 + 1.0 / (2.0 * scale) + (double) h * (double) rs);
		bool ok = true;
		for (long long S = -h; S + h < 257LL * scale && ok; S++) {
			const float v = fmaf((float) S, rs, b); // (the device multiplies S 2^-24 by 2^24 rs: the same product)
			const float r = rintf(v);
			const long long got = r <= 0.0f ? 0 : r >= 255.0f ? 255 : (long long) r;
			long long want = (S + h) / scale;
			want = want > 255 ? 255 : want;
			ok = got == want;
		}
		// below -h the value only falls (rs > 0): one probe at the most negative sum a mask can make is enough
		ok = ok && fmaf(-16777216.0f, rs, b) < 0.5f;
		if (!ok)
			return false;
		it = good.emplace(std::make_pair(scale, h), std::make_pair(ldexpf(rs, 24), b)).first;
	}
	*k1 = it->second.first;
	*bias = it->second.second;
	return true;
}

// the operands live on the device for the life of the process, one set per (device, mask, edge geometry):
// table 0 the mask's own, tables 1 .. 3 pass 1's for the tiles whose windows hang over an edge of the image
struct CmTables {
	unsigned int *tz;
	int edge_wave[3];
};
std::mutex cm_mutex;
std::map<std::vector<int>, CmTables> cm_cache;
// least recently used first.  The key holds the edge geometry, which changes with nearly every image width: a
// process that blurs images of arbitrary widths would otherwise grow this (16 KB x rows of device memory an
// entry) without bound.  An evicted table may still be read by a kernel some thread has queued, so
// eviction -- rare: CM_CACHE_MAX distinct (mask, width) pairs later -- waits for the device before it frees.
std::list<std::vector<int>> cm_order;
constexpr size_t CM_CACHE_MAX = 128;

// c: rows x n taps (rows = 1: the separable mask); the set for variant v, mask row i starts at table (v rows + i)
int cm_tables_device(const int *c, int n, int rows, int hp, int width, CmTables *out)
{
	// the tiles (32 columns each) whose 64-column window, which starts hp columns before the tile, leaves the image
	int waves[3] = { -1, -1, -1 }, lo[3] = { 0, 0, 0 }, hi[3] = { 64, 64, 64 };
	int count = 0;
	const int last = (width - 1) / 32;
	for (int W = 0; W <= last && count < 3; W++) {
		const int first_col = 32 * W - hp; // image column of window position 0
		const int l = first_col < 0 ? -first_col : 0, h = width - first_col < 64 ? width - first_col : 64;
		if (l > 0 || h < 64) {
			waves[count] = W;
			lo[count] = l;
			hi[count] = h;
			count++;
		}
		if (W == 0 && last > 2)
			W = last - 2; // (only tile 0 and the last two can: hp <= 16)
	}
	std::vector<int> key(c, c + (size_t) n * rows);
	key.push_back(rows);
	key.push_back(hp);
	key.push_back(current_device());
	for (int k = 0; k < 3; k++) {
		key.push_back(waves[k]);
		key.push_back(lo[k]);
		key.push_back(hi[k]);
	}
	std::lock_guard<std::mutex> lock(cm_mutex);
	auto it = cm_cache.find(key);
	if (it == cm_cache.end()) {
		const size_t one = 4 * 64 * 4;
		std::vector<unsigned int> host(4 * rows * one);
		for (int i = 0; i < rows; i++) {
			if (!cm_tables(c + (size_t) i * n, n, hp, -1000, 1000, host.data() + i * one))
				return 1;
			for (int k = 0; k < 3; k++)
				if (!cm_tables(c + (size_t) i * n, n, hp, waves[k] < 0 ? -1000 : lo[k], waves[k] < 0 ? 1000 : hi[k],
						host.data() + ((k + 1) * rows + i) * one))
					return 1;
		}
		unsigned int *d = (unsigned int *) upload(host.data(), host.size() * sizeof(unsigned int));
		if (!d)
			return -1;
		CmTables t;
		t.tz = d;
		for (int k = 0; k < 3; k++)
			t.edge_wave[k] = waves[k];
		it = cm_cache.emplace(key, t).first;
		cm_order.push_back(key);
		while (cm_cache.size() > CM_CACHE_MAX && !cm_order.empty()) {
			auto old = cm_cache.find(cm_order.front());
			cm_order.pop_front();
			if (old == cm_cache.end() || old == it)
				continue;
			(void) hipDeviceSynchronize();
			vips_hip_free(old->second.tz);
			cm_cache.erase(old);
		}
	}
	else {
		// (touch: to the young end)
		for (auto o = cm_order.begin(); o != cm_order.end(); ++o)
			if (*o == key) {
				cm_order.splice(cm_order.end(), cm_order, o);
				break;
			}
	}
	*out = it->second;
	return 0;
}

// what both kernels need of the images: the staging geometry, the rounding, the segments; false: not their case
bool cm_geometry(const _VipsHipImage *in, const _VipsHipImage *out, const _VipsHipConv *c, int half, int mh, CmArgs *pa,
	size_t *plds, bool *pwide)
{
	const int B = in->bands * (in->format == VIPS_HIP_FORMAT_USHORT ? 2 : 1); // bytes a pixel: the kernel's byte planes
	CmArgs &a = *pa;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.height = in->height;
	a.half = half;
	a.hp = (a.half + 3) & ~3;
	if (a.hp < 4)
		a.hp = 4;
	a.mh = mh;
	a.stage_rows = CM_ROWS + (mh ? mh - 1 : 0);
	a.row_lead = mh ? mh / 2 : a.hp;
	// the 16-column steps of the 64-column window that hold a tap of some output: columns up to 31 + hp + half
	a.ksteps = (31 + a.hp + a.half) / 16 + 1;
	a.strips = (a.width + CM_BW - 1) / CM_BW;
	// 16-byte staging units when every row start is a multiple of 16, else dwords
	const bool wide = !(((uintptr_t) in->data | in->stride) & 15) && !getenv("VIPS_HIP_CONV_MFMA_NARROW");
	const int U = wide ? 16 : 4;
	const int lead = a.hp * B; // bytes in front of column X0 (X0 B is a multiple of 128 B: of 16)
	a.e_dw = (((lead + U - 1) / U) * U - lead) / 4;
	const int units = (4 * a.e_dw + (CM_BW + 2 * a.hp) * B + U - 1) / U;
	a.in_dw = units * (U / 4);
	{
		// the pitch: a whole number of units; an odd number of 16-byte units (of dword pairs for dword units)
		int pu = units;
		if (wide)
			pu += !(pu & 1);
		else {
			pu += pu & 1;
			if (!((pu / 2) & 1))
				pu += 2;
		}
		a.in_pitch = pu * (U / 4);
	}
	// (whole instructions of 64 units: the last one of a chunk may run past the staged rows)
	a.in_buf = ((a.stage_rows * (a.in_pitch / (U / 4)) + 63) / 64) * 64 * (U / 4);
	a.fin64 = 0;
	a.div_inv = a.rnd_half = 0.0;
	if (in->format == VIPS_HIP_FORMAT_USHORT) {
		// (S + rnd) / scale for 0 < S + rnd < 2^31 as a multiplication: m = ceil(2^(31 + l) / scale), l = ceil(log2
		// scale); the error m scale - 2^(31 + l) is below scale <= 2^l, so (S + rnd) m / 2^(31 + l) and (S + rnd) /
		// scale have the same floor (Granlund & Montgomery 1994, theorem 4.2 with N = 31)
		a.rnd = c->rounding;
		a.fin64 = getenv("VIPS_HIP_U16_FIN64") ? atoi(getenv("VIPS_HIP_U16_FIN64")) : 1; // (0.369 -> 0.334-0.344 ms, sigma 8 on 8192 x 8192 x 3: profiles/r06h_fin64.txt)
		a.div_inv = 1.0 / (double) c->scale_i;
		a.rnd_half = (double) c->rounding + 0.5;
		if (c->scale_i == 1) {
			a.div_m = 0;
			a.div_s = 0;
		}
		else {
			int l = 0;
			while ((1LL << l) < c->scale_i)
				l++;
			const unsigned long long m = ((1ULL << (31 + l)) + (unsigned long long) c->scale_i - 1) / (unsigned long long) c->scale_i;
			if (m >= (1ULL << 32) || l < 1)
				return false;
			a.div_m = (unsigned int) m;
			a.div_s = l - 1;
		}
	}
	else if (!cm_rounding(c->scale_i, c->rounding, &a.k1, &a.bias))
		return false;
	// (the waves' output tiles: 32 rows, 16 on ushort images)
	const int tile_rows = in->format == VIPS_HIP_FORMAT_USHORT ? 16 : CM_ROWS;
	const size_t lds = (size_t) (2 * a.in_buf + (CM_NT / 64) * tile_rows * (8 * B + 1)) * sizeof(unsigned int);
	if (lds > 160 * 1024)
		return false;
	// segments: one residency round of blocks (LDS allows 3 per CU); a separable segment re-makes one chunk of 32 rows
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
		if (!mh && seg_chunks < 2 && chunks >= 2)
			seg_chunks = 2;
		a.seg_rows = CM_ROWS * seg_chunks;
		a.segs = (a.height + a.seg_rows - 1) / a.seg_rows;
	}
	*plds = lds;
	*pwide = wide;
	return true;
}

// what both kernels ask of the images and the plan's rounding
bool cm_common(const _VipsHipImage *in, const _VipsHipImage *out, const _VipsHipConv *c, double offset2, bool u16 = false)
{
	const char *env = getenv("VIPS_HIP_CONV_U8_MFMA");
	if ((env && atoi(env) == 0) || getenv("VIPS_HIP_NO_CONV_U8"))
		return false;
	const int format = u16 ? VIPS_HIP_FORMAT_USHORT : VIPS_HIP_FORMAT_UCHAR;
	if (c->precision != VIPS_HIP_PRECISION_INTEGER || in->format != format || out->format != format)
		return false;
	if (in->bands != out->bands || in->width != out->width || in->height != out->height)
		return false;
	if (in->bands < 1 || in->bands > 4)
		return false;
	if (((uintptr_t) in->data | (uintptr_t) out->data | in->stride | out->stride) & 3)
		return false;
	// (the staging's lane offsets are 32-bit: 40 rows of the image; tiny images stay with the packed-byte kernels)
	if ((long long) in->stride * 42 >= (1LL << 31) || in->width < 32 || in->height < 8)
		return false;
	// the rounding of conv_u8_body.h: offset 0, 1 <= scale <= 8000, numerators below 2^24
	// (the rounding term: convi's scale / 2, or conva's (divisor + 1) / 2 -- approx.hip's fast path --: whatever it is,
	// cm_rounding walks every sum for it before the kernel is enabled)
	if (c->offset_i != 0 || (int) rint(offset2) != 0 || c->scale_i < 1 || c->scale_i > 8000 || c->rounding < 0 ||
		c->rounding > c->scale_i)
		return false;
	long long abs_sum = 0;
	for (int k = 0; k < c->nnz; k++) {
		if (c->coeffi[k] <= -2048 || c->coeffi[k] >= 2048) // an exact half
			return false;
		abs_sum += c->coeffi[k] < 0 ? -c->coeffi[k] : c->coeffi[k];
	}
	// (ushort: the two byte planes' sums each below 2^24, and 256 S_hi + S_lo + rounding in an int as in the reference)
	if (u16 && abs_sum * 65535 + c->rounding >= (1LL << 31))
		return false;
	return abs_sum * 255 + c->rounding < (1LL << 24);
}

} // namespace

// Both passes of vips_convsep / vips_gaussblur (precision integer) on a uchar image, on the matrix cores.
// 1 = not this kernel's case (nothing launched), 0 = done, -1 = error.
int conv_u8_mfma_sep_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c, double offset2)
{
	if (c->mask_height != 1 || c->nnz != c->mask_width || !(c->mask_width & 1) || c->mask_width > 33 || c->mask_width < 3)
		return 1;
	if (!cm_common(in, out, c, offset2))
		return 1;
	const int B = in->bands, n = c->mask_width;
	CmArgs a;
	size_t lds;
	bool wide;
	if (!cm_geometry(in, out, c, n / 2, 0, &a, &lds, &wide))
		return 1;
	CmTables tabs;
	{
		const int r = cm_tables_device(c->coeffi.data(), n, 1, a.hp, a.width, &tabs);
		if (r)
			return r;
	}
	a.tz = tabs.tz;
	for (int k = 0; k < 3; k++)
		a.edge_wave[k] = tabs.edge_wave[k];
	Gate gate("conv_u8_mfma_sep");
	return cm_launch(B, wide, false, a, a.strips * a.segs, lds);
}

// The same on a ushort image: its bytes as 2 x bands planes of a uchar image through the same staging and the same
// operands, two exact products per sample and pass (low and high bytes), put together as integers where a pass
// rounds (conv_u8_mfma_body.h cm_fin16).  1 = not this kernel's case (nothing launched), 0 = done, -1 = error.
int conv_u16_mfma_sep_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c, double offset2)
{
	if (c->mask_height != 1 || c->nnz != c->mask_width || !(c->mask_width & 1) || c->mask_width > 33 || c->mask_width < 3)
		return 1;
	if (getenv("VIPS_HIP_NO_CONV_U16_MFMA") || !cm_common(in, out, c, offset2, true))
		return 1;
	const int n = c->mask_width;
	CmArgs a;
	size_t lds;
	bool wide;
	if (!cm_geometry(in, out, c, n / 2, 0, &a, &lds, &wide))
		return 1;
	CmTables tabs;
	{
		const int r = cm_tables_device(c->coeffi.data(), n, 1, a.hp, a.width, &tabs);
		if (r)
			return r;
	}
	a.tz = tabs.tz;
	for (int k = 0; k < 3; k++)
		a.edge_wave[k] = tabs.edge_wave[k];
	Gate gate("conv_u16_mfma_sep");
	return cm_launch16(in->bands, wide, a, a.strips * a.segs, lds);
}

// vips_conv (precision integer) with a two-dimensional mask of up to 9 rows and 33 columns on a uchar image: one
// Toeplitz product per mask row.  1 = not this kernel's case, 0 = done, -1 = error.
int conv_u8_mfma_2d_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c)
{
	const int mw = c->mask_width, mh = c->mask_height;
	if (!(mw & 1) || !(mh & 1) || mw > 33 || mh > 9 || mh < 3 || mw < 1)
		return 1;
	if (!cm_common(in, out, c, 0.0))
		return 1;
	const int B = in->bands;
	CmArgs a;
	size_t lds;
	bool wide;
	if (!cm_geometry(in, out, c, mw / 2, mh, &a, &lds, &wide))
		return 1;
	// the dense mask, row by row (zero taps were squeezed out of the plan)
	std::vector<int> dense((size_t) mw * mh, 0);
	for (int k = 0; k < c->nnz; k++)
		dense[c->pos[k]] = c->coeffi[k];
	CmTables tabs;
	{
		const int r = cm_tables_device(dense.data(), mw, mh, a.hp, a.width, &tabs);
		if (r)
			return r;
	}
	a.tz = tabs.tz;
	for (int k = 0; k < 3; k++)
		a.edge_wave[k] = tabs.edge_wave[k];
	Gate gate("conv_u8_mfma_2d");
	return cm_launch(B, wide, true, a, a.strips * a.segs, lds);
}

} // namespace vh
