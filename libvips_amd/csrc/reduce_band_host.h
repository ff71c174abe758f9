// Host side of reduce_band_body.h: the banded coefficient matrix cut into 1 KiB blocks, geometry, launch.
// Included by reduce_band.hip (which defines rb_launch() as a kernel launch) and by
// tests/emul/reduce_band_emul.cpp (host fiber runs).
#pragma once

#include "reduce_band_body.h"
#include "reduce_u8.h"
#include "resample.h"

#include <cstdlib>
#include <cstring>
#include <tuple>
#include <vector>

namespace vh {

// defined by the including file; 0 on success
static int rb_launch(const RbArgs &a, int groups, int wblocks, bool u16);
static int rbh_launch(int bands, const RbhArgs &a, int grid, bool u16);

namespace {

// a small integer as IEEE half bits (|v| < 2048: exact)
unsigned int rb_half_bits(int v)
{
	if (v == 0)
		return 0;
	const unsigned int sign = v < 0 ? 0x8000u : 0u;
	const unsigned int m = (unsigned int) (v < 0 ? -v : v);
	int e = 0;
	while ((m >> (e + 1)) != 0)
		e++; // m in [2^e, 2^(e+1)), e <= 10
	return sign | ((unsigned int) (e + 15) << 10) | ((m << (10 - e)) & 0x3ffu);
}

// The device blob of a plan for `out_height` rows: nblocks RbBlock records, then the coefficient blocks.
// Coefficient block (block g, step j): lane (m = l & 31, hf), slot idx <-> image row 16 (s0 + j) + 8 hf + idx
// (rows outside the image: the edge row, loaded again), tap row - first(y) of output row y = 32 g + m.
// nullptr in *blob (and 0 returned): not this kernel's case; -1: error
int rb_plan(_VipsHipReduce *r, int out_height, int tile, const unsigned char **blob)
{
	std::lock_guard<std::mutex> lock(r->mutex);
	const auto key = std::make_tuple(-(1 << 25), out_height, tile);
	auto it = r->pos_cache.find(key);
	if (it != r->pos_cache.end()) {
		*blob = (const unsigned char *) it->second;
		return 0;
	}
	*blob = nullptr;
	const int n = r->n_point;
	bool ok = true;
	// exact halves, sums below 2^24
	for (int ph = 0; ph <= 64 && ok; ph++) {
		long long abs_sum = 0;
		for (int k = 0; k < n; k++) {
			const int c = r->matrixs[(size_t) ph * n + k];
			ok = ok && c > -2048 && c < 2048;
			abs_sum += c < 0 ? -c : c;
		}
		ok = ok && abs_sum * 255 + 2048 < (1LL << 24);
	}
	std::vector<unsigned char> bytes;
	if (ok) {
		std::vector<ReducePos> pos;
		reduce_positions(r, 0, out_height, tile, pos);
		const int nblocks = (out_height + 31) / 32;
		std::vector<RbBlock> blk(nblocks);
		std::vector<unsigned int> tab;
		for (int g = 0; g < nblocks; g++) {
			const int y0 = 32 * g, y1 = out_height < y0 + 32 ? out_height : y0 + 32;
			int lo = pos[y0].first, hi = pos[y0].first + n - 1;
			for (int y = y0; y < y1; y++) {
				lo = pos[y].first < lo ? pos[y].first : lo;
				hi = pos[y].first + n - 1 > hi ? pos[y].first + n - 1 : hi;
			}
			const int s0 = lo >= 0 ? lo / 16 : -((-lo + 15) / 16), s1 = hi >= 0 ? hi / 16 : -((-hi + 15) / 16);
			blk[g].s0 = s0;
			blk[g].ns = s1 - s0 + 1;
			blk[g].tab = (int) (tab.size() / 256);
			blk[g].pad = 0;
			tab.resize(tab.size() + (size_t) blk[g].ns * 256, 0u);
			unsigned int *t = tab.data() + (size_t) blk[g].tab * 256;
			for (int j = 0; j < blk[g].ns; j++)
				for (int l = 0; l < 64; l++) {
					const int m = l & 31, hf = l >> 5, y = y0 + m;
					for (int idx = 0; idx < 8; idx++) {
						const int row = 16 * (s0 + j) + 8 * hf + idx;
						int c = 0;
						if (y < y1) {
							const int k = row - pos[y].first;
							if (k >= 0 && k < n)
								c = r->matrixs[(size_t) pos[y].phase * n + k];
						}
						t[(j * 64 + l) * 4 + (idx >> 1)] |= rb_half_bits(c) << (16 * (idx & 1));
					}
				}
		}
		if (tab.size() * 4 > (64u << 20))
			ok = false; // (a table that size is not a thumbnail's)
		else {
			bytes.resize(blk.size() * sizeof(RbBlock) + tab.size() * 4);
			memcpy(bytes.data(), blk.data(), blk.size() * sizeof(RbBlock));
			memcpy(bytes.data() + blk.size() * sizeof(RbBlock), tab.data(), tab.size() * 4);
		}
	}
	void *d = nullptr;
	if (ok) {
		d = upload(bytes.data(), bytes.size());
		if (!d)
			return -1;
	}
	r->pos_cache[key] = (ReducePos *) d; // (nullptr: tried, not this kernel's case)
	*blob = (const unsigned char *) d;
	return 0;
}

// the same for the horizontal pass: x tiles of 16 output columns, steps of 32 pixels, blocks for
// v_mfma_f32_16x16x32_f16 -- lane (m = l & 15, kg = l >> 4), slot idx <-> pixel 32 (s0 + j) + 8 kg + idx, tap
// pixel - first(x) of output column x = 16 X + m
int rbh_plan(_VipsHipReduce *r, int out_width, int tile, const unsigned char **blob)
{
	std::lock_guard<std::mutex> lock(r->mutex);
	const auto key = std::make_tuple(-(1 << 26), out_width, tile);
	auto it = r->pos_cache.find(key);
	if (it != r->pos_cache.end()) {
		*blob = (const unsigned char *) it->second;
		return 0;
	}
	*blob = nullptr;
	const int n = r->n_point;
	bool ok = true;
	for (int ph = 0; ph <= 64 && ok; ph++) {
		long long abs_sum = 0;
		for (int k = 0; k < n; k++) {
			const int c = r->matrixs[(size_t) ph * n + k];
			ok = ok && c > -2048 && c < 2048;
			abs_sum += c < 0 ? -c : c;
		}
		ok = ok && abs_sum * 255 + 2048 < (1LL << 24);
	}
	std::vector<unsigned char> bytes;
	if (ok) {
		std::vector<ReducePos> pos;
		reduce_positions(r, 0, out_width, tile, pos);
		const int xtiles = (out_width + 15) / 16;
		std::vector<RbBlock> blk(xtiles);
		std::vector<unsigned int> tab;
		for (int X = 0; X < xtiles; X++) {
			const int x0 = 16 * X, x1 = out_width < x0 + 16 ? out_width : x0 + 16;
			int lo = pos[x0].first, hi = pos[x0].first + n - 1;
			for (int x = x0; x < x1; x++) {
				lo = pos[x].first < lo ? pos[x].first : lo;
				hi = pos[x].first + n - 1 > hi ? pos[x].first + n - 1 : hi;
			}
			const int s0 = lo >= 0 ? lo / 32 : -((-lo + 31) / 32), s1 = hi >= 0 ? hi / 32 : -((-hi + 31) / 32);
			blk[X].s0 = s0;
			blk[X].ns = s1 - s0 + 1;
			blk[X].tab = (int) (tab.size() / 256);
			blk[X].pad = 0;
			tab.resize(tab.size() + (size_t) blk[X].ns * 256, 0u);
			unsigned int *t = tab.data() + (size_t) blk[X].tab * 256;
			for (int j = 0; j < blk[X].ns; j++)
				for (int l = 0; l < 64; l++) {
					const int m = l & 15, kg = l >> 4, x = x0 + m;
					for (int idx = 0; idx < 8; idx++) {
						const int px = 32 * (s0 + j) + 8 * kg + idx;
						int c = 0;
						if (x < x1) {
							const int k = px - pos[x].first;
							if (k >= 0 && k < n)
								c = r->matrixs[(size_t) pos[x].phase * n + k];
						}
						t[(j * 64 + l) * 4 + (idx >> 1)] |= rb_half_bits(c) << (16 * (idx & 1));
					}
				}
		}
		if (tab.size() * 4 > (64u << 20))
			ok = false;
		else {
			bytes.resize(blk.size() * sizeof(RbBlock) + tab.size() * 4);
			memcpy(bytes.data(), blk.data(), blk.size() * sizeof(RbBlock));
			memcpy(bytes.data() + blk.size() * sizeof(RbBlock), tab.data(), tab.size() * 4);
		}
	}
	void *d = nullptr;
	if (ok) {
		d = upload(bytes.data(), bytes.size());
		if (!d)
			return -1;
	}
	r->pos_cache[key] = (ReducePos *) d;
	*blob = (const unsigned char *) d;
	return 0;
}

} // namespace

// vips_reduceh of whole rows of a uchar image with a coefficient row per output column; 1 = done, 0 = not this
// kernel's case, -1 = error
int reduceh_band_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	const char *env = getenv("VIPS_HIP_REDUCE_BAND");
	if (env && atoi(env) == 0)
		return 0;
	const bool u16 = in->format == VIPS_HIP_FORMAT_USHORT;
	if ((in->format != VIPS_HIP_FORMAT_UCHAR && !u16) || out->format != in->format || in->bands != out->bands ||
		in->bands < 1 || in->bands > 4 || out->width < 1)
		return 0;
	// whole rows (any range of them)
	if (in->left || out->left || in->width != in->im_width || out->width != out->im_width ||
		out->top < in->top || out->top + out->height > in->top + in->height)
		return 0;
	if (((uintptr_t) in->data | (uintptr_t) in->stride) & 3)
		return 0;
	if ((long long) in->width * in->bands * (u16 ? 2 : 1) >= (1LL << 31) || in->width < 1)
		return 0;
	// (a ushort row of whole dwords: an even number of samples, or the clamped edge loads would split one)
	if (u16 && (((uintptr_t) out->data | (uintptr_t) out->stride) & 1))
		return 0;
	const unsigned char *blob;
	if (rbh_plan(r, out->width, tile, &blob))
		return -1;
	if (!blob)
		return 0;
	RbhArgs a;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data + (long long) (out->top - in->top) * in->stride;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.out_width = out->width;
	a.rows = out->height;
	a.xtiles = (out->width + 15) / 16;
	a.ytiles = (out->height + 15) / 16;
	a.out_dwords = !(((uintptr_t) out->data | (uintptr_t) out->stride) & 3);
	a.blk = (const RbBlock *) blob;
	a.tab = (const unsigned int *) (blob + (size_t) a.xtiles * sizeof(RbBlock));
	const int groups = (a.xtiles + 3) / 4; // a block of 4 waves: 4 neighbouring x tiles of the same rows
	Gate gate(u16 ? "reduceh_u16_band" : "reduceh_u8_band");
	const int rc = rbh_launch(in->bands, a, groups * a.ytiles, u16);
	return rc ? -1 : 1;
}

int shrinkv_reducev_band_try(_VipsHipReduce *r, int vs, int mid_height, const VipsHipRegion *in, const VipsHipRegion *out, int tile,
	bool premul);

// vips_reducev of a whole uchar image with a coefficient row per output row; 1 = done, 0 = not this
// kernel's case, -1 = error
int reducev_band_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	return shrinkv_reducev_band_try(r, 1, 0, in, out, tile, false);
}

// ... with a vips_shrinkv(vs, ceil) in front, the two as one kernel: `in` is the image BEFORE the shrink, `r` the
// plan of the reduce on the image after it (mid_height rows); vs = 1: the reduce alone
// premul: vips_premultiply(uchar fast path, max_alpha 255) of an RGBA image in front of both, on the loaded pixels
int shrinkv_reducev_band_try(_VipsHipReduce *r, int vs, int mid_height, const VipsHipRegion *in, const VipsHipRegion *out, int tile,
	bool premul)
{
	const char *env = getenv("VIPS_HIP_REDUCE_BAND");
	if (env && atoi(env) == 0)
		return 0;
	if (premul && (vs < 2 || in->bands != 4 || in->format != VIPS_HIP_FORMAT_UCHAR || getenv("VIPS_HIP_NO_BAND_PREMUL")))
		return 0;
	const bool u16 = in->format == VIPS_HIP_FORMAT_USHORT;
	if ((in->format != VIPS_HIP_FORMAT_UCHAR && !u16) || out->format != in->format || in->bands != out->bands)
		return 0;
	if (in->left || in->top || out->left || out->top || in->width != in->im_width || in->height != in->im_height ||
		out->width != out->im_width || out->height != out->im_height || in->width != out->width)
		return 0;
	if ((((uintptr_t) in->data | in->stride | (uintptr_t) out->data | out->stride) & 3) ||
		(in->width * in->bands * (u16 ? 2 : 1)) % 4)
		return 0;
	if (vs != 1 && (u16 || vs < 2 || vs > 16 || mid_height < 1))
		return 0;
	// (lane offsets are 32-bit: 16 rows of the image, 16 vs with a shrink in front)
	if ((long long) in->stride * (16 * vs + 1) >= (1LL << 31) || out->height < 1 || in->height < 1)
		return 0;
	const unsigned char *blob;
	if (rb_plan(r, out->height, tile, &blob))
		return -1;
	if (!blob)
		return 0;
	RbArgs a;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.row_bytes = in->width * in->bands * (u16 ? 2 : 1);
	a.height = in->height;
	a.out_height = out->height;
	a.vs = vs;
	a.mid_height = vs > 1 ? mid_height : in->height;
	a.mult = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) vs));
	a.strips = (a.row_bytes + 127) / 128;
	a.nblocks = (out->height + 31) / 32;
	a.blk = (const RbBlock *) blob;
	a.tab = (const unsigned int *) (blob + (size_t) a.nblocks * sizeof(RbBlock));
	const int groups = (((a.strips + 3) / 4) + 7) & ~7; // blocks of 4 waves = 4 neighbouring strips; a multiple of 8
	Gate gate(vs > 1 ? "shrinkv_reducev_u8_band" : u16 ? "reducev_u16_band" : "reducev_u8_band");
	a.alternate = !getenv("VIPS_HIP_BAND_NO_ALTERNATE");
	a.premul = premul ? 1 : 0;
	const int rc = rb_launch(a, groups, a.nblocks, u16);
	return rc ? -1 : 1;
}

} // namespace vh
