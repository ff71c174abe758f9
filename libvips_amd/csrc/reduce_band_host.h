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
static int rb_launch(const RbArgs &a, int grid);

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

} // namespace

// vips_reducev of a whole uchar image with a coefficient row per output row; 1 = done, 0 = not this
// kernel's case, -1 = error
int reducev_band_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	const char *env = getenv("VIPS_HIP_REDUCE_BAND");
	if (env && atoi(env) == 0)
		return 0;
	if (in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR || in->bands != out->bands)
		return 0;
	if (in->left || in->top || out->left || out->top || in->width != in->im_width || in->height != in->im_height ||
		out->width != out->im_width || out->height != out->im_height || in->width != out->width)
		return 0;
	if ((((uintptr_t) in->data | in->stride | (uintptr_t) out->data | out->stride) & 3) || (in->width * in->bands) % 4)
		return 0;
	// (lane offsets are 32-bit: 16 rows of the image)
	if ((long long) in->stride * 17 >= (1LL << 31) || out->height < 1 || in->height < 1)
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
	a.row_bytes = in->width * in->bands;
	a.height = in->height;
	a.out_height = out->height;
	a.strips = (a.row_bytes + 127) / 128;
	a.nblocks = (out->height + 31) / 32;
	a.blk = (const RbBlock *) blob;
	a.tab = (const unsigned int *) (blob + (size_t) a.nblocks * sizeof(RbBlock));
	const int groups = (a.strips + 3) / 4; // a block of 4 waves: 4 neighbouring strips
	Gate gate("reducev_u8_band");
	const int rc = rb_launch(a, groups * a.nblocks);
	return rc ? -1 : 1;
}

} // namespace vh
