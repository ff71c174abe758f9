// vips_reducev / vips_shrinkv / vips_reduceh / vips_shrinkh on ushort images, streaming: the kernel
// bodies, written against gcn.h (product) / tests/emul/gcn.h (host fibers, CPU suite).
//
// The general kernels of resample.hip give a thread one output element and let it walk its taps
// through global memory (2 bytes per lane and tap, every input row read n / shrink times).  Here
//
//   reducev  a thread owns 4 element columns (one 8-byte load per row: a wave reads 512
//            contiguous bytes) and the block walks DOWN the rows of a segment, two rows at a time:
//            the pair of rows of a column is one dword of two signed 16-bit lanes (p - 32768), and
//            every output row in flight takes its two taps from it with ONE v_dot2_i32_i16.  Which
//            outputs are in flight, with which coefficient pair, where a sum starts (at
//            2048 + 32768 * the sum of its coefficients: the rounding term of templates.h:152-157
//            and the bias back) and where it retires is a host-made schedule, one record per pair
//            of rows, read as scalars: outputs live in 8 static accumulator slots (y mod 8), no
//            row is read twice within a segment, any shrink factor and any kernel.  Sums are
//            32-bit and wrap exactly as the reference's int does (reducev.cpp:462-471).
//   shrinkv  the same walk with plain sums, shrinkv.c:233-244's multiply-high at the end of a box.
//   reduceh  a block stages the bytes its 64 output pixels x 4 rows need in LDS with coalesced
//            loads -- 8 bytes of padding after every 64, so that the windows of neighbouring
//            outputs (64 bytes apart for RGBA and a shrink of 8) fall in different banks -- and a
//            thread sums one output pixel's taps from there (reduceh.cpp:278-322).
//   shrinkh  a thread sums the hshrink pixels of one output pixel straight from global memory
//            (they are contiguous, and so are neighbouring threads' boxes), shrinkh.c:98-112.
#pragma once

#include "gcn.h"

namespace vh {

constexpr int R16_NT = 256;
constexpr int R16_SLOTS = 8;   // outputs in flight per column
constexpr int R16_PF = 3;      // pairs of rows travelling per lane

// one pair of input rows of the vertical reduce
struct R16Pair {
	unsigned int c2[R16_SLOTS]; // per slot: coefficients of (row, row + 1) as two i16, 0 = not in flight
	int init[R16_SLOTS];        // per slot that starts here: what its sum starts at
	int yret[R16_SLOTS];        // per slot that retires here: its output row
	unsigned int start_mask, ret_mask;
	int pad[2];
};

struct R16VArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int row_bytes;         // of a row of either image: a multiple of 8
	int in_height, out_height;
	int strips, segs, seg_rows;
	int r_base;            // input row of pair 0's first row (may be negative)
	const R16Pair *sched;  // reducev: the schedule
	const int *seg_pairs;  // reducev: [2 seg], [2 seg + 1] = first and last pair of a segment
	int vshrink;           // shrinkv
	unsigned int mult;     // shrinkv: ceil(2^32 / vshrink) (vshrink >= 2)
	int *counter;
	int off_slot;
};

// ---- reducev
static __device__ __forceinline__ void reducev16_body(const R16VArgs &a, int item)
{
	const int t = tid();
	const int strip = item % a.strips, seg = item / a.strips;
	const int col = strip * (R16_NT * 8) + 8 * t;
	const bool live = col < a.row_bytes;
	const unsigned int off = (unsigned int) (live ? col : a.row_bytes - 8);
	const int ya = seg * a.seg_rows, yb = min(ya + a.seg_rows, a.out_height);
	const int p0 = uniform_load(a.seg_pairs + 2 * seg), p1 = uniform_load(a.seg_pairs + 2 * seg + 1);

	auto load = [&](int row, unsigned int (&w)[2]) {
		const int rc = min(max(row, 0), a.in_height - 1);
		const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) rc * a.in_stride;
		gload64(line, off, w);
	};
	unsigned int raw[R16_PF][2][2];
#pragma unroll
	for (int k = 0; k < R16_PF; k++) {
		load(a.r_base + 2 * (p0 + k), raw[k][0]);
		load(a.r_base + 2 * (p0 + k) + 1, raw[k][1]);
	}
	int acc[R16_SLOTS][4];
#pragma unroll
	for (int j = 0; j < R16_SLOTS; j++)
#pragma unroll
		for (int e = 0; e < 4; e++)
			acc[j][e] = 0;
	for (int p = p0; p <= p1; p++) {
		// (read one pair ahead the records cost 28 more scalar registers and the kernel ran 18 % slower)
		const R16Pair rec = uniform_load(a.sched + p);
		// the pair's 4 columns: (row, row + 1) as signed 16-bit lanes
		unsigned int pk[4];
		pk[0] = perm(raw[0][1][0], raw[0][0][0], 0x05040100u) ^ 0x80008000u;
		pk[1] = perm(raw[0][1][0], raw[0][0][0], 0x07060302u) ^ 0x80008000u;
		pk[2] = perm(raw[0][1][1], raw[0][0][1], 0x05040100u) ^ 0x80008000u;
		pk[3] = perm(raw[0][1][1], raw[0][0][1], 0x07060302u) ^ 0x80008000u;
#pragma unroll
		for (int k = 0; k + 1 < R16_PF; k++)
#pragma unroll
			for (int h = 0; h < 2; h++) {
				raw[k][h][0] = raw[k + 1][h][0];
				raw[k][h][1] = raw[k + 1][h][1];
			}
		if (p + R16_PF <= p1) {
			load(a.r_base + 2 * (p + R16_PF), raw[R16_PF - 1][0]);
			load(a.r_base + 2 * (p + R16_PF) + 1, raw[R16_PF - 1][1]);
		}
		const unsigned int start = rec.start_mask, ret = rec.ret_mask;
#pragma unroll
		for (int j = 0; j < R16_SLOTS; j++) {
			if (start & (1u << j)) {
				const int v = rec.init[j];
#pragma unroll
				for (int e = 0; e < 4; e++)
					acc[j][e] = v;
			}
			const unsigned int c2 = rec.c2[j];
#pragma unroll
			for (int e = 0; e < 4; e++)
				acc[j][e] = dot2(pk[e], c2, acc[j][e]);
			if (ret & (1u << j)) {
				const int y = rec.yret[j];
				if (y >= ya && y < yb && live) {
					unsigned int o[2];
#pragma unroll
					for (int d = 0; d < 2; d++) {
						const int lo = min(max(acc[j][2 * d] >> 12, 0), 65535);
						const int hi = min(max(acc[j][2 * d + 1] >> 12, 0), 65535);
						o[d] = (unsigned int) lo | ((unsigned int) hi << 16);
					}
					const gptr_out line = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride;
					gstore32(line + col, o[0]);
					gstore32(line + col + 4, o[1]);
				}
			}
		}
	}
}

// ---- shrinkv: no row is shared between outputs, so the simple map is the best one: a block per
// (4 KB of an output row), 16 bytes per lane, the rows of the box read one after the other (the
// blocks of a row are neighbours in launch order: whole rows stream from memory in order)
static __device__ __forceinline__ void shrinkv16_body(const R16VArgs &a, int bx, int by, int gy)
{
	const int col = (bx * R16_NT + tid()) * 16;
	if (col >= a.row_bytes)
		return;
	const int vs = a.vshrink;
	const bool whole = col + 16 <= a.row_bytes; // (rows are whole 8-byte groups: the last lane may hold one)
	for (int y = by; y < a.out_height; y += gy) {
		unsigned int s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		for (int k = 0; k < vs; k++) {
			const int rc = min(y * vs + k, a.in_height - 1);
			const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) rc * a.in_stride;
			unsigned int w[4] = { 0, 0, 0, 0 };
			if (whole)
				gload128(line, (unsigned int) col, w);
			else {
				unsigned int h[2];
				gload64(line, (unsigned int) col, h);
				w[0] = h[0];
				w[1] = h[1];
			}
#pragma unroll
			for (int d = 0; d < 4; d++) {
				s[2 * d] += w[d] & 0xffffu;
				s[2 * d + 1] += w[d] >> 16;
			}
		}
		// shrinkv.c:233-244: ((sum + vshrink / 2) * ceil(2^32 / vshrink)) >> 32, truncated to ushort
		unsigned int o[4];
#pragma unroll
		for (int d = 0; d < 4; d++) {
			const unsigned int x0 = s[2 * d] + (unsigned int) (vs >> 1), x1 = s[2 * d + 1] + (unsigned int) (vs >> 1);
			const unsigned int q0 = vs == 1 ? x0 : umulhi(x0, a.mult), q1 = vs == 1 ? x1 : umulhi(x1, a.mult);
			o[d] = (q0 & 0xffffu) | (q1 << 16);
		}
		const gptr_out line = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride;
		if (whole)
			gstore128(line + col, o);
		else {
			gstore32(line + col, o[0]);
			gstore32(line + col + 4, o[1]);
		}
	}
}

// ---- reducev on UCHAR images with a coefficient row per output row (any shrink, any kernel): the
// same walk down a segment from the same kind of schedule.  A lane owns 8 byte columns (one 8-byte
// load per row); the pair of rows of a column is one dword of two 16-bit lanes (the bytes zero
// extended: v_perm), every output row in flight takes its two taps from it with one
// v_dot2_i32_i16.  Sums start at 2048 (templates.h:152-157's rounding term; no bias: a byte is a
// non-negative 16-bit value), retire as clip(sum >> 12) to 0 .. 255.  reducev_u8_kernel
// (reduce_u8.hip) gives a thread one output row and re-reads its n_point input rows through L2.
static __device__ __forceinline__ void reducev8_body(const R16VArgs &a, int item)
{
	const int t = tid();
	const int strip = item % a.strips, seg = item / a.strips;
	const int col = strip * (R16_NT * 8) + 8 * t;
	const bool live = col < a.row_bytes;
	const unsigned int off = (unsigned int) (live ? col : a.row_bytes - 8);
	const int ya = seg * a.seg_rows, yb = min(ya + a.seg_rows, a.out_height);
	const int p0 = uniform_load(a.seg_pairs + 2 * seg), p1 = uniform_load(a.seg_pairs + 2 * seg + 1);

	auto load = [&](int row, unsigned int (&w)[2]) {
		const int rc = min(max(row, 0), a.in_height - 1);
		const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) rc * a.in_stride;
		gload64(line, off, w);
	};
	unsigned int raw[R16_PF][2][2];
#pragma unroll
	for (int k = 0; k < R16_PF; k++) {
		load(a.r_base + 2 * (p0 + k), raw[k][0]);
		load(a.r_base + 2 * (p0 + k) + 1, raw[k][1]);
	}
	int acc[R16_SLOTS][8];
#pragma unroll
	for (int j = 0; j < R16_SLOTS; j++)
#pragma unroll
		for (int e = 0; e < 8; e++)
			acc[j][e] = 0;
	for (int p = p0; p <= p1; p++) {
		const R16Pair rec = uniform_load(a.sched + p);
		// the pair's 8 columns: (row, row + 1) as two 16-bit lanes
		unsigned int pk[8];
#pragma unroll
		for (int e = 0; e < 8; e++)
			pk[e] = perm(raw[0][1][e >> 2], raw[0][0][e >> 2], 0x0c000c00u | (unsigned int) (e & 3) | ((4u + (e & 3)) << 16));
#pragma unroll
		for (int k = 0; k + 1 < R16_PF; k++)
#pragma unroll
			for (int h = 0; h < 2; h++) {
				raw[k][h][0] = raw[k + 1][h][0];
				raw[k][h][1] = raw[k + 1][h][1];
			}
		if (p + R16_PF <= p1) {
			load(a.r_base + 2 * (p + R16_PF), raw[R16_PF - 1][0]);
			load(a.r_base + 2 * (p + R16_PF) + 1, raw[R16_PF - 1][1]);
		}
		const unsigned int start = rec.start_mask, ret = rec.ret_mask;
#pragma unroll
		for (int j = 0; j < R16_SLOTS; j++) {
			if (start & (1u << j)) {
				const int v = rec.init[j];
#pragma unroll
				for (int e = 0; e < 8; e++)
					acc[j][e] = v;
			}
			const unsigned int c2 = rec.c2[j];
#pragma unroll
			for (int e = 0; e < 8; e++)
				acc[j][e] = dot2(pk[e], c2, acc[j][e]);
			if (ret & (1u << j)) {
				const int y = rec.yret[j];
				if (y >= ya && y < yb && live) {
					unsigned int o[2] = { 0, 0 };
#pragma unroll
					for (int e = 0; e < 8; e++) {
						int v = acc[j][e] >> 12;
						opaque(v); // (shift and clamp kept apart: v_ashr_pk_u8_i32, NOTES 3.1 "toolchain findings")
						o[e >> 2] |= (unsigned int) min(max(v, 0), 255) << (8 * (e & 3));
					}
					const gptr_out line = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride;
					gstore32(line + col, o[0]);
					gstore32(line + col + 4, o[1]);
				}
			}
		}
	}
}

static __device__ __forceinline__ void reducev8_block(const R16VArgs &a, unsigned int *lds)
{
	int *slot = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(lds) + a.off_slot);
	for (;;) {
		const int item = next_item(a.counter, slot);
		if (item >= a.strips * a.segs)
			return;
		reducev8_body(a, item);
	}
}

static __device__ __forceinline__ void reducev16_block(const R16VArgs &a, unsigned int *lds)
{
	int *slot = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(lds) + a.off_slot);
	for (;;) {
		const int item = next_item(a.counter, slot);
		if (item >= a.strips * a.segs)
			return;
		reducev16_body(a, item);
	}
}

// ---- reduceh: 64 output pixels x 4 rows per block trip
constexpr int R16H_PX = 64, R16H_ROWS = 4;

struct R16Pos {
	int first, phase; // (= ReducePos, resample.h)
};

struct R16HArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int in_width, out_width, height, bands;
	int n_point;
	int span_dwords;     // staged dwords per row, padded layout included
	const R16Pos *pos;   // per output pixel: first tap (unclamped), coefficient row
	const short *table;  // [phase][n_point]
	int hshrink;         // shrinkh
	unsigned int mult;
	int vshrink, in_height; // shrinkbox16: the vertical box in front of the horizontal one
	unsigned int multv;
	int aligned16;          // shrinkbox16: rows start on 16 bytes (the box's bytes as they lie, 16 per load)
	int x_first;            // shrinkbox16: the first output column this launch makes (behind shrinkbox16c: the ragged one)
	int x_full;             // shrinkbox16c: output columns whose box lies inside the image
};

// byte offset of input byte `b` of a staged row (8 bytes of padding after every 64)
VH_DEV int r16h_pad(int b) { return b + ((b >> 6) << 3); }

// B ushorts to a pixel of the output
template <int B>
VH_DEV void r16_store_px(gptr_out dst, const unsigned int (&v)[B])
{
	if constexpr (B % 2 == 0) {
#pragma unroll
		for (int d = 0; d < B / 2; d++)
			gstore32(dst + 4 * d, (v[2 * d] & 0xffffu) | (v[2 * d + 1] << 16));
	}
	else {
#pragma unroll
		for (int b = 0; b < B; b++)
			gstore16(dst + 2 * b, (unsigned short) v[b]);
	}
}

template <int B>
static __device__ __forceinline__ void reduceh16_body(const R16HArgs &a, int bx, int by, int gy, unsigned int *lds)
{
	const int t = tid();
	const int x0 = bx * R16H_PX;
	const int nx = min(R16H_PX, a.out_width - x0);
	constexpr int PB = 2 * B; // bytes per pixel
	// the pixels the block's taps touch, clamped into the image (vips_embed COPY)
	const int p_lo = min(max(a.pos[x0].first, 0), a.in_width - 1);
	const int p_hi = min(max(a.pos[x0 + nx - 1].first + a.n_point - 1, 0), a.in_width - 1);
	// whole dwords from the one holding the first byte (rows are whole dwords: host)
	const int byte_lo = (p_lo * PB) & ~3, skew = p_lo * PB - byte_lo, byte_hi = (p_hi + 1) * PB;
	const int ndw = (byte_hi - byte_lo + 3) >> 2;
	unsigned char *stage = reinterpret_cast<unsigned char *>(lds);
	const int row_pitch = a.span_dwords * 4;

	// thread -> (output pixel, row of the trip)
	const int px = t & (R16H_PX - 1), rr = t >> 6;
	const bool mine = px < nx;
	const int first = mine ? a.pos[x0 + px].first : 0;
	const short *c = a.table + (size_t) (mine ? a.pos[x0 + px].phase : 0) * a.n_point;
	for (int y0 = by * R16H_ROWS; y0 < a.height; y0 += gy * R16H_ROWS) {
		barrier();
#pragma unroll
		for (int r = 0; r < R16H_ROWS; r++) {
			const int y = min(y0 + r, a.height - 1);
			const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) y * a.in_stride + byte_lo;
			for (int i = t; i < ndw; i += R16_NT)
				*reinterpret_cast<unsigned int *>(stage + r * row_pitch + r16h_pad(4 * i)) = gload32(line, (unsigned int) (4 * i));
		}
		barrier();
		const int y = y0 + rr;
		if (mine && y < a.height) {
			int sum[B];
#pragma unroll
			for (int b = 0; b < B; b++)
				sum[b] = 0;
			const unsigned char *row = stage + rr * row_pitch;
			for (int k = 0; k < a.n_point; k++) {
				const int s = min(max(first + k, 0), a.in_width - 1);
				const int ck = c[k];
				const int at = (s - p_lo) * PB + skew;
				// (a pixel of 2 or 6 bytes can straddle a padded 64-byte group: pad per element)
#pragma unroll
				for (int b = 0; b < B; b++)
					sum[b] += ck * (int) *reinterpret_cast<const unsigned short *>(row + r16h_pad(at + 2 * b));
			}
			const gptr_out dst = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride + (long long) (x0 + px) * PB;
			unsigned int v[B];
#pragma unroll
			for (int b = 0; b < B; b++)
				v[b] = (unsigned int) min(max((sum[b] + 2048) >> 12, 0), 65535);
			r16_store_px<B>(dst, v);
		}
	}
}

// ---- shrinkh: a thread per output pixel, 4 rows per trip
template <int B>
static __device__ __forceinline__ void shrinkh16_body(const R16HArgs &a, int bx, int by, int gy)
{
	const int x = bx * R16_NT + tid();
	if (x >= a.out_width)
		return;
	constexpr int PB = 2 * B;
	const int hs = a.hshrink;
	for (int y0 = by * R16H_ROWS; y0 < a.height; y0 += gy * R16H_ROWS) {
		unsigned int s[R16H_ROWS][B];
#pragma unroll
		for (int r = 0; r < R16H_ROWS; r++)
#pragma unroll
			for (int b = 0; b < B; b++)
				s[r][b] = (unsigned int) (hs >> 1);
		// a box that lies inside the image and is whole 16-byte groups (RGBA and an even shrink, ...):
		// its bytes as they lie, 16 per load
		const int box_bytes = hs * PB;
		const bool wide = B % 2 == 0 && box_bytes % 16 == 0 && (x + 1) * hs <= a.in_width;
		if (wide) {
#pragma unroll
			for (int r = 0; r < R16H_ROWS; r++) {
				const int y = min(y0 + r, a.height - 1);
				const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) y * a.in_stride;
				for (int c = 0; c < box_bytes; c += 16) {
					unsigned int w[4];
					gload128(line, (unsigned int) (x * box_bytes + c), w);
#pragma unroll
					for (int d = 0; d < 4; d++) {
						// dword d of a 16-byte group: elements 2 d, 2 d + 1 -> bands (2 d) % B, (2 d + 1) % B
						s[r][(2 * d) % B] += w[d] & 0xffffu;
						s[r][(2 * d + 1) % B] += w[d] >> 16;
					}
				}
			}
		}
		for (int k = 0; k < (wide ? 0 : hs); k++) {
			const int px = min(x * hs + k, a.in_width - 1);
#pragma unroll
			for (int r = 0; r < R16H_ROWS; r++) {
				const int y = min(y0 + r, a.height - 1);
				const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) y * a.in_stride;
				if constexpr (B % 2 == 0) {
#pragma unroll
					for (int d = 0; d < B / 2; d++) {
						const unsigned int w = gload32(line, (unsigned int) (px * PB + 4 * d));
						s[r][2 * d] += w & 0xffffu;
						s[r][2 * d + 1] += w >> 16;
					}
				}
				else {
#pragma unroll
					for (int b = 0; b < B; b++)
						s[r][b] += gload16(line, (unsigned int) (px * PB + 2 * b));
				}
			}
		}
#pragma unroll
		for (int r = 0; r < R16H_ROWS; r++) {
			const int y = y0 + r;
			if (y < a.height) {
				const gptr_out dst = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride + (long long) x * PB;
				unsigned int v[B];
#pragma unroll
				for (int b = 0; b < B; b++)
					v[b] = (hs == 1 ? s[r][b] : umulhi(s[r][b], a.mult)) & 0xffffu;
				r16_store_px<B>(dst, v);
			}
		}
	}
}

// ---- vips_shrink (shrink.c:77-119: shrinkv, then shrinkh) in ONE kernel (round 6): a thread per output pixel sums
// the vshrink rows of each of its hshrink columns, rounds that sum the way shrinkv does (shrinkv.c:233-244), adds the
// rounded column sums and rounds as shrinkh does (shrinkh.c:98-112) -- the image is read once and the 1 / vshrink-size
// intermediate (537 MB for 16384^2 RGBA ushort by 4) never exists: 0.62 ms for the pair of kernels, of which the
// second one's re-read was a quarter.
template <int B>
static __device__ __forceinline__ void shrinkbox16_body(const R16HArgs &a, int bx, int by, int gy)
{
	const int x = a.x_first + bx * R16_NT + tid();
	if (x >= a.out_width)
		return;
	constexpr int PB = 2 * B;
	const int hs = a.hshrink, vs = a.vshrink;
	const int box_bytes = hs * PB;
	const bool wide = a.aligned16 && B % 2 == 0 && box_bytes % 16 == 0 && (x + 1) * hs <= a.in_width;
	auto round_v = [&](unsigned int sum) -> unsigned int {
		const unsigned int v = sum + (unsigned int) (vs >> 1);
		return (vs == 1 ? v : umulhi(v, a.multv)) & 0xffffu;
	};
	for (int y = by; y < a.height; y += gy) {
		unsigned int tot[B];
#pragma unroll
		for (int b = 0; b < B; b++)
			tot[b] = (unsigned int) (hs >> 1);
		if (wide) {
			// 16 bytes = 8 elements of the box at a time (their bands: element e of a group is band e % B, B = 2 or 4)
			for (int c = 0; c < box_bytes; c += 16) {
				unsigned int v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
				for (int k = 0; k < vs; k++) {
					const int row = min(y * vs + k, a.in_height - 1);
					const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) row * a.in_stride;
					unsigned int w[4];
					gload128(line, (unsigned int) (x * box_bytes + c), w);
#pragma unroll
					for (int d = 0; d < 4; d++) {
						v[2 * d] += w[d] & 0xffffu;
						v[2 * d + 1] += w[d] >> 16;
					}
				}
#pragma unroll
				for (int e = 0; e < 8; e++)
					tot[e % B] += round_v(v[e]);
			}
		}
		else {
			for (int k = 0; k < hs; k++) {
				const int px = min(x * hs + k, a.in_width - 1);
				unsigned int v[B];
#pragma unroll
				for (int b = 0; b < B; b++)
					v[b] = 0;
				for (int j = 0; j < vs; j++) {
					const int row = min(y * vs + j, a.in_height - 1);
					const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) row * a.in_stride;
#pragma unroll
					for (int b = 0; b < B; b++)
						v[b] += gload16(line, (unsigned int) (px * PB + 2 * b));
				}
#pragma unroll
				for (int b = 0; b < B; b++)
					tot[b] += round_v(v[b]);
			}
		}
		const gptr_out dst = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride + (long long) x * PB;
		unsigned int o[B];
#pragma unroll
		for (int b = 0; b < B; b++)
			o[b] = (hs == 1 ? tot[b] : umulhi(tot[b], a.mult)) & 0xffffu;
		r16_store_px<B>(dst, o);
	}
}

// ... with the lanes on CONSECUTIVE 16-byte groups of the row, as shrinkv16 has them (a thread per output pixel reads
// 32 bytes a row for RGBA by 4: every other 16 bytes of a wave's load instruction -- 0.61 ms, no faster than the two
// kernels): a lane sums its group's 8 elements down the vshrink rows, rounds them (shrinkv), adds the pixels of its
// group per band, and the L = box bytes / 16 lanes of a box add theirs with wave shifts; the box's first lane rounds
// (shrinkh) and stores.  Boxes of 16, 32 or 64 bytes, 1 / 2 / 4 bands, rows on 16 bytes; the ragged last column of a
// ceil shrink is shrinkbox16_body's.
template <int B, int L>
static __device__ __forceinline__ void shrinkbox16c_body(const R16HArgs &a, int bx, int by, int gy)
{
	const int lane = bx * R16_NT + tid();
	const int vs = a.vshrink, hs = a.hshrink;
	const unsigned int col = (unsigned int) lane * 16u;
	const unsigned int full_bytes = (unsigned int) a.x_full * (unsigned int) (hs * 2 * B);
	const bool live = col + 16u <= full_bytes;
	const unsigned int lcol = live ? col : 0u;
	for (int y = by; y < a.height; y += gy) {
		unsigned int s[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
		for (int k = 0; k < vs; k++) {
			const int row = min(y * vs + k, a.in_height - 1);
			const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) row * a.in_stride;
			unsigned int w[4];
			gload128(line, lcol, w);
#pragma unroll
			for (int d = 0; d < 4; d++) {
				s[2 * d] += w[d] & 0xffffu;
				s[2 * d + 1] += w[d] >> 16;
			}
		}
		unsigned int p[B];
#pragma unroll
		for (int b = 0; b < B; b++)
			p[b] = 0;
#pragma unroll
		for (int e = 0; e < 8; e++) {
			const unsigned int v = s[e] + (unsigned int) (vs >> 1);
			p[e % B] += (vs == 1 ? v : umulhi(v, a.multv)) & 0xffffu;
		}
		if constexpr (L >= 2) {
#pragma unroll
			for (int b = 0; b < B; b++)
				p[b] += lane_next(p[b]);
		}
		if constexpr (L >= 4) {
#pragma unroll
			for (int b = 0; b < B; b++)
				p[b] += lane_from(p[b], 2);
		}
		if (live && (tid() & (L - 1)) == 0) {
			const int x = lane / L;
			const gptr_out dst = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride + (long long) x * (2 * B);
			unsigned int o[B];
#pragma unroll
			for (int b = 0; b < B; b++) {
				const unsigned int t = p[b] + (unsigned int) (hs >> 1);
				o[b] = (hs == 1 ? t : umulhi(t, a.mult)) & 0xffffu;
			}
			r16_store_px<B>(dst, o);
		}
	}
}

} // namespace vh
