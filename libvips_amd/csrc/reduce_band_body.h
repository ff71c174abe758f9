// vips_reducev on uchar images with a coefficient row PER OUTPUT ROW (a fractional shrink: what every
// vipsthumbnail --size N with N not a divisor of the image lands on, resample/reducev.cpp:517-619, thumbnail.c:413)
// on the matrix cores: out = A x in with A the banded matrix that holds, for output row y, the n_point
// coefficients cy[ty(y)][k] at input rows iy(y) + k -- a GEMM whose left matrix is made on the host.
//
// Exact integers in f32 as in reduce_u8.hip / conv_u8_mfma_body.h: a byte is the f16 denormal 0x00pp, a
// coefficient |c| < 2048 an exact half, sums below 2^24; retire = fma(acc, 4096, 2^-13) + v_cvt_pk_u8_f32 =
// clip((S + 2048) >> 12) (templates.h:152-157).
//
// A WAVE owns 128 byte columns (a dword per lane and row: lanes 0 .. 31; the two halves of the wave take
// different rows) and one block of 32 output rows.  It walks the input rows its block touches in steps of 16
// (aligned to 16 in the image, so the coefficient blocks do not depend on the strip): 8 dword loads per lane --
// whole 128-byte lines -- one v_perm per two bytes to make 4 operands (byte column c of the 8 rows), the step's
// 1 KiB coefficient block (L2-resident, the same for every strip), 4 v_mfma_f32_32x32x16_f16.  No LDS, no
// barrier, nothing shared between waves; rows above / below the image are the edge row read again (vips_embed
// COPY, reducev.cpp:975-980).  Neighbouring blocks read 16 % of their rows twice (from L2).
// Written against gcn.h (product) / tests/emul/gcn.h (host fibers, CPU suite).
#pragma once

#include "conv_u8_body.h" // (cu8_interleave)

namespace vh {

constexpr int RB_NT = 256; // 4 waves: 4 neighbouring strips

struct RbBlock {
	int s0, ns;       // first 16-row step (row 16 s0, may be negative), steps
	int tab;          // index of the block's first coefficient block (1 KiB each)
	int pad;
};

struct RbArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int row_bytes;    // bytes per row: a multiple of 4
	int height, out_height;
	int vs;           // > 1: a vips_shrinkv(vs) in front (resize.c:207-228): the rows the products read are box sums
	int mid_height;   // ... of vs image rows each, rounded as shrinkv.c:158-165 does; the height after it
	unsigned int mult; // 2^32 / (256 vs)
	int strips;       // of 128 bytes
	int nblocks;      // of 32 output rows
	int alternate;    // every other block of rows is walked from the bottom up
	int premul;       // RGBA uchar, box kernel: vips_premultiply's uchar fast path (max_alpha 255) on every loaded pixel
	const RbBlock *blk;
	const unsigned int *tab; // [coefficient block][64 lanes][4 dwords]
};

// S 2^-24 (two exact sums: of the low bytes and of the high bytes of ushort samples) -> clip((256 S_hi + S_lo +
// 2048) >> 12, 0, 65535) (templates.h:152-157 for unsigned short): exact 32-bit integers (the sums are below 2^24)
VH_DEV unsigned int rb_fin16(float lo, float hi)
{
	const int s = (vh::cvt_i32(hi * 16777216.0f) << 8) + vh::cvt_i32(lo * 16777216.0f);
	return (unsigned int) min(max((s + 2048) >> 12, 0), 65535);
}

// one wave: strip x block.  U16: ushort samples -- the strip's 128 byte columns are 64 samples, byte column 2 c of
// a lane's dword the low bytes of its sample c, 2 c + 1 the high bytes: the same four products, then both sums of
// a sample are put together as integers
// shrinkv's rounding on two 16-bit sums in one dword (bytes 0 and 2 of a dword column, or 1 and 3):
// ((sum + vs / 2) * (2^32 / (256 vs))) >> 24, shrinkv.c:158-165; the result again as two 16-bit lanes
template <int VS>
VH_DEV unsigned int rb_box2(unsigned int sums, unsigned int mult)
{
	if constexpr ((VS & (VS - 1)) == 0) {
		constexpr int SH = VS == 2 ? 1 : VS == 4 ? 2 : VS == 8 ? 3 : 4;
		constexpr unsigned int RND = (unsigned int) (VS / 2) * 0x00010001u;
		return ((sums + RND) >> SH) & 0x00ff00ffu;
	}
	else {
		const unsigned int lo = (((sums & 0xffffu) + VS / 2) * mult) >> 24;
		const unsigned int hi = (((sums >> 16) + VS / 2) * mult) >> 24;
		return lo | (hi << 16);
	}
}

// the same wave with a vips_shrinkv(VS) in front -- vips_resize's vertical half for a size that does not divide the
// image (resize.c:207-228: shrinkv, then reducev): a row the products read is the box sum of VS image rows, made
// from the loaded dwords as two 16-bit lanes per dword (even and odd byte columns) and rounded as shrinkv does; the
// 1 / VS-size intermediate image never exists.  Rows past the image are the last row (shrinkv's own embed, ceil).
// vips_premultiply on one RGBA uchar pixel (premultiply.c:163-176 with the table of :252-258 for max_alpha 255:
// scale[a] = (int) (256 a / 255.0) = a + (a == 255)): c' = (c scale + 128) >> 8, alpha kept.  R and B together in the
// 16-bit halves of one 24-bit multiply (255 * 256 + 128 < 2^16), G on its own.
VH_DEV unsigned int rb_premul(unsigned int v)
{
	const unsigned int al = v >> 24;
	const unsigned int sc = al + (al == 255u ? 1u : 0u);
	const unsigned int rb = (((v & 0x00ff00ffu) * sc + 0x00800080u) >> 8) & 0x00ff00ffu;
	const unsigned int g = ((((v >> 8) & 0xffu) * sc + 128u) >> 8) << 8;
	return rb | g | (v & 0xff000000u);
}

template <int VS>
VH_DEV void reducev_box_band_wave(const RbArgs &a, int strip, int g, bool rev)
{
	const int lane = tid() & 63, n = lane & 31, hf = lane >> 5;
	const int xb = min(strip * 128 + 4 * n, a.row_bytes - 4);
	const bool live = strip * 128 + 4 * n < a.row_bytes;
	const RbBlock b = uniform_load(&a.blk[g]);
	const gptr_in gin = gptr_in_of((unsigned long long) a.in);
	const gptr_in gtab = gptr_in_of((unsigned long long) a.tab);
	const unsigned int lane_off = (unsigned int) (8 * hf * VS * (int) a.in_stride + xb);

	// the image rows under shrunk rows 16 s + 8 hf + i, i = 0 .. 7, CH rows of each box at a time (rows k0 .. k0 +
	// CH - 1 of the VS; past VS: row VS - 1 again, masked out of the sum): a chunk of 8 CH dwords is in flight while
	// the products of the step before run -- all 8 VS rows at once do not fit the registers of three blocks a CU
	// past VS = 4.  One loop over (step, chunk) pairs, not unrolled: the same registers for every chunk.
	constexpr int CH = VS <= 4 ? VS : 4, NC = (VS + CH - 1) / CH;
	auto load = [&](int s, int k0, unsigned int (&raw)[8 * CH]) {
		const int m0 = 16 * s;
		// (opaque: or the row offsets, the same at every step, are kept in registers across the loop and spill)
		int st = (int) a.in_stride;
		opaque_uniform(st);
		if (m0 >= 0 && (m0 + 16) * VS <= a.height) {
			const gptr_in base = gin + (long long) m0 * VS * st;
#pragma unroll
			for (int i = 0; i < 8; i++)
#pragma unroll
				for (int k = 0; k < CH; k++)
					raw[i * CH + k] = gload32(base + (long long) (i * VS + min(k0 + k, VS - 1)) * st, lane_off);
		}
		else {
			// (32-bit offsets from the first row the step can read: the host checked 16 VS rows fit)
			const int r0 = min(min(max(m0, 0), a.mid_height - 1) * VS, a.height - 1);
			const gptr_in base = gin + (long long) r0 * st;
#pragma unroll
			for (int i = 0; i < 8; i++) {
				const int m = min(max(m0 + 8 * hf + i, 0), a.mid_height - 1);
#pragma unroll
				for (int k = 0; k < CH; k++) {
					const int r = min(m * VS + min(k0 + k, VS - 1), a.height - 1);
					raw[i * CH + k] = gload32(base, (unsigned int) ((r - r0) * st + xb));
				}
				sched_fence(); // (a lane offset per load: not all 8 CH of them at once)
			}
		}
	};

	float acc[4][16];
#pragma unroll
	for (int c = 0; c < 4; c++)
#pragma unroll
		for (int r = 0; r < 16; r++)
			acc[c][r] = 0.0f;
	unsigned int raw[8 * CH], ev[8], od[8], A[4];
#pragma unroll
	for (int i = 0; i < 8; i++)
		ev[i] = od[i] = 0;
	// (rev: the steps from the last to the first -- sums of integers, exact in any order)
	auto step_of = [&](int j) { return rev ? b.ns - 1 - j : j; };
	load(b.s0 + step_of(0), 0, raw);
	int j = 0, c = 0;
#pragma nounroll
	for (int t = 0; t < b.ns * NC; t++) {
		if (c == 0)
			gload128(gtab, (unsigned int) (((b.tab + step_of(j)) * 64 + lane) * 16), A);
#pragma unroll
		for (int k = 0; k < CH; k++) {
			const unsigned int keep = NC * CH == VS || c * CH + k < VS ? 0x00ff00ffu : 0u;
#pragma unroll
			for (int i = 0; i < 8; i++) {
				// (the premultiply of an RGBA thumbnail -- thumbnail.c:848-860 -- where the pixel is USED: the loads
				// of the next chunk stay in flight; a dword is a pixel, rows are whole pixels: host)
				const unsigned int w = a.premul ? rb_premul(raw[i * CH + k]) : raw[i * CH + k];
				ev[i] += w & keep;
				if constexpr (NC * CH == VS)
					od[i] += perm(w, w, 0x0c030c01u); // (bytes 1 and 3 in one instruction)
				else
					od[i] += (w >> 8) & keep;
			}
		}
		int c1 = c + 1, j1 = j;
		if (c1 == NC) {
			c1 = 0;
			j1++;
		}
		// (past the last chunk: the last one again -- every path through the loop writes raw[], which keeps it
		// in one set of registers)
		if (NC > 1)
			load(b.s0 + step_of(min(j1, b.ns - 1)), (j1 < b.ns ? c1 : NC - 1) * CH, raw);
		else if (j1 < b.ns)
			load(b.s0 + step_of(j1), 0, raw);
		if (c == NC - 1) {
			unsigned int cur[8];
#pragma unroll
			for (int i = 0; i < 8; i++) {
				cur[i] = rb_box2<VS>(ev[i], a.mult) | (rb_box2<VS>(od[i], a.mult) << 8);
				ev[i] = od[i] = 0;
			}
#pragma unroll
			for (int q4 = 0; q4 < 4; q4++) {
				unsigned int Bop[4];
#pragma unroll
				for (int q = 0; q < 4; q++)
					Bop[q] = perm(cur[2 * q + 1], cur[2 * q], 0x0c000c00u | ((4u + (unsigned int) q4) << 16) | (unsigned int) q4);
				mfma_32x32x16_f16(A, Bop, acc[q4]);
			}
		}
		c = c1;
		j = j1;
	}
	const gptr_out gout = gptr_out_of((unsigned long long) a.out);
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const int y = 32 * g + (r & 3) + 8 * (r >> 2) + 4 * hf;
		unsigned int w = 0;
#pragma unroll
		for (int c = 0; c < 4; c++)
			w = cvt_pk_u8(__builtin_fmaf(acc[c][r], 4096.0f, 0x1p-13f), (unsigned int) c, w);
		if (live && y < a.out_height)
			gstore32(gout + (long long) y * a.out_stride + xb, w);
	}
}

// Every other block of rows is walked from the bottom up: two blocks that share rows (the taps of neighbouring
// blocks overlap: 13 of 64 + 13 rows at a residual of 2) then read them at about the same time -- one at its end
// as the other at its end, or both at their start -- and the second read is an L2 hit.  With every block walking
// down, a block read the shared rows ~20 us after its neighbour had: 279 MB fetched for an image of 201, 209 MB now
// (profiles/r05k_band_traffic.txt; 0.0472 -> 0.0396 ms).  Two or four blocks a wave one after the other instead
// (the shared rows re-read by the same CU) were slower than either.
VH_DEV bool rb_bottom_up(const RbArgs &a, int g) { return a.alternate && (g & 1); }

template <bool U16>
VH_DEV void reducev_band_wave(const RbArgs &a, int strip, int g, bool rev)
{
	const int lane = tid() & 63, n = lane & 31, hf = lane >> 5;
	const int xb = min(strip * 128 + 4 * n, a.row_bytes - 4); // (lanes past the row: its last dword, not stored)
	const bool live = strip * 128 + 4 * n < a.row_bytes;
	const RbBlock b = uniform_load(&a.blk[g]);
	const gptr_in gin = gptr_in_of((unsigned long long) a.in);
	const gptr_in gtab = gptr_in_of((unsigned long long) a.tab);
	const unsigned int lane_off = (unsigned int) (8 * hf * (int) a.in_stride + xb);

	// rows 16 s + 8 hf + idx of the image, idx = 0 .. 7
	auto load = [&](int s, unsigned int (&d)[8]) {
		const int r0 = 16 * s;
		if (r0 >= 0 && r0 + 16 <= a.height) {
			const gptr_in base = gin + (long long) r0 * a.in_stride;
#pragma unroll
			for (int i = 0; i < 8; i++)
				d[i] = gload32(base + (long long) i * a.in_stride, lane_off);
		}
		else {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				const int r = min(max(r0 + 8 * hf + i, 0), a.height - 1);
				d[i] = gload32(gin + (long long) r * a.in_stride, (unsigned int) xb);
			}
		}
	};

	float acc[4][16];
	unsigned int cur[8], nxt[8];
	// (rev: the steps from the last to the first -- sums of integers, exact in any order)
	auto step_of = [&](int j) { return rev ? b.ns - 1 - j : j; };
	load(b.s0 + step_of(0), cur);
	for (int j = 0; j < b.ns; j++) {
		if (j + 1 < b.ns)
			load(b.s0 + step_of(j + 1), nxt);
		unsigned int A[4];
		gload128(gtab, (unsigned int) (((b.tab + step_of(j)) * 64 + lane) * 16), A);
#pragma unroll
		for (int c = 0; c < 4; c++) {
			// byte column c of the 8 rows as halves: dword q = rows 2 q, 2 q + 1
			unsigned int Bop[4];
#pragma unroll
			for (int q = 0; q < 4; q++)
				Bop[q] = perm(cur[2 * q + 1], cur[2 * q], 0x0c000c00u | ((4u + (unsigned int) c) << 16) | (unsigned int) c);
			if (j == 0)
				mfma_32x32x16_f16_first(A, Bop, acc[c]);
			else
				mfma_32x32x16_f16(A, Bop, acc[c]);
		}
#pragma unroll
		for (int i = 0; i < 8; i++)
			cur[i] = nxt[i];
	}
	// register r: output row 32 g + (r & 3) + 8 (r >> 2) + 4 hf of the lane's 4 byte columns
	const gptr_out gout = gptr_out_of((unsigned long long) a.out);
#pragma unroll
	for (int r = 0; r < 16; r++) {
		const int y = 32 * g + (r & 3) + 8 * (r >> 2) + 4 * hf;
		unsigned int w = 0;
		if constexpr (U16)
			w = rb_fin16(acc[0][r], acc[1][r]) | (rb_fin16(acc[2][r], acc[3][r]) << 16);
		else {
#pragma unroll
			for (int c = 0; c < 4; c++)
				w = cvt_pk_u8(__builtin_fmaf(acc[c][r], 4096.0f, 0x1p-13f), (unsigned int) c, w);
		}
		if (live && y < a.out_height)
			gstore32(gout + (long long) y * a.out_stride + xb, w);
	}
}

// ---------------------------------------------------------------------------------------------------------
// vips_reduceh with a coefficient row per output column (resample/reduceh.cpp:216-335), the same way: out^T =
// A x in^T with A[x_out][x_in] the banded matrix of the coefficients.  Here the sum runs ALONG the rows, so a
// lane holds 8 neighbouring pixels of ITS row (lane & 15 = the row: 16 rows per wave, v_mfma_f32_16x16x32_f16) --
// 8 B contiguous bytes per lane and step, de-interleaved into one operand per band by v_perm.  The input of a
// horizontal pass is small (the vertical pass ran first, reduce.c:98-121) and every line a lane touches is used
// whole by its next steps, so there is no LDS staging either.  A wave makes 16 output columns of 16 rows; the
// result comes out with 4 neighbouring columns per lane: 4 B bytes, one store.
// Steps are aligned to 32 pixels of the image; a step that is not inside the image reads pixel by pixel, clamped
// (vips_embed COPY, reduceh.cpp:515-520).

struct RbhArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int width;        // input pixels per row
	int out_width;
	int rows;         // rows of the pass
	int xtiles;       // of 16 output columns
	int ytiles;       // of 16 rows
	int out_dwords;   // output rows start on dwords: a lane's 4 pixels leave as whole dwords
	const RbBlock *blk; // per x tile: first 32-pixel step (may be negative), steps, first coefficient block
	const unsigned int *tab;
};

// N dwords (up to 8) at a 4-byte aligned offset
template <int N>
VH_DEV void gload_dwords_n(gptr_in base, unsigned int off, unsigned int (&w)[N])
{
	if constexpr (N <= 4)
		gload_dwords<N>(base, off, w);
	else {
		unsigned int lo[4], hi[N - 4];
		gload_dwords<4>(base, off, lo);
		gload_dwords<N - 4>(base, off + 16, hi);
#pragma unroll
		for (int i = 0; i < 4; i++)
			w[i] = lo[i];
#pragma unroll
		for (int i = 0; i < N - 4; i++)
			w[4 + i] = hi[i];
	}
}

// 4 pixels x B bands (B dwords as they lie in memory) -> band b as 4 halves 0x00pp
template <int B>
VH_DEV void rbh_halves(const unsigned int *raw, int b, unsigned int &a0, unsigned int &a1)
{
#pragma unroll
	for (int q = 0; q < 2; q++) {
		const int e0 = (2 * q) * B + b, e1 = (2 * q + 1) * B + b;
		const unsigned int v =
			perm(raw[e1 >> 2], raw[e0 >> 2], 0x0c000c00u | ((4u + (unsigned int) (e1 & 3)) << 16) | (unsigned int) (e0 & 3));
		if (q == 0)
			a0 = v;
		else
			a1 = v;
	}
}

// N dwords (up to 16) at a 4-byte aligned offset / address
template <int N>
VH_DEV void gload_dwords_long(gptr_in base, unsigned int off, unsigned int (&w)[N])
{
#pragma unroll
	for (int k = 0; k < N; k += 4) {
		constexpr int dummy = 0;
		(void) dummy;
		if (N - k >= 4) {
			unsigned int t[4];
			gload_dwords<4>(base, off + 4 * k, t);
#pragma unroll
			for (int i = 0; i < 4; i++)
				w[k + i] = t[i];
		}
		else {
			unsigned int t[2];
			gload_dwords<2>(base, off + 4 * k, t);
			w[k] = t[0];
			if (k + 1 < N)
				w[k + 1] = t[1];
		}
	}
}

// ushort: 4 pixels x B bands (2 B dwords) -> the low (part 0) or high (part 1) bytes of band b as 4 halves 0x00pp
template <int B>
VH_DEV void rbh_halves16(const unsigned int *raw, int b, int part, unsigned int &a0, unsigned int &a1)
{
#pragma unroll
	for (int q = 0; q < 2; q++) {
		const int e0 = 2 * ((2 * q) * B + b) + part, e1 = 2 * ((2 * q + 1) * B + b) + part;
		const unsigned int v =
			perm(raw[e1 >> 2], raw[e0 >> 2], 0x0c000c00u | ((4u + (unsigned int) (e1 & 3)) << 16) | (unsigned int) (e0 & 3));
		if (q == 0)
			a0 = v;
		else
			a1 = v;
	}
}

// the same for ushort images (rows, widths and strides in SAMPLES of 2 bytes where the uchar form has bytes): two
// products per band and step -- the low bytes' and the high bytes' -- put together as integers at the end
template <int B>
VH_DEV void reduceh16_band_wave(const RbhArgs &a, int xt, int yt)
{
	static_assert(B % 2 == 0 || B == 1 || B == 3, "bands");
	const int lane = tid() & 63, n = lane & 15, kg = lane >> 4;
	const int y = 16 * yt + n, yc = min(y, a.rows - 1);
	const RbBlock b = uniform_load(&a.blk[xt]);
	const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) yc * a.in_stride;
	const gptr_in gtab = gptr_in_of((unsigned long long) a.tab);
	constexpr int ND = 4 * B; // dwords of 8 pixels

	auto load = [&](int s, unsigned int (&d)[ND]) {
		const int p0 = 32 * s;
		if (p0 >= 0 && p0 + 32 <= a.width)
			gload_dwords_long<ND>(line, (unsigned int) ((p0 + 8 * kg) * B * 2), d);
		else {
#pragma unroll
			for (int w = 0; w < ND; w++) {
				unsigned int v = 0;
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const int e = 2 * w + k; // sample of the group
					const int px = min(max(p0 + 8 * kg + e / B, 0), a.width - 1);
					v |= gload16(line, (unsigned int) ((px * B + e % B) * 2)) << (16 * k);
				}
				d[w] = v;
			}
		}
	};

	float acc[B][2][4];
	unsigned int cur[ND], nxt[ND];
	load(b.s0, cur);
	for (int j = 0; j < b.ns; j++) {
		if (j + 1 < b.ns)
			load(b.s0 + j + 1, nxt);
		unsigned int A[4];
		gload128(gtab, (unsigned int) (((b.tab + j) * 64 + lane) * 16), A);
#pragma unroll
		for (int bb = 0; bb < B; bb++)
#pragma unroll
			for (int part = 0; part < 2; part++) {
				unsigned int D[4];
				rbh_halves16<B>(cur, bb, part, D[0], D[1]);
				rbh_halves16<B>(cur + 2 * B, bb, part, D[2], D[3]);
				if (j == 0)
					mfma_16x16x32_f16_first(A, D, acc[bb][part]);
				else
					mfma_16x16x32_f16(A, D, acc[bb][part]);
			}
#pragma unroll
		for (int i = 0; i < ND; i++)
			cur[i] = nxt[i];
	}
	// register r: output column 16 xt + 4 kg + r of row y; the lane's 4 pixels x B samples in memory order
	unsigned int w[2 * B];
#pragma unroll
	for (int d = 0; d < 2 * B; d++) {
		const int e0 = 2 * d, e1 = 2 * d + 1; // samples (pixel e / B, band e % B)
		w[d] = rb_fin16(acc[e0 % B][0][e0 / B], acc[e0 % B][1][e0 / B]) | (rb_fin16(acc[e1 % B][0][e1 / B], acc[e1 % B][1][e1 / B]) << 16);
	}
	const int x0 = 16 * xt + 4 * kg;
	if (y < a.rows && x0 < a.out_width) {
		const gptr_out p = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride + (long long) x0 * B * 2;
		if (x0 + 4 <= a.out_width && a.out_dwords) {
#pragma unroll
			for (int d = 0; d < 2 * B; d++)
				gstore32(p + 4 * d, w[d]);
		}
		else
			for (int e = 0; e < min(4, a.out_width - x0) * B; e++)
				gstore16(p + 2 * e, (unsigned short) (w[e >> 1] >> (16 * (e & 1))));
	}
}

template <int B>
VH_DEV void reduceh_band_wave(const RbhArgs &a, int xt, int yt)
{
	const int lane = tid() & 63, n = lane & 15, kg = lane >> 4;
	const int y = 16 * yt + n, yc = min(y, a.rows - 1);
	const RbBlock b = uniform_load(&a.blk[xt]);
	const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) yc * a.in_stride;
	const gptr_in gtab = gptr_in_of((unsigned long long) a.tab);

	// pixels 32 s + 8 kg .. + 7 of the lane's row as 2 B dwords
	auto load = [&](int s, unsigned int (&d)[2 * B]) {
		const int p0 = 32 * s;
		if (p0 >= 0 && p0 + 32 <= a.width)
			gload_dwords_n<2 * B>(line, (unsigned int) ((p0 + 8 * kg) * B), d);
		else {
#pragma unroll
			for (int w = 0; w < 2 * B; w++) {
				unsigned int v = 0;
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int e = 4 * w + k;
					const int px = min(max(p0 + 8 * kg + e / B, 0), a.width - 1);
					v |= (unsigned int) gload8(line, (unsigned int) (px * B + e % B)) << (8 * k);
				}
				d[w] = v;
			}
		}
	};

	float acc[B][4];
	unsigned int cur[2 * B], nxt[2 * B];
	load(b.s0, cur);
	for (int j = 0; j < b.ns; j++) {
		if (j + 1 < b.ns)
			load(b.s0 + j + 1, nxt);
		unsigned int A[4];
		gload128(gtab, (unsigned int) (((b.tab + j) * 64 + lane) * 16), A);
#pragma unroll
		for (int bb = 0; bb < B; bb++) {
			unsigned int D[4];
			rbh_halves<B>(cur, bb, D[0], D[1]);
			rbh_halves<B>(cur + B, bb, D[2], D[3]);
			if (j == 0)
				mfma_16x16x32_f16_first(A, D, acc[bb]);
			else
				mfma_16x16x32_f16(A, D, acc[bb]);
		}
#pragma unroll
		for (int i = 0; i < 2 * B; i++)
			cur[i] = nxt[i];
	}
	// register r: output column 16 xt + 4 kg + r of row y
	unsigned int P[B], w[B];
#pragma unroll
	for (int bb = 0; bb < B; bb++) {
		unsigned int v = 0;
#pragma unroll
		for (int r = 0; r < 4; r++)
			v = cvt_pk_u8(__builtin_fmaf(acc[bb][r], 4096.0f, 0x1p-13f), (unsigned int) r, v);
		P[bb] = v;
	}
	cu8_interleave<B>(P, w);
	const int x0 = 16 * xt + 4 * kg;
	if (y < a.rows && x0 < a.out_width) {
		const gptr_out p = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride + (long long) x0 * B;
		if (x0 + 4 <= a.out_width && a.out_dwords)
			gstore_dwords<B>(p, w);
		else
			for (int e = 0; e < min(4, a.out_width - x0) * B; e++)
				gstore8(p + e, (unsigned char) (w[e >> 2] >> (8 * (e & 3))));
	}
}

} // namespace vh
