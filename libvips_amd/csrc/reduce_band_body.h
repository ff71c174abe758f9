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

#include "gcn.h"

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
	int strips;       // of 128 bytes
	int nblocks;      // of 32 output rows
	const RbBlock *blk;
	const unsigned int *tab; // [coefficient block][64 lanes][4 dwords]
};

// one wave: strip x block
VH_DEV void reducev_band_wave(const RbArgs &a, int strip, int g)
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
	load(b.s0, cur);
	for (int j = 0; j < b.ns; j++) {
		if (j + 1 < b.ns)
			load(b.s0 + j + 1, nxt);
		unsigned int A[4];
		gload128(gtab, (unsigned int) (((b.tab + j) * 64 + lane) * 16), A);
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
#pragma unroll
		for (int c = 0; c < 4; c++)
			w = cvt_pk_u8(__builtin_fmaf(acc[c][r], 4096.0f, 0x1p-13f), (unsigned int) c, w);
		if (live && y < a.out_height)
			gstore32(gout + (long long) y * a.out_stride + xb, w);
	}
}

} // namespace vh
