// Both passes of vips_convsep / vips_gaussblur (precision integer, convsep.c:61-118, convi.c:698-716) on
// uchar images ON THE MATRIX CORES: a separable convolution is a banded (Toeplitz) matrix times the image,
// twice.  north_star: "MFMA used only where a large conv re-cast as im2col x GEMM actually wins" -- a 29-tap
// gaussian is 58 multiply-adds per output byte; as v_dot4 on packed bytes (conv_u8_body.h) that is ~25 vector
// instructions per output byte at half rate (0.4 - 0.75 ms for 8192^2 x 3, 6 - 13 % of HBM); as
// v_mfma_f32_32x32x16_f16 it is 8 matrix instructions per 1024 output bytes.
//
// Exact integers in f32, as in reduce_u8.hip: a pixel byte p is the f16 DENORMAL 0x00pp = p 2^-24 (no
// conversion: a v_perm puts a zero byte above it), a coefficient |c| < 2048 is an exact half, every product
// and every partial sum is an integer below 2^24 times 2^-24.  The accumulators start at (scale / 2) 2^-24, so
// acc 2^24 = sum + scale / 2 and the rounding of conv_u8_body.h -- RN(x RN(1 / scale) - 0.5 + 1 / (2 scale)) =
// floor(x / scale) for 0 <= x < 2^24, scale <= 8000, a negative numerator clips to 0 -- is one v_fma_f32
// (by 2^24 RN(1 / scale): the same real product) and one v_cvt_pk_u8_f32.
//
// A block of 4 waves owns 128 output columns and streams down a segment of rows in chunks of 32:
//   stage    the chunk's 32 input rows x (128 + 2 hp) columns (hp = half rounded up to 4: row starts are
//            dwords) go from global memory straight into LDS (global_load_lds_dword: no register, no wait
//            until the chunk is needed; the next chunk travels while this one is computed), rows and
//            columns outside the image clamped to its edge (vips_embed COPY);
//   pass 1   wave w makes mid[y][x] for its 32 columns: C1[y][x] = sum_u A1[y][u] T[u][x] over the 64 window
//            columns u.  A1: lane (y = lane & 31, hf = lane >> 5) reads two groups of 4 pixels x B bands of ITS
//            row from LDS per 16 columns, v_perm makes them halves per band; T[u][x] = c[u - x - (hp - half)],
//            made on the host, 16 VGPRs for the launch.  4 matrix instructions per band;
//   rounding the 16 accumulators of a lane are mid rows (r & 3) + 8 (r >> 2) + 4 hf of column lane & 31:
//            rounded to bytes in 16-bit lanes they ARE the operand of pass 2 (slot <-> row is a bijection
//            both operands share, the sum does not care about its order) -- no transpose, no LDS;
//   pass 2   out[x][y] = sum_v A2[x][v] T[v][y] over 64 mid rows v: the previous chunk's 32 and this one's
//            (chunks start at row Ya - hp, and pass 1 takes its pixels in the slot order the accumulators
//            come out in, so ONE Toeplitz operand serves both passes); output rows Ya + 32 (c - 1) .. + 32
//            after chunk c.  The result has lane & 31 = the ROW and 4
//            neighbouring columns per register quad: bytes, bands interleaved, 4 B bytes to LDS;
//   store    the block copies the 32 x 128 output pixels out of LDS in whole rows of dwords.
// Written against gcn.h (product) / tests/emul/gcn.h (host fibers, CPU suite).
#pragma once

#include "conv_u8_body.h"

namespace vh {

constexpr int CM_NT = 256;  // threads per block: 4 waves side by side
constexpr int CM_BW = 128;  // output columns per block
constexpr int CM_ROWS = 32; // rows per chunk

struct CmArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int width, height;
	int half, hp;        // taps / 2; rounded up to a multiple of 4
	int strips, segs;    // blocks across; segments down
	int seg_rows;        // output rows per segment: a multiple of 32
	int in_dw;           // dwords of a staged row that hold pixels: (128 + 2 hp) B / 4
	int in_pitch;        // ... and its pitch in LDS (in_pitch / 2 odd: a wave's 8-byte reads meet no bank twice)
	int in_buf;          // dwords per staging buffer: 32 in_pitch rounded up to a whole number of 256
	int out_pitch;       // dwords per output row in LDS: 32 B + 2
	float acc0;          // (scale / 2) 2^-24
	float k1, bias;      // 2^24 RN(1 / scale), -0.5 + 1 / (2 scale)
	const unsigned int *tz;      // the Toeplitz operand: [4 k-steps][64 lanes][4 dwords]
};

// 4 pixels x B bands (B dwords as they lie in memory) -> band b as 4 halves 0x00pp: dword q = pixels 2 q, 2 q + 1
template <int B>
VH_DEV void cm_halves(const unsigned int (&raw)[B], int b, unsigned int &a0, unsigned int &a1)
{
#pragma unroll
	for (int q = 0; q < 2; q++) {
		const int e0 = (2 * q) * B + b, e1 = (2 * q + 1) * B + b;
		// bytes: [e0, 0, e1, 0]; perm(hi, lo, sel): selector 0..3 = a byte of lo, 4..7 = of hi, 0x0c = 0x00
		const unsigned int v =
			perm(raw[e1 >> 2], raw[e0 >> 2], 0x0c000c00u | ((4u + (unsigned int) (e1 & 3)) << 16) | (unsigned int) (e0 & 3));
		if (q == 0)
			a0 = v;
		else
			a1 = v;
	}
}

// one work item: strip x segment
template <int B>
VH_DEV void conv_u8_mfma_item(const CmArgs &a, int item, unsigned int *lds)
{
	const int t = tid(), lane = t & 63, wv = wave_index(), n = lane & 31, hf = lane >> 5;
	const int strip = item % a.strips, seg = item / a.strips;
	const int X0 = strip * CM_BW;
	const int Xs = X0 - a.hp; // first staged column
	const int Ya = seg * a.seg_rows, Yb = min(Ya + a.seg_rows, a.height);
	const int nchunks = (Yb - Ya + CM_ROWS - 1) / CM_ROWS + 1;
	unsigned int *lds_out = lds + 2 * a.in_buf;
	const bool interior = Xs >= 0 && Xs + CM_BW + 2 * a.hp <= a.width;
	const gptr_in gin = gptr_in_of((unsigned long long) a.in);

	// the Toeplitz operand of both passes
	unsigned int T[4][4];
#pragma unroll
	for (int s = 0; s < 4; s++)
		gload128(gptr_in_of((unsigned long long) a.tz), (unsigned int) ((s * 64 + lane) * 16), T[s]);

	// chunk c: mid rows Ya - hp + 32 c .. + 32 = the same input rows, clamped
	auto stage = [&](int c) {
		unsigned int *dst = lds + (c & 1) * a.in_buf;
		const int r0 = Ya - a.hp + CM_ROWS * c;
		const int r0c = min(max(r0, 0), a.height - 1);
		// (the uniform base carries the row and the strip: lane offsets stay below 32 strides)
		const gptr_in base = gin + (long long) r0c * a.in_stride + (long long) Xs * B;
		// dword d of the tile -> (row, col); a wave's instruction j covers d = 64 j .. 64 j + 63
		int d = 64 * wv + lane;
		int row = d / a.in_pitch, col = d - row * a.in_pitch;
		const int step_row = 256 / a.in_pitch, step_col = 256 - step_row * a.in_pitch;
		for (int j = wv; 64 * j < a.in_buf; j += 4) {
			const int rr = min(row, CM_ROWS - 1), cc = min(col, a.in_dw - 1); // (padding, the tail: any valid dword)
			const int rc = min(max(r0 + rr, 0), a.height - 1) - r0c;
			if (interior)
				lds_dma_dword(base, (unsigned int) (rc * (int) a.in_stride + 4 * cc), dst + 64 * j);
			else {
				// a strip over the left or right edge of the image: byte by byte, columns clamped
				const gptr_in line = gin + (long long) (rc + r0c) * a.in_stride;
				unsigned int w = 0;
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const int e = 4 * cc + k;
					const int px = min(max(Xs + e / B, 0), a.width - 1);
					w |= (unsigned int) gload8(line, (unsigned int) (px * B + e % B)) << (8 * k);
				}
				dst[64 * j + lane] = w;
			}
			col += step_col;
			row += step_row;
			if (col >= a.in_pitch) {
				col -= a.in_pitch;
				row++;
			}
		}
	};

	unsigned int mid_prev[B][8];
#pragma unroll
	for (int b = 0; b < B; b++)
#pragma unroll
		for (int q = 0; q < 8; q++)
			mid_prev[b][q] = 0;

	stage(0);
	for (int c = 0; c < nchunks; c++) {
		wait_vmem0(); // this wave's share of chunk c has landed (and its stores of the last output rows are done)
		barrier();    // ... everybody's; and every wave is past the reads of the buffer chunk c + 1 goes to
		if (c + 1 < nchunks)
			stage(c + 1);
		// the lane's pixels of row n: per 16 window columns the groups 4 hf .. + 3 and 8 + 4 hf .. + 3
		const unsigned int *src = lds + (c & 1) * a.in_buf + n * a.in_pitch + B * (8 * wv + hf);
		unsigned int raw[4][2][B];
#pragma unroll
		for (int s = 0; s < 4; s++)
#pragma unroll
			for (int g = 0; g < 2; g++)
#pragma unroll
				for (int i = 0; i < B; i++)
					raw[s][g][i] = src[B * (4 * s + 2 * g) + i];
		unsigned int P[B][4]; // [band][quad of columns 8 j + 4 hf .. + 3 of row n]
#pragma unroll
		for (int b = 0; b < B; b++) {
			// ---- pass 1
			float acc[16];
#pragma unroll
			for (int r = 0; r < 16; r++)
				acc[r] = a.acc0;
#pragma unroll
			for (int s = 0; s < 4; s++) {
				unsigned int A[4];
				cm_halves<B>(raw[s][0], b, A[0], A[1]);
				cm_halves<B>(raw[s][1], b, A[2], A[3]);
				mfma_32x32x16_f16(A, T[s], acc);
			}
			// rows (r & 3) + 8 (r >> 2) + 4 hf of column n, rounded, as halves: pass 2's operand
			unsigned int mid_cur[8];
#pragma unroll
			for (int q = 0; q < 8; q++) {
				unsigned int w = cvt_pk_u8(__builtin_fmaf(acc[2 * q], a.k1, a.bias), 0u, 0u);
				mid_cur[q] = cvt_pk_u8(__builtin_fmaf(acc[2 * q + 1], a.k1, a.bias), 2u, w);
			}
			// ---- pass 2: output rows Ya + 32 (c - 1) .. + 32
			if (c >= 1) {
#pragma unroll
				for (int r = 0; r < 16; r++)
					acc[r] = a.acc0;
				unsigned int A[4];
#pragma unroll
				for (int s = 0; s < 2; s++) {
#pragma unroll
					for (int q = 0; q < 4; q++)
						A[q] = mid_prev[b][4 * s + q];
					mfma_32x32x16_f16(A, T[s], acc);
				}
#pragma unroll
				for (int s = 0; s < 2; s++) {
#pragma unroll
					for (int q = 0; q < 4; q++)
						A[q] = mid_cur[4 * s + q];
					mfma_32x32x16_f16(A, T[2 + s], acc);
				}
#pragma unroll
				for (int j = 0; j < 4; j++) {
					unsigned int w = 0;
#pragma unroll
					for (int i = 0; i < 4; i++)
						w = cvt_pk_u8(__builtin_fmaf(acc[4 * j + i], a.k1, a.bias), (unsigned int) i, w);
					P[b][j] = w;
				}
			}
#pragma unroll
			for (int q = 0; q < 8; q++)
				mid_prev[b][q] = mid_cur[q];
			sched_fence(); // (one band's accumulators at a time: interleaved, the bands do not fit 168 registers)
		}
		if (c >= 1) {
			unsigned int *orow = lds_out + n * a.out_pitch + B * (8 * wv + hf);
#pragma unroll
			for (int j = 0; j < 4; j++) {
				unsigned int Pj[B], w[B];
#pragma unroll
				for (int b = 0; b < B; b++)
					Pj[b] = P[b][j];
				cu8_interleave<B>(Pj, w);
#pragma unroll
				for (int b = 0; b < B; b++)
					orow[2 * B * j + b] = w[b];
			}
			barrier();
			// the block's 32 rows x 32 B dwords, whole rows of dwords
			const int y0 = Ya + CM_ROWS * (c - 1);
			const gptr_out gout = gptr_out_of((unsigned long long) a.out);
			const int row_bytes = a.width * B;
#pragma nounroll // (unrolled, the 4 B rows / columns / pointers are loop invariants that get spilled)
			for (int i = 0; i < 4 * B; i++) {
				const int idx = t + CM_NT * i;
				const int row = idx / (32 * B), col = idx - row * (32 * B);
				const int y = y0 + row;
				const int xb = X0 * B + 4 * col; // byte of the row
				if (y < Yb && xb < row_bytes) {
					const unsigned int w = lds_out[row * a.out_pitch + col];
					const gptr_out p = gout + (long long) y * a.out_stride + xb;
					if (xb + 4 <= row_bytes)
						gstore32(p, w);
					else
						for (int e = 0; e < row_bytes - xb; e++)
							gstore8(p + e, (unsigned char) (w >> (8 * e)));
				}
			}
		}
	}
	wait_vmem0();
	barrier(); // (the next item of a persistent block stages into the buffers at once)
}

} // namespace vh
