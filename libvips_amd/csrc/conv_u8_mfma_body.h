// Both passes of vips_convsep / vips_gaussblur (precision integer, convsep.c:61-118, convi.c:698-716) on
// uchar images ON THE MATRIX CORES: a separable convolution is a banded (Toeplitz) matrix times the image,
// twice.  north_star: "MFMA used only where a large conv re-cast as im2col x GEMM actually wins" -- a 29-tap
// gaussian is 58 multiply-adds per output byte; as v_dot4 on packed bytes (conv_u8_body.h) that is ~25 vector
// instructions per output byte at half rate (0.4 - 0.75 ms for 8192^2 x 3, 6 - 13 % of HBM); as
// v_mfma_f32_32x32x16_f16 it is 8 matrix instructions per 1024 output bytes.
//
// Exact integers in f32, as in reduce_u8.hip: a pixel byte p is the f16 DENORMAL 0x00pp = p 2^-24 (no
// conversion: a v_perm puts a zero byte above it), a coefficient |c| < 2048 is an exact half, every product
// and every partial sum is an integer below 2^24 times 2^-24.  Rounding as in conv_u8_body.h: clip(floor((S +
// scale / 2) / scale)) = RNE(S RN(1 / scale) + bias) saturated to 0 .. 255, bias = -0.5 + 1 / (2 scale) + (scale / 2)
// RN(1 / scale) -- one v_fma_f32 (by 2^24 RN(1 / scale): the same real product) and one v_cvt_pk_u8_f32; the host
// walks every sum that does not saturate (S + scale / 2 < 257 scale) before it enables the kernel for a scale.
//
// A block of 4 waves owns 128 output columns and streams down a segment of rows in chunks of 32:
//   stage    the chunk's 32 input rows x (128 + 2 hp) columns (hp = half rounded up to 4: row starts are
//            dwords) go from global memory straight into LDS (global_load_lds_dword: no register, no wait
//            until the chunk is needed; the next chunk travels while this one is computed), rows and
//            rows outside the image clamped to its edge (vips_embed COPY); columns outside it are not made at
//            all: their taps are folded into the edge column's coefficient (Th below);
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
//            neighbouring columns per register quad: bytes, bands interleaved, 4 B bytes to the wave's own
//            32 x 32-pixel tile in LDS;
//   store    the wave copies its tile out, 8 bytes per lane, whole tile rows per instruction (no barrier: one
//            barrier per chunk, for the staged rows, is all the block shares).
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
	int mh;              // 0: separable (both passes); else the rows of a two-dimensional mask (one pass)
	int ksteps;          // 16-column steps of the window that hold a tap: 3 or 4
	int stage_rows;      // rows staged per chunk: 32, + mh - 1 for a two-dimensional mask
	int row_lead;        // rows between a chunk's first staged row and its first output row: hp, or mh / 2
	int e_dw;            // dwords between the staged row's first byte (a multiple of the staging unit) and column X0 - hp
	int in_dw;           // dwords of a staged row that hold pixels: e_dw + (128 + 2 hp) B / 4, in whole units
	int in_pitch;        // ... and its pitch in LDS (a whole number of units; an odd number of them, or of dword pairs)
	int in_buf;          // dwords per staging buffer: 32 in_pitch, + the last instruction's overrun
	float k1, bias;      // 2^24 RN(1 / scale); -0.5 + 1 / (2 scale) + (scale / 2) RN(1 / scale): checked on the host
	int rnd;             // ushort: (S + rnd) / scale as ((S + rnd) * div_m) >> (32 + div_s), S + rnd < 2^31 (div_m = 0: scale 1)
	unsigned int div_m;
	int div_s;
	int fin64;           // ushort: the quotient in doubles instead (cm_fin16)
	double div_inv;      // RN(1 / scale)
	double rnd_half;     // rnd + 0.5
	const unsigned int *tz;      // the Toeplitz operands: 4 tables of [4 k-steps][64 lanes][4 dwords]: the mask's, and
	int edge_wave[3];            // pass 1's for the tiles 32 edge_wave[k] .. + 32 whose windows hang over an edge (-1: none)
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

// ushort (conv_u8_mfma_item<B, .., U16 = true>: B = 2 x bands BYTE planes, plane 2 b the low bytes of band b): the two
// exact sums of a sample -- of the low and of the high bytes, each n 2^-24 -- to clip((256 S_hi + S_lo + rnd) / scale,
// 0, 65535) in 32-bit integers (convi.c:698-716 for unsigned short: int sums, C division -- a negative numerator
// gives a quotient <= 0, clipped to 0 either way); the host keeps 256 S_hi + S_lo + rnd below 2^31
// (fin64: the same quotient in doubles -- S + rnd + 0.5 exactly (both sums are integers below 2^24 in units of
// 2^-24, so 2^32 hi + 2^24 lo + rnd + 0.5 is a half-integer below 2^33), times RN(1 / scale): (S + rnd + 0.5) / scale
// lies at least 0.5 / scale from an integer on either side and the two roundings move the product by less than
// 2^31 2^-52 / scale, so its floor is the quotient's; a negative numerator converts to 0.  Seven instructions, none at
// a quarter rate, against ~15 issue slots)
VH_DEV unsigned int cm_fin16(float lo, float hi, const CmArgs &a)
{
	if (a.fin64) {
		const double t = __builtin_fma((double) hi, 4294967296.0, a.rnd_half);
		const double u = __builtin_fma((double) lo, 16777216.0, t);
		return min(vh::cvt_u32(u * a.div_inv), 65535u);
	}
	const int s = (vh::cvt_i32(hi * 16777216.0f) << 8) + vh::cvt_i32(lo * 16777216.0f) + a.rnd;
	if (s <= 0)
		return 0u;
	const unsigned int q = a.div_m ? (unsigned int) (((unsigned long long) (unsigned int) s * a.div_m) >> 32) >> a.div_s : (unsigned int) s;
	return min(q, 65535u);
}

// 4 pixels x NB bands of ushort samples given as byte planes (lo[b], hi[b]: the 4 pixels' low / high bytes of band
// b) -> the 2 NB dwords as they lie in memory (sample (p, b) = halfword p NB + b)
template <int NB>
VH_DEV void cm_interleave16(const unsigned int (&lo)[NB], const unsigned int (&hi)[NB], unsigned int (&w)[2 * NB])
{
	// pl[b][h]: pixels 2 h, 2 h + 1 of band b as two ushorts
	unsigned int pl[NB][2];
#pragma unroll
	for (int b = 0; b < NB; b++) {
		pl[b][0] = perm(hi[b], lo[b], 0x05010400u);
		pl[b][1] = perm(hi[b], lo[b], 0x07030602u);
	}
#pragma unroll
	for (int d = 0; d < 2 * NB; d++) {
		const int h0 = 2 * d, h1 = 2 * d + 1;
		const int p0 = h0 / NB, b0 = h0 % NB, p1 = h1 / NB, b1 = h1 % NB;
		w[d] = perm(pl[b1][p1 >> 1], pl[b0][p0 >> 1],
			((4u + 2u * (unsigned int) (p1 & 1) + 1u) << 24) | ((4u + 2u * (unsigned int) (p1 & 1)) << 16) |
				((2u * (unsigned int) (p0 & 1) + 1u) << 8) | (2u * (unsigned int) (p0 & 1)));
	}
}

// one work item: strip x segment
// MODE: 0 separable; 3 / 5: a two-dimensional mask of that many rows whose window takes 3 steps, its operands in
// registers for the whole item; -1: any other two-dimensional mask (operands fetched as they are used)
template <int B, bool WIDE, int MODE, bool U16 = false>
VH_DEV void conv_u8_mfma_item(const CmArgs &a, int item, unsigned int *lds)
{
	constexpr bool TWOD = MODE != 0;
	static_assert(!U16 || (MODE == 0 && B % 2 == 0), "ushort: the separable form, B byte planes");
	const int t = tid(), lane = t & 63, wv = wave_index(), n = lane & 31, hf = lane >> 5;
	const int strip = item % a.strips, seg = item / a.strips;
	const int X0 = strip * CM_BW;
	const int sb = (X0 - a.hp) * B - 4 * a.e_dw; // first staged byte of a row: a multiple of the staging unit
	const int Ya = seg * a.seg_rows, Yb = min(Ya + a.seg_rows, a.height);
	const int nchunks = (Yb - Ya + CM_ROWS - 1) / CM_ROWS + (TWOD ? 0 : 1);
	const int SR = a.stage_rows;
	const gptr_in gin = gptr_in_of((unsigned long long) a.in);
	const gptr_out gout = gptr_out_of((unsigned long long) a.out);

	// the Toeplitz operand of both passes, and pass 1's where the wave's window hangs over the left or right edge
	// of the image: there every column outside is the edge column (vips_embed COPY), so its taps are ADDED to
	// the edge column's and it gets zero -- no pixel is clamped, whatever the staging put there counts for nothing
	unsigned int T[4][4], Th[4][4];
	int which = 0; // which of the 4 operand sets the wave's tile across the image takes
	{
		const int W = 4 * strip + wv;
#pragma unroll
		for (int k = 0; k < 3; k++)
			which = a.edge_wave[k] == W ? k + 1 : which;
		if constexpr (!TWOD) {
#pragma unroll
			for (int s = 0; s < 4; s++) {
				gload128(gptr_in_of((unsigned long long) a.tz), (unsigned int) ((s * 64 + lane) * 16), T[s]);
				gload128(gptr_in_of((unsigned long long) a.tz), (unsigned int) (((4 * which + s) * 64 + lane) * 16), Th[s]);
			}
		}
	}

	// the wave's own output tile in LDS (32 rows x 8 B dwords, pitch odd) and how its lanes copy it out:
	// 4 B lanes x 8 bytes make a row of the tile, 64 / (4 B) rows per store instruction
	// (ushort: the tile leaves in two halves of 16 rows -- half the LDS, and two blocks a CU at 3 bands)
	constexpr int OP = 8 * B + 1, UNITS = 4 * B, RPI = 64 / UNITS, TR = U16 ? 16 : CM_ROWS;
	unsigned int *otile = lds + 2 * a.in_buf + wv * (TR * OP);
	const int o_rsub = lane / UNITS, o_c2 = lane - o_rsub * UNITS;
	const int o_xb = (X0 + 32 * wv) * B + 8 * o_c2; // byte of the output row
	const int row_bytes = a.width * B;
	const unsigned int o_voff = (unsigned int) (o_rsub * (int) a.out_stride + o_xb);

	// chunk c: mid rows Ya - hp + 32 c .. + 32 = the same input rows, clamped.  The tile (32 rows x in_pitch
	// dwords) is a line of units of U = 16 bytes (4 when base or stride are not multiples of 16): unit u =
	// (row u / upr, column unit u % upr); a wave's instruction moves 64 consecutive units, lane by lane, so the
	// LDS side is linear and the row / column of a lane's unit is two additions and a wrap per instruction
	constexpr int U = WIDE ? 16 : 4;
	const int upr = a.in_pitch / (U / 4), uvalid = a.in_dw / (U / 4);
	const int row_units_last = ((a.width * B) / U - 1) * U; // byte of the last whole unit of an image row
	// (a chunk whose 32 rows lie inside the image: a lane's offsets are the same for every such chunk)
	constexpr int MAXI = 6; // instructions per wave and chunk the offsets are kept for
	const int ninstr = (SR * upr + 255) / 256;
	unsigned int voff_in[MAXI];
	{
		int u = 64 * wv + lane;
		int row = u / upr, cu = u - row * upr;
		const int step_row = 256 / upr, step_cu = 256 - step_row * upr;
#pragma unroll
		for (int i = 0; i < MAXI; i++) {
			const int rr = min(row, SR - 1), cc = min(cu, uvalid - 1);
			voff_in[i] = (unsigned int) (rr * (int) a.in_stride + min(max(sb + U * cc, 0), row_units_last));
			cu += step_cu;
			row += step_row;
			if (cu >= upr) {
				cu -= upr;
				row++;
			}
		}
	}
	auto stage = [&](int c) {
		unsigned int *dst = lds + (c & 1) * a.in_buf;
		const int r0 = Ya - a.row_lead + CM_ROWS * c;
		if (r0 >= 0 && r0 + SR <= a.height && ninstr <= MAXI) {
			const gptr_in base = gin + (long long) r0 * a.in_stride;
#pragma unroll
			for (int i = 0; i < MAXI; i++)
				if (i < ninstr && 64 * (wv + 4 * i) < SR * upr) {
					if constexpr (WIDE)
						lds_dma_x4(base, voff_in[i], dst + 256 * (wv + 4 * i));
					else
						lds_dma_dword(base, voff_in[i], dst + 64 * (wv + 4 * i));
				}
			return;
		}
		const int r0c = min(max(r0, 0), a.height - 1);
		// (the uniform base carries the row and the strip: lane offsets stay below 33 strides)
		const gptr_in base = gin + (long long) r0c * a.in_stride;
		int u = 64 * wv + lane;
		int row = u / upr, cu = u - row * upr;
		const int step_row = 256 / upr, step_cu = 256 - step_row * upr;
		for (int j = wv; 64 * j < SR * upr; j += 4) {
			const int rr = min(row, SR - 1), cc = min(cu, uvalid - 1); // (padding, the tail: any valid unit)
			const int rc = min(max(r0 + rr, 0), a.height - 1);
			// (a strip over the left or right edge of the image: the unit's address clamped into the row -- what
			// lands in columns outside the image meets a zero coefficient, see Th)
			const int ub = min(max(sb + U * cc, 0), row_units_last);
			const unsigned int voff = (unsigned int) ((rc - r0c) * (int) a.in_stride + ub);
			if constexpr (WIDE)
				lds_dma_x4(base, voff, dst + 256 * j);
			else
				lds_dma_dword(base, voff, dst + 64 * j);
			cu += step_cu;
			row += step_row;
			if (cu >= upr) {
				cu -= upr;
				row++;
			}
		}
	};

	// a small two-dimensional mask: its operands for the whole item
	unsigned int T2[MODE > 0 ? MODE : 1][3][4];
	if constexpr (MODE > 0) {
#pragma unroll
		for (int i = 0; i < MODE; i++)
#pragma unroll
			for (int s = 0; s < 3; s++)
				gload128(gptr_in_of((unsigned long long) a.tz), (unsigned int) ((((which * MODE + i) * 4 + s) * 64 + lane) * 16), T2[i][s]);
	}

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
		unsigned int P[B][4]; // [band][quad of columns 8 j + 4 hf .. + 3 of row n]
		if constexpr (TWOD) {
			// ---- a two-dimensional mask: one product per mask row i, rows n + i of the staged chunk, all into
			// the same accumulators; the operand roles of pass 2 (lane & 31 = the output ROW)
			float acc[B][16];
#pragma unroll
			for (int b = 0; b < B; b++)
#pragma unroll
				for (int r = 0; r < 16; r++)
					acc[b][r] = 0.0f;
			auto step = [&](const unsigned int *src, int s, const unsigned int (&Ti)[4]) {
				unsigned int raw[2][B];
#pragma unroll
				for (int g = 0; g < 2; g++)
#pragma unroll
					for (int k = 0; k < B; k++)
						raw[g][k] = src[B * (4 * s + 2 * g) + k];
#pragma unroll
				for (int b = 0; b < B; b++) {
					unsigned int A[4];
					cm_halves<B>(raw[0], b, A[0], A[1]);
					cm_halves<B>(raw[1], b, A[2], A[3]);
					mfma_32x32x16_f16(Ti, A, acc[b]);
				}
			};
			if constexpr (MODE > 0) {
#pragma unroll
				for (int i = 0; i < MODE; i++) {
					const unsigned int *src = lds + (c & 1) * a.in_buf + (n + i) * a.in_pitch + a.e_dw + B * (8 * wv + hf);
#pragma unroll
					for (int s = 0; s < 3; s++)
						step(src, s, T2[i][s]);
				}
			}
			else {
				// (the next step's operand travels while this step's products are made)
				unsigned int Tn[4];
				gload128(gptr_in_of((unsigned long long) a.tz), (unsigned int) ((((which * a.mh) * 4) * 64 + lane) * 16), Tn);
				for (int i = 0; i < a.mh; i++) {
					const unsigned int *src = lds + (c & 1) * a.in_buf + (n + i) * a.in_pitch + a.e_dw + B * (8 * wv + hf);
					for (int s = 0; s < a.ksteps; s++) {
						unsigned int Ti[4];
#pragma unroll
						for (int k = 0; k < 4; k++)
							Ti[k] = Tn[k];
						int in = i, sn = s + 1;
						if (sn == a.ksteps) {
							sn = 0;
							in = i + 1 < a.mh ? i + 1 : i;
						}
						gload128(gptr_in_of((unsigned long long) a.tz), (unsigned int) ((((which * a.mh + in) * 4 + sn) * 64 + lane) * 16), Tn);
						step(src, s, Ti);
					}
				}
			}
#pragma unroll
			for (int b = 0; b < B; b++)
#pragma unroll
				for (int j = 0; j < 4; j++) {
					unsigned int w = 0;
#pragma unroll
					for (int k = 0; k < 4; k++)
						w = cvt_pk_u8(__builtin_fmaf(acc[b][4 * j + k], a.k1, a.bias), (unsigned int) k, w);
					P[b][j] = w;
				}
		}
		else {
		// the lane's pixels of row n: per 16 window columns the groups 4 hf .. + 3 and 8 + 4 hf .. + 3
		const unsigned int *src = lds + (c & 1) * a.in_buf + n * a.in_pitch + a.e_dw + B * (8 * wv + hf);
		unsigned int raw[4][2][B];
#pragma unroll
		for (int s = 0; s < 4; s++)
#pragma unroll
			for (int g = 0; g < 2; g++)
#pragma unroll
				for (int i = 0; i < B; i++)
					raw[s][g][i] = src[B * (4 * s + 2 * g) + i];
		if constexpr (U16) {
#pragma unroll
		for (int b = 0; b < B / 2; b++) {
			// ---- pass 1: the low and the high bytes of band b, each its own exact product
			float acc[2][16];
#pragma unroll
			for (int part = 0; part < 2; part++)
#pragma unroll
				for (int s = 0; s < 4; s++) {
					if (s == 3 && a.ksteps < 4)
						continue;
					unsigned int A[4];
					cm_halves<B>(raw[s][0], 2 * b + part, A[0], A[1]);
					cm_halves<B>(raw[s][1], 2 * b + part, A[2], A[3]);
					if (s == 0)
						mfma_32x32x16_f16_first(A, Th[s], acc[part]);
					else
						mfma_32x32x16_f16(A, Th[s], acc[part]);
				}
			// rows (r & 3) + 8 (r >> 2) + 4 hf of column n as ushorts: their bytes are pass 2's two operands
			unsigned int mid_cur[2][8];
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const unsigned int v0 = cm_fin16(acc[0][2 * q], acc[1][2 * q], a), v1 = cm_fin16(acc[0][2 * q + 1], acc[1][2 * q + 1], a);
				mid_cur[0][q] = (v0 & 255u) | ((v1 & 255u) << 16);
				mid_cur[1][q] = (v0 >> 8) | ((v1 >> 8) << 16);
			}
			// ---- pass 2: output rows Ya + 32 (c - 1) .. + 32
			if (c >= 1) {
#pragma unroll
				for (int part = 0; part < 2; part++) {
					unsigned int A[4];
#pragma unroll
					for (int s = 0; s < 2; s++) {
#pragma unroll
						for (int q = 0; q < 4; q++)
							A[q] = mid_prev[2 * b + part][4 * s + q];
						if (s == 0)
							mfma_32x32x16_f16_first(A, T[s], acc[part]);
						else
							mfma_32x32x16_f16(A, T[s], acc[part]);
					}
#pragma unroll
					for (int s = 0; s < 2; s++) {
						if (s == 1 && a.ksteps < 4)
							continue;
#pragma unroll
						for (int q = 0; q < 4; q++)
							A[q] = mid_cur[part][4 * s + q];
						mfma_32x32x16_f16(A, T[2 + s], acc[part]);
					}
				}
#pragma unroll
				for (int j = 0; j < 4; j++) {
					unsigned int wl = 0, wh = 0;
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const unsigned int v = cm_fin16(acc[0][4 * j + i], acc[1][4 * j + i], a);
						wl |= (v & 255u) << (8 * i);
						wh |= (v >> 8) << (8 * i);
					}
					P[2 * b][j] = wl;
					P[2 * b + 1][j] = wh;
				}
			}
#pragma unroll
			for (int part = 0; part < 2; part++)
#pragma unroll
				for (int q = 0; q < 8; q++)
					mid_prev[2 * b + part][q] = mid_cur[part][q];
			sched_fence(); // (one band's accumulators at a time)
		}
		}
		else {
#pragma unroll
		for (int b = 0; b < B; b++) {
			// ---- pass 1 (the accumulators start at 0: the rounding constant is in a.bias)
			float acc[16];
#pragma unroll
			for (int s = 0; s < 4; s++) {
				// (a short mask's window ends before the last 16-column step: its operand is all zeros)
				if (s == 3 && a.ksteps < 4)
					continue;
				unsigned int A[4];
				cm_halves<B>(raw[s][0], b, A[0], A[1]);
				cm_halves<B>(raw[s][1], b, A[2], A[3]);
				if (s == 0)
					mfma_32x32x16_f16_first(A, Th[s], acc);
				else
					mfma_32x32x16_f16(A, Th[s], acc);
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
				unsigned int A[4];
#pragma unroll
				for (int s = 0; s < 2; s++) {
#pragma unroll
					for (int q = 0; q < 4; q++)
						A[q] = mid_prev[b][4 * s + q];
					if (s == 0)
						mfma_32x32x16_f16_first(A, T[s], acc);
					else
						mfma_32x32x16_f16(A, T[s], acc);
				}
#pragma unroll
				for (int s = 0; s < 2; s++) {
					if (s == 1 && a.ksteps < 4)
						continue; // (the same for the rows: nothing of this chunk's second half is a tap yet)
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
			sched_fence(); // (one band's accumulators at a time)
		}
		}
		}
		if (TWOD || c >= 1) {
#pragma unroll
			for (int th = 0; th < CM_ROWS / TR; th++) {
			// row n, columns 8 j + 4 hf .. + 3 of the wave's tile: bands interleaved, B dwords
			if (TR == CM_ROWS || (n >> 4) == th) {
				unsigned int *orow = otile + (n & (TR - 1)) * OP + B * hf;
#pragma unroll
				for (int j = 0; j < 4; j++) {
					unsigned int Pj[B], w[B];
#pragma unroll
					for (int b = 0; b < B; b++)
						Pj[b] = P[b][j];
					if constexpr (U16) {
						unsigned int lo[B / 2], hi[B / 2];
#pragma unroll
						for (int b = 0; b < B / 2; b++) {
							lo[b] = Pj[2 * b];
							hi[b] = Pj[2 * b + 1];
						}
						cm_interleave16<B / 2>(lo, hi, w);
					}
					else
						cu8_interleave<B>(Pj, w);
#pragma unroll
					for (int b = 0; b < B; b++)
						orow[2 * B * j + b] = w[b];
				}
			}
			wave_lds_fence(); // (the tile is the wave's own: no barrier)
			const int y0 = Ya + CM_ROWS * (TWOD ? c : c - 1) + TR * th;
			if (y0 + TR <= Yb && (X0 + 32 * wv + 32) * B <= row_bytes) {
				// the whole tile lies inside the image: no lane tests anything but its row of the last instruction
				const gptr_out tile_out = gout + (long long) y0 * a.out_stride;
#pragma unroll
				for (int i = 0; i < (TR + RPI - 1) / RPI; i++) {
					const int row = RPI * i + o_rsub;
					if (o_rsub < RPI && (RPI * (i + 1) <= TR || row < TR)) {
						unsigned int w[2];
						w[0] = otile[row * OP + 2 * o_c2];
						w[1] = otile[row * OP + 2 * o_c2 + 1];
						gstore_dwords<2>(tile_out + (unsigned int) (RPI * i * (int) a.out_stride) + o_voff, w);
					}
				}
			}
			else
#pragma nounroll
			for (int i = 0; i < (TR + RPI - 1) / RPI; i++) {
				const int row = RPI * i + o_rsub;
				const int y = y0 + row;
				if (o_rsub < RPI && row < TR && y < Yb && o_xb < row_bytes) {
					unsigned int w[2];
					w[0] = otile[row * OP + 2 * o_c2];
					w[1] = otile[row * OP + 2 * o_c2 + 1];
					const gptr_out p = gout + (long long) y * a.out_stride + o_xb;
					if (o_xb + 8 <= row_bytes)
						gstore_dwords<2>(p, w);
					else
						for (int e = 0; e < row_bytes - o_xb; e++)
							gstore8(p + e, (unsigned char) (w[e >> 2] >> (8 * (e & 3))));
				}
			}
			wave_lds_fence(); // (the reads are done before the next rows overwrite the tile)
			}
		}
	}
	wait_vmem0();
	barrier(); // (the next item of a persistent block stages into the buffers at once)
}

} // namespace vh
