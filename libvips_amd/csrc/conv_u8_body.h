// Integer convolution of uchar images on packed bytes (v_dot4_i32_i8), streaming: the bodies of
//   * vips_convsep / vips_gaussblur with precision=integer -- both passes in one kernel, the
//     intermediate image rounded to uchar exactly as the first vips_convi writes it
//     (convsep.c:61-118, convi.c:698-716), and
//   * vips_conv with precision=integer and a small two-dimensional mask (convi.c:753-857, C path)
// written against gcn.h (the product) / tests/emul/gcn.h (host fibers, CPU suite).
//
// A WAVE owns a strip of 256 pixel columns (of which `pd` lanes = 4 pd columns on each side are
// halo) and streams down a segment of rows; nothing is shared between waves, so there is no LDS
// staging and no barrier in the row loop (a first build staged rows in LDS for the whole block: 12
// waves per CU fitted, each waiting on a barrier per 4 rows -- 2.4 x slower than its instruction
// count).  A lane owns 4 pixels of every row: it loads their `B` dwords, turns them into B planar
// dwords (4 pixels of one band each, v_perm) and biases them to signed bytes (p ^ 0x80 = p - 128).
// Everything is then dot products of 4 bytes:
//
//   horizontal  outputs x0 .. x0 + 3 of a band need the planar bytes x0 - h .. x0 + 3 + h: the
//               lane's own dword and those of its neighbours (v_mov_dpp wave_shr / wave_shl for
//               lane +- 1, ds_bpermute further out), ND dwords; output x0 + a is the sum over them of
//               v_dot4_i32_i8(window[j], cvec[a][j]) -- the mask, shifted by a bytes and cut into
//               dwords on the host, 4 x ND coefficient dwords in scalar registers.  No byte is
//               extracted, no data is shifted: at most 4 ND dot4 per 4 outputs, i.e. (n + 3) / 4 + 1
//               instructions per output for n taps (coefficient dwords that hold no tap of a short
//               mask are compiled out: template H, a 3-wide mask takes 6 of the 12).
//   vertical    (separable) the 4 rows a thread has just made -- one dword per row and band --
//               are transposed (8 v_perm per band) into one dword per COLUMN holding the column's
//               4 rows, and kept in a ring of ND quads that is private to the lane (it owns the
//               same 4 columns in every row): in registers for ND = 3 (the quad loop is unrolled
//               ND times, the rotation is static), in LDS for longer masks.  An output quad is then
//               the same 4 ND dot4 per column with the SAME coefficient dwords.
//   2-D masks   the window dwords of the mask's MH rows rotate through registers (the row loop is
//               unrolled MH times); an output row is the sum over them of the horizontal form.
//
// Rounding: the reference computes clip((sum + scale / 2) / scale + offset) with C division.
// With offset = 0 a negative numerator clips to 0 whatever its quotient, and for x >= 0 the
// quotient floor(x / scale) is RN(x * RN(1 / scale) - 0.5 + 1 / (2 scale)): the float error is
// below 3.1e-5 for quotients up to 256 (x < 2^24 is exact in float) and the nearest wrong integer
// is 1 / (2 scale) away -- the host takes this path for scale <= 8000 only.  That is
// v_cvt_f32_i32, v_fma_f32, v_cvt_pk_u8_f32 (round to nearest, saturate to 0 .. 255, and the
// byte lands in place): three instructions per output byte, the accumulators start at
// scale / 2 + 128 * sum(mask) (the bias of the signed bytes).
#pragma once

#include "gcn.h"

namespace vh {

constexpr int CU8_MAXND = 9;    // window dwords: masks up to 33 taps wide
constexpr int CU8_MAXMH = 7;    // rows of a 2-D mask
constexpr int CU8_CVEC = 96;    // coefficient dwords: 4 ND (separable) or MH 4 ND (2-D, ND = 3)
constexpr int CU8_NT = 256;     // threads per block: 4 independent waves


struct Cu8Args {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int width, height;
	int half, pd;    // taps / 2 of the horizontal window; halo lanes either side of a wave: ceil(half / 4)
	int vhalf, hq;   // taps / 2 of the vertical window; halo quads ceil(vhalf / 4)
	int wout;        // output columns per wave: 4 (64 - 2 pd)
	int strips;      // blocks across the image: 4 waves each
	int segs, seg_rows; // segments down the image; output rows per segment (a multiple of 4)
	int off_ring;    // byte offset of the lanes' private rings in LDS (separable, ND > 3)
	int off_slot;    // ... of the work-item slot
	int *counter;    // work items are dealt from here (zeroed before the launch)
	int acc0;        // accumulators start at scale / 2 + 128 * sum(mask)
	float rscale, bias; // RN(1 / scale), -0.5 + 1 / (2 scale)
	unsigned int cvec[CU8_CVEC];
};

// B interleaved dwords (4 pixels) -> B planar dwords (byte m of P[b] = band b of pixel m)
template <int B>
VH_DEV void cu8_planar(const unsigned int (&w)[B], unsigned int (&P)[B])
{
	if constexpr (B == 1)
		P[0] = w[0];
	else {
#pragma unroll
		for (int b = 0; b < B; b++) {
			const int i0 = b, i1 = B + b, i2 = 2 * B + b, i3 = 3 * B + b;
			const unsigned int lo = perm(w[i1 >> 2], w[i0 >> 2], 0x0c0c0000u | ((4u + (i1 & 3)) << 8) | (unsigned int) (i0 & 3));
			const unsigned int hi = perm(w[i3 >> 2], w[i2 >> 2], 0x0c0c0000u | ((4u + (i3 & 3)) << 8) | (unsigned int) (i2 & 3));
			P[b] = perm(hi, lo, 0x05040100u);
		}
	}
}

// ... and back
template <int B>
VH_DEV void cu8_interleave(const unsigned int (&P)[B], unsigned int (&w)[B])
{
	if constexpr (B == 1)
		w[0] = P[0];
	else {
#pragma unroll
		for (int d = 0; d < B; d++) {
			// byte e = 4 d + k of the group is band e % B of pixel e / B
			const int e0 = 4 * d, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3;
			const unsigned int lo = perm(P[e1 % B], P[e0 % B], 0x0c0c0000u | ((4u + (e1 / B)) << 8) | (unsigned int) (e0 / B));
			const unsigned int hi = perm(P[e3 % B], P[e2 % B], 0x0c0c0000u | ((4u + (e3 / B)) << 8) | (unsigned int) (e2 / B));
			w[d] = perm(hi, lo, 0x05040100u);
		}
	}
}

// four sums -> one dword of four rounded, clipped bytes
VH_DEV unsigned int cu8_round4(const int (&s)[4], const Cu8Args &a)
{
	unsigned int r = 0;
#pragma unroll
	for (int k = 0; k < 4; k++)
		r = cvt_pk_u8(__builtin_fmaf((float) s[k], a.rscale, a.bias), (unsigned int) k, r);
	return r;
}

// the 4 pixels at px0 .. px0 + 3 of row `row` (both clamped into the image: vips_embed COPY) as B dwords
template <int B>
VH_DEV void cu8_load(const Cu8Args &a, int row, int px0, unsigned int (&w)[B])
{
	const int rc = min(max(row, 0), a.height - 1);
	const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) rc * a.in_stride;
	if (px0 >= 0 && px0 + 4 <= a.width)
		gload_dwords<B>(line, (unsigned int) (px0 * B), w); // (one load of B dwords, not B loads 4 B apart)
	else {
#pragma unroll
		for (int d = 0; d < B; d++) {
			unsigned int v = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int e = 4 * d + k;
				const int px = min(max(px0 + e / B, 0), a.width - 1);
				v |= (unsigned int) gload8(line, (unsigned int) (px * B + e % B)) << (8 * k);
			}
			w[d] = v;
		}
	}
}

// the 4 pixels at px0 of output row `row`: whole dwords inside the strip's share of the image, bytes at its ragged end
template <int B>
VH_DEV void cu8_store(const Cu8Args &a, int row, int px0, int px_end, const unsigned int (&w)[B])
{
	const gptr_out line = gptr_out_of((unsigned long long) a.out) + (long long) row * a.out_stride;
	if (px0 + 4 <= px_end)
		gstore_dwords<B>(line + px0 * B, w);
	else {
#pragma unroll
		for (int d = 0; d < B; d++)
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int e = 4 * d + k;
				if (px0 + e / B < px_end)
					gstore8(line + (px0 * B + e), (unsigned char) (w[d] >> (8 * k)));
			}
	}
}

// does the coefficient dword of output k, window dword j hold any tap of a mask of half-width H?
// (tap = 4 (j - PD) + byte - k + H must lie in 0 .. 2 H; H < 0: unknown, keep all)
constexpr bool cu8_live(int H, int PD, int k, int j)
{
	if (H < 0)
		return true;
	for (int bl = 0; bl < 4; bl++) {
		const int tap = 4 * (j - PD) + bl - k + H;
		if (tap >= 0 && tap <= 2 * H)
			return true;
	}
	return false;
}

// the window of a planar dword: own[PD] = the lane's, the rest from its neighbours
template <int ND>
VH_DEV void cu8_window(unsigned int own, unsigned int (&w)[ND])
{
	constexpr int PD = ND / 2;
#pragma unroll
	for (int j = 0; j < ND; j++) {
		if (j == PD)
			w[j] = own;
		else if (j == PD - 1)
			w[j] = lane_prev(own);
		else if (j == PD + 1)
			w[j] = lane_next(own);
		else
			w[j] = lane_from(own, j - PD);
	}
}

// ---- separable: both passes.  One work item = a block's 4 wave strips x one segment of rows.
// RG: the lanes' rings of transposed quads in registers (ND quads x B bands x 4 dwords: always for
// ND = 3; for longer masks it trades registers -- 192 for 3 bands and ND = 9, two waves per SIMD --
// for the 110 KB of LDS that allowed ONE block of four waves per CU)
template <int B, int ND, int H, bool RG = true>
static __device__ __forceinline__ void conv_u8_sep_body(const Cu8Args &a, int item, unsigned int *lds)
{
	constexpr bool REGS = ND == 3 || RG;
	const int t = tid(), lane = t & 63;
	const int strip = item % a.strips, seg = item / a.strips;
	const int X0 = (strip * (CU8_NT / 64) + (t >> 6)) * a.wout;  // the wave's first output column
	const int px0 = X0 - 4 * a.pd + 4 * lane;                   // this lane's 4 columns
	const int px_end = min(X0 + a.wout, a.width);
	const bool writer = lane >= a.pd && lane < 64 - a.pd && px0 < px_end;
	const int Qa = (seg * a.seg_rows) >> 2, Qb = min(Qa + (a.seg_rows >> 2), (a.height + 3) >> 2);
	uint4 *RING = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(lds) + a.off_ring); // [ND][B][CU8_NT]
	uint4 ring[REGS ? ND : 1][REGS ? B : 1];

	// two quads travelling per lane
	unsigned int cur[4][B], nxt[4][B];
#pragma unroll
	for (int r = 0; r < 4; r++) {
		cu8_load<B>(a, 4 * (Qa - a.hq) + r, px0, cur[r]);
		cu8_load<B>(a, 4 * (Qa - a.hq + 1) + r, px0, nxt[r]);
	}
	for (int base = Qa - a.hq; base < Qb + a.hq; base += ND) {
#pragma unroll
		for (int s = 0; s < ND; s++) {
			const int qi = base + s; // its ring slot: s
			if (qi < Qb + a.hq) {
				// the quad's rows, planar and signed; the next quad starts travelling
				unsigned int P[4][B];
#pragma unroll
				for (int r = 0; r < 4; r++) {
					cu8_planar<B>(cur[r], P[r]);
#pragma unroll
					for (int b = 0; b < B; b++)
						P[r][b] ^= 0x80808080u;
				}
#pragma unroll
				for (int r = 0; r < 4; r++)
#pragma unroll
					for (int b = 0; b < B; b++)
						cur[r][b] = nxt[r][b];
				if (qi + 2 < Qb + a.hq) {
#pragma unroll
					for (int r = 0; r < 4; r++)
						cu8_load<B>(a, 4 * (qi + 2) + r, px0, nxt[r]);
				}
				// horizontal pass, transposed into the ring
#pragma unroll
				for (int b = 0; b < B; b++) {
					unsigned int hrow[4];
#pragma unroll
					for (int r = 0; r < 4; r++) {
						unsigned int w[ND];
						cu8_window<ND>(P[r][b], w);
						int sum[4];
#pragma unroll
						for (int k = 0; k < 4; k++) {
							sum[k] = a.acc0;
#pragma unroll
							for (int j = 0; j < ND; j++)
								if (cu8_live(H, ND / 2, k, j))
									sum[k] = dot4(w[j], a.cvec[k * ND + j], sum[k]);
						}
						hrow[r] = cu8_round4(sum, a) ^ 0x80808080u;
					}
					// 4 rows x 4 columns -> 4 columns x 4 rows
					const unsigned int t01l = perm(hrow[1], hrow[0], 0x05010400u), t01h = perm(hrow[1], hrow[0], 0x07030602u);
					const unsigned int t23l = perm(hrow[3], hrow[2], 0x05010400u), t23h = perm(hrow[3], hrow[2], 0x07030602u);
					uint4 col;
					col.x = perm(t23l, t01l, 0x05040100u);
					col.y = perm(t23l, t01l, 0x07060302u);
					col.z = perm(t23h, t01h, 0x05040100u);
					col.w = perm(t23h, t01h, 0x07060302u);
					if constexpr (REGS)
						ring[s][b] = col;
					else
						RING[(s * B + b) * CU8_NT + t] = col;
				}
				// vertical pass of output quad qi - hq: ring slots s + 1 .. s + ND (mod ND), oldest first
				const int Qo = qi - a.hq;
				if (Qo >= Qa && writer) {
					unsigned int orow[4][B];
#pragma unroll
					for (int b = 0; b < B; b++) {
						int sum[4][4]; // [output row][column]
#pragma unroll
						for (int k = 0; k < 4; k++)
#pragma unroll
							for (int c = 0; c < 4; c++)
								sum[k][c] = a.acc0;
#pragma unroll
						for (int j = 0; j < ND; j++) {
							uint4 q;
							if constexpr (REGS)
								q = ring[(s + 1 + j) % ND][b];
							else
								q = RING[(((s + 1 + j) % ND) * B + b) * CU8_NT + t];
#pragma unroll
							for (int k = 0; k < 4; k++) {
								if (!cu8_live(H, ND / 2, k, j))
									continue;
								const unsigned int cv = a.cvec[k * ND + j];
								sum[k][0] = dot4(q.x, cv, sum[k][0]);
								sum[k][1] = dot4(q.y, cv, sum[k][1]);
								sum[k][2] = dot4(q.z, cv, sum[k][2]);
								sum[k][3] = dot4(q.w, cv, sum[k][3]);
							}
						}
#pragma unroll
						for (int k = 0; k < 4; k++)
							orow[k][b] = cu8_round4(sum[k], a);
					}
#pragma unroll
					for (int k = 0; k < 4; k++)
						if (4 * Qo + k < a.height) {
							unsigned int w[B];
							cu8_interleave<B>(orow[k], w);
							cu8_store<B>(a, 4 * Qo + k, px0, px_end, w);
						}
				}
			}
		}
	}
}

// ---- a two-dimensional mask of MH rows, at most 9 columns (ND = 3)
template <int B, int MH, int H>
static __device__ __forceinline__ void conv_u8_2d_body(const Cu8Args &a, int item, unsigned int *lds)
{
	constexpr int ND = 3;
	(void) lds;
	const int t = tid(), lane = t & 63;
	const int strip = item % a.strips, seg = item / a.strips;
	const int X0 = (strip * (CU8_NT / 64) + (t >> 6)) * a.wout;
	const int px0 = X0 - 4 * a.pd + 4 * lane;
	const int px_end = min(X0 + a.wout, a.width);
	const bool writer = lane >= a.pd && lane < 64 - a.pd && px0 < px_end;
	const int Ya = seg * a.seg_rows, Yb = min(Ya + a.seg_rows, a.height);

	unsigned int win[MH][B][ND]; // the window dwords of the last MH rows, slot = row mod MH (static)
	// PF rows travelling per lane (with 2 a wave waited for memory two thirds of its time: rocprofv3
	// SQ_WAIT_ANY, profiles/r04_ops_pmc.txt)
	constexpr int PF = 4;
	unsigned int raw[PF][B];
#pragma unroll
	for (int r = 0; r < PF; r++)
		cu8_load<B>(a, Ya - a.vhalf + r, px0, raw[r]);
	for (int base = Ya - a.vhalf; base < Yb + a.vhalf; base += MH) {
#pragma unroll
		for (int s = 0; s < MH; s++) {
			const int yi = base + s; // input row, window slot s
			if (yi < Yb + a.vhalf) {
				unsigned int P[B];
				cu8_planar<B>(raw[0], P);
#pragma unroll
				for (int r = 0; r + 1 < PF; r++)
#pragma unroll
					for (int b = 0; b < B; b++)
						raw[r][b] = raw[r + 1][b];
				if (yi + PF < Yb + a.vhalf)
					cu8_load<B>(a, yi + PF, px0, raw[PF - 1]);
#pragma unroll
				for (int b = 0; b < B; b++)
					cu8_window<ND>(P[b] ^ 0x80808080u, win[s][b]);
				const int y = yi - a.vhalf; // the output row whose last mask row this is
				if (y >= Ya && writer) {
					unsigned int orow[B];
#pragma unroll
					for (int b = 0; b < B; b++) {
						int sum[4] = { a.acc0, a.acc0, a.acc0, a.acc0 };
#pragma unroll
						for (int i = 0; i < MH; i++) {
							const int ws = (s + 1 + i) % MH; // mask row i: input row y - vhalf + i
#pragma unroll
							for (int j = 0; j < ND; j++)
#pragma unroll
								for (int c = 0; c < 4; c++)
									if (cu8_live(H, 1, c, j))
										sum[c] = dot4(win[ws][b][j], a.cvec[(i * 4 + c) * ND + j], sum[c]);
						}
						orow[b] = cu8_round4(sum, a);
					}
					unsigned int w[B];
					cu8_interleave<B>(orow, w);
					cu8_store<B>(a, y, px0, px_end, w);
				}
			}
		}
	}
}

// a persistent block: work items (strip, segment) until the counter runs out
template <int B, int ND, int H, bool RG = true>
static __device__ __forceinline__ void conv_u8_sep_block(const Cu8Args &a, unsigned int *lds)
{
	int *slot = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(lds) + a.off_slot);
	for (;;) {
		const int item = next_item(a.counter, slot);
		if (item >= a.strips * a.segs)
			return;
		conv_u8_sep_body<B, ND, H, RG>(a, item, lds);
	}
}

template <int B, int MH, int H>
static __device__ __forceinline__ void conv_u8_2d_block(const Cu8Args &a, unsigned int *lds)
{
	int *slot = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(lds) + a.off_slot);
	for (;;) {
		const int item = next_item(a.counter, slot);
		if (item >= a.strips * a.segs)
			return;
		conv_u8_2d_body<B, MH, H>(a, item, lds);
	}
}

} // namespace vh
