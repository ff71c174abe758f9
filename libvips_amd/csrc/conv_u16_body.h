// Integer convolution of USHORT images with a small two-dimensional mask (vips_conv with
// precision=integer, convi.c:698-716,753-857, C path), streaming, on packed 16-bit lanes: the kernel
// body, written against gcn.h (product) / tests/emul/gcn.h (host fibers, CPU suite).  The sibling of
// conv_u8_body.h's conv_u8_2d: the general kernel (conv.hip `convi`) gives a thread one output
// element and reads its taps through L1 / L2 (11-17 % of HBM on 8192^2 x 3).
//
// A WAVE owns a strip of 128 pixel columns (one halo lane on each side) and streams down a segment of
// rows; nothing is shared between waves: no LDS, no barrier.  A lane owns 2 pixels of every row: it
// loads their B dwords, turns them into B planar dwords (2 pixels of one band each, v_perm) and biases
// them to signed 16-bit lanes (p ^ 0x8000 = p - 32768).  Outputs x0, x0 + 1 of a band need the planar
// pixels x0 - 2 .. x0 + 3 (masks up to 5 wide): the lane's own dword and its two neighbours' (DPP wave
// shifts); output x0 + c is the sum over the mask's rows and the three window dwords of
// v_dot2_i32_i16(window, coefficient pair) -- the mask row shifted by c pixels and cut into pairs on the
// host, scalar operands.  The window dwords of the mask's MH rows rotate through registers (the row
// loop is unrolled MH times).
//
// Rounding: clip((sum + rounding) / scale + offset) with C division.  With offset = 0 a negative
// numerator clips to 0 whatever its quotient; for 0 <= x < 2^31 and scale >= 2
//     floor(x / scale) = (x * m) >> (31 + l),   l = ceil(log2 scale), m = ceil(2^(31 + l) / scale) < 2^32
// (the error x e / 2^(31 + l) with e < 1 stays below 2^-l <= 1 / scale), i.e. v_mul_hi_u32 and a shift;
// the host checks the identity on the multiples of the scale and their predecessors before it takes
// this path.  The accumulators start at rounding + 32768 * sum(mask) (the bias of the signed lanes).
#pragma once

#include "gcn.h"

namespace vh {

constexpr int CU16_MAXMH = 5;  // rows of the mask
constexpr int CU16_CVEC = 32;  // coefficient dwords: MH x 2 outputs x 3 window dwords
constexpr int CU16_NT = 256;   // threads per block: 4 independent waves

struct Cu16Args {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int width, height;
	int half, vhalf;   // taps / 2 of the mask's width and height
	int wout;          // output columns per wave: 2 * 62
	int strips, segs, seg_rows;
	int off_slot;
	int *counter;
	int acc0;          // rounding + 32768 * sum(mask)
	unsigned int mult; // ceil(2^(31 + l) / scale), or 0: scale 1
	int shift;         // l - 1
	unsigned int cvec[CU16_CVEC];
};

// B interleaved dwords (2 pixels) -> B planar dwords (low half: band b of pixel 0, high half: of pixel 1)
template <int B>
VH_DEV void cu16_planar(const unsigned int (&w)[B], unsigned int (&P)[B])
{
	if constexpr (B == 1)
		P[0] = w[0];
	else {
#pragma unroll
		for (int b = 0; b < B; b++) {
			const int e0 = b, e1 = B + b; // elements of the 2 B the lane holds
			const unsigned int l0 = 2u * (e0 & 1), h0 = 4u + 2u * (e1 & 1);
			P[b] = perm(w[e1 >> 1], w[e0 >> 1], l0 | ((l0 + 1) << 8) | (h0 << 16) | ((h0 + 1) << 24));
		}
	}
}

// ... and back
template <int B>
VH_DEV void cu16_interleave(const unsigned int (&P)[B], unsigned int (&w)[B])
{
	if constexpr (B == 1)
		w[0] = P[0];
	else {
#pragma unroll
		for (int d = 0; d < B; d++) {
			const int e0 = 2 * d, e1 = e0 + 1; // element e: band e % B of pixel e / B
			const unsigned int l0 = 2u * (e0 / B), h0 = 4u + 2u * (e1 / B);
			w[d] = perm(P[e1 % B], P[e0 % B], l0 | ((l0 + 1) << 8) | (h0 << 16) | ((h0 + 1) << 24));
		}
	}
}

// the 2 pixels at px0, px0 + 1 of row `row` (both clamped into the image: vips_embed COPY) as B dwords
template <int B>
VH_DEV void cu16_load(const Cu16Args &a, int row, int px0, unsigned int (&w)[B])
{
	const int rc = min(max(row, 0), a.height - 1);
	const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) rc * a.in_stride;
	if (px0 >= 0 && px0 + 2 <= a.width)
		gload_dwords<B>(line, (unsigned int) (px0 * 2 * B), w);
	else {
#pragma unroll
		for (int d = 0; d < B; d++) {
			unsigned int v = 0;
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const int e = 2 * d + k;
				const int px = min(max(px0 + e / B, 0), a.width - 1);
				v |= gload16(line, (unsigned int) ((px * B + e % B) * 2)) << (16 * k);
			}
			w[d] = v;
		}
	}
}

template <int B>
VH_DEV void cu16_store(const Cu16Args &a, int row, int px0, int px_end, const unsigned int (&w)[B])
{
	const gptr_out line = gptr_out_of((unsigned long long) a.out) + (long long) row * a.out_stride;
	if (px0 + 2 <= px_end)
		gstore_dwords<B>(line + px0 * 2 * B, w);
	else {
#pragma unroll
		for (int d = 0; d < B; d++)
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const int e = 2 * d + k;
				if (px0 + e / B < px_end)
					gstore16(line + (px0 * B + e) * 2, (unsigned short) (w[d] >> (16 * k)));
			}
	}
}

// does the coefficient pair of output c, window dword j hold any tap of a mask of half-width H?
// (pixel x0 - 2 + 2 j + h is tap 2 j + h - 2 - c + H; H < 0: unknown, keep all)
constexpr bool cu16_live(int H, int c, int j)
{
	if (H < 0)
		return true;
	for (int h = 0; h < 2; h++) {
		const int tap = 2 * j + h - 2 - c + H;
		if (tap >= 0 && tap <= 2 * H)
			return true;
	}
	return false;
}

// clip((x + 0) / scale) of a numerator that already holds the rounding term, to 0 .. 65535
VH_DEV unsigned int cu16_fin(int x, const Cu16Args &a)
{
	const unsigned int n = (unsigned int) max(x, 0);
	const unsigned int q = a.mult ? umulhi(n, a.mult) >> a.shift : n;
	return min(q, 65535u);
}

template <int B, int MH, int H>
static __device__ __forceinline__ void conv_u16_2d_body(const Cu16Args &a, int item)
{
	const int t = tid(), lane = t & 63;
	const int strip = item % a.strips, seg = item / a.strips;
	const int X0 = (strip * (CU16_NT / 64) + (t >> 6)) * a.wout;
	const int px0 = X0 - 2 + 2 * lane; // one halo lane either side
	const int px_end = min(X0 + a.wout, a.width);
	const bool writer = lane >= 1 && lane < 63 && px0 < px_end;
	const int Ya = seg * a.seg_rows, Yb = min(Ya + a.seg_rows, a.height);

	unsigned int win[MH][B][3]; // the window dwords of the last MH rows, slot = row mod MH (static)
	constexpr int PF = 4;       // rows travelling per lane
	unsigned int raw[PF][B];
#pragma unroll
	for (int r = 0; r < PF; r++)
		cu16_load<B>(a, Ya - a.vhalf + r, px0, raw[r]);
	for (int base = Ya - a.vhalf; base < Yb + a.vhalf; base += MH) {
#pragma unroll
		for (int s = 0; s < MH; s++) {
			const int yi = base + s; // input row, window slot s
			if (yi < Yb + a.vhalf) {
				unsigned int P[B];
				cu16_planar<B>(raw[0], P);
#pragma unroll
				for (int r = 0; r + 1 < PF; r++)
#pragma unroll
					for (int b = 0; b < B; b++)
						raw[r][b] = raw[r + 1][b];
				if (yi + PF < Yb + a.vhalf)
					cu16_load<B>(a, yi + PF, px0, raw[PF - 1]);
#pragma unroll
				for (int b = 0; b < B; b++) {
					const unsigned int own = P[b] ^ 0x80008000u;
					win[s][b][0] = lane_prev(own);
					win[s][b][1] = own;
					win[s][b][2] = lane_next(own);
				}
				const int y = yi - a.vhalf; // the output row whose last mask row this is
				if (y >= Ya && writer) {
					unsigned int orow[B];
#pragma unroll
					for (int b = 0; b < B; b++) {
						int sum[2] = { a.acc0, a.acc0 };
#pragma unroll
						for (int i = 0; i < MH; i++) {
							const int ws = (s + 1 + i) % MH; // mask row i: input row y - vhalf + i
#pragma unroll
							for (int j = 0; j < 3; j++)
#pragma unroll
								for (int c = 0; c < 2; c++)
									if (cu16_live(H, c, j))
										sum[c] = dot2(win[ws][b][j], a.cvec[(i * 2 + c) * 3 + j], sum[c]);
						}
						orow[b] = cu16_fin(sum[0], a) | (cu16_fin(sum[1], a) << 16);
					}
					unsigned int w[B];
					cu16_interleave<B>(orow, w);
					cu16_store<B>(a, y, px0, px_end, w);
				}
			}
		}
	}
}

template <int B, int MH, int H>
static __device__ __forceinline__ void conv_u16_2d_block(const Cu16Args &a, unsigned int *lds)
{
	int *slot = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(lds) + a.off_slot);
	for (;;) {
		const int item = next_item(a.counter, slot);
		if (item >= a.strips * a.segs)
			return;
		conv_u16_2d_body<B, MH, H>(a, item);
	}
}

} // namespace vh
