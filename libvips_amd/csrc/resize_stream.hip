// vips_resize() of uchar images by an even integer factor, ALL FOUR operations in one streaming
// kernel, and a batch of same-sized images in one launch (BASELINE configs 1 and 4):
//
//   vips_shrinkv(vs, ceil) -> vips_reducev(2.0) -> vips_shrinkh(hs, ceil) -> vips_reduceh(2.0)
//   (resample/resize.c:207-228 chains shrinkv.c:158-268, reducev.cpp:418-459, shrinkh.c:78-156 and
//   reduceh.cpp:216-255 through three intermediate images; each rounds to uchar).
//
// With `gap` = 2 a resize by 1/(2k) runs a box shrink of k and a residual reduce of exactly 2 on
// each axis: the reduce then steps two pixels per output with ONE coefficient phase, and the
// whole chain streams.  A 256-thread block owns a strip of output columns and a segment of output
// rows of one image and walks down the input rows; lane t owns bytes [8t, 8t + 8) of the strip's
// span of every row (one global_load_dwordx2 per row: a wave reads 512 contiguous bytes; a
// scanline is a byte array to the two vertical operations, so any band count works).
//
//   vertical: the vs rows of a box are summed as two 16-bit lanes per dword and rounded the way
//   shrinkv does; two shrunk rows make one i16 pair per byte column and feed the 7 reducev sums
//   they are taps (2q, 2q + 1) of with v_dot2_i32_i16 (7 accumulator rows per column rotate
//   statically through the unrolled body of 7 pairs); one row of the vertically resized image
//   retires per pair into LDS.  Row registers are refilled in place with the next pair's rows, so
//   2 vs loads per lane are always in flight.  Nothing is read twice within a segment.
//   horizontal, every 7 retired rows: shrinkh box sums from that LDS slab into a second one, then
//   the 13 reduceh taps from there, bytes stored straight to the output image.  While a block is
//   in its horizontal pass it issues no loads, so the pass is kept short: when a box is whole
//   dwords (hs * bands a multiple of 4, e.g. 4 x RGB = 12 bytes) its sums are v_dot4_u32_u8 of
//   the slab's dwords with 0/1 byte masks (band b of dword j), the shrunk pixels go to the second
//   slab as 16-bit lanes, band-planar, indexed by UNCLAMPED column (edge columns are written
//   twice), and a reduceh output is 7 v_dot2_i32_i16 per row on consecutive dwords (HF below;
//   any other geometry takes the byte-by-byte form).
//
// The 1/vs-size and the two further intermediate images of the chain never exist: an 8192 x 8192 x 3
// image is read once (201 MB) and 3 MB are written.  Every rounding is the separate operations'
// own, so the result is theirs bit for bit (tests/test_resample_gpu.py compares it with the
// compiled reference and with the unfused kernels).
#include "resample.h"
#include "reduce_u8.h"
#include "kernel_stmt.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace vh {

constexpr int RS_SPAN = 2048; // bytes of a row a strip covers
constexpr int RS_NT = 256;    // a strip's outputs are at most this many band elements wide
constexpr int RS_MAXB = 64;        // images per launch
constexpr int RS_NP = 7;           // most coefficient pairs of a vertical reduce (13 taps) = most rows per slab

struct StreamArgs {
	long long in_stride, out_stride;
	int width, height, bands; // input images
	int h1, w3;               // height after shrinkv, width after shrinkh
	int out_width, out_height;
	int hs;
	unsigned int mult_v, mult_h; // 2^32 / (256 * shrink), shrinkv.c:201 / shrinkh.c:141
	int fv, fh;                  // first tap of output row 0 / output column 0
	int n_h;                     // horizontal taps (<= 13)
	int tw, seg;                 // output columns per strip, output rows per segment
	int s_pitch;                 // bytes per row of the shrinkh slab
	int debug;                   // $VIPS_HIP_STREAM_DEBUG: 1 skip the horizontal pass, 2 skip its stores (HF form) (timing only)
	int nstrips, nsegs, n_images; // the launch: strips x segments x images
	int grouped;                  // the strips of a (segment, image) on one XCD
	unsigned int cv[RS_NP];      // vertical taps (2q, 2q + 1) as i16 pairs
	short ch[16];                // horizontal taps (those past n_h are 0: read in pairs by the HF pass)
	int s_len;                   // HF: dwords per band and row of the shrinkh slab
	int o_pitch, burst;          // HF: bytes per staged output row, slabs the stage holds (0: store row by row)
	int window;                  // HF: the staged rows leave when the 100 MHz clock crosses a multiple of 2^window ticks
};

struct StreamPtrs {
	const unsigned char *in[RS_MAXB];
	unsigned char *out[RS_MAXB];
};

typedef short rs_short2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ int rs_dot2(unsigned int pix, unsigned int coef, int acc)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(rs_short2, pix), __builtin_bit_cast(rs_short2, coef), acc, false);
}

// (sum + 2048) >> 12, clip (templates.h:152-157); see fin_u8 in reduce_u8.hip for the asm
static __device__ __forceinline__ unsigned int rs_fin(int s)
{
	s = (s + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT;
	VH_VECTOR1(s);
	return (unsigned int) min(max(s, 0), 255);
}

// two i32 sums, already shifted, as one dword of two saturated bytes: {0, 0, sat_u8(hi), sat_u8(lo)}
static __device__ __forceinline__ unsigned int rs_sat2(int lo, int hi)
{
	// the low halves of the two sums side by side (|sum >> 12| < 2^15), then v_sat_pk_u8_i16
	const unsigned int both = __builtin_amdgcn_perm((unsigned int) hi, (unsigned int) lo, 0x05040100u);
	unsigned int r;
	VH_SAT_PK_U8_I16(r, both);
	return r;
}

// The box rounding of shrinkv on two 16-bit sums held in one dword (bytes 0 and 2 of a dword
// column, or 1 and 3): ((sum + vs/2) * (2^32 / (256 vs))) >> 24, shrinkv.c:158-165; the result
// again as two 16-bit lanes.
template <int VS>
static __device__ __forceinline__ unsigned int rs_box2(unsigned int sums, unsigned int mult)
{
	if constexpr ((VS & (VS - 1)) == 0) {
		// a power of two: the multiplier is 2^(24 - log2 vs), both lanes shift at once
		constexpr int SH = VS == 1 ? 0 : VS == 2 ? 1 : VS == 4 ? 2 : VS == 8 ? 3 : 4;
		constexpr unsigned int RND = (unsigned int) (VS / 2) * 0x00010001u;
		return ((sums + RND) >> SH) & 0x00ff00ffu;
	}
	else {
		const unsigned int lo = (((sums & 0xffffu) + VS / 2) * mult) >> 24;
		const unsigned int hi = (((sums >> 16) + VS / 2) * mult) >> 24;
		return lo | (hi << 16);
	}
}

// DW = dwords of a row per lane (2: 256 threads with 7 x 8 sums each; 1: 512 threads, half the
// registers, twice the waves)
// NP = coefficient pairs of the vertical reduce = output rows in flight per column = rows per
// slab: 7 (13 taps: lanczos3), 5 (9 taps: lanczos2, cubic, mitchell), 3 (5 taps: linear)
// HF = the dword form of the horizontal pass (see the head of the file)
// (the compiler gathers a pair's 2 VS refills in front of the pair: 2 VS to 4 VS loads in flight per
// lane; pinned behind their rows' uses -- 2 VS at all times -- the kernel ran 0.5 % faster, within
// what the boxes differ by: profiles/r03_probes.txt)
template <int VS, int DW, int NP, bool HF>
__global__ void __launch_bounds__(RS_SPAN / (4 * DW))
resize_stream_u8(StreamArgs a, StreamPtrs ptrs_by_value)
{
	constexpr int NT = RS_SPAN / (4 * DW);
	constexpr int NB = 4 * DW; // byte columns per lane
	VH_DYNAMIC_LDS(unsigned int, rs_lds);
	unsigned char *T = reinterpret_cast<unsigned char *>(rs_lds); // NP rows of RS_SPAN bytes
	unsigned char *S = T + NP * RS_SPAN;                        // NP rows of s_pitch bytes
	unsigned int *HM = reinterpret_cast<unsigned int *>(S + NP * a.s_pitch); // HF: byte masks [band][dword of a box]
	unsigned int *CLK = HM + 32;                                             // HF: the clock slot every wave of the block acts on
	unsigned char *O = reinterpret_cast<unsigned char *>(HM + 36);           // HF: burst * NP staged output rows
	(void) ptrs_by_value;
	// the image pointers where they lie in the kernarg segment (a by-value array indexed
	// dynamically would be copied to scratch)
	typedef const unsigned long long __attribute__((address_space(4))) *KernargPtrs;
	static_assert(sizeof(StreamArgs) % 8 == 0, "kernarg layout");
	const KernargPtrs kp = (KernargPtrs) ((const char __attribute__((address_space(4))) *)
											  __builtin_amdgcn_kernarg_segment_ptr() +
		sizeof(StreamArgs));
	// (pointers made from integers are generic to the compiler: say they are global, or every
	// access is a flat_load)
	typedef const unsigned char __attribute__((address_space(1))) *GlobalIn;
	typedef unsigned char __attribute__((address_space(1))) *GlobalOut;
	typedef const unsigned int __attribute__((address_space(1))) *GlobalIn1;
	typedef unsigned int rs_uint2 __attribute__((ext_vector_type(2)));
	typedef const rs_uint2 __attribute__((address_space(1))) *GlobalIn2;
	// block -> (strip, segment, image).  Workgroup w runs on XCD w % 8: the strips of one
	// (segment, image) go to ONE XCD, next to each other in launch order, so the input lines two
	// neighbouring strips share (the 11-column halo, and the 128-byte line their border falls
	// in) come from HBM once and from that XCD's L2 the second time (dealt strip by strip round
	// the XCDs the kernel fetched 1.16 x the image: rocprofv3 FETCH_SIZE, profiles/r02b_c4_pmc.txt)
	// (a launch of few units -- one image -- deals its blocks round the XCDs one by one instead:
	// with two blocks per CU the even spread matters more than the shared lines, 0.048 against
	// 0.058 ms for one 8192 x 8192 x 3 image)
	const int wg = blockIdx.x;
	int strip, unit;
	if (a.grouped) {
		const int grp = (wg >> 3) / a.nstrips;
		strip = (wg >> 3) - grp * a.nstrips;
		unit = grp * 8 + (wg & 7);
	}
	else {
		unit = wg / a.nstrips;
		strip = wg - unit * a.nstrips;
	}
	if (unit >= a.nsegs * a.n_images)
		return;
	const int img = unit / a.nsegs;
	const int seg_i = unit - img * a.nsegs;
	const GlobalIn in = (GlobalIn) kp[img];
	const GlobalOut out = (GlobalOut) kp[RS_MAXB + img];

	const int t = threadIdx.x;
	const int B = a.bands;
	if constexpr (HF) {
		// byte i of dword j of a box belongs to band (4 j + i) % B (ordered before its readers by
		// the first slab's barrier)
		if (t < 32) {
			const int b = t >> 3, j = t & 7;
			unsigned int m = 0;
#pragma unroll
			for (int i = 0; i < 4; i++)
				if ((4 * j + i) % B == b)
					m |= 1u << (8 * i);
			HM[t] = m;
		}
	}
	const int x0 = strip * a.tw, nx = min(a.tw, a.out_width - x0);
	const int y0 = seg_i * a.seg, ny = min(a.seg, a.out_height - y0);
	// columns of the shrinkh image the strip's taps touch, and the input bytes under them
	const int c_lo = min(max(2 * x0 + a.fh, 0), a.w3 - 1);
	const int c_hi = min(max(2 * (x0 + nx - 1) + a.fh + a.n_h - 1, 0), a.w3 - 1);
	const int ncol = c_hi - c_lo + 1;
	const int row_bytes = a.width * B;
	// the strip's span starts at the dword holding its first byte, or earlier when it would run
	// over the end of the row (row_bytes % 4 == 0: host); a row shorter than the span is one strip
	// whose lanes beyond the row re-read its last dword (nothing reads what they make)
	const int start_al = max(min((c_lo * a.hs * B) & ~3, row_bytes - RS_SPAN), 0);
	const GlobalIn span = in + start_al; // uniform: rows load as scalar base + lane offset
	const unsigned int lane_off = (unsigned int) min(4 * DW * t, row_bytes - 4 * DW - start_al);

	struct Row {
		unsigned int w[DW];
	};
	// input row k of shrunk row `r` (any integer: rows clamp to the shrunk image, boxes to the input)
	auto load = [&](int r, int k) -> Row {
		const int rc = min(max(r, 0), a.h1 - 1);
		const int row = min(rc * VS + k, a.height - 1);
		// (the host checked height * stride < 2^32: a 32-bit scalar multiply)
		const unsigned int row_off = (unsigned int) row * (unsigned int) a.in_stride;
		// global_load_dword(x2) v, v_off, s[span]: one 32-bit add per load (as span + row_off +
		// lane_off the compiler keeps a 64-bit vector address per row instead)
		const unsigned int off = row_off + lane_off;
		Row v;
		if constexpr (DW == 2) {
			const rs_uint2 x = *(GlobalIn2) (span + off);
			v.w[0] = x.x;
			v.w[DW - 1] = x.y;
		}
		else
			v.w[0] = *(GlobalIn1) (span + off);
		return v;
	};

	// NP output rows in flight per byte column; a sum starts at the rounding term of its final
	// (sum + 2048) >> 12 (templates.h:152-157)
	int acc[NP][NB];
	int half = INTERPOLATE_SCALE >> 1;
	VH_VECTOR1(half); // one register for all sums to start from
#pragma unroll
	for (int s = 0; s < NP; s++)
#pragma unroll
		for (int b = 0; b < NB; b++)
			acc[s][b] = 0;

	// pair j of the segment = shrunk rows r0 + 2j, r0 + 2j + 1; it is tap pair q of output row
	// y0 + j - q.  Pair -1 is a dummy that lets a slab of NP output rows end with a body.
	const int r0 = 2 * y0 + a.fv;
	Row ring[2][VS];
#pragma unroll
	for (int h = 0; h < 2; h++)
#pragma unroll
		for (int k = 0; k < VS; k++)
			ring[h][k] = load(r0 - 2 + h, k);

	// HF: output rows wait in LDS and leave `burst` slabs at a time -- a trickle of small writes
	// into the streaming read costs a fifth of its rate on this part (the stores of 1.5 % of the
	// bytes: 0.0404 -> 0.0337 ms per image without them), a burst now and then far less (the same
	// finding as reduce_u8.hip's output stage)
	//
	// ... and the bursts of ALL blocks fall together: a block writes what it holds when the chip-wide
	// 100 MHz clock (s_memrealtime) crosses a multiple of 2^window ticks (and when its stage is
	// full, and at its end).  HBM pays for every change of direction on a channel; output rows are
	// spread over all channels, so only writes that arrive together from the whole chip share
	// those turnarounds (tools/c4_load_probe: the stores of this kernel cost 18 % of the read rate
	// trickling, 10 % in per-block bursts, 6 % in chip-wide windows 82 us apart).
	int staged = 0, flushed = 0;
	unsigned int slot = 0; // (the first slab's reading differs: a first, short burst)
	const int nbody = (ny + NP - 1) / NP + 1;
	for (int n = 0; n < nbody; n++) {
#pragma unroll
		for (int p = 0; p < NP; p++) {
			const int j = NP * n + p - 1;
			// the two shrunk rows: per dword column the even bytes (0, 2) and the odd bytes (1, 3)
			// as 16-bit lanes: [h][2 d] even, [h][2 d + 1] odd
			unsigned int sb[2][2 * DW];
#pragma unroll
			for (int h = 0; h < 2; h++) {
				unsigned int e[DW], o[DW];
#pragma unroll
				for (int d = 0; d < DW; d++)
					e[d] = o[d] = 0;
#pragma unroll
				for (int k = 0; k < VS; k++) {
					const Row w = ring[h][k];
#pragma unroll
					for (int d = 0; d < DW; d++) {
						e[d] += w.w[d] & 0x00ff00ffu;
						o[d] += __builtin_amdgcn_perm(0u, w.w[d], 0x0c030c01u);
					}
					ring[h][k] = load(r0 + 2 * (j + 1) + h, k);
				}
#pragma unroll
				for (int d = 0; d < DW; d++) {
					sb[h][2 * d] = rs_box2<VS>(e[d], a.mult_v);
					sb[h][2 * d + 1] = rs_box2<VS>(o[d], a.mult_v);
				}
			}
			// byte column b of the lane: dword b / 4, lane (b % 4) / 2 of its even / odd sums
#pragma unroll
			for (int b = 0; b < NB; b++) {
				const int w = (b >> 2) * 2 + (b & 1);
				const unsigned int pk = __builtin_amdgcn_perm(sb[1][w], sb[0][w], (b & 2) ? 0x07060302u : 0x05040100u);
#pragma unroll
				for (int q = 0; q < NP; q++) {
					const int slot = (p - 1 - q + 2 * NP) % NP;
					if (q == 0)
						VH_DOT2_SCALAR_COEF(acc[slot][b], pk, a.cv[0], half);
					else
						acc[slot][b] = rs_dot2(pk, a.cv[q], acc[slot][b]);
				}
			}
			// output row y0 + j - (NP - 1) is complete: slab row p
			unsigned int packed[DW];
#pragma unroll
			for (int d = 0; d < DW; d++)
				packed[d] = rs_sat2(acc[p][4 * d] >> INTERPOLATE_SHIFT, acc[p][4 * d + 1] >> INTERPOLATE_SHIFT) |
					(rs_sat2(acc[p][4 * d + 2] >> INTERPOLATE_SHIFT, acc[p][4 * d + 3] >> INTERPOLATE_SHIFT) << 16);
			if constexpr (DW == 2)
				*reinterpret_cast<uint2 *>(T + p * RS_SPAN + 8 * t) = make_uint2(packed[0], packed[DW - 1]);
			else
				*reinterpret_cast<unsigned int *>(T + p * RS_SPAN + 4 * t) = packed[0];
			// keep the pairs apart: left alone the scheduler hoists the loads of several pairs to
			// the top of the body and holds their destinations (50 more registers)
			__builtin_amdgcn_sched_barrier(0);
		}
		if (n == 0 || (a.debug & 1))
			continue;

		// ---- the slab: output rows y0 + NP (n - 1) ...
		const int yb = NP * (n - 1);
		const int nr = min(NP, ny - yb);
		__syncthreads();
		// (everything the horizontal pass derives from the thread index is made anew per slab:
		// hoisted out of the row loop it would sit in ~30 registers through the vertical pass)
		int th = t;
		VH_VECTOR1(th);
		if constexpr (!HF) {
			// shrinkh: thread = one band element of the shrunk rows, all slab rows
			{
				const int per_row = ncol * B;
				const unsigned int magic = (65536u + B - 1) / B;
				for (int e = th; e < per_row; e += NT) {
					const int c = (int) ((e * magic) >> 16);
					const int b = e - c * B;
					const int px0 = (c_lo + c) * a.hs;
					const unsigned char *src = T + b - start_al;
					unsigned int sum[NP];
	#pragma unroll
					for (int r = 0; r < NP; r++)
						sum[r] = (unsigned int) (a.hs / 2);
	#pragma unroll 4
					for (int k = 0; k < a.hs; k++) {
						const int off = min(px0 + k, a.width - 1) * B;
	#pragma unroll
						for (int r = 0; r < NP; r++)
							sum[r] += src[r * RS_SPAN + off];
					}
	#pragma unroll
					for (int r = 0; r < NP; r++)
						S[r * a.s_pitch + e] = (unsigned char) ((sum[r] * a.mult_h) >> 24);
				}
			}
			__syncthreads();
			// reduceh: thread = one band element of the output rows (nx * B <= 256), all slab rows
			if (th < nx * B) {
				const unsigned int magic = (65536u + B - 1) / B;
				const int x = (int) ((th * magic) >> 16);
				const int b = th - x * B;
				const int f = 2 * (x0 + x) + a.fh;
				int sum[NP];
	#pragma unroll
				for (int r = 0; r < NP; r++)
					sum[r] = 0;
	#pragma unroll
				for (int k = 0; k < 13; k++) {
					if (k < a.n_h) {
						const int off = (min(max(f + k, 0), a.w3 - 1) - c_lo) * B + b;
						const int ck = a.ch[k];
	#pragma unroll
						for (int r = 0; r < NP; r++)
							sum[r] += ck * (int) S[r * a.s_pitch + off];
					}
				}
				const GlobalOut dst = out + (long long) (y0 + yb) * a.out_stride + (long long) x0 * B + th;
	#pragma unroll
				for (int r = 0; r < NP; r++)
					if (r < nr)
						dst[(long long) r * a.out_stride] = (unsigned char) rs_fin(sum[r]);
			}
		}
		else {
			// shrinkh: thread = one band of one UNCLAMPED shrunk column u (column 2 x0 + fh + u of
			// the shrunk image, clamped into it: reduceh's edge taps then read plain consecutive
			// columns), band-major so that a wave's lanes walk consecutive boxes; all slab rows
			const int len = 2 * nx + a.n_h - 1;
			const int total = len * B;
			const int ndw = (a.hs * B) >> 2;
			unsigned short *S16 = reinterpret_cast<unsigned short *>(S);
			const int band_pitch = 2 * a.s_len; // 16-bit lanes per band and row
			for (int e = th; e < total; e += NT) {
				const int b = (e >= len) + (e >= 2 * len) + (e >= 3 * len);
				const int u = e - b * len;
				const int col = min(max(2 * x0 + a.fh + u, 0), a.w3 - 1);
				const unsigned int *src = reinterpret_cast<const unsigned int *>(T + col * a.hs * B - start_al);
				unsigned int sum[NP];
#pragma unroll
				for (int r = 0; r < NP; r++)
					sum[r] = (unsigned int) (a.hs / 2);
#pragma unroll
				for (int j = 0; j < 8; j++)
					if (j < ndw) {
						const unsigned int m = HM[b * 8 + j];
#pragma unroll
						for (int r = 0; r < NP; r++)
							sum[r] = __builtin_amdgcn_udot4(src[r * (RS_SPAN / 4) + j], m, sum[r], false);
					}
#pragma unroll
				for (int r = 0; r < NP; r++)
					S16[(r * B + b) * band_pitch + u] = (unsigned short) ((sum[r] * a.mult_h) >> 24);
			}
			// (ONE reading of the clock per block and slab: waves that read it for themselves could
			// disagree about the slot, and about the barrier below)
			if (th == 0)
				*CLK = (unsigned int) (__builtin_amdgcn_s_memrealtime() >> a.window);
			__syncthreads();
			// reduceh: thread = one band element of the output rows; taps (2 q, 2 q + 1) of output
			// x are the two lanes of dword x + q of the band's row
			if (th < nx * B) {
				const unsigned int magic = (65536u + B - 1) / B;
				const int x = (int) ((th * magic) >> 16);
				const int b = th - x * B;
				const unsigned int *row = reinterpret_cast<const unsigned int *>(S) + b * a.s_len + x;
				const int nq = (a.n_h + 1) >> 1;
				int sum[NP];
#pragma unroll
				for (int r = 0; r < NP; r++)
					sum[r] = 0;
#pragma unroll
				for (int q = 0; q < 7; q++)
					if (q < nq) {
						const unsigned int ck = (unsigned int) (unsigned short) a.ch[2 * q] |
							((unsigned int) (unsigned short) a.ch[2 * q + 1] << 16);
#pragma unroll
						for (int r = 0; r < NP; r++)
							sum[r] = rs_dot2(row[r * B * a.s_len + q], ck, sum[r]);
					}
				if (a.burst > 0) {
					// into the stage, every row at the byte offset its global address has in a dword
					// (the burst then moves whole dwords)
					const unsigned int lo = (unsigned int) (unsigned long long) (out + (long long) x0 * B);
#pragma unroll
					for (int r = 0; r < NP; r++)
						if (r < nr) {
							const unsigned int mis = (lo + (unsigned int) (y0 + yb + r) * (unsigned int) a.out_stride) & 3u;
							O[(staged + r) * a.o_pitch + mis + th] = (unsigned char) rs_fin(sum[r]);
						}
				}
				else {
					const GlobalOut dst = out + (long long) (y0 + yb) * a.out_stride + (long long) x0 * B + th;
#pragma unroll
					for (int r = 0; r < NP; r++)
						if (r < nr && !(a.debug & 2)) // (debug 2: everything but the stores -- timing only)
							dst[(long long) r * a.out_stride] = (unsigned char) rs_fin(sum[r]);
				}
			}
			if (a.burst > 0) {
				staged += nr;
				const unsigned int now = *CLK;
				if (now != slot || staged + NP > a.burst * NP || n == nbody - 1) {
					slot = now;
					__syncthreads();
					// a wave per row, a lane per dword of the row's window [address & ~3, ...): whole
					// dwords inside the fragment as dwords, its ragged ends byte by byte
					const int nb = nx * B;
					for (int row = th >> 6; row < staged; row += NT / 64) {
						const long long first = (long long) (y0 + flushed + row) * a.out_stride + (long long) x0 * B;
						const int mis = (int) ((unsigned int) (unsigned long long) (out + first) & 3u);
						const GlobalOut g = out + first - mis;
						const unsigned char *src = O + row * a.o_pitch;
						for (int d = th & 63; 4 * d < mis + nb; d += 64) {
							const int b0 = 4 * d;
							if (a.debug & 2)
								continue;
							if (b0 >= mis && b0 + 4 <= mis + nb)
								*(unsigned int __attribute__((address_space(1))) *) (g + b0) = *reinterpret_cast<const unsigned int *>(src + b0);
							else
								for (int k = 0; k < 4; k++)
									if (b0 + k >= mis && b0 + k < mis + nb)
										g[b0 + k] = src[b0 + k];
						}
					}
					flushed += staged;
					staged = 0;
				}
			}
		}
	}
}

namespace {

// positions 2 k + first0 with one phase
bool stream_regular(const std::vector<ReducePos> &pos, int *first0, int *phase)
{
	if (pos.empty())
		return false;
	*first0 = pos[0].first;
	*phase = pos[0].phase;
	for (size_t k = 0; k < pos.size(); k++)
		if (pos[k].first != *first0 + 2 * (int) k || pos[k].phase != *phase)
			return false;
	return true;
}

template <int VS, int NP>
void stream_launch_np(const StreamArgs &a, const StreamPtrs &p, dim3 grid, size_t lds, int dw, bool hf)
{
	if (dw == 2 && NP == RS_NP)
		hipLaunchKernelGGL((resize_stream_u8<VS, NP == RS_NP ? 2 : 1, NP, false>), grid, dim3(RS_SPAN / 8, 1, 1), lds, stream(), a, p);
	else if (hf)
		hipLaunchKernelGGL((resize_stream_u8<VS, 1, NP, true>), grid, dim3(RS_SPAN / 4, 1, 1), lds, stream(), a, p);
	else
		hipLaunchKernelGGL((resize_stream_u8<VS, 1, NP, false>), grid, dim3(RS_SPAN / 4, 1, 1), lds, stream(), a, p);
}

template <int VS>
void stream_launch(const StreamArgs &a, const StreamPtrs &p, dim3 grid, size_t lds, int dw, int np, bool hf)
{
	if (np == 7)
		stream_launch_np<VS, 7>(a, p, grid, lds, dw, hf);
	else if (np == 5)
		stream_launch_np<VS, 5>(a, p, grid, lds, dw, hf);
	else
		stream_launch_np<VS, 3>(a, p, grid, lds, dw, hf);
}

} // namespace

// The whole downsizing chain of vips_resize on n uchar images of one geometry.  `rv` was built
// for the image after shrinkv(vs) (height h1), `rh` for the one after shrinkh(hs) (width w3).
// 1 = handled, 0 = not this kernel's case (nothing launched), -1 = error.
int resize_stream_u8_try(_VipsHipReduce *rv, int vs, _VipsHipReduce *rh, int hs, int h1, int w3,
	const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n, int tile)
{
	if (getenv("VIPS_HIP_NO_RESIZE_STREAM") || n < 1)
		return 0;
	const VipsHipRegion *i0 = in[0], *o0 = out[0];
	for (int i = 0; i < n; i++) {
		const VipsHipRegion *ri = in[i], *ro = out[i];
		if (ri->format != VIPS_HIP_FORMAT_UCHAR || ro->format != VIPS_HIP_FORMAT_UCHAR || ri->bands != ro->bands ||
			ri->bands < 1 || ri->bands > 4)
			return 0;
		if (ri->left != 0 || ri->top != 0 || ri->width != ri->im_width || ri->height != ri->im_height ||
			ro->left != 0 || ro->top != 0 || ro->width != ro->im_width || ro->height != ro->im_height)
			return 0;
		if (ri->width != i0->width || ri->height != i0->height || ri->bands != i0->bands || ri->stride != i0->stride ||
			ro->width != o0->width || ro->height != o0->height || ro->stride != o0->stride)
			return 0;
		if (((uintptr_t) ri->data & 3) || (ri->stride & 3))
			return 0;
	}
	const int B = i0->bands;
	const long long row_bytes = (long long) i0->width * B;
	if ((row_bytes & 3) || row_bytes < 8 || row_bytes > 0x3fffffffLL)
		return 0;
	if ((unsigned long long) i0->stride * (unsigned long long) i0->height > 0xffffffffULL)
		return 0;
	if (vs != 1 && vs != 2 && vs != 3 && vs != 4 && vs != 5 && vs != 6 && vs != 8)
		return 0;
	if (hs < 1 || hs > 64)
		return 0;
	if (rv->in_size != h1 || rv->out_size != o0->height || rh->in_size != w3 || rh->out_size != o0->width)
		return 0;
	// 13, 9 or 5 vertical taps (7, 5, 3 coefficient pairs); at most 13 horizontal ones
	if ((rv->n_point != 13 && rv->n_point != 9 && rv->n_point != 5) || rh->n_point > 13 || rh->n_point < 1)
		return 0;
	const int np = (rv->n_point + 1) / 2;
	std::vector<ReducePos> pv, ph;
	reduce_positions(rv, 0, o0->height, tile, pv);
	reduce_positions(rh, 0, o0->width, 0, ph);
	int fv, fh, phase_v, phase_h;
	if (!stream_regular(pv, &fv, &phase_v) || !stream_regular(ph, &fh, &phase_h))
		return 0;

	StreamArgs a;
	memset(&a, 0, sizeof(a));
	a.in_stride = (long long) i0->stride;
	a.out_stride = (long long) o0->stride;
	a.width = i0->width;
	a.height = i0->height;
	a.bands = B;
	a.h1 = h1;
	a.w3 = w3;
	a.out_width = o0->width;
	a.out_height = o0->height;
	a.hs = hs;
	a.mult_v = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) vs));
	a.mult_h = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) hs));
	a.fv = fv;
	a.fh = fh;
	a.n_h = rh->n_point;
	// the widest strip whose span fits: (2 tw + n_h - 1) shrunk columns of hs pixels, + 3 bytes of alignment
	int tw = (int) (((RS_SPAN - 3) / ((long long) hs * B) - (a.n_h - 1)) / 2);
	if (tw > RS_NT / B)
		tw = RS_NT / B;
	if (tw > o0->width)
		tw = o0->width;
	if (tw < 4)
		return 0;
	a.tw = tw;
	a.s_pitch = ((2 * tw + a.n_h - 1) * B + 3) & ~3;
	// the dword form of the horizontal pass: boxes of whole dwords (at most 8), none clipped by the
	// right edge of the image (the byte form clamps pixel by pixel there)
	int dw = getenv("VIPS_HIP_STREAM_DW") ? atoi(getenv("VIPS_HIP_STREAM_DW")) : 1;
	if (row_bytes < RS_SPAN)
		dw = 1; // (a row shorter than the span: lanes beyond it clamp dword by dword)
	const char *hf_env = getenv("VIPS_HIP_STREAM_HF");
	const bool hf = dw == 1 && (hs * B) % 4 == 0 && hs * B <= 32 && i0->width % hs == 0 && !(hf_env && atoi(hf_env) == 0);
	if (hf) {
		// 2 tw + n_h - 1 columns and the lane a last odd tap's partner reads; an odd number of dwords
		// per band (the three bands of a wave's lanes then fall on different banks)
		a.s_len = ((2 * tw + a.n_h - 1 + 1 + 1) / 2) | 1;
		a.s_pitch = B * a.s_len * 4;
		a.o_pitch = (tw * B + 3 + 3) & ~3;
		a.burst = getenv("VIPS_HIP_STREAM_BURST") ? atoi(getenv("VIPS_HIP_STREAM_BURST")) : 14;
		if (a.burst < 0 || a.burst > 32)
			a.burst = 14;
		a.window = getenv("VIPS_HIP_STREAM_WINDOW") ? atoi(getenv("VIPS_HIP_STREAM_WINDOW")) : 13;
		if (a.window < 4 || a.window > 40)
			a.window = 13;
	}
	const int nstrips = (o0->width + tw - 1) / tw;
	// segments: enough blocks to fill the chip several times, but tall (a segment re-reads 6 pairs)
	long long want = getenv("VIPS_HIP_STREAM_BLOCKS") ? atoll(getenv("VIPS_HIP_STREAM_BLOCKS")) : 2048;
	int nsegs = (int) ((want + (long long) nstrips * n - 1) / ((long long) nstrips * n));
	int seg = (o0->height + nsegs - 1) / nsegs;
	// (shortest segment: 21 output rows -- one 4096 x 4096 x 3 image, BASELINE config 1: 0.0359 ms against
	// 0.0385 with 28 and 0.0405 with 14, tools/time_c1.py)
	const int seg_min = getenv("VIPS_HIP_STREAM_SEG") ? atoi(getenv("VIPS_HIP_STREAM_SEG")) : 21;
	if (seg < seg_min)
		seg = seg_min;
	seg = (seg + np - 1) / np * np;
	nsegs = (o0->height + seg - 1) / seg;
	a.seg = seg;
	a.debug = getenv("VIPS_HIP_STREAM_DEBUG") ? atoi(getenv("VIPS_HIP_STREAM_DEBUG")) : 0;
	const short *cvs = &rv->matrixs[(size_t) phase_v * rv->n_point];
	for (int q = 0; q < np; q++) {
		const unsigned int lo = (unsigned short) cvs[2 * q];
		const unsigned int hi = 2 * q + 1 < rv->n_point ? (unsigned short) cvs[2 * q + 1] : 0u;
		a.cv[q] = lo | (hi << 16);
	}
	const short *chs = &rh->matrixs[(size_t) phase_h * rh->n_point];
	for (int k = 0; k < rh->n_point; k++)
		a.ch[k] = chs[k];
	// ($VIPS_HIP_STREAM_LDSPAD: unused LDS bytes per block, to hold the blocks per CU down in experiments)
	// (the stage shrinks until the block's LDS is within the 64 KB a launch gets without asking)
	while (hf && a.burst > 1 && (size_t) np * RS_SPAN + (size_t) np * a.s_pitch + 144 + (size_t) a.burst * np * a.o_pitch > 65536)
		a.burst--;
	const size_t lds = (size_t) np * RS_SPAN + (size_t) np * a.s_pitch +
		(hf ? 36 * sizeof(unsigned int) + (size_t) a.burst * np * a.o_pitch : 0) +
		(getenv("VIPS_HIP_STREAM_LDSPAD") ? (size_t) atoi(getenv("VIPS_HIP_STREAM_LDSPAD")) : 0);

	Gate gate("resize_stream_u8");
	for (int base = 0; base < n; base += RS_MAXB) {
		const int count = n - base < RS_MAXB ? n - base : RS_MAXB;
		StreamPtrs p;
		memset(&p, 0, sizeof(p));
		for (int i = 0; i < count; i++) {
			p.in[i] = (const unsigned char *) in[base + i]->data;
			p.out[i] = (unsigned char *) out[base + i]->data;
		}
		a.nstrips = nstrips;
		a.nsegs = nsegs;
		a.n_images = count;
		const long long units = (long long) nsegs * count;
		a.grouped = units >= 64;
		const long long blocks = (a.grouped ? (units + 7) / 8 * 8 : units) * nstrips;
		if (blocks > 0x7fffffffLL) {
			error("resize", "image too large");
			return -1;
		}
		const dim3 grid((unsigned int) blocks, 1, 1);
		switch (vs) {
		case 1:
			stream_launch<1>(a, p, grid, lds, dw, np, hf);
			break;
		case 2:
			stream_launch<2>(a, p, grid, lds, dw, np, hf);
			break;
		case 3:
			stream_launch<3>(a, p, grid, lds, dw, np, hf);
			break;
		case 4:
			stream_launch<4>(a, p, grid, lds, dw, np, hf);
			break;
		case 5:
			stream_launch<5>(a, p, grid, lds, dw, np, hf);
			break;
		case 6:
			stream_launch<6>(a, p, grid, lds, dw, np, hf);
			break;
		default:
			stream_launch<8>(a, p, grid, lds, dw, np, hf);
			break;
		}
		if (hipGetLastError() != hipSuccess) {
			error("resize", "kernel launch failed");
			return -1;
		}
	}
	return 1;
}

} // namespace vh
