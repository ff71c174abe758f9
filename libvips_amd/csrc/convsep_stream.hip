// Separable convolution on float images, both passes -- and optionally the colourspace
// conversion that follows -- in one streaming kernel (gfx950).  BASELINE config 3:
// vips_gaussblur(sigma 8) + vips_colourspace(sRGB -> Lab) on 32768 x 32768 x 3 float.
//
// vips_convsep (convsep.c:61-118) is conv(M) then conv(rot90 M, offset 0); vips_gaussblur
// (gaussblur.c:71-116) builds M with vips_gaussmat; on a float image both precisions sum in
// double (convi.c:721-741, convf.c:163-181): 2 x n double multiply-adds per element, so the
// pipeline is bound by the FP64 pipe, not by HBM (v_fma_f64 issues at 5 cycles per wave64 on
// this part; the f64 matrix instruction runs on the same pipe, a banded-Toeplitz recast wastes
// a third of it and it does not overlap with v_fma_f64: tools/mfma_f64_probe.hip).  The job of
// this kernel is to issue little else:
//
//   * a 768-thread block owns a strip of up to 768 row elements (x * bands + band) and streams
//     down a segment of rows, 8 rows per step, ONE barrier per step: in a step every thread
//     runs the vertical pass of the previous group of rows, then the horizontal pass of this
//     one (and the colour epilogue of the one before), on double-buffered LDS rows;
//   * horizontal: a thread makes 8 neighbouring same-band outputs of one row from a sliding
//     window of 12 doubles; the staged rows are planar per band in LDS, so the window arrives
//     as aligned ds_read_b128 and every element is converted to double once per thread
//     (4.5 reads and converts per output where the round-1 kernel had 8); the sum is rounded to
//     float exactly as the first operation's output image would be and crosses LDS once;
//   * vertical: a thread owns one element column and keeps the partial sums of the 32 output
//     rows in flight in REGISTERS (slot = output row mod 32): an intermediate value is read from
//     LDS once, converted once and fed to the n accumulators it belongs to (tap = row - output
//     row).  The rotation is static (the step bodies are unrolled over 32 rows), so registers
//     never move and the taps are scalar operands at fixed kernarg offsets.  No ring of
//     intermediate rows in LDS, nothing re-read;
//   * taps beyond n are never multiplied (0 * inf would poison a sum the reference never
//     touches): whole groups of 4 taps are compiled in (template NG), the last group's 1..4
//     taps are wave-uniform branches;
//   * epilogue (EPI): the blurred rows go to LDS instead of HBM, and the next step converts
//     them pixel by pixel with the colour route of colour_device.h and writes the final image:
//     the 12.9 GB intermediate image of C3 never exists.
//
// Arithmetic, bit-exact with the two- (three-) operation reference:
//   MODE 1  convi on float input (convi.c:721-741): double sum of (double) int coefficient *
//           pixel in mask order, sum / scale + offset, cast to float.  int * float is exact in
//           double (|c| < 2^29 checked on the host), so fma(c, v, sum) rounds exactly like the
//           reference's separate multiply and add; the division by the constant scale is
//           Markstein's correctly rounded 3-operation form.
//   MODE 2  convf (convf.c:163-181): sum seeded with the offset, coefficient = mask / scale
//           (double), separate multiply and add.
//   MODE 3  the library's default float mode (vips_hip_set_exact_float(0)) for either precision:
//           the sum is seeded with the offset, the coefficients are mask / scale in double and
//           every tap is one fused multiply-add -- no division, no separate offset addition, no
//           second rounding per tap.  The double sum differs from the reference's by a few
//           units of 2^-53 relative to its terms, so the float it rounds to is the reference's
//           except where the reference's sum lies that close to a rounding boundary, and then it
//           is the neighbouring float: within 1 ULP for sums that do not cancel (every
//           gaussian), the tolerance BASELINE.json's north_star grants float paths.  "That close"
//           is not rare: integer taps on pixels that are themselves short binary fractions (a
//           cast uchar image, or the first pass's float output) put the quotient sum / scale
//           EXACTLY on a float midpoint for about one element in a thousand, which the reference
//           (exact sum, correctly rounded division) breaks to even and any other arithmetic
//           breaks by its error's sign: 0.06 % of the elements of BASELINE config 3's blur differ
//           by 1 ULP (tools/c3_fast_diff.py).  Behind a colour route that truncates to a table
//           index (sRGB -> scRGB) such a pixel can land in the next table entry -- 21 of 268 M
//           pixels at 16384^2, a whole table step in Lab -- so MODE 3 is never used with the
//           colour epilogue: the fused blur + colourspace kernel always reproduces the
//           reference's bits.
// Each output sums its taps in mask order in both passes (rows arrive in tap order).
#include "colour_device.h"
#include "conv.h"
#include "convsep_int_body.h"
#include "convsep_int_host.h"

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace vh {

constexpr int SS_T = 8;       // rows per step = outputs per thread in the horizontal pass
constexpr int SS_SLOTS = 32;  // vertical accumulators = longest mask
constexpr int SS_SLACK = 32;  // staged pixels per plane beyond the strip's own (halo + window over-read)
constexpr int SS_LDS_MAX = 158 * 1024;
constexpr bool SS_INT_DEFAULT = true; // the integer horizontal pass unless $VIPS_HIP_STREAM_INT=0

struct StreamArgs {
	const float *in;
	float *out;
	long long in_stride, out_stride; // bytes
	int width, height, bands;
	int n, half, rem; // rem = taps in the last group of 4: n - 4 * (NG - 1)
	int w;            // strip width in elements: a multiple of 8 * bands, <= the block size
	int pxw;          // strip width in pixels
	int spw;          // staged pixels per (row, band) plane: pxw + SS_SLACK
	int step_rows, step_px; // block size / pxw and % pxw: the epilogue walks its items without dividing
	int ring;         // staged-row buffers in LDS (2..4): the LDS-DMA runs ring - 1 steps ahead
	int order;        // with an epilogue: 1 = request a phase's rows after the epilogue, 0 = before
	int strips, segs, seg_rows;
	int has_scale;    // scale != 1 (an integer for the kernel to test: a scalar compare and branch;
	                  // the f64 compare makes a lane mask that is kept -- spilled -- for every use)
	double scale, rscale;
	double offset1, offset2;
	int *counter;
	double coef[SS_SLOTS]; // taps in mask order
	// the integer horizontal pass (convsep_int_body.h): 0 = never, 1 = for waves whose windows hold
	// nothing but the integers 0 .. 255
	int int_h;
	float scale_f, rscale_f;
	unsigned int coefi[HINT_SETS];
};

template <int MODE>
static __device__ __forceinline__ double ss_mac(double s, double c, double v)
{
	if constexpr (MODE == 1 || MODE == 3)
		return __fma_rn(c, v, s);
	else
		return __dadd_rn(s, __dmul_rn(c, v));
}

// a / y for the constant y = a.scale, correctly rounded (Markstein): r = RN(1 / y)
static __device__ __forceinline__ double ss_div_scale(double s, const StreamArgs &a)
{
	const double q0 = __dmul_rn(s, a.rscale);
	const double e = __fma_rn(-a.scale, q0, s);
	const double q1 = __fma_rn(e, a.rscale, q0);
	// IEEE division's special cases (infinite or NaN sums) in one instruction
	return __builtin_amdgcn_div_fixup(q1, a.scale, s);
}

// what a pass stores
template <int MODE>
static __device__ __forceinline__ float ss_fin(double s, const StreamArgs &a, int pass)
{
	if constexpr (MODE == 1) {
		double q = s;
		int has_scale = a.has_scale;
		VH_SCALAR(has_scale); // (tested here, as a scalar: hoisted, the test is a lane mask that spills)
		if (has_scale) {
			asm volatile(""); // a wave-uniform branch, not two selects per output
			q = ss_div_scale(s, a);
		}
		// (+ offset even when it is zero: a branch around the addition -- it is a no-op for a zero
		// offset and a positive scale -- measured 3 % slower than the 16 additions per step)
		return (float) __dadd_rn(q, pass == 1 ? a.offset1 : a.offset2);
	}
	else
		return (float) s;
}

// One dword per lane from global memory straight into LDS: lane i's dword lands at LDS byte
// address lds_dst + 4 * i (lds_dst wave-uniform), read from src + voff (src wave-uniform, an SGPR
// pair; voff per lane).  M0 carries the LDS address and belongs to the compiler: saved, written
// and restored inside the one statement (cdna_hip_programming.md, LDS-DMA recipe).
static __device__ __forceinline__ void ss_dma_dword(const char *src, unsigned int voff, unsigned int lds_dst)
{
	VH_LDS_DMA_DWORD(src, voff, lds_dst);
}

typedef const double __attribute__((address_space(4))) *SsCoefs;
typedef float float4v __attribute__((ext_vector_type(4)));

// ---- horizontal pass: 8 same-band outputs from a window of 12 doubles.  The window is
// circular with static indices (the groups are unrolled): element e of the staged row lives in
// win[e mod 12], the quad that arrives during group g replaces the quad group g has just
// finished with -- no register moves.
template <int MODE, int NG>
static __device__ __forceinline__ void ss_hpass(const StreamArgs &a, const double *kc, const float *base, float *xd)
{
	double win[12], hacc[SS_T];
	{
		const float4v r0 = *reinterpret_cast<const float4v *>(base);
		const float4v r1 = *reinterpret_cast<const float4v *>(base + 4);
		const float4v r2 = *reinterpret_cast<const float4v *>(base + 8);
#pragma unroll
		for (int m = 0; m < 4; m++) {
			win[m] = (double) r0[m];
			win[4 + m] = (double) r1[m];
			win[8 + m] = (double) r2[m];
		}
	}
	const double seed = MODE >= 2 ? a.offset1 : 0.0;
#pragma unroll
	for (int k = 0; k < SS_T; k++)
		hacc[k] = seed;
#pragma unroll
	for (int g = 0; g < NG; g++) {
		float4v nx = { 0.0f, 0.0f, 0.0f, 0.0f };
		if (g + 1 < NG)
			nx = *reinterpret_cast<const float4v *>(base + 12 + 4 * g);
		// this group's 4 taps: two broadcast ds_read_b128 (the taps live in LDS, not in SGPRs:
		// 29 doubles held as scalars across the unrolled code spill into VGPR lanes and come
		// back one v_readlane per multiply-add -- 460 per phase, measured)
		double cg[4];
#pragma unroll
		for (int ii = 0; ii < 4; ii++)
			cg[ii] = kc[4 * g + ii];
		int rem = a.rem;
		if (g + 1 == NG)
			VH_SCALAR(rem); // (compared here as a scalar: hoisted, each test is a lane mask that spills)
#pragma unroll
		for (int ii = 0; ii < 4; ii++) {
			if (g + 1 < NG || ii < rem) { // whole groups unconditional, the last one tap by tap
				// (a real branch: if-converted, the absent taps cost 8 multiply-adds and 16 selects each)
				if (g + 1 == NG)
					asm volatile("");
				const double c = cg[ii];
#pragma unroll
				for (int k = 0; k < SS_T; k++)
					hacc[k] = ss_mac<MODE>(hacc[k], c, win[(4 * g + ii + k) % 12]);
			}
		}
		if (g + 1 < NG) {
			// elements 4g + 12 .. 4g + 15 take the places of 4g .. 4g + 3
#pragma unroll
			for (int m = 0; m < 4; m++)
				win[(4 * g + m) % 12] = (double) nx[m];
		}
		// one group's LDS read in flight at a time: left alone the scheduler hoists all ten
		// reads (40 registers) above the first multiply-add and the accumulators spill
		__builtin_amdgcn_sched_barrier(0);
	}
#pragma unroll
	for (int k = 0; k < SS_T; k++)
		xd[k * a.bands] = ss_fin<MODE>(hacc[k], a, 1);
}

// ---- horizontal pass on packed bytes (convsep_int_body.h) when every lane's window holds nothing
// but the integers 0 .. 255; false (wave-uniform, nothing stored) when one does not
static __device__ __forceinline__ bool ss_hpass_int(const StreamArgs &a, const unsigned int *kci, const float *base, float *xd)
{
	unsigned int w[9], bad = 0;
#pragma unroll
	for (int m = 0; m < 9; m++) {
		const float4v r = *reinterpret_cast<const float4v *>(base + 4 * m);
		w[m] = hint_pack4(r[0], r[1], r[2], r[3], bad);
		// (three window quads in flight at a time: the 32 vertical accumulators stay live across this)
		if (m % 3 == 2)
			__builtin_amdgcn_sched_barrier(0);
	}
	if (__builtin_amdgcn_ballot_w64(bad != 0)) {
		if (a.int_h != 2)
			return false;
		// ($VIPS_HIP_STREAM_INT=2, tests only: a refused window poisons its outputs instead of
		// taking the double path, so that a clean result proves which path made it)
#pragma unroll
		for (int k = 0; k < SS_T; k++)
			xd[k * a.bands] = __builtin_nanf("");
		return true;
	}
	hint_outputs(w, kci, a.scale_f, a.rscale_f, xd, a.bands);
	return true;
}

// ---- vertical pass of the 8 rows of step q (Q4 = q mod 4 fixes the slot rotation).
// Intermediate row m = 8q + r is tap d of output row m - d, kept in slot (m - d) mod 32.  A slot
// is handed to a new output row every 32 rows: the row that finished in it (output m - 32,
// complete since row m - 32 + n - 1) is rounded and stored right before the slot is seeded
// again, so the slot to retire is the static ROT whatever the mask length (a segment simply
// runs 32 rows past its last output row instead of n - 1).
template <int MODE, int NG, int EPI, int Q4>
static __device__ __forceinline__ void ss_vpass(const StreamArgs &a, const double *kc, double (&acc)[SS_SLOTS],
	const float *xs, float *os, int q, int rows_out, int y0, int e0, int t, bool store)
{
	float v[SS_T];
#pragma unroll
	for (int r = 0; r < SS_T; r++)
		v[r] = xs[r * a.w];
	// the taps, from their LDS table into registers for the 8 rows of the step (broadcast reads)
	double kr[4 * NG];
#pragma unroll
	for (int d = 0; d < 4 * NG; d++)
		kr[d] = kc[d];
	const double seed = MODE >= 2 ? a.offset2 : 0.0;
#pragma unroll
	for (int r = 0; r < SS_T; r++) {
		const int ROT = (Q4 * SS_T + r) & (SS_SLOTS - 1); // constant after unrolling
		const double dv = (double) v[r];

		const int j = q * SS_T + r - SS_SLOTS;
		if (j >= 0 && j < rows_out) {
			const float o = ss_fin<MODE>(acc[ROT], a, 2);
			if (EPI)
				os[r * a.w] = o;
			else if (store) {
				float *dst = reinterpret_cast<float *>(reinterpret_cast<char *>(a.out) + (long long) (y0 + j) * a.out_stride);
				dst[e0 + t] = o;
			}
		}
#pragma unroll
		for (int d = 0; d < 4 * (NG - 1); d++) {
			const int slot = (ROT - d) & (SS_SLOTS - 1);
			acc[slot] = ss_mac<MODE>(d == 0 ? seed : acc[slot], kr[d], dv);
		}
		int rem = a.rem;
		VH_SCALAR(rem); // (see ss_hpass)
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int d = 4 * (NG - 1) + k;
			const int slot = (ROT - d) & (SS_SLOTS - 1);
			if (k < rem) {
				asm volatile(""); // a real branch (see ss_hpass)
				acc[slot] = ss_mac<MODE>(d == 0 ? seed : acc[slot], kr[d], dv);
			}
		}
		__builtin_amdgcn_sched_barrier(0);
	}
}

// EPIF: the form of the colour epilogue -- 0: a thread's pixels one after the other, every wave before its
// vertical pass; 1: side by side and staggered over the waves of a SIMD (see there).  Form 1 is built, tested
// and measured for BASELINE config 3's instantiation (integer mask of 29 taps, sRGB -> Lab) only.
template <int MODE, int NG, int EPI, int SS_NT, int EPIF = 0>
__global__ void __launch_bounds__(SS_NT)
convsep_stream(StreamArgs a, RouteArgs route)
{
	VH_DYNAMIC_LDS(float, ss_lds);
	__shared__ int s_item;
	__shared__ __attribute__((aligned(16))) double s_coef[SS_SLOTS];
	__shared__ float s_v2y[EPI == 2 ? 256 : 1]; // the sRGB -> scRGB table of the spelled-out epilogue
	__shared__ __attribute__((aligned(16))) unsigned int s_coefi[MODE == 1 ? HINT_SETS : 1];
	const int in_row = a.bands * a.spw;   // floats per staged row (planar per band)
	const int in_buf = SS_T * in_row;
	const int x_buf = SS_T * a.w;
	float *s_in = ss_lds;                 // [ring][8][bands][spw]
	float *s_x = s_in + a.ring * in_buf;  // [2][8][w]   horizontal results, rounded to float
	float *s_o = s_x + 2 * x_buf;         // [2][8][w]   blurred rows for the epilogue (EPI)

	const int t = threadIdx.x;
	if (t < SS_SLOTS)
		s_coef[t] = a.coef[t];
	if (EPI == 2 && t < 256)
		s_v2y[t] = route.tables.v2Y_8[t];
	if (MODE == 1 && t < HINT_SETS)
		s_coefi[t] = a.coefi[t];
	const double *kc = s_coef; // (the first barrier of the work loop publishes it)
	const int E = a.width * a.bands;
	const int items = a.strips * a.segs;

	// horizontal pass: thread -> (row of the step, band, chunk of 8 pixels)
	const int chunks = a.w / SS_T;
	const int hrow = t / chunks;
	const int hc = t - hrow * chunks;
	const int hband = hc % a.bands;
	const int hpx0 = (hc / a.bands) * SS_T;
	const bool active = t < a.w;

	for (;;) {
		__syncthreads();
		if (t == 0)
			s_item = atomicAdd(a.counter, 1);
		__syncthreads();
		const int item = __builtin_amdgcn_readfirstlane(s_item);
		if (item >= items)
			return;
		const int strip = item % a.strips;
		const int seg = item / a.strips;
		const int e0 = strip * a.w;
		const int ne = min(a.w, E - e0);
		const int px_base = strip * a.pxw;
		const int y0 = seg * a.seg_rows;
		const int rows_out = min(a.seg_rows, a.height - y0);
		const int y_first = y0 - a.half;
		// the vertical pass stores output row j when row j + 32 arrives (ss_vpass)
		const int steps = (rows_out + SS_SLOTS + SS_T - 1) / SS_T;

		// Staging: the rows of a step go from HBM straight into LDS (global_load_lds_dword, no
		// registers, nothing for the compiler to spill or to wait on), planar per band.  LDS
		// position L of a staged row (band L / spw, staged pixel L mod spw) is filled by thread
		// L mod 768 -- the LDS-DMA writes a wave's 64 dwords contiguously, the SOURCE address is
		// per lane -- from the image column clamped to the image (vips_embed COPY).  At most
		// two positions per thread and row (bands * spw <= 2 * 768).
		const int staged = a.bands * a.spw;
		unsigned int goff[2]; // byte offsets in a source row
#pragma unroll
		for (int jj = 0; jj < 2; jj++) {
			const int L = min(t + jj * SS_NT, staged - 1);
			const int b = L / a.spw;
			const int sp = L - b * a.spw;
			const int px = min(max(px_base - a.half + sp, 0), a.width - 1);
			goff[jj] = (unsigned int) (px * a.bands + b) * 4u;
		}
		const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
		const bool first = wv * 64 < staged;                // wave-uniform: narrow strips stage < 768 positions
		const bool first_lane = t < staged;
		const bool second = SS_NT + wv * 64 < staged;       // wave-uniform
		const bool second_lane = SS_NT + t < staged;
		const unsigned int lds_in = VH_LDS_ADDR(s_in); // LDS byte address
		auto dma_rows = [&](int q, int slot) __attribute__((always_inline)) {
			// (opaque copies: hoisted out of the phase loop, the eight row offsets and the eight
			// first rows sit in scalar registers that spill, and every use is a v_readlane --
			// re-made here they are a scalar multiply and add each)
			int in_row_o = in_row, y_first_o = y_first;
			VH_SCALAR2(in_row_o, y_first_o);
			const unsigned int buf = lds_in + (unsigned int) (slot * in_buf + wv * 64) * 4u;
			// (the lane predicate once around the eight rows, not around each)
			if (first && first_lane) {
#pragma unroll
				for (int i = 0; i < SS_T; i++) {
					const int row = min(max(y_first_o + q * SS_T + i, 0), a.height - 1);
					const char *src = reinterpret_cast<const char *>(a.in) + (long long) row * a.in_stride;
					ss_dma_dword(src, goff[0], buf + (unsigned int) (i * in_row_o) * 4u);
				}
			}
			if (second && second_lane) {
#pragma unroll
				for (int i = 0; i < SS_T; i++) {
					const int row = min(max(y_first_o + q * SS_T + i, 0), a.height - 1);
					const char *src = reinterpret_cast<const char *>(a.in) + (long long) row * a.in_stride;
					ss_dma_dword(src, goff[1], buf + (unsigned int) (i * in_row_o + SS_NT) * 4u);
				}
			}
		};

		double acc[SS_SLOTS];
#pragma unroll
		for (int sl = 0; sl < SS_SLOTS; sl++)
			acc[sl] = 0.0;
		// the integer horizontal pass is tried until a window of this wave fails its test (a float
		// image proper fails in the first phase and pays for one test per work item)
		int int_ok = MODE == 1 ? a.int_h : 0;

		// The LDS-DMA runs ring - 1 steps ahead of the horizontal pass (one step ahead its
		// latency showed: 3.4 us per step of fixed cost whatever the mask).  hslot / dslot: the
		// ring slots the horizontal pass reads / the DMA fills in the coming phase.
		for (int q = 0; q < a.ring - 1 && q < steps; q++)
			dma_rows(q, q);
		int hslot = 0, dslot = a.ring - 1;
		const int phases = steps + (EPI ? 2 : 1);
		// One phase = what runs between two barriers.  The slot rotation of the vertical pass
		// has period 4 steps; four consecutive phases are spelled out so that it is static
		// (arms of a switch that differ only in a register index get merged into one body with
		// a dynamic index, i.e. scratch memory).
		// this thread's offsets into the staged rows / the exchange rows
		int h_in_off = hrow * in_row + hband * a.spw + hpx0;
		int h_x_off = hrow * a.w + hpx0 * a.bands + hband;
		int tt = t;
		const int er0 = t / a.pxw, ex0 = t - er0 * a.pxw; // epilogue item of this thread
		auto phase = [&](auto p4c, int p) __attribute__((always_inline)) {
			constexpr int Q4 = (decltype(p4c)::value + 3) & 3; // (p - 1) mod 4
			// The rows of step p were requested ring - 1 phases ago by LDS-DMA, which the compiler
			// does not count.  Memory operations complete in issue order, so "at most N
			// outstanding" with N = the DMA instructions this wave issued SINCE (8 or 16 per
			// phase) means the rows of step p -- and whatever stores came before them -- have
			// landed, while the younger DMA batches and stores stay in flight.
			// (near the end of a segment the younger batches do not exist: wait for everything)
			if (a.ring == 2 || p + a.ring - 2 >= steps)
				VH_WAIT_VMCNT(0);
			else if (a.ring == 3) {
				if (second)
					VH_WAIT_VMCNT(16);
				else
					VH_WAIT_VMCNT(8);
			}
			else {
				if (second)
					VH_WAIT_VMCNT(32);
				else
					VH_WAIT_VMCNT(16);
			}
			__syncthreads();
			// Keep the compiler from hoisting every derived address out of the phase loop: held
			// across the loop they spill; re-deriving them costs a few adds per phase.
			VH_VECTOR5(goff[0], goff[1], h_in_off, h_x_off, tt);
			// The rows of step p + ring - 1 start travelling into the buffer the horizontal pass of
			// step p - 1 read before the barrier.  With an epilogue they are requested AFTER it: the
			// epilogue's table gathers and stores wait with s_waitcnt vmcnt(0) -- the compiler does
			// not know about the LDS-DMA -- so requested first, every wave sat out the HBM latency of
			// the rows it had just asked for, once per phase.
			auto request_rows = [&]() __attribute__((always_inline)) {
				if (p + a.ring - 1 < steps)
					dma_rows(p + a.ring - 1, dslot);
				dslot = dslot + 1 == a.ring ? 0 : dslot + 1;
			};
			if (!EPI || a.order == 0)
				request_rows();
			// (the epilogue of step p - 2 reads the blurred rows the vertical pass wrote in the phase
			// before; the vertical pass below writes the other buffer)
			// ---- colour epilogue of step p - 2: blurred rows (LDS) -> the route -> the final image.
			// One pixel per item, items dealt round the block (a 4-pixels-per-thread form with
			// wide loads and stores measured slower: a third of the threads idle while the rest
			// run four conversions back to back; so did thread -> (row, every 96th pixel), which
			// needs no wrap-around arithmetic but makes every wave run three turns: +4 %).
			auto epilogue = [&]() __attribute__((always_inline)) {
				if (!(EPI && p >= 2 && p - 2 < steps))
					return;
				const int q = p - 2;
				const float *os = s_o + (q & 1) * x_buf;
				if constexpr (EPI == 2 && EPIF == 1) {
					// The items of a thread side by side (at most 3: 8 rows of <= 256 pixels dealt round
					// 768 threads): every table read of the three pixels is in flight before the first
					// is used.  One after the other, each item sat out its own LDS and L2 round trips
					// (about a microsecond) with every wave of the block in step behind the barrier.
					// An item beyond the step's rows or the strip computes on clamped indices and
					// stores nothing: no branch between the reads.
					constexpr int NI = 3;
					int ir[NI], ix[NI];
					bool ok[NI];
					{
						int r = er0, x = ex0, idx = tt;
						// (re-derived in every phase: hoisted out of the phase loop, the items' store
						// addresses are 64-bit pairs that spill)
						VH_VECTOR2(r, x);
#pragma unroll
						for (int i = 0; i < NI; i++) {
							if (x >= a.pxw) {
								x -= a.pxw;
								r++;
							}
							const int j = q * SS_T + r - SS_SLOTS;
							ok[i] = idx < SS_T * a.pxw && j >= 0 && j < rows_out && px_base + x < a.width;
							ir[i] = min(r, SS_T - 1);
							ix[i] = x;
							idx += SS_NT;
							r += a.step_rows;
							x += a.step_px;
						}
					}
					float fx[NI], fy[NI], fz[NI];
					float2 tx[NI], ty[NI], tz[NI];
#pragma unroll
					for (int i = 0; i < NI; i++) {
						const float *src = os + ir[i] * a.w + 3 * ix[i];
						Px v;
						v.a = s_v2y[load_as_uchar_like<float>(src[0], 255)];
						v.b = s_v2y[load_as_uchar_like<float>(src[1], 255)];
						v.c = s_v2y[load_as_uchar_like<float>(src[2], 255)];
						v = step_scRGB2XYZ(v);
						const int jx = cbrt_index_finite<0>(v.a, fx[i]);
						const int jy = cbrt_index_finite<1>(v.b, fy[i]);
						const int jz = cbrt_index_finite<2>(v.c, fz[i]);
						__builtin_memcpy(&tx[i], route.tables.cbrt + jx, sizeof(float2));
						__builtin_memcpy(&ty[i], route.tables.cbrt + jy, sizeof(float2));
						__builtin_memcpy(&tz[i], route.tables.cbrt + jz, sizeof(float2));
					}
					__builtin_amdgcn_sched_barrier(0);
					char *row0 = reinterpret_cast<char *>(a.out) + (long long) (y0 + q * SS_T - SS_SLOTS) * a.out_stride;
#pragma unroll
					for (int i = 0; i < NI; i++) {
						const float cbx = cbrt_finish(tx[i], fx[i]);
						const float cby = cbrt_finish(ty[i], fy[i]);
						const float cbz = cbrt_finish(tz[i], fz[i]);
						const float o0 = __fsub_rn(__fmul_rn(116.0F, cby), 16.0F);
						const float o1 = __fmul_rn(500.0F, __fsub_rn(cbx, cby));
						const float o2 = __fmul_rn(200.0F, __fsub_rn(cby, cbz));
						if (ok[i]) {
							const unsigned int off = (unsigned int) ir[i] * (unsigned int) a.out_stride + 12u * (unsigned int) (px_base + ix[i]);
							float *dst = reinterpret_cast<float *>(row0 + off);
							dst[0] = o0;
							dst[1] = o1;
							dst[2] = o2;
						}
					}
					return;
				}
				int r = er0, x = ex0; // (row, pixel) of item tt, stepping by one block of items
				for (int idx = tt; idx < SS_T * a.pxw; idx += SS_NT, r += a.step_rows, x += a.step_px) {
					if (x >= a.pxw) {
						x -= a.pxw;
						r++;
					}
					const int j = q * SS_T + r - SS_SLOTS;
					const int px = px_base + x;
					if (j >= 0 && j < rows_out && px < a.width) {
						const float *src = os + r * a.w + 3 * x;
						float o0, o1, o2;
						if constexpr (EPI == 2) {
							// sRGB -> scRGB -> XYZ -> Lab spelled out (what route_pixel does for these
							// steps, without its step loop and the scalars that loop keeps live)
							Px v;
							v.a = s_v2y[load_as_uchar_like<float>(src[0], 255)];
							v.b = s_v2y[load_as_uchar_like<float>(src[1], 255)];
							v.c = s_v2y[load_as_uchar_like<float>(src[2], 255)];
							// (table values: small finite numbers, so the quotients of XYZ2Lab need no
							// inf / NaN care)
							v = step_XYZ2Lab<true>(step_scRGB2XYZ(v), route.tables.cbrt);
							o0 = v.a;
							o1 = v.b;
							o2 = v.c;
						}
						else
							route_pixel<float, float>(route, route.tables.v2Y_8, route.tables.Y2v_8, src[0], src[1], src[2], o0, o1, o2);
						// one 12-byte store at a scalar base (the step's first row) + a 32-bit lane offset
						// (the host checked 8 rows of the output fit 32 bits)
						char *row0 = reinterpret_cast<char *>(a.out) + (long long) (y0 + q * SS_T - SS_SLOTS) * a.out_stride;
						const unsigned int off = (unsigned int) r * (unsigned int) a.out_stride + 12u * (unsigned int) px;
						float *dst = reinterpret_cast<float *>(row0 + off);
						dst[0] = o0;
						dst[1] = o1;
						dst[2] = o2;
					}
				}
			};
			// (form 1: the waves of a SIMD -- wave, wave + 4, wave + 8 -- do not all wait for their table reads
			// at once: two run the epilogue before their vertical pass, the third between its vertical and
			// its horizontal pass; the three stages of a phase are independent of each other.  Measured on
			// BASELINE config 3, profiles/r04_calls/r04p_c3.txt: pixels side by side alone 15.19 ms, one wave
			// first and two between 14.34, every wave between 13.85, two first and one between 13.08 --
			// against 13.68 for form 0)
			const bool epi_first = EPIF == 0 || wv < 8;
			if (EPI && a.order != 0 && !epi_first)
				request_rows();
			if (epi_first)
				epilogue();
			if (EPI && a.order != 0 && epi_first)
				request_rows();
			// ---- vertical pass of step p - 1
			if (p >= 1 && p - 1 < steps && active) {
				const int q = p - 1;
				ss_vpass<MODE, NG, EPI, Q4>(a, kc, acc, s_x + (q & 1) * x_buf + tt, s_o + (q & 1) * x_buf + tt, q,
					rows_out, y0, e0, tt, tt < ne);
			}
			if (!epi_first)
				epilogue();
			// ---- horizontal pass of step p
			if (p < steps && active) {
				bool done = false;
				if (MODE == 1 && int_ok) {
					done = ss_hpass_int(a, s_coefi, s_in + hslot * in_buf + h_in_off, s_x + (p & 1) * x_buf + h_x_off);
					int_ok = done;
				}
				if (!done)
					ss_hpass<MODE, NG>(a, kc, s_in + hslot * in_buf + h_in_off, s_x + (p & 1) * x_buf + h_x_off);
			}
			hslot = hslot + 1 == a.ring ? 0 : hslot + 1;
		};
		for (int p0 = 0; p0 < phases; p0 += 4) {
			phase(std::integral_constant<int, 0>{}, p0);
			if (p0 + 1 < phases)
				phase(std::integral_constant<int, 1>{}, p0 + 1);
			if (p0 + 2 < phases)
				phase(std::integral_constant<int, 2>{}, p0 + 2);
			if (p0 + 3 < phases)
				phase(std::integral_constant<int, 3>{}, p0 + 3);
		}
	}
}

template <int MODE, int NG, int EPI, int NT, int EPIF = 0>
static int ss_launch(const StreamArgs &a, const RouteArgs &route, size_t lds, int grid, const char *gate_name)
{
	if constexpr (MODE == 1 && NG == 8 && EPI == 2 && EPIF == 0) {
		// $VIPS_HIP_STREAM_EPI=0: the older form of the epilogue
		const char *form_env = getenv("VIPS_HIP_STREAM_EPI");
		if (!form_env || atoi(form_env) != 0)
			return ss_launch<MODE, NG, EPI, NT, 1>(a, route, lds, grid, gate_name);
	}
	static bool attr_done = false; // one attribute per instantiation
	if (!attr_done) {
		VH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&convsep_stream<MODE, NG, EPI, NT, EPIF>),
			hipFuncAttributeMaxDynamicSharedMemorySize, SS_LDS_MAX));
		attr_done = true;
	}
	Gate gate(gate_name);
	hipLaunchKernelGGL((convsep_stream<MODE, NG, EPI, NT, EPIF>), dim3(grid), dim3(NT), lds, stream(), a, route);
	VH_CHECK(hipGetLastError());
	return 0;
}

template <int MODE, int EPI, int NT>
static int ss_launch_ng(int ng, const StreamArgs &a, const RouteArgs &route, size_t lds, int grid, const char *gate_name)
{
	switch (ng) {
	case 1: return ss_launch<MODE, 1, EPI, NT>(a, route, lds, grid, gate_name);
	case 2: return ss_launch<MODE, 2, EPI, NT>(a, route, lds, grid, gate_name);
	case 3: return ss_launch<MODE, 3, EPI, NT>(a, route, lds, grid, gate_name);
	case 4: return ss_launch<MODE, 4, EPI, NT>(a, route, lds, grid, gate_name);
	case 5: return ss_launch<MODE, 5, EPI, NT>(a, route, lds, grid, gate_name);
	case 6: return ss_launch<MODE, 6, EPI, NT>(a, route, lds, grid, gate_name);
	case 7: return ss_launch<MODE, 7, EPI, NT>(a, route, lds, grid, gate_name);
	default: return ss_launch<MODE, 8, EPI, NT>(a, route, lds, grid, gate_name);
	}
}

template <int NT>
static int ss_launch_mode(bool integer, bool fast, int epi, int ng, const StreamArgs &a, const RouteArgs &route, size_t lds,
	int grid)
{
	const char *plain = integer ? "convsep_stream_convi" : "convsep_stream_convf";
	const char *colour = integer ? "convsep_stream_convi_colour" : "convsep_stream_convf_colour";
	if (fast && !epi)
		return ss_launch_ng<3, 0, NT>(ng, a, route, lds, grid, plain);
	if (integer) {
		if (epi == 2)
			return ss_launch_ng<1, 2, NT>(ng, a, route, lds, grid, colour);
		return epi ? ss_launch_ng<1, 1, NT>(ng, a, route, lds, grid, colour) : ss_launch_ng<1, 0, NT>(ng, a, route, lds, grid, plain);
	}
	if (epi == 2)
		return ss_launch_ng<2, 2, NT>(ng, a, route, lds, grid, colour);
	return epi ? ss_launch_ng<2, 1, NT>(ng, a, route, lds, grid, colour) : ss_launch_ng<2, 0, NT>(ng, a, route, lds, grid, plain);
}

// Both passes of a separable convolution on a float image (and, with route_steps, the colour
// route behind it: `out` then has the route's interpretation).  Returns 1 when the case is
// outside this kernel (the caller takes the older paths), 0 on success, -1 on error.
int convsep_stream_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c, double offset2,
	const int *route_steps, int n_route)
{
	if (getenv("VIPS_HIP_NO_STREAM_CONVSEP") || getenv("VIPS_HIP_NO_FUSED_CONVSEP"))
		return 1;
	const bool integer = c->precision == VIPS_HIP_PRECISION_INTEGER;
	if (in->format != VIPS_HIP_FORMAT_FLOAT || out->format != VIPS_HIP_FORMAT_FLOAT)
		return 1;
	if (c->mask_height != 1 || c->mask_width > SS_SLOTS || c->mask_width < 1 || c->nnz != c->mask_width)
		return 1;
	if (in->bands < 1 || in->bands > 4 || in->width != out->width || in->height != out->height || in->bands != out->bands)
		return 1;
	if ((long long) in->width * in->bands >= (1LL << 30))
		return 1;
	const bool epi = n_route > 0;
	if (epi && in->bands != 3)
		return 1;
	// the epilogue addresses the rows of a step with 32-bit offsets
	if (epi && (unsigned long long) out->stride * (SS_T + 1) > 0xffffffffULL)
		return 1;
	const int n = c->mask_width;

	StreamArgs a;
	a.in = (const float *) in->data;
	a.out = (float *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.height = in->height;
	a.bands = in->bands;
	a.n = n;
	a.half = n / 2;
	const int ng = (n + 3) / 4;
	a.rem = n - 4 * (ng - 1);
	for (int k = 0; k < SS_SLOTS; k++)
		a.coef[k] = 0.0;
	if (integer && c->scale_i == 0)
		return 1;
	// MODE 3 (see the head of the file) unless the caller asked for the reference's bits; masks
	// whose taps differ in sign can cancel, and a cancelling sum has no 1 ULP bound: exact arithmetic
	bool fast = !vips_hip_get_exact_float() && n_route == 0;
	for (int k = 0; k < n && fast; k++) {
		const double ck = integer ? (double) c->coeffi[k] : c->coefff[k];
		const double c0 = integer ? (double) c->coeffi[0] : c->coefff[0];
		if ((ck < 0.0) != (c0 < 0.0))
			fast = false;
	}
	for (int k = 0; k < n; k++) {
		if (integer) {
			// the product must be exact for the fused multiply-add to round like mul + add
			if (c->coeffi[k] >= (1 << 29) || c->coeffi[k] <= -(1 << 29))
				return 1;
			a.coef[k] = fast ? (double) c->coeffi[k] / (double) c->scale_i : (double) c->coeffi[k];
		}
		else
			a.coef[k] = c->coefff[k];
	}
	a.scale = integer ? (double) c->scale_i : 1.0;
	a.rscale = 1.0 / a.scale;
	a.has_scale = a.scale != 1.0;
	a.offset1 = integer ? (double) c->offset_i : c->offset;
	a.offset2 = integer ? (double) (int) rint(offset2) : offset2;
	a.int_h = 0;
	a.scale_f = a.rscale_f = 1.0f;
	memset(a.coefi, 0, sizeof(a.coefi));
	{
		// $VIPS_HIP_STREAM_INT: 0 = never take the integer horizontal pass, 1 = take it where it applies
		// (2: and poison what it refuses -- tests)
		const char *int_env = getenv("VIPS_HIP_STREAM_INT");
		HintTables ht;
		if (integer && !fast && (int_env ? atoi(int_env) != 0 : SS_INT_DEFAULT) &&
			hint_prepare(c->coeffi.data(), n, c->scale_i, c->offset_i, &ht)) {
			a.int_h = int_env && atoi(int_env) == 2 ? 2 : 1;
			a.scale_f = ht.scale;
			a.rscale_f = ht.rscale;
			memcpy(a.coefi, ht.coefi, sizeof(a.coefi));
		}
	}

	// threads per block = widest strip: 768 (168 registers per thread, 3 waves per SIMD; the
	// compiler spills ~130 registers outside the hot loops).  A 512-thread build (256 registers,
	// 2 waves per SIMD, no spills) measured 30-50 % slower (16384^2 x 3 float, sigma 8: blur 4.25
	// ms against 3.31, blur + sRGB->Lab 7.5 against 5.03): the kernel is bound by VALU issue and
	// hides its LDS and barrier stalls behind other waves, so the third wave per SIMD is worth
	// more than the spills cost.  Only the 768 build is instantiated.
	const int nt = 768;
	const int unit = SS_T * a.bands;
	const long long E = (long long) a.width * a.bands;
	long long w = nt / unit * unit;
	if (E < w)
		w = (E + unit - 1) / unit * unit;
	a.w = (int) w;
	a.pxw = a.w / a.bands;
	a.spw = a.pxw + SS_SLACK;
	a.step_rows = nt / a.pxw;
	a.step_px = nt % a.pxw;
	a.strips = (int) ((E + w - 1) / w);
	// ~4 work items per CU; segments long enough that the n - 1 warm-up rows and the pipeline
	// fill stay cheap, whole steps of 8 rows
	int want_segs = (256 * 4 + a.strips - 1) / a.strips;
	int seg_rows = (a.height + want_segs - 1) / want_segs;
	if (seg_rows < 16 * n)
		seg_rows = 16 * n;
	seg_rows = (seg_rows + SS_T - 1) / SS_T * SS_T;
	if (seg_rows > a.height)
		seg_rows = a.height;
	a.seg_rows = seg_rows;
	a.segs = (a.height + seg_rows - 1) / seg_rows;
	const int items = a.strips * a.segs;

	RouteArgs route;
	memset(&route, 0, sizeof(route));
	if (epi && colour_route_prepare(route_steps, n_route, &route))
		return -1;

	const size_t in_bytes = (size_t) SS_T * a.bands * a.spw * sizeof(float);
	const size_t x_bytes = (size_t) 2 * SS_T * a.w * (epi ? 2 : 1) * sizeof(float);
	const char *ring_env = getenv("VIPS_HIP_STREAM_RING");
	int ring = ring_env ? atoi(ring_env) : 4;
	ring = ring < 2 ? 2 : ring > 4 ? 4 : ring;
	while (ring > 2 && ring * in_bytes + x_bytes > (size_t) SS_LDS_MAX)
		ring--;
	a.ring = ring;
	{
		const char *order_env = getenv("VIPS_HIP_STREAM_ORDER");
		a.order = order_env ? atoi(order_env) : 1;
	}
	const size_t lds = ring * in_bytes + x_bytes;
	if (lds > (size_t) SS_LDS_MAX)
		return 1;

	int *counter = (int *) vips_hip_malloc(sizeof(int));
	if (!counter)
		return -1;
	a.counter = counter;
	if (hipMemsetAsync(counter, 0, sizeof(int), stream()) != hipSuccess) {
		vips_hip_free(counter);
		return hip_failed(hipErrorUnknown, "hipMemsetAsync");
	}
	const int grid = items < 256 ? items : 256;
	// the route of BASELINE config 3 has its own build of the epilogue
	const int epi_kind = !epi ? 0
		: (n_route == 3 && route_steps[0] == VIPS_HIP_COLOUR_sRGB2scRGB && route_steps[1] == VIPS_HIP_COLOUR_scRGB2XYZ &&
			  route_steps[2] == VIPS_HIP_COLOUR_XYZ2Lab)
		? 2
		: 1;
	const int r = ss_launch_mode<768>(integer, fast, epi_kind, ng, a, route, lds, grid);
	vips_hip_free(counter);
	return r;
}

} // namespace vh
