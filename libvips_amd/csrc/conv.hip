// convi (C path) / convf generates for gfx950, plus vips_gaussmat.
//
// Arithmetic contracts (reference: libvips/convolution):
//   convi, integer input   sum(int64) of coeff[i] * p[off[i]] over the non-zero mask
//                          elements in row-major order, (sum + scale/2) / scale + offset
//                          with C (truncating) division, clip to the format
//                          (convi.c:698-716, 743-751); coeff = rint(mask), scale/offset =
//                          rint() of the mask's scale/offset (convi.c:760-762,886-894)
//   convi, float input     double sum, sum / scale + offset           (convi.c:721-741)
//   convf                  double sum seeded with offset, coeff = mask / scale,
//                          separate mul and add in mask order; integer inputs give
//                          float output                         (convf.c:163-181,300-355)
// The "convolution" is a correlation: out(x, y) = sum mask(i, j) * in(x + i - mw/2,
// y + j - mh/2), input coordinates clamped to the image, which is the
// vips_embed(VIPS_EXTEND_COPY) of convi.c:1142-1147 / convf.c:331-336.
#include "conv.h"

#include <climits>
#include <type_traits>
#include <cmath>

namespace vh {

struct ConvArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int in_left, in_top, im_width, im_height;
	int in_right; // in_left + window width
	int out_left, out_top, out_width, out_height;
	int epp;
	int nnz;
	int half_w, half_h;
	const void *coeff;
	const void *dense;
	const double *dense8; // convf: rows padded with zeros to np8 = roundup(mask_width, 8) + 16 doubles
	int np8;
	int mask_width, mask_height;
	const short *dx, *dy;
	int scale_i, rounding, offset_i;
	double offset;
	int narrow; // integer conv on 8/16-bit pixels whose sums fit 32 bits
};

template <typename T>
struct ConvClip;
#define CONV_CLIP(TYPE, LO, HI) \
	template <> \
	struct ConvClip<TYPE> { \
		static __device__ __forceinline__ TYPE run(long long v) \
		{ \
			v = v > (long long) (HI) ? (long long) (HI) : v; \
			v = v < (long long) (LO) ? (long long) (LO) : v; \
			return (TYPE) v; \
		} \
	};
CONV_CLIP(unsigned char, 0, UCHAR_MAX)
CONV_CLIP(signed char, SCHAR_MIN, SCHAR_MAX)
CONV_CLIP(unsigned short, 0, USHRT_MAX)
CONV_CLIP(short, SHRT_MIN, SHRT_MAX)
#undef CONV_CLIP
// CLIP_NONE: the int64 is assigned straight to the 32-bit type (convi.c:832-838)
template <>
struct ConvClip<unsigned int> {
	static __device__ __forceinline__ unsigned int run(long long v) { return (unsigned int) v; }
};
template <>
struct ConvClip<int> {
	static __device__ __forceinline__ int run(long long v) { return (int) v; }
};

// MODE 0: convi on an integer format.  MODE 1: convi on float/double.  MODE 2: convf.
template <typename TIN, typename TOUT, int MODE>
__global__ void __launch_bounds__(256)
conv_general(ConvArgs a)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.out_width * a.epp)
		return;
	const int x = e / a.epp;
	const int b = e - x * a.epp;
	const int gx = a.out_left + x - a.half_w;
	for (int y = blockIdx.y; y < a.out_height; y += gridDim.y) {
		const int gy = a.out_top + y - a.half_h;
		TOUT *dst = (TOUT *) (a.out + (long long) y * a.out_stride);
		if constexpr (MODE == 0) {
			const int *c = (const int *) a.coeff;
			long long sum = 0;
			for (int i = 0; i < a.nnz; i++) {
				const int col = min(max(gx + a.dx[i], 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + a.dy[i], 0), a.im_height - 1) - a.in_top;
				const TIN *src = (const TIN *) (a.in + row * a.in_stride);
				sum += (long long) c[i] * (long long) src[(long long) col * a.epp + b];
			}
			sum = ((sum + a.rounding) / a.scale_i) + a.offset_i;
			dst[e] = (TOUT) ConvClip<TIN>::run(sum);
		}
		else if constexpr (MODE == 4) {
			const int *c = (const int *) a.coeff;
			int sum = a.rounding;
			for (int i = 0; i < a.nnz; i++) {
				const int col = min(max(gx + a.dx[i], 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + a.dy[i], 0), a.im_height - 1) - a.in_top;
				const TIN *src = (const TIN *) (a.in + row * a.in_stride);
				sum += c[i] * (int) src[(long long) col * a.epp + b];
			}
			dst[e] = (TOUT) ConvClip<TIN>::run((long long) ((sum >> a.scale_i) + a.offset_i));
		}
		else if constexpr (MODE == 1) {
			const int *c = (const int *) a.coeff;
			double sum = 0;
			for (int i = 0; i < a.nnz; i++) {
				const int col = min(max(gx + a.dx[i], 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + a.dy[i], 0), a.im_height - 1) - a.in_top;
				const TIN *src = (const TIN *) (a.in + row * a.in_stride);
				sum = __dadd_rn(sum, __dmul_rn((double) c[i], (double) src[(long long) col * a.epp + b]));
			}
			sum = __dadd_rn(__ddiv_rn(sum, (double) a.scale_i), (double) a.offset_i);
			dst[e] = (TOUT) sum;
		}
		else {
			const double *c = (const double *) a.coeff;
			double sum = a.offset;
			for (int i = 0; i < a.nnz; i++) {
				const int col = min(max(gx + a.dx[i], 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + a.dy[i], 0), a.im_height - 1) - a.in_top;
				const TIN *src = (const TIN *) (a.in + row * a.in_stride);
				sum = __dadd_rn(sum, __dmul_rn(c[i], (double) src[(long long) col * a.epp + b]));
			}
			dst[e] = (TOUT) sum;
		}
	}
}

template <typename TIN, typename TOUT, int MODE>
static int launch_conv(const ConvArgs &a, const char *gate_name)
{
	const int ne = a.out_width * a.epp;
	dim3 block(256, 1, 1);
	dim3 grid((ne + 255) / 256, a.out_height < 32768 ? a.out_height : 32768, 1);
	Gate gate(gate_name);
	hipLaunchKernelGGL((conv_general<TIN, TOUT, MODE>), grid, block, 0, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}


// ---------------------------------------------------------- register-tiled kernels
//
// conv_general re-reads every tap from memory for every output.  Here a thread makes
// TILE outputs that are neighbours along the mask's long axis (same band), so one load
// feeds TILE multiply-adds: a sliding window of TILE inputs lives in registers and the tap
// loop slides it one pixel per tap.  The window's slots rotate with the tap index, so
// the loop is unrolled TILE-fold with static register names (no moves).
//   ALONG_Y = false: any mask, tile along x (each mask row restarts the window)
//   ALONG_Y = true:  n x 1 masks (the second pass of convsep / gaussblur), tile along y,
//                    perfectly coalesced: adjacent threads are adjacent elements of a row
// Every output still sums its non-zero taps in row-major mask order, so results are
// bit-identical to conv_general (and to the reference).
constexpr int CONV_TILE = 8;

template <typename TIN, int MODE>
struct ConvAcc;
template <typename TIN>
struct ConvAcc<TIN, 0> {
	typedef long long acc_t;
	typedef int coef_t;
	typedef long long win_t;
	static __device__ __forceinline__ acc_t seed(const ConvArgs &) { return 0; }
	static __device__ __forceinline__ win_t widen(TIN v) { return (long long) v; }
	static __device__ __forceinline__ acc_t mac(acc_t s, int c, long long v) { return s + (long long) c * v; }
	static __device__ __forceinline__ TIN fin(acc_t s, const ConvArgs &a)
	{
		s = ((s + a.rounding) / a.scale_i) + a.offset_i;
		return ConvClip<TIN>::run(s);
	}
};
// MODE 3: MODE 0 for 8- and 16-bit pixels when every partial sum fits 32 bits (checked on the
// host): 24-bit multiply-adds at full rate instead of 64-bit ones at a quarter of it.  Same
// integers, same truncating division, so the same result.
template <typename TIN>
struct ConvAcc<TIN, 3> {
	typedef int acc_t;
	typedef int coef_t;
	typedef int win_t;
	static __device__ __forceinline__ acc_t seed(const ConvArgs &) { return 0; }
	static __device__ __forceinline__ win_t widen(TIN v) { return (int) v; }
	static __device__ __forceinline__ acc_t mac(acc_t s, int c, int v) { return s + __mul24(c, v); }
	static __device__ __forceinline__ TIN fin(acc_t s, const ConvArgs &a)
	{
		s = ((s + a.rounding) / a.scale_i) + a.offset_i;
		return ConvClip<TIN>::run((long long) s);
	}
};
// MODE 4: the Highway variant of convi on uchar (convi_hwy.cpp:264-273): int32 sum of pixel x
// 8-bit mantissa seeded with the rounding term, ARITHMETIC shift by the shared exponent (a floor,
// where the C path's division truncates), offset, clip.  scale_i carries the exponent.
template <typename TIN>
struct ConvAcc<TIN, 4> {
	typedef int acc_t;
	typedef int coef_t;
	typedef int win_t;
	static __device__ __forceinline__ acc_t seed(const ConvArgs &a) { return a.rounding; }
	static __device__ __forceinline__ win_t widen(TIN v) { return (int) v; }
	static __device__ __forceinline__ acc_t mac(acc_t s, int c, int v) { return s + __mul24(c, v); }
	static __device__ __forceinline__ TIN fin(acc_t s, const ConvArgs &a)
	{
		return ConvClip<TIN>::run((long long) ((s >> a.scale_i) + a.offset_i));
	}
};
template <typename TIN>
struct ConvAcc<TIN, 1> {
	typedef double acc_t;
	typedef int coef_t;
	typedef double win_t;
	static __device__ __forceinline__ acc_t seed(const ConvArgs &) { return 0.0; }
	static __device__ __forceinline__ win_t widen(TIN v) { return (double) v; }
	static __device__ __forceinline__ acc_t mac(acc_t s, int c, double v)
	{
		return __dadd_rn(s, __dmul_rn((double) c, v));
	}
	static __device__ __forceinline__ TIN fin(acc_t s, const ConvArgs &a)
	{
		return (TIN) __dadd_rn(__ddiv_rn(s, (double) a.scale_i), (double) a.offset_i);
	}
};
template <typename TIN>
struct ConvAcc<TIN, 2> {
	typedef double acc_t;
	typedef double coef_t;
	typedef double win_t;
	static __device__ __forceinline__ acc_t seed(const ConvArgs &a) { return a.offset; }
	static __device__ __forceinline__ win_t widen(TIN v) { return (double) v; }
	static __device__ __forceinline__ acc_t mac(acc_t s, double c, double v)
	{
		return __dadd_rn(s, __dmul_rn(c, v));
	}
	static __device__ __forceinline__ double fin(acc_t s, const ConvArgs &) { return s; }
};

template <typename TIN, typename TOUT, int MODE, bool ALONG_Y>
__global__ void __launch_bounds__(256)
conv_tiled(ConvArgs a)
{
	typedef ConvAcc<TIN, MODE> A;
	typedef typename A::acc_t acc_t;
	typedef typename A::coef_t coef_t;
	typedef typename A::win_t win_t;
	constexpr int T = CONV_TILE;
	const coef_t *__restrict__ dense = (const coef_t *) a.dense;

	// thread -> (tile index along the tiled axis, element across it)
	int e, x0, y0;
	if (ALONG_Y) {
		e = blockIdx.x * blockDim.x + threadIdx.x; // element of the row: x * epp + b
		if (e >= a.out_width * a.epp)
			return;
		x0 = e / a.epp;
		y0 = blockIdx.y * T;
	}
	else {
		const int id = blockIdx.x * blockDim.x + threadIdx.x;
		const int tiles = (a.out_width + T - 1) / T;
		if (id >= tiles * a.epp)
			return;
		const int tile = id / a.epp;
		e = id - tile * a.epp; // band
		x0 = tile * T;
		y0 = blockIdx.y;
	}
	const int b = ALONG_Y ? e - x0 * a.epp : e;

	acc_t acc[T];
#pragma unroll
	for (int k = 0; k < T; k++)
		acc[k] = A::seed(a);

	// input coordinates of tap (0, 0) of output (x0, y0)
	const int gx = a.out_left + x0 - a.half_w;
	const int gy = a.out_top + y0 - a.half_h;

	auto fetch = [&](int col, int row) -> win_t {
		const int cc = min(max(col, 0), a.im_width - 1) - a.in_left;
		const int rr = min(max(row, 0), a.im_height - 1) - a.in_top;
		const TIN *src = (const TIN *) (a.in + rr * a.in_stride);
		return A::widen(src[(long long) cc * a.epp + b]);
	};

	const int n_long = ALONG_Y ? a.mask_height : a.mask_width;
	const int n_short = ALONG_Y ? 1 : a.mask_height;
	for (int j = 0; j < n_short; j++) {
		// window slot s holds input (tap + k) with s = (tap + k) % T
		win_t w[T];
#pragma unroll
		for (int k = 0; k < T - 1; k++)
			w[k] = ALONG_Y ? fetch(gx, gy + k) : fetch(gx + k, gy + j);
		for (int i0 = 0; i0 < n_long; i0 += T) {
#pragma unroll
			for (int ii = 0; ii < T; ii++) {
				const int i = i0 + ii;
				if (i < n_long) {
					// the newest element of this tap's window goes into slot (ii + T - 1) % T
					w[(ii + T - 1) % T] =
						ALONG_Y ? fetch(gx, gy + i + T - 1) : fetch(gx + i + T - 1, gy + j);
					const coef_t c = ALONG_Y ? dense[i] : dense[j * a.mask_width + i];
					if (c != 0) {
#pragma unroll
						for (int k = 0; k < T; k++)
							acc[k] = A::mac(acc[k], c, w[(ii + k) % T]);
					}
				}
			}
		}
	}

#pragma unroll
	for (int k = 0; k < T; k++) {
		const int ox = ALONG_Y ? x0 : x0 + k;
		const int oy = ALONG_Y ? y0 + k : y0;
		if (ox < a.out_width && oy < a.out_height) {
			TOUT *dst = (TOUT *) (a.out + (long long) oy * a.out_stride);
			dst[(long long) ox * a.epp + b] = (TOUT) A::fin(acc[k], a);
		}
	}
}

template <typename TIN, typename TOUT, int MODE>
static int launch_conv_tiled(const ConvArgs &a, const char *gate_name)
{
	dim3 block(256, 1, 1);
	Gate gate(gate_name);
	if (a.mask_width == 1) {
		const int ne = a.out_width * a.epp;
		dim3 grid((ne + 255) / 256, (a.out_height + CONV_TILE - 1) / CONV_TILE, 1);
		if (grid.y > 65535)
			return 1;
		hipLaunchKernelGGL((conv_tiled<TIN, TOUT, MODE, true>), grid, block, 0, stream(), a);
	}
	else {
		const int ids = ((a.out_width + CONV_TILE - 1) / CONV_TILE) * a.epp;
		dim3 grid((ids + 255) / 256, a.out_height, 1);
		if (grid.y > 65535)
			return 1;
		hipLaunchKernelGGL((conv_tiled<TIN, TOUT, MODE, false>), grid, block, 0, stream(), a);
	}
	VH_CHECK(hipGetLastError());
	return 0;
}

// convf on an integer image, taps in groups of 8: the 8 coefficients of a group are one scalar
// load, and the 8 window elements of the NEXT group are fetched before the current group's 128
// multiplies and adds are issued, so a load's latency hides behind arithmetic of the same wave
// (conv_tiled waits for one load per tap: 45 % of the FP64 mul+add rate on C5).  The window is
// two groups of 8 values whose roles swap every group (two groups per loop trip keep the
// register names static).  Zero taps are multiplied through instead of skipped: for integer
// pixels c * v = 0 exactly and sum + 0 = sum, so the result equals the reference's sum over
// its squeezed non-zero taps bit for bit.  Float images keep conv_tiled (0 * inf).
template <typename TIN, int EPP, bool INTERIOR>
struct ConvGroupLoad {
	// elements gx + i0 .. gx + i0 + 7 of row `row` (band b).  INTERIOR (the whole block: no tap,
	// nor the two groups read ahead, leaves the window): step a pointer, immediate offsets (EPP
	// is a template argument).  Otherwise clamp every column to [0, im_width) and to the last
	// element any tap of the thread needs (the window holds no more).  The choice is made per
	// wave, outside the loops, so that no load sits in a divergent branch (a branch join
	// forces the wait for the loads just issued, i.e. no prefetch).
	static __device__ __forceinline__ void run(const ConvArgs &a, const TIN *row, int b, int gx, int i0,
		int last, int e_base, TIN (&raw)[8])
	{
		const int epp = EPP ? EPP : a.epp;
		if (INTERIOR) {
			const TIN *p = row + (e_base + i0 * epp);
#pragma unroll
			for (int m = 0; m < 8; m++)
				raw[m] = p[m * epp];
		}
		else {
#pragma unroll
			for (int m = 0; m < 8; m++) {
				const int col = min(gx + i0 + m, last);
				const int cc = min(max(col, 0), a.im_width - 1) - a.in_left;
				raw[m] = row[cc * epp + b];
			}
		}
	}
};

template <typename TIN, int EPP, bool INTERIOR, bool FUSED>
static __device__ __forceinline__ void convf_grouped_rows(const ConvArgs &a, double (&acc)[CONV_TILE], int b,
	int gx, int gy, int last, int e_base)
{
	constexpr int T = CONV_TILE;
	typedef ConvGroupLoad<TIN, EPP, INTERIOR> Load;
	const int n = a.mask_width;
	for (int j = 0; j < a.mask_height; j++) {
		const int rr = min(max(gy + j, 0), a.im_height - 1) - a.in_top;
		const TIN *row = (const TIN *) (a.in + rr * a.in_stride);
		const double *crow = a.dense8 + (size_t) j * a.np8;
		double g0[8], g1[8];
		TIN raw[8];
		Load::run(a, row, b, gx, 0, last, e_base, raw);
#pragma unroll
		for (int m = 0; m < 8; m++)
			g0[m] = (double) raw[m];
		Load::run(a, row, b, gx, 8, last, e_base, raw);
#pragma unroll
		for (int m = 0; m < 8; m++)
			g1[m] = (double) raw[m];
		// the taps too are fetched a group ahead (scalar loads into SGPR operands)
		double c0[8], c1[8];
#pragma unroll
		for (int m = 0; m < 8; m++)
			c0[m] = crow[m];
		for (int i0 = 0; i0 < n; i0 += 16) {
#pragma unroll
			for (int m = 0; m < 8; m++)
				c1[m] = crow[i0 + 8 + m];
			Load::run(a, row, b, gx, i0 + 16, last, e_base, raw);
#pragma unroll
			for (int ii = 0; ii < 8; ii++)
#pragma unroll
				for (int k = 0; k < T; k++)
					acc[k] = FUSED ? __fma_rn(c0[ii], ii + k < 8 ? g0[ii + k] : g1[ii + k - 8], acc[k])
								   : __dadd_rn(acc[k], __dmul_rn(c0[ii], ii + k < 8 ? g0[ii + k] : g1[ii + k - 8]));
#pragma unroll
			for (int m = 0; m < 8; m++)
				g0[m] = (double) raw[m];
			if (i0 + 8 >= n)
				break;
#pragma unroll
			for (int m = 0; m < 8; m++)
				c0[m] = crow[i0 + 16 + m];
			Load::run(a, row, b, gx, i0 + 24, last, e_base, raw);
#pragma unroll
			for (int ii = 0; ii < 8; ii++)
#pragma unroll
				for (int k = 0; k < T; k++)
					acc[k] = FUSED ? __fma_rn(c1[ii], ii + k < 8 ? g1[ii + k] : g0[ii + k - 8], acc[k])
								   : __dadd_rn(acc[k], __dmul_rn(c1[ii], ii + k < 8 ? g1[ii + k] : g0[ii + k - 8]));
#pragma unroll
			for (int m = 0; m < 8; m++)
				g1[m] = (double) raw[m];
		}
	}
}

// FUSED: v_fma_f64 instead of v_mul_f64 + v_add_f64 -- one rounding per tap instead of two, twice
// the DP instruction rate; the float output then differs from the reference's by at most 1 ULP
// (vips_hip_set_exact_float, include/vips_hip.h).
template <typename TIN, typename TOUT, int EPP, bool FUSED>
__global__ void __launch_bounds__(256)
convf_grouped(ConvArgs a)
{
	constexpr int T = CONV_TILE;
	const int epp = EPP ? EPP : a.epp;
	const int tiles = (a.out_width + T - 1) / T;
	const int id_lo = blockIdx.x * blockDim.x;
	const int id = id_lo + threadIdx.x;
	if (id >= tiles * epp)
		return;
	const int tile = id / epp;
	const int b = id - tile * epp;
	const int x0 = tile * T;
	const int y0 = blockIdx.y;

	double acc[T];
#pragma unroll
	for (int k = 0; k < T; k++)
		acc[k] = a.offset;

	const int gx = a.out_left + x0 - a.half_w;
	const int gy = a.out_top + y0 - a.half_h;
	const int last = gx + a.mask_width - 1 + T - 1;
	const int e_base = (gx - a.in_left) * epp + b;
	// one choice per wave (a wave-uniform branch): is every lane interior?  The loop reads up
	// to 16 elements past the last tap's window.
	const bool mine = gx >= a.in_left && gx >= 0 && last + 16 < a.in_right && last + 16 < a.im_width;
	if (__all(mine))
		convf_grouped_rows<TIN, EPP, true, FUSED>(a, acc, b, gx, gy, last, e_base);
	else
		convf_grouped_rows<TIN, EPP, false, FUSED>(a, acc, b, gx, gy, last, e_base);

#pragma unroll
	for (int k = 0; k < T; k++) {
		const int ox = x0 + k;
		if (ox < a.out_width) {
			TOUT *dst = (TOUT *) (a.out + (long long) y0 * a.out_stride);
			dst[(long long) ox * epp + b] = (TOUT) acc[k];
		}
	}
}


// convf on a one-band 8/16-bit image, masks 8..64 wide (BASELINE config 5: 31x31 on ushort):
// a thread makes 8 x R outputs (8 along x, R rows), a block 2048 x R.  Input rows stream
// through LDS one at a time (staged once per block with clamped columns, double-buffered, one
// barrier per row): a thread reads its window for a group of 8 taps as two aligned
// ds_read_b128 instead of eight scattered global loads, converts it once and uses it for the
// R output rows the input row belongs to (mask row j = q - i for output row i), so
// per input row a thread issues ~8 global loads and ~60 converts for up to 256 * R double
// multiply-adds.  The taps of (mask row, group) are 8 scalar operands fetched a step ahead.
// Every output still sums its taps in row-major mask order (zero taps multiplied through: exact
// for integer pixels), so with FUSED = false the result is the reference's bit for bit
// (convf.c:163-181); FUSED = true uses v_fma_f64 (see convf_grouped).
constexpr int CONVF_LDS_SPAN = 256 * 8;   // output columns per block
constexpr int CONVF_LDS_MAXW = 64;        // widest mask
constexpr int CONVF_LDS_WIN = CONVF_LDS_SPAN + CONVF_LDS_MAXW + 16; // staged elements per row

template <typename TIN, int R, bool FUSED>
__global__ void __launch_bounds__(256)
convf_rows_lds(ConvArgs a)
{
	constexpr int T = 8;
	__shared__ __attribute__((aligned(16))) TIN rows[2][CONVF_LDS_WIN];
	const int t = threadIdx.x;
	const int bx0 = blockIdx.x * CONVF_LDS_SPAN; // first output column of the block (region coords)
	const int y0 = blockIdx.y * R;
	const int gx = a.out_left + bx0 - a.half_w; // image column of staged element 0
	const int gy = a.out_top + y0 - a.half_h;   // image row of step q = 0
	const int groups = (a.mask_width + 7) / 8;
	const int win = CONVF_LDS_SPAN + 8 * groups + 8; // staged elements a row needs (incl. group over-read)
	// input rows the block's output rows need: a last block of fewer than R rows stops early, so
	// that in region mode nothing below the window's last needed row is read
	const int steps = a.mask_height + min(R, a.out_height - y0) - 1;

	double acc[R][T];
#pragma unroll
	for (int i = 0; i < R; i++)
#pragma unroll
		for (int k = 0; k < T; k++)
			acc[i][k] = a.offset;

	// this thread's share of a staged row: elements 8t .. 8t+7, and (t < 16) 2048 + 8t .. + 7
	int col[8], colx[8];
#pragma unroll
	for (int m = 0; m < 8; m++) {
		col[m] = min(max(gx + 8 * t + m, 0), a.im_width - 1) - a.in_left;
		colx[m] = min(max(gx + CONVF_LDS_SPAN + 8 * t + m, 0), a.im_width - 1) - a.in_left;
		// never outside the window (elements past the last tap multiply zero coefficients)
		col[m] = min(max(col[m], 0), a.in_right - a.in_left - 1);
		colx[m] = min(max(colx[m], 0), a.in_right - a.in_left - 1);
	}
	const bool extra = CONVF_LDS_SPAN + 8 * t < win;
	TIN stage[8], stagex[8];
	auto fetch = [&](int q) {
		const int rr = min(max(gy + q, 0), a.im_height - 1) - a.in_top;
		const TIN *row = (const TIN *) (a.in + (long long) rr * a.in_stride);
#pragma unroll
		for (int m = 0; m < 8; m++)
			stage[m] = row[col[m]];
		if (extra) {
#pragma unroll
			for (int m = 0; m < 8; m++)
				stagex[m] = row[colx[m]];
		}
	};
	fetch(0);
	for (int q = 0; q < steps; q++) {
		TIN *buf = rows[q & 1];
#pragma unroll
		for (int m = 0; m < 8; m++)
			buf[8 * t + m] = stage[m];
		if (extra) {
#pragma unroll
			for (int m = 0; m < 8; m++)
				buf[CONVF_LDS_SPAN + 8 * t + m] = stagex[m];
		}
		__syncthreads();
		if (q + 1 < steps)
			fetch(q + 1);
		for (int tg = 0; tg < groups; tg++) {
			// window elements 8t + 8tg .. + 15 (two aligned 16-byte reads for 16-bit pixels)
			double d[16];
#pragma unroll
			for (int m = 0; m < 16; m++)
				d[m] = (double) buf[8 * t + 8 * tg + m];
#pragma unroll
			for (int i = 0; i < R; i++) {
				const int j = q - i; // mask row this input row is for output row i
				if (j >= 0 && j < a.mask_height) {
					const double *crow = a.dense8 + (size_t) j * a.np8 + 8 * tg;
					double c[8];
#pragma unroll
					for (int m = 0; m < 8; m++)
						c[m] = crow[m];
#pragma unroll
					for (int ii = 0; ii < 8; ii++)
#pragma unroll
						for (int k = 0; k < T; k++)
							acc[i][k] = FUSED ? __fma_rn(c[ii], d[ii + k], acc[i][k])
											  : __dadd_rn(acc[i][k], __dmul_rn(c[ii], d[ii + k]));
				}
			}
		}
	}

#pragma unroll
	for (int i = 0; i < R; i++) {
		const int oy = y0 + i;
		if (oy >= a.out_height)
			break;
		float *dst = (float *) (a.out + (long long) oy * a.out_stride) + bx0 + 8 * t;
#pragma unroll
		for (int k = 0; k < T; k++)
			if (bx0 + 8 * t + k < a.out_width)
				dst[k] = (float) acc[i][k];
	}
}

template <typename TIN>
static int launch_convf_rows_lds(const ConvArgs &a, const char *name)
{
	constexpr int R = 4;
	dim3 grid((a.out_width + CONVF_LDS_SPAN - 1) / CONVF_LDS_SPAN, (a.out_height + R - 1) / R, 1);
	if (grid.y > 65535)
		return 1;
	Gate gate(name);
	if (vips_hip_get_exact_float())
		hipLaunchKernelGGL((convf_rows_lds<TIN, R, false>), grid, dim3(256, 1, 1), 0, stream(), a);
	else
		hipLaunchKernelGGL((convf_rows_lds<TIN, R, true>), grid, dim3(256, 1, 1), 0, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

// The tiled kernel wins whenever a thread's TILE outputs share taps; masks that are a
// single element (or tiny images) keep the general kernel.
template <typename TIN, typename TOUT, int MODE>
static int launch_conv_best(const ConvArgs &a, const char *name)
{
	if constexpr (MODE == 2 && std::is_integral<TIN>::value && sizeof(TIN) <= 2 && std::is_same<TOUT, float>::value) {
		// one band, masks 8..64 wide, images wide enough to fill blocks: the LDS row-streaming kernel
		if (a.epp == 1 && a.mask_width >= 8 && a.mask_width <= CONVF_LDS_MAXW && a.mask_height >= 4 &&
			a.out_width >= 512 && a.dense8 && !getenv("VIPS_HIP_NO_GROUPED_CONV") && !getenv("VIPS_HIP_NO_LDS_CONV")) {
			const int r = launch_convf_rows_lds<TIN>(a, name);
			if (r <= 0)
				return r;
		}
	}
	if constexpr (MODE == 2 && std::is_integral<TIN>::value) {
		if (a.mask_width >= 8 && a.out_height <= 65535 && a.dense8 && !getenv("VIPS_HIP_NO_GROUPED_CONV")) {
			const int ids = ((a.out_width + CONV_TILE - 1) / CONV_TILE) * a.epp;
			dim3 grid((ids + 255) / 256, a.out_height, 1);
			Gate gate(name);
#define GROUPED(EPP, FUSED) \
	hipLaunchKernelGGL((convf_grouped<TIN, TOUT, EPP, FUSED>), grid, dim3(256, 1, 1), 0, stream(), a)
			if (vips_hip_get_exact_float()) {
				switch (a.epp) {
				case 1: GROUPED(1, false); break;
				case 3: GROUPED(3, false); break;
				case 4: GROUPED(4, false); break;
				default: GROUPED(0, false); break;
				}
			}
			else {
				switch (a.epp) {
				case 1: GROUPED(1, true); break;
				case 3: GROUPED(3, true); break;
				case 4: GROUPED(4, true); break;
				default: GROUPED(0, true); break;
				}
			}
#undef GROUPED
			VH_CHECK(hipGetLastError());
			return 0;
		}
	}
	if (a.mask_width * a.mask_height >= 3 && a.out_height <= 65535 * (a.mask_width == 1 ? CONV_TILE : 1)) {
		if constexpr (MODE == 0 && sizeof(TIN) <= 2) {
			if (a.narrow) {
				const int r = launch_conv_tiled<TIN, TOUT, 3>(a, name);
				if (r <= 0)
					return r;
			}
		}
		const int r = launch_conv_tiled<TIN, TOUT, MODE>(a, name);
		if (r <= 0)
			return r;
	}
	return launch_conv<TIN, TOUT, MODE>(a, name);
}

// vips_convi_intize, convi.c:925-1120 (HAVE_HWY branch).  false: the mask is refused.
static bool conv_hwy_intize(_VipsHipConv *c, const double *mask, double scale)
{
	const int n = c->mask_width * c->mask_height;
	std::vector<double> scaled(n);
	for (int i = 0; i < n; i++)
		scaled[i] = mask[i] / scale;
	double mx = scaled[0];
	for (int i = 1; i < n; i++)
		mx = scaled[i] > mx ? scaled[i] : mx;
	// the max rounded up to a power of two is the exponent every element shares; + 1 keeps an
	// exact power of two inside signed 8 bits after the * 128
	const double fshift = ceil(log2(mx) + 1);
	if (!(fshift <= 6 && fshift >= -24)) // NaN / -inf (no positive element) lands here too
		return false;
	const int shift = (int) fshift;
	if (ceil(log2((double) n)) > 10)
		return false;
	c->hwy_exp = 7 - shift;
	for (int i = 0; i < n; i++) {
		const double m = rint(128 * scaled[i] * pow(2, -shift));
		if (m < -128 || m > 127)
			return false;
		if (m != 0) {
			c->hwy_mant.push_back((int) m);
			c->hwy_pos.push_back(i);
		}
	}
	if (c->hwy_mant.empty()) {
		c->hwy_mant.push_back(0);
		c->hwy_pos.push_back(0);
	}
	// refuse masks that come out more than 2 grey levels wrong on a flat image
	double true_sum = 0;
	int int_sum = 0;
	for (size_t i = 0; i < c->hwy_mant.size(); i++) {
		true_sum += 128 * scaled[c->hwy_pos[i]];
		int_sum += 128 * c->hwy_mant[i];
	}
	const int true_value = (int) (true_sum < 0 ? 0 : (true_sum > 255 ? 255 : true_sum));
	int int_value = (int_sum + (1 << (c->hwy_exp - 1))) >> c->hwy_exp;
	int_value = int_value < 0 ? 0 : (int_value > 255 ? 255 : int_value);
	return abs(true_value - int_value) <= 2;
}

static int conv_hwy_tables(_VipsHipConv *c)
{
	std::lock_guard<std::mutex> lock(c->mutex);
	if (c->d_hwy_coeff)
		return 0;
	const size_t nnz = c->hwy_mant.size();
	std::vector<short> dx(nnz), dy(nnz);
	std::vector<int> dense((size_t) c->mask_width * c->mask_height, 0);
	for (size_t i = 0; i < nnz; i++) {
		dx[i] = (short) (c->hwy_pos[i] % c->mask_width);
		dy[i] = (short) (c->hwy_pos[i] / c->mask_width);
		dense[c->hwy_pos[i]] = c->hwy_mant[i];
	}
	c->d_hwy_dx = (short *) upload(dx.data(), dx.size() * sizeof(short));
	c->d_hwy_dy = (short *) upload(dy.data(), dy.size() * sizeof(short));
	c->d_hwy_dense = upload(dense.data(), dense.size() * sizeof(int));
	c->d_hwy_coeff = upload(c->hwy_mant.data(), nnz * sizeof(int));
	if (!c->d_hwy_dx || !c->d_hwy_dy || !c->d_hwy_dense || !c->d_hwy_coeff)
		return -1;
	return 0;
}

static int conv_tables(_VipsHipConv *c)
{
	std::lock_guard<std::mutex> lock(c->mutex);
	if (c->d_coeff)
		return 0;
	std::vector<short> dx(c->nnz), dy(c->nnz);
	for (int i = 0; i < c->nnz; i++) {
		dx[i] = (short) (c->pos[i] % c->mask_width);
		dy[i] = (short) (c->pos[i] / c->mask_width);
	}
	c->d_dx = (short *) upload(dx.data(), dx.size() * sizeof(short));
	c->d_dy = (short *) upload(dy.data(), dy.size() * sizeof(short));
	if (c->precision == VIPS_HIP_PRECISION_INTEGER)
		c->d_coeff = upload(c->coeffi.data(), c->coeffi.size() * sizeof(int));
	else
		c->d_coeff = upload(c->coefff.data(), c->coefff.size() * sizeof(double));
	// dense copy (zeros kept, they are skipped at run time) for the register-tiled kernels
	const int ne = c->mask_width * c->mask_height;
	if (c->precision == VIPS_HIP_PRECISION_INTEGER) {
		std::vector<int> dense(ne, 0);
		for (int i = 0; i < c->nnz; i++)
			dense[c->pos[i]] = c->coeffi[i];
		c->d_dense = upload(dense.data(), dense.size() * sizeof(int));
	}
	else {
		std::vector<double> dense(ne, 0.0);
		for (int i = 0; i < c->nnz; i++)
			dense[c->pos[i]] = c->coefff[i];
		c->d_dense = upload(dense.data(), dense.size() * sizeof(double));
		// rows padded to whole groups of 8 taps plus one group (the grouped kernel reads ahead)
		c->np8 = (c->mask_width + 7) / 8 * 8 + 16;
		std::vector<double> dense8((size_t) c->np8 * c->mask_height, 0.0);
		for (int i = 0; i < c->nnz; i++)
			dense8[(size_t) (c->pos[i] / c->mask_width) * c->np8 + c->pos[i] % c->mask_width] = c->coefff[i];
		c->d_dense8 = (double *) upload(dense8.data(), dense8.size() * sizeof(double));
		if (!c->d_dense8)
			return -1;
	}
	if (!c->d_dx || !c->d_dy || !c->d_coeff || !c->d_dense)
		return -1;
	return 0;
}

} // namespace vh

using namespace vh;

extern "C" {

VipsHipConv *vips_hip_conv_new(const double *mask, int mask_width, int mask_height, double scale,
	double offset, int precision)
{
	const char *domain = precision == VIPS_HIP_PRECISION_INTEGER ? "convi" : "convf";
	if (!mask || mask_width <= 0 || mask_height <= 0) {
		error(domain, "bad mask");
		return nullptr;
	}
	// vips_check_matrix, iofuncs/error.c:1196
	if (mask_width > 100000 || mask_height > 100000) {
		error(domain, "matrix image too large");
		return nullptr;
	}
	if (precision != VIPS_HIP_PRECISION_INTEGER && precision != VIPS_HIP_PRECISION_FLOAT) {
		error(domain, "precision 'approximate' (vips_conva) is outside the HIP path");
		return nullptr;
	}
	if (mask_width > 32767 || mask_height > 32767) {
		error(domain, "mask too large for the HIP path");
		return nullptr;
	}
	VipsHipConv *c = new VipsHipConv;
	c->precision = precision;
	c->mask_width = mask_width;
	c->mask_height = mask_height;
	c->scale = scale;
	c->offset = offset;
	c->d_coeff = nullptr;
	c->d_dx = c->d_dy = nullptr;
	c->d_dense = nullptr;
	c->d_dense8 = nullptr;
	c->np8 = 0;
	c->hwy_ok = c->no_vector = false;
	c->hwy_exp = 0;
	c->d_hwy_coeff = c->d_hwy_dense = nullptr;
	c->d_hwy_dx = c->d_hwy_dy = nullptr;
	c->scale_i = c->rounding = c->offset_i = 0;
	const int ne = mask_width * mask_height;
	if (precision == VIPS_HIP_PRECISION_INTEGER) {
		// vips_convi_gen reads scale/offset from the ORIGINAL mask (convi.c:760-762):
		// the scale adjustment vips__image_intize computes (:909-915) is not used by
		// the C path.  Elements are rint()ed (:886-889), zeros squeezed out (:1191-1209).
		c->scale_i = (int) rint(scale);
		c->rounding = c->scale_i / 2;
		c->offset_i = (int) rint(offset);
		if (c->scale_i == 0) {
			error(domain, "mask scale rounds to zero");
			delete c;
			return nullptr;
		}
		for (int i = 0; i < ne; i++) {
			const double v = rint(mask[i]);
			if (v) {
				c->coeffi.push_back((int) v);
				c->pos.push_back(i);
			}
		}
		if (c->coeffi.empty()) {
			c->coeffi.push_back(0);
			c->pos.push_back(0);
		}
		c->nnz = (int) c->coeffi.size();
		c->hwy_ok = conv_hwy_intize(c, mask, scale);
	}
	else {
		// convf.c:300-323: bake the scale into the mask, keep the non-zero elements
		for (int i = 0; i < ne; i++) {
			const double v = mask[i] / scale;
			if (v) {
				c->coefff.push_back(v);
				c->pos.push_back(i);
			}
		}
		if (c->coefff.empty()) {
			c->coefff.push_back(0);
			c->pos.push_back(0);
		}
		c->nnz = (int) c->coefff.size();
	}
	return c;
}

void vips_hip_conv_free(VipsHipConv *c)
{
	if (!c)
		return;
	vips_hip_free(c->d_coeff);
	vips_hip_free(c->d_dx);
	vips_hip_free(c->d_dy);
	vips_hip_free(c->d_dense);
	vips_hip_free(c->d_dense8);
	vips_hip_free(c->d_hwy_coeff);
	vips_hip_free(c->d_hwy_dense);
	vips_hip_free(c->d_hwy_dx);
	vips_hip_free(c->d_hwy_dy);
	delete c;
}

int vips_hip_conv_get_nnz(const VipsHipConv *c)
{
	return c ? c->nnz : -1;
}

// The Highway-variant coefficients of an INTEGER plan, for inspection (host only): returns the
// number of non-zero mantissas (written to mant[] / pos[] when max allows), or 0 when
// vips_convi_intize refuses the mask, -1 on error.
int vips_hip_conv_get_vector(const VipsHipConv *c, int *exp, int *mant, int *pos, int max)
{
	if (!c || c->precision != VIPS_HIP_PRECISION_INTEGER) {
		error("convi", "not an integer conv plan");
		return -1;
	}
	if (!c->hwy_ok)
		return 0;
	const int n = (int) c->hwy_mant.size();
	if (exp)
		*exp = c->hwy_exp;
	if (mant && pos && max >= n)
		for (int i = 0; i < n; i++) {
			mant[i] = c->hwy_mant[i];
			pos[i] = c->hwy_pos[i];
		}
	return n;
}

int vips_hip_conv_out_format(const VipsHipConv *c, int format)
{
	if (!c)
		return -1;
	// convf.c:354-355
	if (c->precision == VIPS_HIP_PRECISION_FLOAT && format_isint(format))
		return VIPS_HIP_FORMAT_FLOAT;
	return format;
}

int vips_hip_conv_gen(const VipsHipConv *conv, const VipsHipRegion *in, const VipsHipRegion *out)
{
	const char *domain = conv && conv->precision == VIPS_HIP_PRECISION_INTEGER ? "convi" : "convf";
	if (ensure_init())
		return -1;
	if (!conv) {
		error("conv", "null conv");
		return -1;
	}
	_VipsHipConv *c = const_cast<_VipsHipConv *>(conv);
	if (plan_device(domain, &conv->device))
		return -1;
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->bands != out->bands || out->format != vips_hip_conv_out_format(c, in->format)) {
		error(domain, "output region has the wrong bands or format");
		return -1;
	}
	if (in->im_width != out->im_width || in->im_height != out->im_height) {
		error(domain, "input and output images must have the same size");
		return -1;
	}
	// the window must cover out rect grown by the mask (convi.c:778-782), clipped
	const int half_w = c->mask_width / 2, half_h = c->mask_height / 2;
	{
		int x0 = out->left - half_w, x1 = out->left + out->width - 1 - half_w + c->mask_width - 1;
		int y0 = out->top - half_h, y1 = out->top + out->height - 1 - half_h + c->mask_height - 1;
		x0 = x0 < 0 ? 0 : x0;
		y0 = y0 < 0 ? 0 : y0;
		x1 = x1 > in->im_width - 1 ? in->im_width - 1 : x1;
		y1 = y1 > in->im_height - 1 ? in->im_height - 1 : y1;
		if (x0 < in->left || y0 < in->top || x1 >= in->left + in->width ||
			y1 >= in->top + in->height) {
			error(domain, "input region too small");
			return -1;
		}
	}
	if (conv_tables(c))
		return -1;

	ConvArgs a;
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_left = in->left;
	a.in_top = in->top;
	a.in_right = in->left + in->width;
	a.im_width = in->im_width;
	a.im_height = in->im_height;
	a.out_left = out->left;
	a.out_top = out->top;
	a.out_width = out->width;
	a.out_height = out->height;
	a.epp = region_elems_per_pel(in);
	a.nnz = c->nnz;
	a.half_w = half_w;
	a.half_h = half_h;
	a.coeff = c->d_coeff;
	a.dense = c->d_dense;
	a.dense8 = c->d_dense8;
	a.np8 = c->np8;
	a.mask_width = c->mask_width;
	a.mask_height = c->mask_height;
	a.dx = c->d_dx;
	a.dy = c->d_dy;
	a.scale_i = c->scale_i;
	a.rounding = c->rounding;
	a.offset_i = c->offset_i;
	a.offset = c->offset;

	const int fmt = format_real(in->format);
	a.narrow = 0;
	if (c->precision == VIPS_HIP_PRECISION_INTEGER && format_sizeof(fmt) <= 2 && !getenv("VIPS_HIP_NO_NARROW_CONV")) {
		const long long maxval = fmt == VIPS_HIP_FORMAT_UCHAR ? 255 : fmt == VIPS_HIP_FORMAT_CHAR ? 128
			: fmt == VIPS_HIP_FORMAT_USHORT ? 65535 : 32768;
		long long abs_sum = 0;
		bool small = true;
		for (int v : c->coeffi) {
			abs_sum += v < 0 ? -(long long) v : v;
			small = small && v > -(1 << 23) && v < (1 << 23);
		}
		const long long off = c->offset_i < 0 ? -(long long) c->offset_i : c->offset_i;
		const long long rnd = c->rounding < 0 ? -(long long) c->rounding : c->rounding;
		a.narrow = small && abs_sum * maxval + rnd < (1LL << 30) && off < (1LL << 30);
	}
	// convi.c:1150-1158: a Highway build takes its vector path for uchar images when vectors are
	// enabled and the intize accepts the mask.  Off unless vips_hip_vector_set_enabled(1).
	if (c->precision == VIPS_HIP_PRECISION_INTEGER && fmt == VIPS_HIP_FORMAT_UCHAR && c->hwy_ok &&
		!c->no_vector && vips_hip_vector_isenabled()) {
		if (conv_hwy_tables(c))
			return -1;
		a.coeff = c->d_hwy_coeff;
		a.dense = c->d_hwy_dense;
		a.dx = c->d_hwy_dx;
		a.dy = c->d_hwy_dy;
		a.nnz = (int) c->hwy_mant.size();
		a.scale_i = c->hwy_exp;
		a.rounding = 1 << (c->hwy_exp - 1);
		a.narrow = 0;
		return launch_conv_best<unsigned char, unsigned char, 4>(a, "convi_vector");
	}
	if (c->precision == VIPS_HIP_PRECISION_INTEGER) {
		switch (fmt) {
		case VIPS_HIP_FORMAT_UCHAR: return launch_conv_best<unsigned char, unsigned char, 0>(a, "convi");
		case VIPS_HIP_FORMAT_CHAR: return launch_conv_best<signed char, signed char, 0>(a, "convi");
		case VIPS_HIP_FORMAT_USHORT: return launch_conv_best<unsigned short, unsigned short, 0>(a, "convi");
		case VIPS_HIP_FORMAT_SHORT: return launch_conv_best<short, short, 0>(a, "convi");
		case VIPS_HIP_FORMAT_UINT: return launch_conv_best<unsigned int, unsigned int, 0>(a, "convi");
		case VIPS_HIP_FORMAT_INT: return launch_conv_best<int, int, 0>(a, "convi");
		case VIPS_HIP_FORMAT_FLOAT: return launch_conv_best<float, float, 1>(a, "convi");
		case VIPS_HIP_FORMAT_DOUBLE: return launch_conv_best<double, double, 1>(a, "convi");
		default: break;
		}
	}
	else {
		switch (fmt) {
		case VIPS_HIP_FORMAT_UCHAR: return launch_conv_best<unsigned char, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_CHAR: return launch_conv_best<signed char, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_USHORT: return launch_conv_best<unsigned short, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_SHORT: return launch_conv_best<short, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_UINT: return launch_conv_best<unsigned int, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_INT: return launch_conv_best<int, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_FLOAT: return launch_conv_best<float, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_DOUBLE: return launch_conv_best<double, double, 2>(a, "convf");
		default: break;
		}
	}
	error(domain, "unsupported band format %d", in->format);
	return -1;
}

// vips_gaussmat_build, create/gaussmat.c:95-167
int vips_hip_gaussmat(double sigma, double min_ampl, int separable, int precision, double *mask,
	int max, double *scale)
{
	const double sig2 = 2. * sigma * sigma;
	double clipped = 8 * sigma;
	clipped = clipped > 5000 ? 5000 : clipped; // VIPS_CLIP(0, 8 * sigma, MASK_SANITY)
	clipped = clipped < 0 ? 0 : clipped;
	const int max_x = (int) clipped;
	int x, y;

	if (precision == VIPS_HIP_PRECISION_APPROXIMATE)
		precision = VIPS_HIP_PRECISION_INTEGER; // "!= FLOAT" rounds, gaussmat.c:152
	for (x = 0; x < max_x; x++) {
		const double v = exp(-((double) (x * x)) / sig2);

		if (v < min_ampl)
			break;
	}
	if (x >= 5000) {
		error("gaussmat", "mask too large");
		return -1;
	}
	const int width = 2 * ((x - 1) > 0 ? (x - 1) : 0) + 1;
	const int height = separable ? 1 : width;
	if (!mask || (long long) width * height > max) {
		if (mask) {
			error("gaussmat", "mask buffer too small (%d x %d)", width, height);
			return -1;
		}
		return width;
	}
	double sum = 0.0;
	for (y = 0; y < height; y++)
		for (x = 0; x < width; x++) {
			const int xo = x - width / 2;
			const int yo = y - height / 2;
			const double distance = xo * xo + yo * yo;
			double v = exp(-distance / sig2);

			if (precision != VIPS_HIP_PRECISION_FLOAT)
				v = rint(20 * v);
			mask[(size_t) y * width + x] = v;
			sum += v;
		}
	if (sum == 0)
		sum = 1;
	if (scale)
		*scale = sum;
	return width;
}

} // extern "C"
