// Separable convolution, both passes in one streaming kernel (gfx950): float images (convi on
// float input, convf) and uchar / ushort / short images (the convi C path).
//
// vips_convsep (convsep.c:61-118) is conv(M) then conv(rot90 M, offset 0); vips_gaussblur
// (gaussblur.c:71-116) builds M with vips_gaussmat.  Run as two operations the float
// intermediate crosses HBM twice; here a 1024-thread block owns a strip of ~1000 row
// elements (x * bands + band) and streams down the rows, 4 at a time:
//
//   load     4 input rows of the strip (+ the (n - 1) * bands halo), prefetched in registers
//            one step ahead, staged in LDS
//   H pass   a thread makes 4 neighbouring same-band outputs of one row from a sliding
//            register window over the LDS row; the result is rounded to float exactly as
//            the first operation's output image would be, and goes into an LDS ring of the
//            last n + 3 intermediate rows
//   V pass   a thread owns one element column and makes the 4 output rows the new rows
//            complete, again from a sliding window, now down the ring
//
// so a pixel is read from HBM once and written once, the vertical halo is the n - 1 warm-up
// rows of a row segment, and each LDS read feeds 4 multiply-adds.  Work items (strip x row
// segment) come from an atomic counter, one persistent block per CU (the ring fills LDS).
//
// Arithmetic (bit-exact with the two-operation reference):
//   MODE 1  convi on float input (convi.c:721-741): double sum of (double) int coefficient
//           * pixel in mask order, sum / scale + offset, cast to float.  int * float is
//           exact in double (|c| < 2^29 checked on the host), so fma(c, v, sum) rounds
//           exactly like the reference's separate multiply and add; the division by the
//           constant scale is Markstein's 3-operation correctly-rounded form.
//   MODE 2  convf (convf.c:163-181): sum seeded with the offset, coefficient = mask / scale
//           (double), separate multiply and add.
//   MODE 0  convi C path on uchar / ushort / short (convi.c:698-716): integer sum of
//           coefficient * pixel, (sum + scale / 2) / scale + offset with C division, clip to
//           the format; the intermediate image has the input's format.  32-bit sums (the host
//           checks sum |c| * max < 2^31; 24-bit multiplies: |c| < 2^23), the truncating
//           division by the constant scale done exactly through a correctly rounded double
//           quotient.  LDS holds the staged rows and the ring as int.
// Masks with zero elements (which the reference squeezes out) and masks longer than 32
// stay on the two-pass path.
#include "conv.h"
#include "kernel_stmt.h"

#include <cstddef>

namespace vh {

constexpr int CS_THREADS = 1024;
constexpr int CS_T = 4;     // rows per step = outputs per thread in either pass
constexpr int CS_MAXN = 32; // longest mask
constexpr int CS_LDS_BYTES = 158 * 1024;
constexpr int CS_PAD = 3 * CS_T * 4; // floats the horizontal window may read past the staged rows

struct ConvSepArgs {
	const void *in;
	void *out;
	long long in_stride, out_stride; // bytes
	int width, height, bands;
	int n, half;
	int se;   // strip width in elements, a multiple of CS_T * bands, <= CS_THREADS
	int ring; // intermediate rows kept: n + CS_T - 1
	int strips, segs, seg_rows;
	double scale, rscale;
	double offset1, offset2;
	int *counter;
	double coef[CS_MAXN + 3 * CS_T]; // taps in mask order, zero padded (read a group ahead)
	int coefi[CS_MAXN + 3 * CS_T];   // MODE 0: the same as integers
	int scale_i, rounding, offset1_i, offset2_i, clip_lo, clip_hi;
};

// MODE 0: int values, int taps, int LDS; MODE 1 / 2: double values and taps, float LDS
template <int MODE>
struct CsTraits {
	typedef double vt;
	typedef double ct;
	typedef float lt;
};
template <>
struct CsTraits<0> {
	typedef int vt;
	typedef int ct;
	typedef int lt;
};

template <int MODE>
static __device__ __forceinline__ typename CsTraits<MODE>::vt cs_mac(typename CsTraits<MODE>::vt s,
	typename CsTraits<MODE>::ct c, typename CsTraits<MODE>::vt v)
{
	if constexpr (MODE == 0)
		return __mul24(c, v) + s;
	else if constexpr (MODE == 1)
		return __fma_rn(c, v, s);
	else
		return __dadd_rn(s, __dmul_rn(c, v));
}

template <int MODE>
static __device__ __forceinline__ typename CsTraits<MODE>::vt cs_seed(const ConvSepArgs &a, int pass)
{
	if constexpr (MODE == 2)
		return pass == 1 ? a.offset1 : a.offset2;
	else
		return 0;
}

// a / y for the constant y = a.scale, correctly rounded (Markstein): r = RN(1 / y)
static __device__ __forceinline__ double cs_div_scale(double s, const ConvSepArgs &a)
{
	const double q0 = __dmul_rn(s, a.rscale);
	const double e = __fma_rn(-a.scale, q0, s);
	const double q1 = __fma_rn(e, a.rscale, q0);
	return isinf(q0) ? q0 : q1;
}

// what the pass stores: the LDS / memory value of one output
template <int MODE>
static __device__ __forceinline__ typename CsTraits<MODE>::lt cs_fin(typename CsTraits<MODE>::vt s,
	const ConvSepArgs &a, int pass)
{
	if constexpr (MODE == 0) {
		// (sum + rounding) / scale with C (truncating) division: |sum| < 2^31, so the correctly
		// rounded double quotient truncates to the exact integer quotient
		int q = s + a.rounding;
		if (a.scale_i != 1)
			q = vh::cvt_i32(cs_div_scale((double) q, a));
		q += pass == 1 ? a.offset1_i : a.offset2_i;
		return min(max(q, a.clip_lo), a.clip_hi);
	}
	else if constexpr (MODE == 1) {
		const double q = a.scale != 1.0 ? cs_div_scale(s, a) : s;
		return (float) __dadd_rn(q, pass == 1 ? a.offset1 : a.offset2);
	}
	else
		return (float) s;
}

template <typename TIO, typename LT>
static __device__ __forceinline__ void cs_load(const ConvSepArgs &a, LT (&pre)[CS_T][2], int q, int rows_in,
	int y_first, const int (&coff)[2], const bool (&cok)[2])
{
#pragma unroll
	for (int i = 0; i < CS_T; i++) {
		const int r = q * CS_T + i;
		if (r < rows_in) {
			const int row = min(max(y_first + r, 0), a.height - 1);
			const TIO *src = reinterpret_cast<const TIO *>(
				reinterpret_cast<const char *>(a.in) + (long long) row * a.in_stride);
#pragma unroll
			for (int j = 0; j < 2; j++)
				if (cok[j])
					pre[i][j] = (LT) src[coff[j]];
		}
	}
}

template <typename LT>
static __device__ __forceinline__ void cs_stage(const LT (&pre)[CS_T][2], LT *s_in, int inw, int t,
	const bool (&cok)[2])
{
#pragma unroll
	for (int i = 0; i < CS_T; i++)
#pragma unroll
		for (int j = 0; j < 2; j++)
			if (cok[j])
				s_in[i * inw + t + j * CS_THREADS] = pre[i][j];
}

// The taps live in the kernel-argument segment and are read with scalar loads at a dynamic
// (wave-uniform) index: coefficients are SGPR operands of the multiply-adds, fetched one
// group ahead.
template <int MODE>
struct CsCoefs {
	typedef const typename CsTraits<MODE>::ct __attribute__((address_space(4))) *ptr;
};

template <int MODE>
static __device__ __forceinline__ void cs_taps(const typename CsTraits<MODE>::vt (&ga)[CS_T],
	const typename CsTraits<MODE>::vt (&gb)[CS_T], const typename CsTraits<MODE>::ct (&c)[CS_T],
	typename CsTraits<MODE>::vt (&acc)[CS_T], int ntaps)
{
#pragma unroll
	for (int ii = 0; ii < CS_T; ii++)
		if (ii < ntaps) {
#pragma unroll
			for (int k = 0; k < CS_T; k++)
				acc[k] = cs_mac<MODE>(acc[k], c[ii], ii + k < CS_T ? ga[ii + k] : gb[ii + k - CS_T]);
		}
}

// CS_T outputs from a sliding window: tap i of output k is element k + i of a sequence read
// through `next()`.  The window is two groups of CS_T values; the group after them is fetched
// (LDS reads) before the current group's 16 multiply-adds are issued and converted to double
// after them, so a read's latency hides behind arithmetic of the same wave.  Two groups of
// taps per loop iteration make the roles of the groups static; whole groups run without any
// condition, the last n mod 8 taps with one test per tap.
template <int MODE, typename Next>
static __device__ __forceinline__ void cs_window(Next next, int n, typename CsCoefs<MODE>::ptr kc,
	typename CsTraits<MODE>::vt (&acc)[CS_T])
{
	typedef typename CsTraits<MODE>::vt vt;
	typename CsTraits<MODE>::vt g0[CS_T], g1[CS_T];
	typename CsTraits<MODE>::ct c0[CS_T], c1[CS_T];
	typename CsTraits<MODE>::lt raw[CS_T];
#pragma unroll
	for (int m = 0; m < CS_T; m++) {
		g0[m] = (vt) next();
		c0[m] = kc[m];
	}
#pragma unroll
	for (int m = 0; m < CS_T; m++)
		g1[m] = (vt) next();
	int i0 = 0;
	for (; i0 + 2 * CS_T <= n; i0 += 2 * CS_T) {
#pragma unroll
		for (int m = 0; m < CS_T; m++) {
			raw[m] = next();
			c1[m] = kc[i0 + CS_T + m];
		}
		cs_taps<MODE>(g0, g1, c0, acc, CS_T);
#pragma unroll
		for (int m = 0; m < CS_T; m++)
			g0[m] = (vt) raw[m];
#pragma unroll
		for (int m = 0; m < CS_T; m++) {
			raw[m] = next();
			c0[m] = kc[i0 + 2 * CS_T + m]; // zero padded past n
		}
		cs_taps<MODE>(g1, g0, c1, acc, CS_T);
#pragma unroll
		for (int m = 0; m < CS_T; m++)
			g1[m] = (vt) raw[m];
	}
	const int rem = n - i0; // 0 .. 7 taps left, window = (g0, g1), their coefficients start in c0
	if (rem > 0) {
#pragma unroll
		for (int m = 0; m < CS_T; m++) {
			raw[m] = next();
			c1[m] = kc[i0 + CS_T + m];
		}
		cs_taps<MODE>(g0, g1, c0, acc, rem);
		if (rem > CS_T) {
#pragma unroll
			for (int m = 0; m < CS_T; m++)
				g0[m] = (vt) raw[m];
			cs_taps<MODE>(g1, g0, c1, acc, rem - CS_T);
		}
	}
}

// horizontal: consecutive same-band elements of a staged row (over-reads the row's end by
// fewer than 3 * CS_T elements of the next row / the pad, never used)
template <typename LT>
struct CsNextH {
	const LT *p;
	int stride;
	__device__ __forceinline__ LT operator()()
	{
		const LT v = *p;
		p += stride;
		return v;
	}
};

// vertical: down the ring of intermediate rows, wrapping at `ring`
template <typename LT>
struct CsNextV {
	const LT *col;
	int pitch, slot, ring;
	__device__ __forceinline__ LT operator()()
	{
		const LT v = col[slot * pitch];
		slot = slot + 1 == ring ? 0 : slot + 1;
		return v;
	}
};

template <int MODE, typename TIO>
__global__ void __launch_bounds__(CS_THREADS)
convsep_kernel(ConvSepArgs a)
{
	typedef typename CsTraits<MODE>::vt vt;
	typedef typename CsTraits<MODE>::lt lt;
	VH_DYNAMIC_LDS(unsigned int, cs_lds_raw);
	lt *cs_lds = reinterpret_cast<lt *>(cs_lds_raw);
	__shared__ int s_item;
	const int inw = a.se + (a.n - 1) * a.bands;
	lt *s_in = cs_lds;                         // [CS_T][inw]
	lt *s_ring = cs_lds + CS_T * inw + CS_PAD; // [ring][se]

	const int t = threadIdx.x;
	const typename CsCoefs<MODE>::ptr kc = (typename CsCoefs<MODE>::ptr) (
		(const char __attribute__((address_space(4))) *) __builtin_amdgcn_kernarg_segment_ptr() +
		(MODE == 0 ? offsetof(ConvSepArgs, coefi) : offsetof(ConvSepArgs, coef)));

	const int E = a.width * a.bands;
	const int items = a.strips * a.segs;
	// horizontal pass: thread -> (row of the step, chunk of CS_T same-band outputs)
	const int chunks = a.se / CS_T;
	const int hrow = t / chunks;
	const int hc = t - hrow * chunks;
	const int hoff = (hc / a.bands) * CS_T * a.bands + hc % a.bands;

	for (;;) {
		__syncthreads();
		if (t == 0)
			s_item = atomicAdd(a.counter, 1);
		__syncthreads();
		// (an LDS read is per-lane to the compiler: say that the work item is wave-uniform)
		const int item = __builtin_amdgcn_readfirstlane(s_item);
		if (item >= items)
			return;
		const int strip = item % a.strips;
		const int seg = item / a.strips;
		const int e0 = strip * a.se;
		const int ne = min(a.se, E - e0);
		const int y0 = seg * a.seg_rows;
		const int rows_out = min(a.seg_rows, a.height - y0);
		const int rows_in = rows_out + a.n - 1;
		const int y_first = y0 - a.half;
		const int steps = (rows_in + CS_T - 1) / CS_T;

		// the (at most two) staged elements per row this thread fetches, columns clamped
		int coff[2];
		bool cok[2];
#pragma unroll
		for (int j = 0; j < 2; j++) {
			const int idx = t + j * CS_THREADS;
			cok[j] = idx < inw;
			const int abs_e = e0 - a.half * a.bands + idx;
			int px = (abs_e + 64 * a.bands) / a.bands - 64;
			const int b = abs_e - px * a.bands;
			px = min(max(px, 0), a.width - 1);
			coff[j] = px * a.bands + b;
		}

		lt pre[CS_T][2];
		cs_load<TIO>(a, pre, 0, rows_in, y_first, coff, cok);
		cs_stage(pre, s_in, inw, t, cok);
		int hslot = 0;    // ring slot of the step's first row (row r lives in slot r mod ring)
		int jb = 1 - a.n; // first output row of the step (negative: warm-up, nothing stored)
		// ring slot of row jb, the first row the step's first output needs; the warm-up
		// steps start "before" row 0 at slots that are never written and never used
		int vslot = ((jb % a.ring) + a.ring) % a.ring;
		for (int q = 0; q < steps; q++) {
			__syncthreads(); // the rows of step q are in s_in

			// ---- horizontal pass
			if (hrow < CS_T) {
				vt hacc[CS_T];
#pragma unroll
				for (int k = 0; k < CS_T; k++)
					hacc[k] = cs_seed<MODE>(a, 1);
				cs_window<MODE>(CsNextH<lt>{ s_in + hrow * inw + hoff, a.bands }, a.n, kc, hacc);
				int slot = hslot + hrow;
				slot = slot >= a.ring ? slot - a.ring : slot;
				lt *dst = s_ring + slot * a.se + hoff;
#pragma unroll
				for (int k = 0; k < CS_T; k++)
					dst[k * a.bands] = cs_fin<MODE>(hacc[k], a, 1);
			}
			// the next step's rows travel while the vertical pass runs
			if (q + 1 < steps)
				cs_load<TIO>(a, pre, q + 1, rows_in, y_first, coff, cok);
			__syncthreads();

			// ---- vertical pass: outputs jb .. jb + 3 (rows jb .. jb + n + 2 of the ring)
			if (t < a.se && jb + CS_T > 0) {
				vt vacc[CS_T];
#pragma unroll
				for (int k = 0; k < CS_T; k++)
					vacc[k] = cs_seed<MODE>(a, 2);
				cs_window<MODE>(CsNextV<lt>{ s_ring + t, a.se, vslot, a.ring }, a.n, kc, vacc);
				if (t < ne) {
#pragma unroll
					for (int k = 0; k < CS_T; k++) {
						const int j = jb + k;
						if (j >= 0 && j < rows_out) {
							TIO *dst = reinterpret_cast<TIO *>(
								reinterpret_cast<char *>(a.out) + (long long) (y0 + j) * a.out_stride);
							dst[e0 + t] = (TIO) cs_fin<MODE>(vacc[k], a, 2);
						}
					}
				}
			}
			if (q + 1 < steps)
				cs_stage(pre, s_in, inw, t, cok);
			hslot += CS_T;
			hslot = hslot >= a.ring ? hslot - a.ring : hslot;
			vslot += CS_T;
			vslot = vslot >= a.ring ? vslot - a.ring : vslot;
			jb += CS_T;
		}
	}
}

template <int MODE, typename TIO>
static int cs_launch(const ConvSepArgs &a, size_t lds, int grid, const char *gate_name)
{
	static bool attr_done = false; // one attribute per instantiation
	if (!attr_done) {
		VH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&convsep_kernel<MODE, TIO>),
			hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS_BYTES));
		attr_done = true;
	}
	Gate gate(gate_name);
	hipLaunchKernelGGL((convsep_kernel<MODE, TIO>), dim3(grid), dim3(CS_THREADS), lds, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

int convsep_f32_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c, double offset2)
{
	if (getenv("VIPS_HIP_NO_FUSED_CONVSEP"))
		return 1;
	const bool integer = c->precision == VIPS_HIP_PRECISION_INTEGER;
	const int fmt = in->format;
	int mode;
	long long maxval = 0;
	if (fmt == VIPS_HIP_FORMAT_FLOAT)
		mode = integer ? 1 : 2;
	else if (integer && (fmt == VIPS_HIP_FORMAT_UCHAR || fmt == VIPS_HIP_FORMAT_USHORT ||
				 fmt == VIPS_HIP_FORMAT_SHORT)) {
		mode = 0;
		maxval = fmt == VIPS_HIP_FORMAT_UCHAR ? 255 : fmt == VIPS_HIP_FORMAT_USHORT ? 65535 : 32768;
	}
	else
		return 1;
	if (out->format != fmt)
		return 1;
	if (c->mask_height != 1 || c->mask_width > CS_MAXN || c->nnz != c->mask_width)
		return 1;
	// short integer masks (sharpen's 3 taps): two register-tiled passes are quicker than the
	// ring's barriers (measured on 8192^2 short: 0.12 ms against 0.22)
	if (mode == 0 && c->mask_width < 7)
		return 1;
	if (in->bands < 1 || in->bands > 4 || in->width != out->width || in->height != out->height)
		return 1;
	if ((long long) in->width * in->bands >= (1LL << 30))
		return 1;
	const int n = c->mask_width;

	ConvSepArgs a;
	a.in = in->data;
	a.out = out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.height = in->height;
	a.bands = in->bands;
	a.n = n;
	a.half = n / 2;
	a.ring = n + CS_T - 1;
	for (int k = 0; k < CS_MAXN + 3 * CS_T; k++) {
		a.coef[k] = 0.0;
		a.coefi[k] = 0;
	}
	long long abs_sum = 0;
	for (int k = 0; k < n; k++) {
		if (integer) {
			// float images: the product must be exact for the fused multiply-add to round like
			// mul + add; integer images: 24-bit multiplies
			const long long lim = mode == 0 ? (1LL << 23) : (1LL << 29);
			if (c->coeffi[k] >= lim || c->coeffi[k] <= -lim)
				return 1;
			a.coef[k] = (double) c->coeffi[k];
			a.coefi[k] = c->coeffi[k];
			abs_sum += c->coeffi[k] < 0 ? -(long long) c->coeffi[k] : c->coeffi[k];
		}
		else
			a.coef[k] = c->coefff[k];
	}
	a.scale = integer ? (double) c->scale_i : 1.0;
	a.rscale = 1.0 / a.scale;
	a.offset1 = integer ? (double) c->offset_i : c->offset;
	a.offset2 = integer ? (double) (int) rint(offset2) : offset2;
	a.scale_i = c->scale_i;
	a.rounding = c->rounding;
	a.offset1_i = c->offset_i;
	a.offset2_i = (int) rint(offset2);
	a.clip_lo = fmt == VIPS_HIP_FORMAT_SHORT ? -32768 : 0;
	a.clip_hi = fmt == VIPS_HIP_FORMAT_UCHAR ? 255 : fmt == VIPS_HIP_FORMAT_USHORT ? 65535 : 32767;
	if (mode == 0) {
		// 32-bit sums: every partial sum, plus the rounding term, stays inside int; the
		// quotient plus the offset too
		const long long bound = abs_sum * maxval + (c->rounding < 0 ? -(long long) c->rounding : c->rounding);
		const long long off = a.offset1_i < 0 ? -(long long) a.offset1_i : a.offset1_i;
		if (bound >= (1LL << 31) - 1 || off >= (1LL << 30) || c->scale_i == 0)
			return 1;
	}

	// strip width: as wide as LDS (the staged rows + the ring) and the block allow
	const int unit = CS_T * a.bands;
	const long long E = (long long) a.width * a.bands;
	long long se = (CS_LDS_BYTES / 4 - CS_PAD - (long long) CS_T * (n - 1) * a.bands) / (CS_T + a.ring);
	if (se > CS_THREADS)
		se = CS_THREADS;
	se = se / unit * unit;
	if (E < se)
		se = (E + unit - 1) / unit * unit;
	if (se < unit)
		return 1;
	a.se = (int) se;
	a.strips = (int) ((E + se - 1) / se);
	// ~12 items per CU: segments long enough that the n - 1 warm-up rows stay cheap
	int want_segs = (256 * 12 + a.strips - 1) / a.strips;
	int seg_rows = (a.height + want_segs - 1) / want_segs;
	if (seg_rows < 16 * n)
		seg_rows = 16 * n;
	if (seg_rows > a.height)
		seg_rows = a.height;
	a.seg_rows = seg_rows;
	a.segs = (a.height + seg_rows - 1) / seg_rows;
	const int items = a.strips * a.segs;

	int *counter = (int *) vips_hip_malloc(sizeof(int));
	if (!counter)
		return -1;
	a.counter = counter;
	if (hipMemsetAsync(counter, 0, sizeof(int), stream()) != hipSuccess) {
		vips_hip_free(counter);
		return hip_failed(hipErrorUnknown, "hipMemsetAsync");
	}
	const int inw = a.se + (n - 1) * a.bands;
	const size_t lds = (size_t) (CS_T * inw + CS_PAD + a.ring * a.se) * sizeof(float);
	const int grid = items < 256 ? items : 256;
	int r;
	if (mode == 1)
		r = cs_launch<1, float>(a, lds, grid, "convsep_f32_convi");
	else if (mode == 2)
		r = cs_launch<2, float>(a, lds, grid, "convsep_f32_convf");
	else if (fmt == VIPS_HIP_FORMAT_UCHAR)
		r = cs_launch<0, unsigned char>(a, lds, grid, "convsep_u8_convi");
	else if (fmt == VIPS_HIP_FORMAT_USHORT)
		r = cs_launch<0, unsigned short>(a, lds, grid, "convsep_u16_convi");
	else
		r = cs_launch<0, short>(a, lds, grid, "convsep_s16_convi");
	vips_hip_free(counter);
	return r;
}

} // namespace vh
