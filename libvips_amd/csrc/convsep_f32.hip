// Separable convolution of float images, both passes in one streaming kernel (gfx950).
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
// Masks with zero elements (which the reference squeezes out) and masks longer than 32
// stay on the two-pass path.
#include "conv.h"

#include <cstddef>

namespace vh {

constexpr int CS_THREADS = 1024;
constexpr int CS_T = 4;     // rows per step = outputs per thread in either pass
constexpr int CS_MAXN = 32; // longest mask
constexpr int CS_LDS_BYTES = 158 * 1024;
constexpr int CS_PAD = 3 * CS_T * 4; // floats the horizontal window may read past the staged rows

struct ConvSepArgs {
	const float *in;
	float *out;
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
};

template <int MODE>
static __device__ __forceinline__ double cs_mac(double s, double c, double v)
{
	if (MODE == 1)
		return __fma_rn(c, v, s);
	return __dadd_rn(s, __dmul_rn(c, v));
}

template <int MODE>
static __device__ __forceinline__ float cs_fin(double s, const ConvSepArgs &a, double offset)
{
	if (MODE == 1) {
		double q = s;
		if (a.scale != 1.0) {
			const double q0 = __dmul_rn(s, a.rscale);
			const double e = __fma_rn(-a.scale, q0, s);
			const double q1 = __fma_rn(e, a.rscale, q0);
			q = isinf(q0) ? q0 : q1;
		}
		return (float) __dadd_rn(q, offset);
	}
	return (float) s;
}

static __device__ __forceinline__ void cs_load(const ConvSepArgs &a, float (&pre)[CS_T][2], int q, int rows_in,
	int y_first, const int (&coff)[2], const bool (&cok)[2])
{
#pragma unroll
	for (int i = 0; i < CS_T; i++) {
		const int r = q * CS_T + i;
		if (r < rows_in) {
			const int row = min(max(y_first + r, 0), a.height - 1);
			const float *src = reinterpret_cast<const float *>(
				reinterpret_cast<const char *>(a.in) + (long long) row * a.in_stride);
#pragma unroll
			for (int j = 0; j < 2; j++)
				if (cok[j])
					pre[i][j] = src[coff[j]];
		}
	}
}

static __device__ __forceinline__ void cs_stage(const float (&pre)[CS_T][2], float *s_in, int inw, int t,
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
typedef const double __attribute__((address_space(4))) *CsCoefs;

template <int MODE>
static __device__ __forceinline__ void cs_taps(const double (&ga)[CS_T], const double (&gb)[CS_T],
	const double (&c)[CS_T], double (&acc)[CS_T], int ntaps)
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
static __device__ __forceinline__ void cs_window(Next next, int n, CsCoefs kc, double (&acc)[CS_T])
{
	double g0[CS_T], g1[CS_T], c0[CS_T], c1[CS_T];
	float raw[CS_T];
#pragma unroll
	for (int m = 0; m < CS_T; m++) {
		g0[m] = (double) next();
		c0[m] = kc[m];
	}
#pragma unroll
	for (int m = 0; m < CS_T; m++)
		g1[m] = (double) next();
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
			g0[m] = (double) raw[m];
#pragma unroll
		for (int m = 0; m < CS_T; m++) {
			raw[m] = next();
			c0[m] = kc[i0 + 2 * CS_T + m]; // zero padded past n
		}
		cs_taps<MODE>(g1, g0, c1, acc, CS_T);
#pragma unroll
		for (int m = 0; m < CS_T; m++)
			g1[m] = (double) raw[m];
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
				g0[m] = (double) raw[m];
			cs_taps<MODE>(g1, g0, c1, acc, rem - CS_T);
		}
	}
}

// horizontal: consecutive same-band elements of a staged row (over-reads the row's end by
// fewer than 3 * CS_T elements of the next row / the pad, never used)
struct CsNextH {
	const float *p;
	int stride;
	__device__ __forceinline__ float operator()()
	{
		const float v = *p;
		p += stride;
		return v;
	}
};

// vertical: down the ring of intermediate rows, wrapping at `ring`
struct CsNextV {
	const float *col;
	int pitch, slot, ring;
	__device__ __forceinline__ float operator()()
	{
		const float v = col[slot * pitch];
		slot = slot + 1 == ring ? 0 : slot + 1;
		return v;
	}
};

template <int MODE>
__global__ void __launch_bounds__(CS_THREADS)
convsep_f32_kernel(ConvSepArgs a)
{
	extern __shared__ __attribute__((aligned(16))) float cs_lds[];
	__shared__ int s_item;
	const int inw = a.se + (a.n - 1) * a.bands;
	float *s_in = cs_lds;                // [CS_T][inw]
	float *s_ring = cs_lds + CS_T * inw + CS_PAD; // [ring][se]

	const int t = threadIdx.x;
	const CsCoefs kc = (CsCoefs) ((const char __attribute__((address_space(4))) *)
								   __builtin_amdgcn_kernarg_segment_ptr() +
		offsetof(ConvSepArgs, coef));

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

		float pre[CS_T][2];
		cs_load(a, pre, 0, rows_in, y_first, coff, cok);
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
				double hacc[CS_T];
				const double seed = MODE == 1 ? 0.0 : a.offset1;
#pragma unroll
				for (int k = 0; k < CS_T; k++)
					hacc[k] = seed;
				cs_window<MODE>(CsNextH{ s_in + hrow * inw + hoff, a.bands }, a.n, kc, hacc);
				int slot = hslot + hrow;
				slot = slot >= a.ring ? slot - a.ring : slot;
				float *dst = s_ring + slot * a.se + hoff;
#pragma unroll
				for (int k = 0; k < CS_T; k++)
					dst[k * a.bands] = cs_fin<MODE>(hacc[k], a, a.offset1);
			}
			// the next step's rows travel while the vertical pass runs
			if (q + 1 < steps)
				cs_load(a, pre, q + 1, rows_in, y_first, coff, cok);
			__syncthreads();

			// ---- vertical pass: outputs jb .. jb + 3 (rows jb .. jb + n + 2 of the ring)
			if (t < a.se && jb + CS_T > 0) {
				double vacc[CS_T];
				const double seed = MODE == 1 ? 0.0 : a.offset2;
#pragma unroll
				for (int k = 0; k < CS_T; k++)
					vacc[k] = seed;
				cs_window<MODE>(CsNextV{ s_ring + t, a.se, vslot, a.ring }, a.n, kc, vacc);
				if (t < ne) {
#pragma unroll
					for (int k = 0; k < CS_T; k++) {
						const int j = jb + k;
						if (j >= 0 && j < rows_out) {
							float *dst = reinterpret_cast<float *>(
								reinterpret_cast<char *>(a.out) + (long long) (y0 + j) * a.out_stride);
							dst[e0 + t] = cs_fin<MODE>(vacc[k], a, a.offset2);
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

template <int MODE>
static int cs_launch(const ConvSepArgs &a, size_t lds, int grid)
{
	static bool attr_done = false; // one attribute per instantiation
	if (!attr_done) {
		VH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&convsep_f32_kernel<MODE>),
			hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS_BYTES));
		attr_done = true;
	}
	Gate gate(MODE == 1 ? "convsep_f32_convi" : "convsep_f32_convf");
	hipLaunchKernelGGL((convsep_f32_kernel<MODE>), dim3(grid), dim3(CS_THREADS), lds, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

int convsep_f32_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c, double offset2)
{
	if (getenv("VIPS_HIP_NO_FUSED_CONVSEP"))
		return 1;
	if (in->format != VIPS_HIP_FORMAT_FLOAT || out->format != VIPS_HIP_FORMAT_FLOAT)
		return 1;
	if (c->mask_height != 1 || c->mask_width > CS_MAXN || c->nnz != c->mask_width)
		return 1;
	if (in->bands < 1 || in->bands > 4 || in->width != out->width || in->height != out->height)
		return 1;
	if ((long long) in->width * in->bands >= (1LL << 30))
		return 1;
	const int n = c->mask_width;
	const int mode = c->precision == VIPS_HIP_PRECISION_INTEGER ? 1 : 2;

	ConvSepArgs a;
	a.in = (const float *) in->data;
	a.out = (float *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.height = in->height;
	a.bands = in->bands;
	a.n = n;
	a.half = n / 2;
	a.ring = n + CS_T - 1;
	for (int k = 0; k < CS_MAXN + 3 * CS_T; k++)
		a.coef[k] = 0.0;
	for (int k = 0; k < n; k++) {
		if (mode == 1) {
			// the product must be exact for the fused multiply-add to round like mul + add
			if (c->coeffi[k] > (1 << 29) || c->coeffi[k] < -(1 << 29))
				return 1;
			a.coef[k] = (double) c->coeffi[k];
		}
		else
			a.coef[k] = c->coefff[k];
	}
	a.scale = mode == 1 ? (double) c->scale_i : 1.0;
	a.rscale = 1.0 / a.scale;
	a.offset1 = mode == 1 ? (double) c->offset_i : c->offset;
	a.offset2 = mode == 1 ? (double) (int) rint(offset2) : offset2;

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
	const int r = mode == 1 ? cs_launch<1>(a, lds, grid) : cs_launch<2>(a, lds, grid);
	vips_hip_free(counter);
	return r;
}

} // namespace vh
