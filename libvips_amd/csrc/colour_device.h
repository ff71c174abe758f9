// Device side of the colour routes, shared by colour.hip (the pointwise route kernels) and
// convsep_stream.hip (the colourspace epilogue fused behind a separable convolution): the
// per-pixel steps with the reference's intermediate types and operation order, and the chain
// for one pixel.  Host side: colour.hip.
#pragma once

#include "colour.h"
#include "kernel_stmt.h"

namespace vh {

struct ColourTables {
	float *v2Y_8;   // 256     sRGB2scRGB, 8 bit    LabQ2sRGB.c:151-159
	float *v2Y_16;  // 65536
	int *Y2v_8;     // 257     scRGB2sRGB           LabQ2sRGB.c:134-149
	int *Y2v_16;    // 65537
	float *cbrt;    // 100000  XYZ2Lab              XYZ2Lab.c:92-106
};

// ---------------------------------------------------------------- the steps

struct Px {
	float a, b, c;
};

// scRGB2XYZ.c:58-82
static __device__ __forceinline__ Px step_scRGB2XYZ(Px p)
{
	// p * VIPS_D65_Y0: float * double(100.0), rounded to float == float multiply
	const float R = __fmul_rn(p.a, 100.0f);
	const float G = __fmul_rn(p.b, 100.0f);
	const float B = __fmul_rn(p.c, 100.0f);
	Px q;
	q.a = __fadd_rn(__fadd_rn(__fmul_rn(0.4124F, R), __fmul_rn(0.3576F, G)), __fmul_rn(0.1805F, B));
	q.b = __fadd_rn(__fadd_rn(__fmul_rn(0.2126F, R), __fmul_rn(0.7152F, G)), __fmul_rn(0.0722F, B));
	q.c = __fadd_rn(__fadd_rn(__fmul_rn(0.0193F, R), __fmul_rn(0.1192F, G)), __fmul_rn(0.9505F, B));
	return q;
}

// a / y for a compile-time constant y, correctly rounded: q0 = a * RN(1/y), one FMA
// residual, one FMA correction (Markstein's theorem: with r = RN(1/y) and q0 within an ulp
// of a/y, RN(q0 + (a - y*q0) * r) = RN(a/y)).  Three DP ops instead of the ~12 of the
// generic IEEE division expansion; the FMAs are the exact-division device, not a fused
// version of reference arithmetic.
static __device__ __forceinline__ double div_const(double a, double y, double r)
{
	const double q0 = __dmul_rn(a, r);
	const double e = __fma_rn(-y, q0, a);
	const double q1 = __fma_rn(e, r, q0);
	// an infinite (or NaN) dividend has no finite residual: v_div_fixup_f64 puts IEEE division's
	// special cases (and its sign of zero) back in one instruction, and passes every other
	// quotient through
	return __builtin_amdgcn_div_fixup(q1, y, a);
}
#define DIV_CONST(A, Y) div_const((A), (Y), 1.0 / (Y))

// the same for a dividend known to be finite and far from overflow (no infinite quotient to keep)
static __device__ __forceinline__ double div_const_finite(double a, double y, double r)
{
	const double q0 = __dmul_rn(a, r);
	const double e = __fma_rn(-y, q0, a);
	return __fma_rn(e, r, q0);
}
#define DIV_CONST_F(A, Y) div_const_finite((A), (Y), 1.0 / (Y))

// (float) ((double) A / C) for C = X0, Y0, Z0 of D65 (WHICH = 0, 1, 2) in two single-precision
// operations.  With R = 1 / C cut into two floats, fmaf(A, Rhi, RN(A * Rlo)) rounds
// A * (Rhi + Rlo) + (an error below 2^-47 of it) once, to float; A / C for a 24-bit A and
// C = 95047 / 1000, 100, 1088827 / 10000 is never closer than 2^-44 (relative) to the midpoint of two
// floats, and the reference's own double quotient is within 2^-52 of it: the same float.  Checked
// against the double division for EVERY float A in [2^-8, 2^27) and the three constants (8 M
// mantissas x 35 exponents, tools/div_f32_check.py); the XYZ of a uchar pixel gives A = 0 or
// 58 <= A < 1.1e7.
template <int WHICH>
static __device__ __forceinline__ float quant_div_finite(float A)
{
	constexpr double C = WHICH == 0 ? 95.0470 : WHICH == 1 ? 100.0 : 108.8827;
	constexpr float rhi = (float) (1.0 / C);
	constexpr float rlo = (float) (1.0 / C - (double) rhi);
	return __builtin_fmaf(A, rhi, __fmul_rn(A, rlo));
}

// vips_col_XYZ2Lab_helper, XYZ2Lab.c:109-138 (D65: include/vips/colour.h:58-60)
// FINITE: v is known to be a small finite number (XYZ of a uchar pixel): no inf / NaN handling
template <int WHICH, bool FINITE = false>
static __device__ __forceinline__ float cbrt_lerp(const float *__restrict__ table, float v)
{
	// nX = QUANT_ELEMENTS * X / X0: (int * float) in float, then / double, back to float
	const float fnum = __fmul_rn(100000.0f, v);
	float n;
	if (FINITE)
		n = quant_div_finite<WHICH>(fnum);
	else {
		const double num = (double) fnum;
		n = (float) (WHICH == 0 ? DIV_CONST(num, 95.0470) : WHICH == 1 ? DIV_CONST(num, 100.0)
																   : DIV_CONST(num, 108.8827));
	}
	// VIPS_CLIP(0, (int) nX, QUANT_ELEMENTS - 2); (int) of NaN / overflow is the x86
	// "integer indefinite" INT_MIN, which the clip turns into 0
	// (v_cvt_i32_f32 saturates and turns NaN into 0: after the clip only n >= 2^31 -- INT_MAX
	// here, INT_MIN there -- needs telling apart)
	int i = vh::cvt_i32(n);
	if (!FINITE)
		i = n >= 2147483648.0f ? 0 : i;
	i = min(max(i, 0), 100000 - 2);
	const float f = __fsub_rn(n, (float) i);
	// table[i] and table[i + 1] in one 8-byte access (dword aligned is enough for global memory)
	float2 pair;
	__builtin_memcpy(&pair, table + i, sizeof(pair));
	return __fadd_rn(pair.x, __fmul_rn(f, __fsub_rn(pair.y, pair.x)));
}

// cbrt_lerp<WHICH, true> in two halves, so that a caller can have the table reads of several pixels
// in flight before it uses the first: the index and fraction, then the interpolation.  The same
// operations in the same order.
template <int WHICH>
static __device__ __forceinline__ int cbrt_index_finite(float v, float &f)
{
	const float n = quant_div_finite<WHICH>(__fmul_rn(100000.0f, v));
	int i = vh::cvt_i32(n);
	i = min(max(i, 0), 100000 - 2);
	f = __fsub_rn(n, (float) i);
	return i;
}
static __device__ __forceinline__ float cbrt_finish(float2 pair, float f)
{
	return __fadd_rn(pair.x, __fmul_rn(f, __fsub_rn(pair.y, pair.x)));
}

template <bool FINITE = false>
static __device__ __forceinline__ Px step_XYZ2Lab(Px p, const float *__restrict__ table)
{
	const float cbx = cbrt_lerp<0, FINITE>(table, p.a);
	const float cby = cbrt_lerp<1, FINITE>(table, p.b);
	const float cbz = cbrt_lerp<2, FINITE>(table, p.c);
	Px q;
	q.a = __fsub_rn(__fmul_rn(116.0F, cby), 16.0F);
	q.b = __fmul_rn(500.0F, __fsub_rn(cbx, cby));
	q.c = __fmul_rn(200.0F, __fsub_rn(cby, cbz));
	return q;
}

// vips_col_Lab2XYZ_helper, Lab2XYZ.c:84-109 -- double arithmetic
static __device__ __forceinline__ Px step_Lab2XYZ(Px p)
{
	const double X0 = 95.0470, Y0 = 100.0, Z0 = 108.8827;
	const float L = p.a, a = p.b, b = p.c;
	double cby, tmp;
	Px q;

	if (L < 8.0) {
		q.b = (float) __ddiv_rn(__dmul_rn((double) L, Y0), 903.3);
		cby = __dadd_rn(__dmul_rn(7.787, __ddiv_rn((double) q.b, Y0)), 16.0 / 116.0);
	}
	else {
		cby = __ddiv_rn(__dadd_rn((double) L, 16.0), 116.0);
		q.b = (float) __dmul_rn(__dmul_rn(__dmul_rn(Y0, cby), cby), cby);
	}

	tmp = __dadd_rn(__ddiv_rn((double) a, 500.0), cby);
	if (tmp < 0.2069)
		q.a = (float) __ddiv_rn(__dmul_rn(X0, __dsub_rn(tmp, 0.13793)), 7.787);
	else
		q.a = (float) __dmul_rn(__dmul_rn(__dmul_rn(X0, tmp), tmp), tmp);

	tmp = __dsub_rn(cby, __ddiv_rn((double) b, 200.0));
	if (tmp < 0.2069)
		q.c = (float) __ddiv_rn(__dmul_rn(Z0, __dsub_rn(tmp, 0.13793)), 7.787);
	else
		q.c = (float) __dmul_rn(__dmul_rn(__dmul_rn(Z0, tmp), tmp), tmp);
	return q;
}

// The same for LabS-coded input (vips_sharpen's way back to sRGB): every division by a constant
// is the correctly rounded three-operation quotient (div_const_finite) instead of the
// ~25-instruction IEEE expansion of `/` -- five divisions per pixel -- and the dark-pixel arms
// (L < 8, f < 0.2069) are only executed by waves that hold such a pixel.  It equals step_Lab2XYZ
// for EVERY (L, a, b) the coding can hold: tools/div_probe.hip compares the two on the device
// over all 2^15 x 2^16 (L, a) and (L, b) pairs (profiles/r02_probes.txt).
static __device__ __forceinline__ Px step_Lab2XYZ_c(Px p)
{
	const double X0 = 95.0470, Y0 = 100.0, Z0 = 108.8827;
	const float L = p.a, a = p.b, b = p.c;
	Px q;

	double cby = DIV_CONST_F(__dadd_rn((double) L, 16.0), 116.0);
	q.b = (float) __dmul_rn(__dmul_rn(__dmul_rn(Y0, cby), cby), cby);
	if (__builtin_amdgcn_ballot_w64(L < 8.0f)) {
		if (L < 8.0f) {
			q.b = (float) DIV_CONST_F(__dmul_rn((double) L, Y0), 903.3);
			cby = __dadd_rn(__dmul_rn(7.787, DIV_CONST_F((double) q.b, 100.0)), 16.0 / 116.0);
		}
	}

	const double fx = __dadd_rn(DIV_CONST_F((double) a, 500.0), cby);
	const double fz = __dsub_rn(cby, DIV_CONST_F((double) b, 200.0));
	q.a = (float) __dmul_rn(__dmul_rn(__dmul_rn(X0, fx), fx), fx);
	q.c = (float) __dmul_rn(__dmul_rn(__dmul_rn(Z0, fz), fz), fz);
	if (__builtin_amdgcn_ballot_w64(fx < 0.2069 || fz < 0.2069)) {
		if (fx < 0.2069)
			q.a = (float) DIV_CONST_F(__dmul_rn(X0, __dsub_rn(fx, 0.13793)), 7.787);
		if (fz < 0.2069)
			q.c = (float) DIV_CONST_F(__dmul_rn(Z0, __dsub_rn(fz, 0.13793)), 7.787);
	}
	return q;
}

// vips_col_XYZ2scRGB, LabQ2sRGB.c:263-283
static __device__ __forceinline__ Px step_XYZ2scRGB(Px p)
{
	// X /= SCALE with SCALE = VIPS_D65_Y0 (double)
	const float X = (float) DIV_CONST((double) p.a, 100.0);
	const float Y = (float) DIV_CONST((double) p.b, 100.0);
	const float Z = (float) DIV_CONST((double) p.c, 100.0);
	Px q;
	q.a = __fadd_rn(__fadd_rn(__fmul_rn(3.240625F, X), __fmul_rn(-1.537208F, Y)), __fmul_rn(-0.498629F, Z));
	q.b = __fadd_rn(__fadd_rn(__fmul_rn(-0.968931F, X), __fmul_rn(1.875756F, Y)), __fmul_rn(0.041518F, Z));
	q.c = __fadd_rn(__fadd_rn(__fmul_rn(0.055710F, X), __fmul_rn(-0.204021F, Y)), __fmul_rn(1.056996F, Z));
	return q;
}

// one channel of vips_col_scRGB2sRGB, LabQ2sRGB.c:290-360
static __device__ __forceinline__ int scRGB2sRGB_channel(const int *__restrict__ lut, float v, int maxval)
{
	float Yf = __fmul_rn(v, (float) maxval);
	if (Yf < 0)
		Yf = 0;
	else if (Yf > maxval)
		Yf = maxval;
	const int Yi = vh::cvt_i32(Yf); // (0 <= Yf <= maxval, or NaN -> 0)
	const int l0 = lut[Yi];
	const float r =
		__fadd_rn((float) l0, __fmul_rn((float) (lut[Yi + 1] - l0), __fsub_rn(Yf, (float) Yi)));
	return vh::cvt_i32(rintf(r));
}

// vips_Lab2LabS_line, Lab2LabS.c:59-73: double multiply, clip, truncate
static __device__ __forceinline__ short lab2labs(float v, double scale, double lo)
{
	double d = __dmul_rn((double) v, scale);
	d = d > 32767.0 ? 32767.0 : d; // VIPS_MIN(B, V)
	d = lo > d ? lo : d;           // VIPS_MAX(A, ...)
	return (short) vh::cvt_i32(d); // (NaN falls through both selects: 0, as the low half of x86's INT_MIN)
}

// ------------------------------------------------------------- pixel IO

// vips_cast semantics (conversion/cast.c:120-330, no shift) from any real format to the
// format a chain's first step wants, one band element at a time.
template <typename T>
static __device__ __forceinline__ int load_as_uchar_like(T v, int maxv)
{
	// CAST_INT_INT with TEMP = int: wraps through int first
	int t = (int) v;
	return min(max(t, 0), maxv);
}
template <>
__device__ __forceinline__ int load_as_uchar_like<float>(float v, int maxv)
{
	// CAST_FLOAT_INT: clip as double, then C truncation.  float -> double is exact and monotonic
	// and both limits are floats, so the same compares and selects on the float give the same
	// value (NaN falls through both, as it does in double)
	float d = v;
	d = (float) maxv < d ? (float) maxv : d;
	d = 0.0f > d ? 0.0f : d;
	return vh::cvt_i32(d);
}
template <>
__device__ __forceinline__ int load_as_uchar_like<double>(double v, int maxv)
{
	double d = v;
	d = (double) maxv < d ? (double) maxv : d;
	d = 0.0 > d ? 0.0 : d;
	return vh::cvt_i32(d);
}

template <typename T>
static __device__ __forceinline__ int load_as_short(T v)
{
	int t = (int) v;
	return min(max(t, (int) SHRT_MIN), (int) SHRT_MAX);
}
template <>
__device__ __forceinline__ int load_as_short<float>(float v)
{
	double d = (double) v;
	d = 32767.0 < d ? 32767.0 : d;
	d = -32768.0 > d ? -32768.0 : d;
	return vh::cvt_i32(d);
}
template <>
__device__ __forceinline__ int load_as_short<double>(double v)
{
	double d = v;
	d = 32767.0 < d ? 32767.0 : d;
	d = -32768.0 > d ? -32768.0 : d;
	return vh::cvt_i32(d);
}

struct RouteArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int width, height;
	int in_bands, out_bands; // bands per pel in memory (3 colour + extra)
	int n_steps;
	int steps[8];
	int extra_bands;      // bands carried through after the 3 colour bands
	double alpha_scale;   // max_alpha_after / max_alpha_before (colour.c:257-273), 1.0 = none
	ColourTables tables;
};

// Extra bands: [vips_linear1(scale) ->] vips_cast(out format) (colour.c:249-296).
template <typename TOUT>
static __device__ __forceinline__ TOUT cast_from_double(double d);
template <>
__device__ __forceinline__ unsigned char cast_from_double<unsigned char>(double d)
{
	d = 255.0 < d ? 255.0 : d;
	d = 0.0 > d ? 0.0 : d;
	return (unsigned char) vh::cvt_i32(d);
}
template <>
__device__ __forceinline__ unsigned short cast_from_double<unsigned short>(double d)
{
	d = 65535.0 < d ? 65535.0 : d;
	d = 0.0 > d ? 0.0 : d;
	return (unsigned short) vh::cvt_i32(d);
}
template <>
__device__ __forceinline__ short cast_from_double<short>(double d)
{
	d = 32767.0 < d ? 32767.0 : d;
	d = -32768.0 > d ? -32768.0 : d;
	return (short) vh::cvt_i32(d);
}
template <>
__device__ __forceinline__ float cast_from_double<float>(double d)
{
	return (float) d;
}

template <typename TOUT, typename TIN>
static __device__ __forceinline__ TOUT cast_int_to(TIN v);
// CAST_INT_INT through TEMP = int
#define CAST_INT_TO(TOUT, LO, HI) \
	template <> \
	__device__ __forceinline__ TOUT cast_int_to<TOUT, unsigned char>(unsigned char v) \
	{ \
		int t = (int) v; \
		return (TOUT) min(max(t, LO), HI); \
	} \
	template <> \
	__device__ __forceinline__ TOUT cast_int_to<TOUT, unsigned short>(unsigned short v) \
	{ \
		int t = (int) v; \
		return (TOUT) min(max(t, LO), HI); \
	} \
	template <> \
	__device__ __forceinline__ TOUT cast_int_to<TOUT, short>(short v) \
	{ \
		int t = (int) v; \
		return (TOUT) min(max(t, LO), HI); \
	}
CAST_INT_TO(unsigned char, 0, 255)
CAST_INT_TO(unsigned short, 0, 65535)
CAST_INT_TO(short, -32768, 32767)
#undef CAST_INT_TO

template <typename TIN, typename TOUT>
struct Carry {
	static __device__ __forceinline__ TOUT run(TIN v, double scale)
	{
		if (scale != 1.0) {
			// vips_linear1, LOOP1 (arithmetic/linear.c:213-223): float a1 = a,
			// q = a1 * (float) p + b1 in float; then vips_cast from float
			const float f = __fadd_rn(__fmul_rn((float) scale, (float) v), 0.0f);
			return cast_from_double<TOUT>((double) f);
		}
		return cast_from_double<TOUT>((double) v);
	}
};

// int -> int carries go through CAST_INT_INT rather than the double clip (same result for
// the in-range values these formats hold, but keep the reference's path)
#define CARRY_INT(TIN, TOUT) \
	template <> \
	struct Carry<TIN, TOUT> { \
		static __device__ __forceinline__ TOUT run(TIN v, double scale) \
		{ \
			if (scale != 1.0) { \
				const float f = __fadd_rn(__fmul_rn((float) scale, (float) v), 0.0f); \
				return cast_from_double<TOUT>((double) f); \
			} \
			return cast_int_to<TOUT, TIN>(v); \
		} \
	};
CARRY_INT(unsigned char, unsigned char)
CARRY_INT(unsigned char, unsigned short)
CARRY_INT(unsigned char, short)
CARRY_INT(unsigned short, unsigned char)
CARRY_INT(unsigned short, unsigned short)
CARRY_INT(unsigned short, short)
CARRY_INT(short, unsigned char)
CARRY_INT(short, unsigned short)
CARRY_INT(short, short)
#undef CARRY_INT

// Routes compiled in: with ROUTE > 0 the steps are constants, the step loop unrolls and its
// branches fold away (ROUTE 0 reads a.steps at run time: any chain).  Row = {n, steps...}.
constexpr int kStaticRoutes[5][5] = {
	{ 0, -1, -1, -1, -1 },
	{ 3, VIPS_HIP_COLOUR_sRGB2scRGB, VIPS_HIP_COLOUR_scRGB2XYZ, VIPS_HIP_COLOUR_XYZ2Lab, -1 },
	{ 4, VIPS_HIP_COLOUR_sRGB2scRGB, VIPS_HIP_COLOUR_scRGB2XYZ, VIPS_HIP_COLOUR_XYZ2Lab, VIPS_HIP_COLOUR_Lab2LabS },
	{ 4, VIPS_HIP_COLOUR_LabS2Lab, VIPS_HIP_COLOUR_Lab2XYZ, VIPS_HIP_COLOUR_XYZ2scRGB, VIPS_HIP_COLOUR_scRGB2sRGB },
	{ 3, VIPS_HIP_COLOUR_Lab2XYZ, VIPS_HIP_COLOUR_XYZ2scRGB, VIPS_HIP_COLOUR_scRGB2sRGB, -1 },
};
constexpr int kStaticRouteCount = 5;

// The chain for one pixel: stored bands in, stored bands out.  v2Y8 / Y2v8: the two 8-bit
// tables wherever the caller keeps them (a.tables' own, or copies in LDS).
template <typename TIN, typename TOUT, int ROUTE = 0>
static __device__ __forceinline__ void route_pixel(const RouteArgs &a, const float *v2Y8, const int *Y2v8, TIN i0,
	TIN i1, TIN i2, TOUT &o0, TOUT &o1, TOUT &o2)
{
	Px v;
	const int n_steps = ROUTE ? kStaticRoutes[ROUTE][0] : a.n_steps;
	const int first = ROUTE ? kStaticRoutes[ROUTE][1] : a.steps[0];
	int s = 0;
	// ---- the first step fixes how the stored bands are interpreted
	if (first == VIPS_HIP_COLOUR_sRGB2scRGB) {
		// vips_colour_code_build casts to uchar (colour.c:428-434), sRGB2scRGB.c:72-90
		v.a = v2Y8[load_as_uchar_like<TIN>(i0, 255)];
		v.b = v2Y8[load_as_uchar_like<TIN>(i1, 255)];
		v.c = v2Y8[load_as_uchar_like<TIN>(i2, 255)];
		s = 1;
	}
	else if (first == VIPS_HIP_COLOUR_sRGB2scRGB16) {
		v.a = a.tables.v2Y_16[load_as_uchar_like<TIN>(i0, 65535)];
		v.b = a.tables.v2Y_16[load_as_uchar_like<TIN>(i1, 65535)];
		v.c = a.tables.v2Y_16[load_as_uchar_like<TIN>(i2, 65535)];
		s = 1;
	}
	else if (first == VIPS_HIP_COLOUR_LabS2Lab) {
		// LabS2Lab.c:55-69 on the vips_cast_short'ed input
		v.a = (float) __ddiv_rn((double) load_as_short<TIN>(i0), 32767.0 / 100.0);
		v.b = (float) __ddiv_rn((double) load_as_short<TIN>(i1), 32768.0 / 128.0);
		v.c = (float) __ddiv_rn((double) load_as_short<TIN>(i2), 32768.0 / 128.0);
		s = 1;
	}
	else {
		// colour transforms see vips_cast_float'ed input (colour.c:343-348)
		v.a = (float) i0;
		v.b = (float) i1;
		v.c = (float) i2;
	}

	// ---- float -> float steps
	int last = -1;
#pragma unroll
	for (int k = 0; k < 4; k++) { // (a ROUTE has at most 4 steps; the run-time loop takes what is left)
		if (!ROUTE)
			break;
		if (k < s || k >= n_steps)
			continue;
		const int st = kStaticRoutes[ROUTE][1 + k];
		if (st == VIPS_HIP_COLOUR_scRGB2XYZ)
			v = step_scRGB2XYZ(v);
		else if (st == VIPS_HIP_COLOUR_XYZ2Lab)
			v = step_XYZ2Lab(v, a.tables.cbrt);
		else if (st == VIPS_HIP_COLOUR_Lab2XYZ)
			v = step_Lab2XYZ(v);
		else if (st == VIPS_HIP_COLOUR_XYZ2scRGB)
			v = step_XYZ2scRGB(v);
		else
			last = st;
	}
	for (; !ROUTE && s < n_steps; s++) {
		const int st = a.steps[s];
		if (st == VIPS_HIP_COLOUR_scRGB2XYZ)
			v = step_scRGB2XYZ(v);
		else if (st == VIPS_HIP_COLOUR_XYZ2Lab)
			v = step_XYZ2Lab(v, a.tables.cbrt);
		else if (st == VIPS_HIP_COLOUR_Lab2XYZ)
			v = step_Lab2XYZ(v);
		else if (st == VIPS_HIP_COLOUR_XYZ2scRGB)
			v = step_XYZ2scRGB(v);
		else
			last = st; // a coding step: must be the final one
	}

	// ---- the last step fixes the stored format
	if (last == VIPS_HIP_COLOUR_scRGB2sRGB || last == VIPS_HIP_COLOUR_scRGB2sRGB16) {
		const bool wide = last == VIPS_HIP_COLOUR_scRGB2sRGB16;
		int r = 0, g = 0, b = 0;
		if (!(isnan(v.a) || isnan(v.b) || isnan(v.c))) {
			// (two arms: the tables may live in different address spaces)
			if (wide) {
				r = scRGB2sRGB_channel(a.tables.Y2v_16, v.a, 65535);
				g = scRGB2sRGB_channel(a.tables.Y2v_16, v.b, 65535);
				b = scRGB2sRGB_channel(a.tables.Y2v_16, v.c, 65535);
			}
			else {
				r = scRGB2sRGB_channel(Y2v8, v.a, 255);
				g = scRGB2sRGB_channel(Y2v8, v.b, 255);
				b = scRGB2sRGB_channel(Y2v8, v.c, 255);
			}
		}
		o0 = (TOUT) r;
		o1 = (TOUT) g;
		o2 = (TOUT) b;
	}
	else if (last == VIPS_HIP_COLOUR_Lab2LabS) {
		o0 = (TOUT) lab2labs(v.a, 32767.0 / 100.0, 0.0);
		o1 = (TOUT) lab2labs(v.b, 32768.0 / 128.0, -32768.0);
		o2 = (TOUT) lab2labs(v.c, 32768.0 / 128.0, -32768.0);
	}
	else {
		o0 = vh::cvt_to<TOUT>(v.a);
		o1 = vh::cvt_to<TOUT>(v.b);
		o2 = vh::cvt_to<TOUT>(v.c);
	}
}

// The two fixed chains of vips_sharpen on uchar sRGB (sharpen.c:214, 285), without the step
// interpreter of route_pixel: sRGB2scRGB -> scRGB2XYZ -> XYZ2Lab -> Lab2LabS and
// LabS2Lab -> Lab2XYZ -> XYZ2scRGB -> scRGB2sRGB.  Same steps, same roundings.
// (v2Y / Y2v: the 8-bit tables, wherever the caller keeps them -- LDS in the sharpen kernel)
// lab2labs for a finite value: the same clip with v_min_f64 / v_max_f64
static __device__ __forceinline__ short lab2labs_finite(float v, double scale, double lo)
{
	const double d = __dmul_rn((double) v, scale);
	return (short) vh::cvt_i32(__builtin_fmax(lo, __builtin_fmin(d, 32767.0)));
}

// step_XYZ2scRGB for finite input
static __device__ __forceinline__ Px step_XYZ2scRGB_finite(Px p)
{
	const float X = (float) DIV_CONST_F((double) p.a, 100.0);
	const float Y = (float) DIV_CONST_F((double) p.b, 100.0);
	const float Z = (float) DIV_CONST_F((double) p.c, 100.0);
	Px q;
	q.a = __fadd_rn(__fadd_rn(__fmul_rn(3.240625F, X), __fmul_rn(-1.537208F, Y)), __fmul_rn(-0.498629F, Z));
	q.b = __fadd_rn(__fadd_rn(__fmul_rn(-0.968931F, X), __fmul_rn(1.875756F, Y)), __fmul_rn(0.041518F, Z));
	q.c = __fadd_rn(__fadd_rn(__fmul_rn(0.055710F, X), __fmul_rn(-0.204021F, Y)), __fmul_rn(1.056996F, Z));
	return q;
}

template <bool WANT_AB>
static __device__ __forceinline__ void srgb8_to_labs(const ColourTables &tb, const float *v2Y, int r, int g, int b,
	short &L, short &A, short &B)
{
	Px v;
	v.a = v2Y[r];
	v.b = v2Y[g];
	v.c = v2Y[b];
	v = step_scRGB2XYZ(v);
	const float cby = cbrt_lerp<1, true>(tb.cbrt, v.b);
	L = lab2labs_finite(__fsub_rn(__fmul_rn(116.0F, cby), 16.0F), 32767.0 / 100.0, 0.0);
	if (WANT_AB) {
		const float cbx = cbrt_lerp<0, true>(tb.cbrt, v.a);
		const float cbz = cbrt_lerp<2, true>(tb.cbrt, v.c);
		A = lab2labs_finite(__fmul_rn(500.0F, __fsub_rn(cbx, cby)), 32768.0 / 128.0, -32768.0);
		B = lab2labs_finite(__fmul_rn(200.0F, __fsub_rn(cby, cbz)), 32768.0 / 128.0, -32768.0);
	}
}

static __device__ __forceinline__ void labs_to_srgb8(const int *Y2v, int L, int A, int B, unsigned char &r,
	unsigned char &g, unsigned char &b)
{
	Px v;
	// LabS2Lab.c:55-69: / (32767 / 100) correctly rounded; / 256 is exact
	v.a = (float) DIV_CONST_F((double) L, 32767.0 / 100.0);
	v.b = __fmul_rn((float) A, 0.00390625f);
	v.c = __fmul_rn((float) B, 0.00390625f);
	v = step_Lab2XYZ_c(v);
	v = step_XYZ2scRGB_finite(v);
	// (finite: the NaN test of vips_col_scRGB2sRGB cannot fire)
	r = (unsigned char) scRGB2sRGB_channel(Y2v, v.a, 255);
	g = (unsigned char) scRGB2sRGB_channel(Y2v, v.b, 255);
	b = (unsigned char) scRGB2sRGB_channel(Y2v, v.c, 255);
}

// Fill the steps and table pointers of a RouteArgs (tables are built and uploaded on first use);
// 0 on success.  Defined in colour.hip.
int colour_route_prepare(const int *steps, int n_steps, RouteArgs *a);

} // namespace vh
