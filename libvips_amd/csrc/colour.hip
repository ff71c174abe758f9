// Colour path for gfx950: the per-scanline process_line functions of colour/
// (colour.c:119-156 drives them one scanline at a time) as ONE kernel per region.
//
// A colourspace conversion in the reference is a chain of images, each step
// rounding its result to float (or uchar / short) in memory:
//     sRGB -> LAB = cast(uchar), sRGB2scRGB, scRGB2XYZ, XYZ2Lab   (colourspace.c:362)
// Every step is per-pixel, so the chain is evaluated here in registers with the same
// intermediate types and the same operation order: one read and one write per pixel
// instead of four of each, identical bits.  Transcendentals never run on the device:
// the LUTs the reference builds with powf()/cbrtf() (LabQ2sRGB.c:130-160,
// XYZ2Lab.c:92-106) are built on the host by the same libm calls and uploaded.
//
// Float arithmetic: the reference is compiled for baseline x86-64 (SSE2, no FMA), so
// every multiply and add below is a separate IEEE operation (__fmul_rn/__fadd_rn,
// __dmul_rn/...; the file is also built with -ffp-contract=off).
#include "colour.h"

#include <climits>
#include <cmath>
#include <mutex>

namespace vh {

// The pointwise kernels below give a thread one element column and RU rows per loop trip,
// all RU loads issued before the first is used: with one load in flight per thread these
// kernels sit at ~1 TB/s (bytes in flight, not bandwidth, bound them); four gets 3-4x.
constexpr int RU = 4;

// grid.y for a row-looping kernel: enough blocks to fill the part, few enough that per-block
// setup is amortised over many rows
static inline int rows_grid(int gx, int height)
{
	const int groups = (height + RU - 1) / RU;
	int gy = 16384 / (gx > 0 ? gx : 1);
	gy = gy < 1 ? 1 : gy;
	return groups < gy ? groups : gy;
}

// ------------------------------------------------------------------ tables

struct ColourTables {
	float *v2Y_8;   // 256     sRGB2scRGB, 8 bit    LabQ2sRGB.c:151-159
	float *v2Y_16;  // 65536
	int *Y2v_8;     // 257     scRGB2sRGB           LabQ2sRGB.c:134-149
	int *Y2v_16;    // 65537
	float *cbrt;    // 100000  XYZ2Lab              XYZ2Lab.c:92-106
};

static std::mutex &g_tables_mutex = *new std::mutex;
static ColourTables g_tables = { nullptr, nullptr, nullptr, nullptr, nullptr };

// calcul_tables(), LabQ2sRGB.c:130-160
static void calcul_tables(int range, std::vector<int> &Y2v, std::vector<float> &v2Y)
{
	Y2v.resize(range + 1);
	v2Y.resize(range);
	for (int i = 0; i < range; i++) {
		float f = (float) i / (range - 1);
		float v;

		if (f <= 0.0031308)
			v = 12.92F * f;
		else
			v = (1.0F + 0.055F) * powf(f, 1.0F / 2.4F) - 0.055F;

		Y2v[i] = rintf((range - 1) * v);
	}
	Y2v[range] = Y2v[range - 1];

	for (int i = 0; i < range; i++) {
		float f = (float) i / (range - 1);

		if (f <= 0.04045)
			v2Y[i] = f / 12.92F;
		else
			v2Y[i] = powf((f + 0.055F) / (1 + 0.055F), 2.4F);
	}
}

static int ensure_tables()
{
	std::lock_guard<std::mutex> lock(g_tables_mutex);
	if (g_tables.cbrt)
		return 0;
	std::vector<int> Y2v;
	std::vector<float> v2Y;
	calcul_tables(256, Y2v, v2Y);
	g_tables.Y2v_8 = (int *) upload(Y2v.data(), Y2v.size() * sizeof(int));
	g_tables.v2Y_8 = (float *) upload(v2Y.data(), v2Y.size() * sizeof(float));
	calcul_tables(65536, Y2v, v2Y);
	g_tables.Y2v_16 = (int *) upload(Y2v.data(), Y2v.size() * sizeof(int));
	g_tables.v2Y_16 = (float *) upload(v2Y.data(), v2Y.size() * sizeof(float));
	// table_init(), XYZ2Lab.c:92-106
	const int QUANT_ELEMENTS = 100000;
	std::vector<float> cb(QUANT_ELEMENTS);
	for (int i = 0; i < QUANT_ELEMENTS; i++) {
		float Y = (double) i / QUANT_ELEMENTS;

		if (Y < 0.008856)
			cb[i] = 7.787F * Y + (16.0F / 116.0F);
		else
			cb[i] = cbrtf(Y);
	}
	float *cbrt = (float *) upload(cb.data(), cb.size() * sizeof(float));
	if (!g_tables.Y2v_8 || !g_tables.v2Y_8 || !g_tables.Y2v_16 || !g_tables.v2Y_16 || !cbrt)
		return -1;
	g_tables.cbrt = cbrt;
	return 0;
}

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
	// an infinite quotient has no finite residual: keep it (the reference gets inf too)
	return isinf(q0) ? q0 : q1;
}
#define DIV_CONST(A, Y) div_const((A), (Y), 1.0 / (Y))

// vips_col_XYZ2Lab_helper, XYZ2Lab.c:109-138 (D65: include/vips/colour.h:58-60)
template <int WHICH>
static __device__ __forceinline__ float cbrt_lerp(const float *__restrict__ table, float v)
{
	// nX = QUANT_ELEMENTS * X / X0: (int * float) in float, then / double, back to float
	const double num = (double) __fmul_rn(100000.0f, v);
	const float n = (float) (WHICH == 0 ? DIV_CONST(num, 95.0470)
							 : WHICH == 1 ? DIV_CONST(num, 100.0)
										  : DIV_CONST(num, 108.8827));
	// VIPS_CLIP(0, (int) nX, QUANT_ELEMENTS - 2); (int) of NaN / overflow is the x86
	// "integer indefinite" INT_MIN, which the clip turns into 0
	int i;
	if (!(n > -2147483904.0f && n < 2147483648.0f))
		i = INT_MIN;
	else
		i = (int) n;
	i = min(max(i, 0), 100000 - 2);
	const float f = __fsub_rn(n, (float) i);
	const float t0 = table[i];
	return __fadd_rn(t0, __fmul_rn(f, __fsub_rn(table[i + 1], t0)));
}

static __device__ __forceinline__ Px step_XYZ2Lab(Px p, const float *__restrict__ table)
{
	const float cbx = cbrt_lerp<0>(table, p.a);
	const float cby = cbrt_lerp<1>(table, p.b);
	const float cbz = cbrt_lerp<2>(table, p.c);
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
	const int Yi = (int) Yf;
	const int l0 = lut[Yi];
	const float r =
		__fadd_rn((float) l0, __fmul_rn((float) (lut[Yi + 1] - l0), __fsub_rn(Yf, (float) Yi)));
	return (int) rintf(r);
}

// vips_Lab2LabS_line, Lab2LabS.c:59-73: double multiply, clip, truncate
static __device__ __forceinline__ short lab2labs(float v, double scale, double lo)
{
	double d = __dmul_rn((double) v, scale);
	d = d > 32767.0 ? 32767.0 : d; // VIPS_MIN(B, V)
	d = lo > d ? lo : d;           // VIPS_MAX(A, ...)
	return (short) d;
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
	// CAST_FLOAT_INT: clip as double, then C truncation
	double d = (double) v;
	d = (double) maxv < d ? (double) maxv : d;
	d = 0.0 > d ? 0.0 : d;
	return (int) d;
}
template <>
__device__ __forceinline__ int load_as_uchar_like<double>(double v, int maxv)
{
	double d = v;
	d = (double) maxv < d ? (double) maxv : d;
	d = 0.0 > d ? 0.0 : d;
	return (int) d;
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
	return (int) d;
}
template <>
__device__ __forceinline__ int load_as_short<double>(double v)
{
	double d = v;
	d = 32767.0 < d ? 32767.0 : d;
	d = -32768.0 > d ? -32768.0 : d;
	return (int) d;
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
	return (unsigned char) d;
}
template <>
__device__ __forceinline__ unsigned short cast_from_double<unsigned short>(double d)
{
	d = 65535.0 < d ? 65535.0 : d;
	d = 0.0 > d ? 0.0 : d;
	return (unsigned short) d;
}
template <>
__device__ __forceinline__ short cast_from_double<short>(double d)
{
	d = 32767.0 < d ? 32767.0 : d;
	d = -32768.0 > d ? -32768.0 : d;
	return (short) d;
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

// The chain for one pixel: stored bands in, stored bands out.
template <typename TIN, typename TOUT>
static __device__ __forceinline__ void route_pixel(const RouteArgs &a, TIN i0, TIN i1, TIN i2,
	TOUT &o0, TOUT &o1, TOUT &o2)
{
	Px v;
	const int first = a.steps[0];
	int s = 0;
	// ---- the first step fixes how the stored bands are interpreted
	if (first == VIPS_HIP_COLOUR_sRGB2scRGB) {
		// vips_colour_code_build casts to uchar (colour.c:428-434), sRGB2scRGB.c:72-90
		v.a = a.tables.v2Y_8[load_as_uchar_like<TIN>(i0, 255)];
		v.b = a.tables.v2Y_8[load_as_uchar_like<TIN>(i1, 255)];
		v.c = a.tables.v2Y_8[load_as_uchar_like<TIN>(i2, 255)];
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
	for (; s < a.n_steps; s++) {
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
		const int *lut = wide ? a.tables.Y2v_16 : a.tables.Y2v_8;
		const int maxval = wide ? 65535 : 255;
		int r = 0, g = 0, b = 0;
		if (!(isnan(v.a) || isnan(v.b) || isnan(v.c))) {
			r = scRGB2sRGB_channel(lut, v.a, maxval);
			g = scRGB2sRGB_channel(lut, v.b, maxval);
			b = scRGB2sRGB_channel(lut, v.c, maxval);
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
		o0 = (TOUT) v.a;
		o1 = (TOUT) v.b;
		o2 = (TOUT) v.c;
	}
}

// One thread per pixel.  TIN/TOUT are the in-memory band formats.
template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256)
colour_route_kernel(RouteArgs a)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
		TIN i0[RU], i1[RU], i2[RU];
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = min(y0 + r, a.height - 1);
			const TIN *p = (const TIN *) (a.in + (long long) y * a.in_stride) + (long long) x * a.in_bands;
			i0[r] = p[0];
			i1[r] = p[1];
			i2[r] = p[2];
		}
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = y0 + r;
			if (y < a.height) {
				const TIN *p = (const TIN *) (a.in + (long long) y * a.in_stride) + (long long) x * a.in_bands;
				TOUT *q = (TOUT *) (a.out + (long long) y * a.out_stride) + (long long) x * a.out_bands;
				route_pixel<TIN, TOUT>(a, i0[r], i1[r], i2[r], q[0], q[1], q[2]);
				for (int e = 0; e < a.extra_bands; e++)
					q[3 + e] = Carry<TIN, TOUT>::run(p[3 + e], a.alpha_scale);
			}
		}
	}
}

// 3 bands -> 3 bands, 4 pixels per thread: the 12 input elements arrive as three aligned
// vector loads and the 12 outputs leave as three aligned vector stores (dword / dwordx2 /
// dwordx4 for 1 / 2 / 4-byte elements).  One element per store instruction, as the scalar
// kernel does for interleaved bands, makes every wave store touch all of the wave's cache
// lines (3x the L1->L2 write requests); this is the shape of every BASELINE colour config
// (C3: float -> float, thumbnails: uchar -> float / short and back).
template <typename T>
struct __attribute__((aligned(4 * sizeof(T)))) Vec4 {
	T v[4];
};

template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256)
colour_route_x4_kernel(RouteArgs a)
{
	const int x4 = blockIdx.x * blockDim.x + threadIdx.x;
	if (x4 * 4 >= a.width)
		return;
	for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
		const Vec4<TIN> *p = (const Vec4<TIN> *) (a.in + (long long) y * a.in_stride) + (long long) x4 * 3;
		Vec4<TOUT> *q = (Vec4<TOUT> *) (a.out + (long long) y * a.out_stride) + (long long) x4 * 3;
		const Vec4<TIN> v0 = p[0], v1 = p[1], v2 = p[2];
		Vec4<TOUT> r0, r1, r2;
		route_pixel<TIN, TOUT>(a, v0.v[0], v0.v[1], v0.v[2], r0.v[0], r0.v[1], r0.v[2]);
		route_pixel<TIN, TOUT>(a, v0.v[3], v1.v[0], v1.v[1], r0.v[3], r1.v[0], r1.v[1]);
		route_pixel<TIN, TOUT>(a, v1.v[2], v1.v[3], v2.v[0], r1.v[2], r1.v[3], r2.v[0]);
		route_pixel<TIN, TOUT>(a, v2.v[1], v2.v[2], v2.v[3], r2.v[1], r2.v[2], r2.v[3]);
		q[0] = r0;
		q[1] = r1;
		q[2] = r2;
	}
}

// ------------------------------------------------------------------- cast

template <typename TIN, typename TOUT>
struct CastOne;

// CAST_INT_INT: through int (<= 16 bit targets) or int64 (32 bit targets)
#define CAST_II(TIN, TOUT, TEMP, LO, HI) \
	template <> \
	struct CastOne<TIN, TOUT> { \
		static __device__ __forceinline__ TOUT run(TIN v) \
		{ \
			TEMP t = (TEMP) v; \
			t = t > (TEMP) (HI) ? (TEMP) (HI) : t; \
			t = t < (TEMP) (LO) ? (TEMP) (LO) : t; \
			return (TOUT) t; \
		} \
	};
#define CAST_II_ALL(TIN) \
	CAST_II(TIN, unsigned char, int, 0, UCHAR_MAX) \
	CAST_II(TIN, signed char, int, SCHAR_MIN, SCHAR_MAX) \
	CAST_II(TIN, unsigned short, int, 0, USHRT_MAX) \
	CAST_II(TIN, short, int, SHRT_MIN, SHRT_MAX) \
	CAST_II(TIN, unsigned int, long long, 0, UINT_MAX) \
	CAST_II(TIN, int, long long, INT_MIN, INT_MAX)
CAST_II_ALL(unsigned char)
CAST_II_ALL(signed char)
CAST_II_ALL(unsigned short)
CAST_II_ALL(short)
CAST_II_ALL(unsigned int)
CAST_II_ALL(int)
#undef CAST_II_ALL
#undef CAST_II

// CAST_FLOAT_INT: clip as double, then C truncation
#define CAST_FI(TIN, TOUT, LO, HI) \
	template <> \
	struct CastOne<TIN, TOUT> { \
		static __device__ __forceinline__ TOUT run(TIN v) \
		{ \
			double d = (double) v; \
			d = (double) (HI) < d ? (double) (HI) : d; \
			d = (double) (LO) > d ? (double) (LO) : d; \
			return (TOUT) d; \
		} \
	};
#define CAST_FI_ALL(TIN) \
	CAST_FI(TIN, unsigned char, 0, UCHAR_MAX) \
	CAST_FI(TIN, signed char, SCHAR_MIN, SCHAR_MAX) \
	CAST_FI(TIN, unsigned short, 0, USHRT_MAX) \
	CAST_FI(TIN, short, SHRT_MIN, SHRT_MAX) \
	CAST_FI(TIN, unsigned int, 0, UINT_MAX) \
	CAST_FI(TIN, int, INT_MIN, INT_MAX)
CAST_FI_ALL(float)
CAST_FI_ALL(double)
#undef CAST_FI_ALL
#undef CAST_FI

// CAST_REAL_FLOAT: plain conversion
#define CAST_RF(TIN, TOUT) \
	template <> \
	struct CastOne<TIN, TOUT> { \
		static __device__ __forceinline__ TOUT run(TIN v) { return (TOUT) v; } \
	};
#define CAST_RF_ALL(TIN) \
	CAST_RF(TIN, float) \
	CAST_RF(TIN, double)
CAST_RF_ALL(unsigned char)
CAST_RF_ALL(signed char)
CAST_RF_ALL(unsigned short)
CAST_RF_ALL(short)
CAST_RF_ALL(unsigned int)
CAST_RF_ALL(int)
CAST_RF_ALL(float)
CAST_RF_ALL(double)
#undef CAST_RF_ALL
#undef CAST_RF

struct CastArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int ne; // elements per row copied
	int height;
	// band remap for extract_band / bandjoin: element e of the output row reads
	// input pel e / out_take, band in_first + e % out_take; writes band out_first + ...
	int in_bands, out_bands, in_first, out_first, take;
};

template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256)
cast_kernel(CastArgs a)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.ne)
		return;
	const int x = e / a.take;
	const int b = e - x * a.take;
	const long long ie = (long long) x * a.in_bands + a.in_first + b;
	const long long oe = (long long) x * a.out_bands + a.out_first + b;
	for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
		TIN v[RU];
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = min(y0 + r, a.height - 1);
			v[r] = ((const TIN *) (a.in + (long long) y * a.in_stride))[ie];
		}
#pragma unroll
		for (int r = 0; r < RU; r++)
			if (y0 + r < a.height)
				((TOUT *) (a.out + (long long) (y0 + r) * a.out_stride))[oe] = CastOne<TIN, TOUT>::run(v[r]);
	}
}

template <typename TIN>
static int launch_cast_out(const CastArgs &a, int out_format)
{
	dim3 block(256, 1, 1);
	dim3 grid((a.ne + 255) / 256, rows_grid((a.ne + 255) / 256, a.height), 1);
	Gate gate("cast");
#define GO(TOUT) \
	hipLaunchKernelGGL((cast_kernel<TIN, TOUT>), grid, block, 0, stream(), a); \
	break;
	switch (out_format) {
	case VIPS_HIP_FORMAT_UCHAR: GO(unsigned char)
	case VIPS_HIP_FORMAT_CHAR: GO(signed char)
	case VIPS_HIP_FORMAT_USHORT: GO(unsigned short)
	case VIPS_HIP_FORMAT_SHORT: GO(short)
	case VIPS_HIP_FORMAT_UINT: GO(unsigned int)
	case VIPS_HIP_FORMAT_INT: GO(int)
	case VIPS_HIP_FORMAT_FLOAT: GO(float)
	case VIPS_HIP_FORMAT_DOUBLE: GO(double)
	default:
		error("cast", "unsupported band format %d", out_format);
		return -1;
	}
#undef GO
	VH_CHECK(hipGetLastError());
	return 0;
}

int band_cast(const VipsHipRegion *in, int in_first, const VipsHipRegion *out, int out_first, int take)
{
	if (ensure_init())
		return -1;
	if (check_region("cast", in) || check_region("cast", out))
		return -1;
	if (format_iscomplex(in->format) || format_iscomplex(out->format)) {
		error("cast", "complex formats are outside the HIP path");
		return -1;
	}
	if (out->left < in->left || out->top < in->top ||
		out->left + out->width > in->left + in->width ||
		out->top + out->height > in->top + in->height) {
		error("cast", "input region too small");
		return -1;
	}
	if (in_first < 0 || take <= 0 || in_first + take > in->bands || out_first < 0 ||
		out_first + take > out->bands) {
		error("cast", "bad band range");
		return -1;
	}
	CastArgs a;
	const int ies = format_sizeof(in->format);
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride +
		(size_t) (out->left - in->left) * in->bands * ies;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.ne = out->width * take;
	a.height = out->height;
	a.in_bands = in->bands;
	a.out_bands = out->bands;
	a.in_first = in_first;
	a.out_first = out_first;
	a.take = take;
	switch (in->format) {
	case VIPS_HIP_FORMAT_UCHAR: return launch_cast_out<unsigned char>(a, out->format);
	case VIPS_HIP_FORMAT_CHAR: return launch_cast_out<signed char>(a, out->format);
	case VIPS_HIP_FORMAT_USHORT: return launch_cast_out<unsigned short>(a, out->format);
	case VIPS_HIP_FORMAT_SHORT: return launch_cast_out<short>(a, out->format);
	case VIPS_HIP_FORMAT_UINT: return launch_cast_out<unsigned int>(a, out->format);
	case VIPS_HIP_FORMAT_INT: return launch_cast_out<int>(a, out->format);
	case VIPS_HIP_FORMAT_FLOAT: return launch_cast_out<float>(a, out->format);
	case VIPS_HIP_FORMAT_DOUBLE: return launch_cast_out<double>(a, out->format);
	default:
		error("cast", "unsupported band format %d", in->format);
		return -1;
	}
}

// ------------------------------------------------------------------ sharpen

struct SharpenArgs {
	const unsigned char *in, *blur;
	unsigned char *out;
	long long in_stride, blur_stride, out_stride;
	int width, height, bands;
	const int *lut;
};

// vips_sharpen_generate, sharpen.c:116-168, on band 0 of a LabS image; the other
// bands are copied (the reference splits them off and joins them back, :274-295).
__global__ void __launch_bounds__(256)
sharpen_kernel(SharpenArgs a)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
		int v1[RU], v2[RU];
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = min(y0 + r, a.height - 1);
			v1[r] = ((const short *) (a.in + (long long) y * a.in_stride))[(long long) x * a.bands];
			v2[r] = ((const short *) (a.blur + (long long) y * a.blur_stride))[x];
		}
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = y0 + r;
			if (y < a.height) {
				const short *p1 = (const short *) (a.in + (long long) y * a.in_stride) + (long long) x * a.bands;
				short *q = (short *) (a.out + (long long) y * a.out_stride) + (long long) x * a.bands;
				const int diff = (v1[r] & 0x7fff) - (v2[r] & 0x7fff);
				int out = v1[r] + a.lut[diff + 32768];
				out = min(max(out, 0), 32767);
				q[0] = (short) out;
				for (int b = 1; b < a.bands; b++)
					q[b] = p1[b];
			}
		}
	}
}

// ------------------------------------------------- premultiply / unpremultiply

struct PremulArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int width, height, bands;
	double max_alpha;
	int scale[256]; // the uchar fast path's table (premultiply.c:252-258, unpremultiply.c:316-323)
};

// fast uchar -> uchar path: premultiply.c:163-176, unpremultiply.c:222-235
__global__ void __launch_bounds__(256)
premul_u8_kernel(PremulArgs a)
{
	__shared__ int scale[256];
	scale[threadIdx.x] = a.scale[threadIdx.x];
	__syncthreads();
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	if (a.bands == 4) {
		for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
			unsigned int vv[RU];
#pragma unroll
			for (int r = 0; r < RU; r++) {
				const int y = min(y0 + r, a.height - 1);
				vv[r] = *reinterpret_cast<const unsigned int *>(a.in + (long long) y * a.in_stride + (long long) x * 4);
			}
#pragma unroll
			for (int r = 0; r < RU; r++) {
				if (y0 + r >= a.height)
					break;
				const unsigned int v = vv[r];
				const unsigned int alpha = v >> 24;
				const int s = scale[alpha];
				const unsigned int cr = ((int) (v & 0xff) * s + 128) >> 8;
				const unsigned int cg = ((int) ((v >> 8) & 0xff) * s + 128) >> 8;
				const unsigned int cb = ((int) ((v >> 16) & 0xff) * s + 128) >> 8;
				*reinterpret_cast<unsigned int *>(a.out + (long long) (y0 + r) * a.out_stride + (long long) x * 4) =
					(cr & 0xff) | ((cg & 0xff) << 8) | ((cb & 0xff) << 16) | (alpha << 24);
			}
		}
		return;
	}
	for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
		const unsigned char *p = a.in + (long long) y * a.in_stride + (long long) x * a.bands;
		unsigned char *q = a.out + (long long) y * a.out_stride + (long long) x * a.bands;
		{
			const int alpha = p[a.bands - 1];
			const int s = scale[alpha];
			for (int i = 0; i < a.bands - 1; i++)
				q[i] = (unsigned char) ((p[i] * s + 128) >> 8);
			q[a.bands - 1] = (unsigned char) alpha;
		}
	}
}

// PRE_MANY / PRE_RGBA (premultiply.c:78-128) and UNPRE / FUNPRE (unpremultiply.c:85-186),
// float output.
// one pixel of NB bands (alpha last); ptr-free so that the vector path stays in registers
template <typename TIN, bool INVERSE, int NB>
static __device__ __forceinline__ void premul_pixel(const TIN (&p)[NB], float (&q)[NB], double max_alpha)
{
	constexpr int ab = NB - 1;
	const TIN alpha = p[ab];
	// VIPS_CLIP(0, alpha, max_alpha) is evaluated in double
	double clip = (double) alpha;
	clip = max_alpha < clip ? max_alpha : clip;
	clip = 0.0 > clip ? 0.0 : clip;
	if (!INVERSE) {
		// IN clip_alpha = CLIP(...); OUT nalpha = (OUT) clip_alpha / max_alpha
		const TIN clip_alpha = (TIN) clip;
		const float nalpha = (float) __ddiv_rn((double) (float) clip_alpha, max_alpha);
#pragma unroll
		for (int i = 0; i < ab; i++)
			q[i] = __fmul_rn((float) p[i], nalpha);
		q[ab] = (float) alpha;
	}
	else {
		float factor;
		if (sizeof(TIN) == 4 && ((TIN) 0.5f != (TIN) 0)) // float input: FUNPRE
			factor = fabs((double) alpha) < 0.01 ? 0.0f : (float) __ddiv_rn(max_alpha, (double) alpha);
		else
			factor = alpha == (TIN) 0 ? 0.0f : (float) __ddiv_rn(max_alpha, (double) alpha);
#pragma unroll
		for (int i = 0; i < ab; i++)
			q[i] = __fmul_rn(factor, (float) p[i]);
		q[ab] = (float) clip;
	}
}

template <typename TIN, bool INVERSE>
__global__ void __launch_bounds__(256)
premul_float_kernel(PremulArgs a)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	// RGBA with aligned rows: one vector load and one 16-byte store per pixel
	if (a.bands == 4 && !(((uintptr_t) a.in | (uintptr_t) a.in_stride) % (4 * sizeof(TIN))) &&
		!(((uintptr_t) a.out | (uintptr_t) a.out_stride) & 15)) {
		for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
			const Vec4<TIN> pv = ((const Vec4<TIN> *) (a.in + (long long) y * a.in_stride))[x];
			float qv[4];
			premul_pixel<TIN, INVERSE, 4>(pv.v, qv, a.max_alpha);
			((float4 *) (a.out + (long long) y * a.out_stride))[x] = make_float4(qv[0], qv[1], qv[2], qv[3]);
		}
		return;
	}
	const int ab = a.bands - 1;
	for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
		const TIN *p = (const TIN *) (a.in + (long long) y * a.in_stride) + (long long) x * a.bands;
		float *q = (float *) (a.out + (long long) y * a.out_stride) + (long long) x * a.bands;
		const TIN alpha = p[ab];
		// VIPS_CLIP(0, alpha, max_alpha) is evaluated in double
		double clip = (double) alpha;
		clip = a.max_alpha < clip ? a.max_alpha : clip;
		clip = 0.0 > clip ? 0.0 : clip;
		if (!INVERSE) {
			// IN clip_alpha = CLIP(...); OUT nalpha = (OUT) clip_alpha / max_alpha
			const TIN clip_alpha = (TIN) clip;
			const float nalpha = (float) __ddiv_rn((double) (float) clip_alpha, a.max_alpha);
			for (int i = 0; i < ab; i++)
				q[i] = __fmul_rn((float) p[i], nalpha);
			q[ab] = (float) alpha;
		}
		else {
			float factor;
			if (sizeof(TIN) == 4 && ((TIN) 0.5f != (TIN) 0)) // float input: FUNPRE
				factor = fabs((double) alpha) < 0.01 ? 0.0f : (float) __ddiv_rn(a.max_alpha, (double) alpha);
			else
				factor = alpha == (TIN) 0 ? 0.0f : (float) __ddiv_rn(a.max_alpha, (double) alpha);
			for (int i = 0; i < ab; i++)
				q[i] = __fmul_rn(factor, (float) p[i]);
			q[ab] = (float) clip;
		}
	}
}

int premultiply_region(const VipsHipRegion *in, const VipsHipRegion *out, double max_alpha, int uchar,
	int inverse)
{
	const char *domain = inverse ? "unpremultiply" : "premultiply";
	if (ensure_init())
		return -1;
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->bands != out->bands || in->bands < 2) {
		error(domain, "need the same number of bands (at least 2) in and out");
		return -1;
	}
	if (out->left < in->left || out->top < in->top ||
		out->left + out->width > in->left + in->width ||
		out->top + out->height > in->top + in->height) {
		error(domain, "input region too small");
		return -1;
	}
	const bool fast = uchar && in->format == VIPS_HIP_FORMAT_UCHAR;
	if (out->format != (fast ? VIPS_HIP_FORMAT_UCHAR : VIPS_HIP_FORMAT_FLOAT)) {
		error(domain, "output region has the wrong format");
		return -1;
	}
	PremulArgs a;
	const int ies = format_sizeof(in->format);
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride +
		(size_t) (out->left - in->left) * in->bands * ies;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = out->width;
	a.height = out->height;
	a.bands = in->bands;
	a.max_alpha = max_alpha;
	for (int i = 0; i < 256; i++) {
		double clip = (double) i;
		clip = max_alpha < clip ? max_alpha : clip;
		clip = 0.0 > clip ? 0.0 : clip;
		if (inverse)
			a.scale[i] = clip == 0 ? 0 : (int) (256 * max_alpha / clip);
		else
			a.scale[i] = (int) (256 * clip / max_alpha);
	}
	dim3 block(256, 1, 1);
	dim3 grid((a.width + 255) / 256, rows_grid((a.width + 255) / 256, a.height), 1);
	Gate gate(inverse ? "unpremultiply" : "premultiply");
	if (fast) {
		if (a.bands == 4 && (((uintptr_t) a.in | (uintptr_t) a.out | a.in_stride | a.out_stride) & 3)) {
			error(domain, "RGBA rows must be 4-byte aligned");
			return -1;
		}
		hipLaunchKernelGGL(premul_u8_kernel, grid, block, 0, stream(), a);
	}
	else {
#define GO(TIN) \
	if (inverse) \
		hipLaunchKernelGGL((premul_float_kernel<TIN, true>), grid, block, 0, stream(), a); \
	else \
		hipLaunchKernelGGL((premul_float_kernel<TIN, false>), grid, block, 0, stream(), a); \
	break;
		switch (in->format) {
		case VIPS_HIP_FORMAT_UCHAR: GO(unsigned char)
		case VIPS_HIP_FORMAT_CHAR: GO(signed char)
		case VIPS_HIP_FORMAT_USHORT: GO(unsigned short)
		case VIPS_HIP_FORMAT_SHORT: GO(short)
		case VIPS_HIP_FORMAT_UINT: GO(unsigned int)
		case VIPS_HIP_FORMAT_INT: GO(int)
		case VIPS_HIP_FORMAT_FLOAT: GO(float)
		default:
			error(domain, "band format %d is outside the HIP path", in->format);
			return -1;
		}
#undef GO
	}
	VH_CHECK(hipGetLastError());
	return 0;
}

// ----------------------------------------------------------- route launcher

int colour_route(const int *steps, int n_steps, double alpha_scale, const VipsHipRegion *in,
	const VipsHipRegion *out)
{
	const char *domain = "colour";
	if (ensure_init())
		return -1;
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (n_steps < 1 || n_steps > 8) {
		error(domain, "bad route length %d", n_steps);
		return -1;
	}
	if (in->bands < 3) {
		error(domain, "image must have at least 3 bands"); // vips_check_bands_atleast
		return -1;
	}
	if (out->bands != in->bands) {
		error(domain, "output must have as many bands as the input");
		return -1;
	}
	if (out->left < in->left || out->top < in->top ||
		out->left + out->width > in->left + in->width ||
		out->top + out->height > in->top + in->height) {
		error(domain, "input region too small");
		return -1;
	}
	// what the last step stores
	const int last = steps[n_steps - 1];
	int want_out;
	if (last == VIPS_HIP_COLOUR_scRGB2sRGB)
		want_out = VIPS_HIP_FORMAT_UCHAR;
	else if (last == VIPS_HIP_COLOUR_scRGB2sRGB16)
		want_out = VIPS_HIP_FORMAT_USHORT;
	else if (last == VIPS_HIP_COLOUR_Lab2LabS)
		want_out = VIPS_HIP_FORMAT_SHORT;
	else
		want_out = VIPS_HIP_FORMAT_FLOAT;
	if (out->format != want_out) {
		error(domain, "output region has format %d, this conversion writes %d", out->format, want_out);
		return -1;
	}
	for (int s = 0; s < n_steps; s++) {
		const int st = steps[s];
		if (st < 0 || st >= VIPS_HIP_COLOUR_LAST) {
			error(domain, "unknown colour step %d", st);
			return -1;
		}
		const bool decoder = st == VIPS_HIP_COLOUR_sRGB2scRGB || st == VIPS_HIP_COLOUR_sRGB2scRGB16 ||
			st == VIPS_HIP_COLOUR_LabS2Lab;
		const bool encoder = st == VIPS_HIP_COLOUR_scRGB2sRGB || st == VIPS_HIP_COLOUR_scRGB2sRGB16 ||
			st == VIPS_HIP_COLOUR_Lab2LabS;
		if ((decoder && s != 0) || (encoder && s != n_steps - 1)) {
			error(domain, "colour step %d cannot sit at position %d of a fused route", st, s);
			return -1;
		}
	}
	if (ensure_tables())
		return -1;

	RouteArgs a;
	const int ies = format_sizeof(in->format);
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride +
		(size_t) (out->left - in->left) * in->bands * ies;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = out->width;
	a.height = out->height;
	a.in_bands = in->bands;
	a.out_bands = out->bands;
	a.n_steps = n_steps;
	for (int s = 0; s < 8; s++)
		a.steps[s] = s < n_steps ? steps[s] : -1;
	a.extra_bands = in->bands - 3;
	a.alpha_scale = alpha_scale;
	a.tables = g_tables;

	dim3 block(256, 1, 1);
	dim3 grid((a.width + 255) / 256, rows_grid((a.width + 255) / 256, a.height), 1);
	Gate gate("colour_route");
	// 3 bands in and out, rows that start and advance on 4-element boundaries: 4 pixels per thread
	const int oes = format_sizeof(want_out);
	const bool x4 = in->bands == 3 && out->bands == 3 && !(a.width & 3) &&
		!((uintptr_t) a.in % (4 * ies)) && !((uintptr_t) a.out % (4 * oes)) &&
		!(a.in_stride % (4 * ies)) && !(a.out_stride % (4 * oes));
	// (one row per block: the table-gathering route kernels measured faster with many short blocks)
	const dim3 grid4((a.width / 4 + 255) / 256, a.height < 32768 ? a.height : 32768, 1);
#define GO(TIN, TOUT) \
	if (x4) \
		hipLaunchKernelGGL((colour_route_x4_kernel<TIN, TOUT>), grid4, block, 0, stream(), a); \
	else \
		hipLaunchKernelGGL((colour_route_kernel<TIN, TOUT>), grid, block, 0, stream(), a)
#define GO_IN(TOUT) \
	switch (in->format) { \
	case VIPS_HIP_FORMAT_UCHAR: GO(unsigned char, TOUT); break; \
	case VIPS_HIP_FORMAT_USHORT: GO(unsigned short, TOUT); break; \
	case VIPS_HIP_FORMAT_SHORT: GO(short, TOUT); break; \
	case VIPS_HIP_FORMAT_FLOAT: GO(float, TOUT); break; \
	default: \
		error(domain, "input band format %d is outside the HIP colour path", in->format); \
		return -1; \
	}
	switch (want_out) {
	case VIPS_HIP_FORMAT_UCHAR: GO_IN(unsigned char) break;
	case VIPS_HIP_FORMAT_USHORT: GO_IN(unsigned short) break;
	case VIPS_HIP_FORMAT_SHORT: GO_IN(short) break;
	default: GO_IN(float) break;
	}
#undef GO_IN
#undef GO
	VH_CHECK(hipGetLastError());
	return 0;
}

} // namespace vh

using namespace vh;

extern "C" {

int vips_hip_colour_gen(int step, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return colour_route(&step, 1, 1.0, in, out);
}

int vips_hip_colour_route_gen(const int *steps, int n_steps, double alpha_scale,
	const VipsHipRegion *in, const VipsHipRegion *out)
{
	return colour_route(steps, n_steps, alpha_scale, in, out);
}

int vips_hip_premultiply_gen(const VipsHipRegion *in, const VipsHipRegion *out, double max_alpha,
	int uchar, int inverse)
{
	return premultiply_region(in, out, max_alpha, uchar, inverse);
}

int vips_hip_cast_gen(const VipsHipRegion *in, const VipsHipRegion *out)
{
	if (in && out && in->bands != out->bands) {
		error("cast", "input and output must have the same number of bands");
		return -1;
	}
	return band_cast(in, 0, out, 0, in ? in->bands : 0);
}

int vips_hip_sharpen_gen(const int *lut_device, const VipsHipRegion *in,
	const VipsHipRegion *blurred_l, const VipsHipRegion *out)
{
	const char *domain = "sharpen";
	if (ensure_init())
		return -1;
	if (check_region(domain, in) || check_region(domain, blurred_l) || check_region(domain, out))
		return -1;
	if (in->format != VIPS_HIP_FORMAT_SHORT || out->format != VIPS_HIP_FORMAT_SHORT ||
		blurred_l->format != VIPS_HIP_FORMAT_SHORT || blurred_l->bands != 1 ||
		in->bands != out->bands) {
		error(domain, "need LabS short images (and a 1-band blurred L)");
		return -1;
	}
	if (!lut_device) {
		error(domain, "null lut");
		return -1;
	}
	for (const VipsHipRegion *r : { in, blurred_l })
		if (out->left < r->left || out->top < r->top || out->left + out->width > r->left + r->width ||
			out->top + out->height > r->top + r->height) {
			error(domain, "input region too small");
			return -1;
		}
	SharpenArgs a;
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride +
		(size_t) (out->left - in->left) * in->bands * 2;
	a.blur = (const unsigned char *) blurred_l->data +
		(size_t) (out->top - blurred_l->top) * blurred_l->stride +
		(size_t) (out->left - blurred_l->left) * 2;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.blur_stride = (long long) blurred_l->stride;
	a.out_stride = (long long) out->stride;
	a.width = out->width;
	a.height = out->height;
	a.bands = in->bands;
	a.lut = lut_device;
	dim3 block(256, 1, 1);
	dim3 grid((a.width + 255) / 256, rows_grid((a.width + 255) / 256, a.height), 1);
	Gate gate("sharpen");
	hipLaunchKernelGGL(sharpen_kernel, grid, block, 0, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

} // extern "C"
