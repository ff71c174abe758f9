// XYZ2Lab's cube-root table (XYZ2Lab.c:92-106: 100000 floats made with the host's cbrtf) WITHOUT
// the table: every entry, bit for bit, from 30 KB that fit in LDS.
//
// Why: a lane that reads table[i] at a random i pulls a 128-byte line through its CU's L1 fill
// path; a wave's gather costs ~146 cycles of that path (tools/gather_probe.hip), the same path the
// streaming loads of an HBM-bound kernel need.  Three such gathers per pixel are what bounds
// vips_colourspace(sRGB -> Lab) and vips_sharpen on this part, not their arithmetic.
//
// How: entry i >= 886 is cbrtf((float) (i / 100000.0)), a smooth function sampled and rounded.
//   * i falls in block k = (bits of (float) i >> 18) - KEY0: the integers that share an exponent
//     and their top 5 mantissa bits, so (i - i0) / i0 < 1 / 32 for the block's first entry i0;
//   * in a block cbrt(x0 (1 + u)) = c0 (1 + u (1/3 - u/9 + 5 u^2 / 81 - 10 u^3 / 243)) to 1e-9
//     (the next term is 22/729 u^5 < 1e-9), c0 = cbrt(x0) and 1 / i0 stored per block as doubles;
//   * rounded to float this is the host's cbrtf within one unit in the last place (cbrtf itself
//     is not correctly rounded); the difference r[i] in {-1, 0, 1} is stored, 2 bits per entry.
// Entries below 886 are the table's linear arm, computed as the table code does.  The host makes
// r[] by running THIS function (same operations, same order, IEEE double: identical on both
// sides) against its own table and refuses the scheme if any entry does not come out: exact by
// construction, checked entry by entry when the tables are made (colour.hip cbrt_exact_tables()).
#pragma once

#include <cstdint>
#include <cstring>

#ifndef VH_CBRT_FN
#define VH_CBRT_FN static inline
#endif

namespace vh {

constexpr int CBRT_N = 100000;        // QUANT_ELEMENTS
constexpr int CBRT_LINEAR = 886;      // entries below are 7.787 Y + 16 / 116 (Y < 0.008856)
constexpr int CBRT_KEY0 = 4375;       // block key of entry 886: (bits of 886.0f) >> 18
constexpr int CBRT_BLOCKS = 218;      // keys 4375 .. 4592 (entry 99999)
constexpr int CBRT_RES_WORDS = CBRT_N / 16 + 2; // 2 bits per entry, 16 per word; a spare word for the pair read

struct CbrtBlockD {
	double c0, inv; // cbrt(i0 / 100000), 1 / i0
};
struct CbrtBlockI {
	int i0, count; // first entry, entries in the block
};
// the same block for the single-precision form below
struct CbrtBlockF {
	float c0, inv; // (float) cbrt(i0 / 100000), (float) (1 / i0)
};

// what a kernel keeps in LDS (or reads through these pointers)
struct CbrtExact {
	const CbrtBlockD *bd;
	const CbrtBlockI *bi;
	const unsigned int *res;
	const float *lin; // the 886 entries of the linear arm as they are, or NULL: computed
	// The single-precision form (round 4): the block polynomial evaluated in float --
	//     u = (float) j * inv,  p = fma(fma(fma(-10/243, u, 5/81), u, -1/9), u, 1/3),  c = fma(c0, u * p, c0)
	// (the last operation rounds c0 (1 + u p) once, so the float it makes is within one unit of
	// the host's cbrtf everywhere, like the double form: measured on the host's table, -1 / 0 / +1 in
	// 13.5 / 72 / 13.5 % of the entries) and its own 2-bit residuals.  Seven float operations per entry
	// where the double form takes eight double ones and a conversion, at half their cost each on
	// this part (tools/valu_probe2.hip).  NULL when this host's cbrtf does not fit it.
	const CbrtBlockF *bf;
	const unsigned int *res32;
};

VH_CBRT_FN unsigned int cbrt_bits(float f)
{
	unsigned int u;
	memcpy(&u, &f, 4);
	return u;
}
VH_CBRT_FN float cbrt_float(unsigned int u)
{
	float f;
	memcpy(&f, &u, 4);
	return f;
}

// the polynomial part: cbrt(x0 (1 + u)) / cbrt(x0)
VH_CBRT_FN double cbrt_poly(double u)
{
	const double p = __builtin_fma(__builtin_fma(__builtin_fma(-10.0 / 243.0, u, 5.0 / 81.0), u, -1.0 / 9.0), u, 1.0 / 3.0);
	return __builtin_fma(u, p, 1.0);
}

// the table's linear arm, as table_init() computes it
VH_CBRT_FN float cbrt_linear(int i)
{
	const double q0 = (double) i * (1.0 / 100000.0);
	const double e = __builtin_fma(-100000.0, q0, (double) i);
	const float Y = (float) __builtin_fma(e, 1.0 / 100000.0, q0); // (double) i / 100000, correctly rounded
	return 7.787F * Y + (16.0F / 116.0F);
}

// entry i >= 886 before its residual: the float the block polynomial rounds to, as bits
VH_CBRT_FN unsigned int cbrt_predict(const CbrtBlockD &bd, const CbrtBlockI &bi, int i)
{
	const double u = (double) (i - bi.i0) * bd.inv;
	return cbrt_bits((float) (bd.c0 * cbrt_poly(u)));
}

// entry i >= 886 before its residual in the single-precision form: j = i - i0 of its block
VH_CBRT_FN unsigned int cbrt_predict32(const CbrtBlockF &bf, int j)
{
	const float u = (float) j * bf.inv;
	const float p = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf((float) (-10.0 / 243.0), u, (float) (5.0 / 81.0)), u, (float) (-1.0 / 9.0)), u,
		(float) (1.0 / 3.0));
	const float s = u * p;
	return cbrt_bits(__builtin_fmaf(bf.c0, s, bf.c0));
}

// cbrt_pair() through the single-precision form (t.bf, t.res32)
VH_CBRT_FN void cbrt_pair32(const CbrtExact &t, int i, float *t0, float *dt)
{
	const float fi = (float) i;
	int k = (int) (cbrt_bits(fi) >> 18) - CBRT_KEY0;
	k = k < 0 ? 0 : k;
	const CbrtBlockF bf = t.bf[k];
	const CbrtBlockI bi = t.bi[k];
	const float next_c0 = t.bf[k + 1].c0; // (one block past the last is stored; its entry 0 is c0 itself)
	const int j = i - bi.i0;
	const unsigned int p0 = cbrt_predict32(bf, j);
	const unsigned int p1s = cbrt_predict32(bf, j + 1);
	const unsigned int p1 = j + 1 < bi.count ? p1s : cbrt_bits(next_c0);
	const unsigned int w = (unsigned int) i >> 4;
	const unsigned long long both = ((unsigned long long) t.res32[w + 1] << 32) | t.res32[w];
	const unsigned int rr = (unsigned int) (both >> (2 * (i & 15)));
	float v0 = cbrt_float(p0 + (rr & 3u) - 1u), v1 = cbrt_float(p1 + ((rr >> 2) & 3u) - 1u);
	if (i < CBRT_LINEAR) {
		v0 = t.lin ? t.lin[i] : cbrt_linear(i);
		if (i + 1 < CBRT_LINEAR)
			v1 = t.lin ? t.lin[i + 1] : cbrt_linear(i + 1);
	}
	*t0 = v0;
	*dt = v1 - v0;
}

// table[i] and table[i + 1] - table[i] for 0 <= i <= 99998
VH_CBRT_FN void cbrt_pair(const CbrtExact &t, int i, float *t0, float *dt)
{
	const float fi = (float) i;
	int k = (int) (cbrt_bits(fi) >> 18) - CBRT_KEY0;
	k = k < 0 ? 0 : k;
	const CbrtBlockD bd = t.bd[k];
	const CbrtBlockI bi = t.bi[k];
	const double next_c0 = t.bd[k + 1].c0; // (one block past the last is stored)
	const int j = i - bi.i0;
	const double u = (double) j * bd.inv;
	const double c = bd.c0 * cbrt_poly(u);
	const double c1s = bd.c0 * cbrt_poly((double) (j + 1) * bd.inv); // (as entry i + 1 evaluates itself)
	const double c1 = j + 1 < bi.count ? c1s : next_c0;
	// the two residuals: 2 bits each, biased by one, entries i and i + 1 side by side
	const unsigned int w = (unsigned int) i >> 4;
	const unsigned long long both = ((unsigned long long) t.res[w + 1] << 32) | t.res[w];
	const unsigned int rr = (unsigned int) (both >> (2 * (i & 15)));
	unsigned int b0 = cbrt_bits((float) c) + (rr & 3u) - 1u;
	unsigned int b1 = cbrt_bits((float) c1) + ((rr >> 2) & 3u) - 1u;
	float v0 = cbrt_float(b0), v1 = cbrt_float(b1);
	if (i < CBRT_LINEAR) {
		v0 = t.lin ? t.lin[i] : cbrt_linear(i);
		if (i + 1 < CBRT_LINEAR)
			v1 = t.lin ? t.lin[i + 1] : cbrt_linear(i + 1);
	}
	*t0 = v0;
	*dt = v1 - v0;
}

} // namespace vh
