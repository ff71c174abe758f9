// The x87 extended format (64 bits of mantissa) in integer arithmetic: what the reference's long
// double sums over double images are made of (reduceh.cpp:196-213; templates.h:533-560).  Used by
// the device (resample.hip reduce_notab_f64) and, compiled as plain C++, by the CPU test that
// checks it against the host's own long double (tests/test_x80.py).
#pragma once

#include <cstring>

#if defined(__HIPCC__)
#define X80_FN static __device__ __forceinline__
#else
#define X80_FN static inline
#endif

X80_FN unsigned long long x80_mulhi(unsigned long long a, unsigned long long b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul64hi(a, b);
#else
	return (unsigned long long) (((unsigned __int128) a * b) >> 64);
#endif
}

X80_FN int x80_clz(unsigned long long v)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __clzll((long long) v);
#else
	return __builtin_clzll(v);
#endif
}

X80_FN unsigned long long x80_bits(double d)
{
	unsigned long long b;
	memcpy(&b, &d, sizeof(b));
	return b;
}

X80_FN double x80_from_bits(unsigned long long b)
{
	double d;
	memcpy(&d, &b, sizeof(d));
	return d;
}

struct X80 {
	unsigned long long m; // bit 63 set, or 0 for zero (zeros are +0: what RN gives exact cancellation)
	int e;                // value = m * 2^(e - 63)
	int s;
};

X80_FN X80 x80_round(unsigned long long hi, unsigned long long lo, int e, int s)
{
	// hi holds the 64 leading bits (bit 63 set), lo what lies beyond
	if (lo > 0x8000000000000000ULL || (lo == 0x8000000000000000ULL && (hi & 1))) {
		hi += 1;
		if (hi == 0) {
			hi = 0x8000000000000000ULL;
			e += 1;
		}
	}
	X80 r = { hi, e, s };
	return r;
}

// a finite double (the caller has sorted out inf / NaN)
X80_FN X80 x80_from_double(double d)
{
	const unsigned long long bits = x80_bits(d);
	const int biased = (int) ((bits >> 52) & 0x7ff);
	unsigned long long frac = bits & 0x000fffffffffffffULL;
	X80 r;
	r.s = (int) (bits >> 63);
	if (biased == 0) {
		if (frac == 0) {
			r.m = 0;
			r.e = 0;
			r.s = 0;
			return r;
		}
		const int lz = x80_clz(frac); // >= 12
		r.m = frac << lz;
		r.e = -1022 - (lz - 11);
		return r;
	}
	r.m = 0x8000000000000000ULL | (frac << 11);
	r.e = biased - 1023;
	return r;
}

X80_FN X80 x80_mul(X80 a, X80 b)
{
	X80 zero = { 0, 0, 0 };
	if (a.m == 0 || b.m == 0)
		return zero;
	const unsigned long long hi = x80_mulhi(a.m, b.m), lo = a.m * b.m;
	if (hi >> 63)
		return x80_round(hi, lo, a.e + b.e + 1, a.s ^ b.s);
	return x80_round((hi << 1) | (lo >> 63), lo << 1, a.e + b.e, a.s ^ b.s);
}

X80_FN X80 x80_add(X80 a, X80 b)
{
	if (a.m == 0)
		return b;
	if (b.m == 0)
		return a;
	// a = the one of larger magnitude
	if (b.e > a.e || (b.e == a.e && b.m > a.m)) {
		const X80 t = a;
		a = b;
		b = t;
	}
	const int d = a.e - b.e;
	// b in 128 bits, shifted right by d with the lost bits jammed into the last one
	unsigned long long bh, bl;
	if (d == 0) {
		bh = b.m;
		bl = 0;
	}
	else if (d < 64) {
		bh = b.m >> d;
		bl = b.m << (64 - d);
	}
	else if (d == 64) {
		bh = 0;
		bl = b.m;
	}
	else if (d < 128) {
		bh = 0;
		bl = (b.m >> (d - 64)) | ((b.m << (128 - d)) != 0 ? 1ULL : 0ULL);
	}
	else {
		bh = 0;
		bl = 1;
	}
	unsigned long long hi, lo;
	int e = a.e;
	if (a.s == b.s) {
		lo = bl; // a's low half is zero
		hi = a.m + bh;
		if (hi < a.m) { // carry out of 128 bits: one place right, jamming
			lo = (lo >> 1) | (hi << 63) | (lo & 1);
			hi = (hi >> 1) | 0x8000000000000000ULL;
			e += 1;
		}
	}
	else {
		lo = 0ULL - bl;
		hi = a.m - bh - (bl != 0 ? 1ULL : 0ULL);
		if (hi == 0 && lo == 0) {
			X80 zero = { 0, 0, 0 };
			return zero;
		}
		// renormalise
		if (hi == 0) {
			hi = lo;
			lo = 0;
			e -= 64;
		}
		const int lz = x80_clz(hi);
		if (lz) {
			hi = (hi << lz) | (lo >> (64 - lz));
			lo <<= lz;
			e -= lz;
		}
	}
	return x80_round(hi, lo, e, a.s);
}

// (double) of an extended value: one rounding to 53 bits (or fewer, for a denormal result)
X80_FN double x80_to_double(X80 a)
{
	if (a.m == 0)
		return 0.0;
	const unsigned long long sign = (unsigned long long) a.s << 63;
	int biased = a.e + 1023;
	if (biased >= 0x7ff)
		return x80_from_bits(sign | 0x7ff0000000000000ULL);
	int shift = 11; // bits to drop
	if (biased < 1) {
		shift += 1 - biased;
		biased = 0;
		if (shift > 64)
			return x80_from_bits(sign);
	}
	unsigned long long kept, rest_top, rest_more;
	if (shift == 64) {
		kept = 0;
		rest_top = a.m >> 63;
		rest_more = a.m << 1;
	}
	else {
		kept = a.m >> shift;
		rest_top = (a.m >> (shift - 1)) & 1;
		rest_more = a.m << (65 - shift);
	}
	if (rest_top && (rest_more != 0 || (kept & 1)))
		kept += 1;
	// normal: kept has its leading one at bit 52 (a carry to bit 53 bumps the exponent, and a
	// denormal that rounds up to 2^52 becomes the smallest normal: both fall out of the addition)
	const unsigned long long bits = biased ? ((unsigned long long) (biased - 1) << 52) + kept : kept;
	return x80_from_bits(sign | bits);
}

