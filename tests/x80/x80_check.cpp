// TEST INFRASTRUCTURE: libvips_amd/csrc/x80.h (the x87 extended format in integer arithmetic, what
// the device uses for double images) against the host's own long double, on random and on
// adversarial operands.  Prints the number of mismatches per operation; exit status 0 when none.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "x80.h"

static X80 from_ld(long double v)
{
	unsigned char raw[16];
	memcpy(raw, &v, sizeof(long double));
	X80 r;
	unsigned short se;
	memcpy(&r.m, raw, 8);
	memcpy(&se, raw + 8, 2);
	r.s = se >> 15;
	r.e = (int) (se & 0x7fff) - 16383;
	if (r.m == 0) {
		r.e = 0;
		r.s = 0;
	}
	return r;
}

static bool same(X80 a, long double v)
{
	const X80 b = from_ld(v);
	if (a.m == 0 && b.m == 0)
		return true;
	return a.m == b.m && a.e == b.e && a.s == b.s;
}

int main()
{
	std::mt19937_64 rng(12345);
	auto rnd_double = [&](int spread) {
		unsigned long long bits = rng();
		const int e = 1023 + (int) (rng() % (2 * spread + 1)) - spread;
		bits = (bits & 0x800fffffffffffffULL) | ((unsigned long long) e << 52);
		if (rng() % 16 == 0)
			bits &= ~((1ULL << (rng() % 53)) - 1); // trailing zeros: exact sums, ties
		double d;
		memcpy(&d, &bits, 8);
		return d;
	};
	auto rnd_ld = [&](int spread) {
		long double v = (long double) rnd_double(spread);
		v += (long double) rnd_double(spread) * 0x1p-53L; // fill the low mantissa bits
		if (rng() % 8 == 0)
			v = (long double) rnd_double(spread);
		return v;
	};
	long bad_mul = 0, bad_add = 0, bad_cvt = 0, bad_from = 0, bad_chain = 0;
	const int N = 4000000;
	for (int i = 0; i < N; i++) {
		const int spread = i % 3 == 0 ? 2 : (i % 3 == 1 ? 40 : 300);
		const long double a = rnd_ld(spread), b = rnd_ld(spread);
		const double d = rnd_double(spread);
		// double -> extended
		if (!same(x80_from_double(d), (long double) d))
			bad_from++;
		// products of an extended coefficient and a double pixel
		volatile long double p = a * (long double) d;
		if (!same(x80_mul(from_ld(a), x80_from_double(d)), p))
			bad_mul++;
		// sums, including near-cancellation
		volatile long double s1 = a + b;
		if (!same(x80_add(from_ld(a), from_ld(b)), s1))
			bad_add++;
		const long double nb = -a * (1.0L + (long double) (int) (rng() % 7 - 3) * 0x1p-60L);
		volatile long double s2 = a + nb;
		if (!same(x80_add(from_ld(a), from_ld(nb)), s2))
			bad_add++;
		// extended -> double, normal and denormal results
		volatile double c1 = (double) a;
		if (x80_bits(x80_to_double(from_ld(a))) != x80_bits(c1))
			bad_cvt++;
		const long double tiny = a * 0x1p-1040L;
		volatile double c2 = (double) tiny;
		if (x80_bits(x80_to_double(from_ld(tiny))) != x80_bits(c2))
			bad_cvt++;
	}
	// whole sums the way reduce_sum runs them
	for (int i = 0; i < 200000; i++) {
		const int n = 5 + (int) (rng() % 45);
		volatile long double sum = 0;
		X80 acc = { 0, 0, 0 };
		for (int k = 0; k < n; k++) {
			const long double c = rnd_ld(1) * 0.1L;
			const double v = rnd_double(i % 2 ? 3 : 30);
			sum = sum + c * (long double) v;
			acc = x80_add(acc, x80_mul(from_ld(c), x80_from_double(v)));
		}
		volatile double want = (double) sum;
		if (x80_bits(x80_to_double(acc)) != x80_bits(want))
			bad_chain++;
	}
	// denormal doubles in
	for (int i = 0; i < 100000; i++) {
		unsigned long long bits = rng() & 0x800fffffffffffffULL;
		double d;
		memcpy(&d, &bits, 8);
		if (!same(x80_from_double(d), (long double) d))
			bad_from++;
	}
	printf("from_double %ld  mul %ld  add %ld  to_double %ld  chains %ld\n", bad_from, bad_mul, bad_add, bad_cvt, bad_chain);
	return bad_from || bad_mul || bad_add || bad_cvt || bad_chain;
}
