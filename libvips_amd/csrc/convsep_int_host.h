// Host side of the integer horizontal pass (convsep_int_body.h): whether a mask qualifies, its
// shifted coefficient dwords, and the check of the single-precision division.  Included by
// convsep_stream.hip and by the CPU test (tests/emul/convsep_int_check.cpp).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>

namespace vh {

struct HintTables {
	unsigned int coefi[32]; // set s (byte shift), dword j, byte b: c[4 j + b - s]
	float scale, rscale;
};

// q = fma(fma(-scale, S r, S), r, S r) against (float) ((double) S / scale + 0.0) for every S in
// 0 .. smax (what the reference stores, convi.c:735-741 with offset 0); the host's fmaf is the
// IEEE operation the device's v_fma_f32 is.  Once per (scale, smax) and process.
static inline bool hint_div_check(int scale_i, unsigned int smax)
{
	static std::mutex lock;
	static std::map<std::pair<int, unsigned int>, bool> seen;
	std::lock_guard<std::mutex> guard(lock);
	const auto key = std::make_pair(scale_i, smax);
	const auto it = seen.find(key);
	if (it != seen.end())
		return it->second;
	bool ok = true;
	const float scale = (float) scale_i;
	const float r = 1.0f / scale;
	const double scale_d = (double) scale_i;
	for (unsigned int S = 0; S <= smax && ok; S++) {
		const float s = (float) S;
		const float q0 = s * r;
		const float e = fmaf(-scale, q0, s);
		const float q = fmaf(e, r, q0);
		const float want = (float) ((double) S / scale_d + 0.0);
		ok = memcmp(&q, &want, 4) == 0;
	}
	seen[key] = ok;
	return ok;
}

// n integer taps, the first pass's scale and offset.  false = the mask does not qualify.
static inline bool hint_prepare(const int *coef, int n, int scale_i, int offset_i, HintTables *t)
{
	if (n < 1 || n > 29 || scale_i < 1 || scale_i >= (1 << 24) || offset_i != 0)
		return false;
	unsigned long long sum = 0;
	for (int k = 0; k < n; k++) {
		if (coef[k] < 0 || coef[k] > 255)
			return false;
		sum += (unsigned long long) coef[k];
	}
	// every sum an exact float, and a v_dot4_u32_u8 chain that cannot wrap
	if (sum * 255ULL >= (1ULL << 24))
		return false;
	if (!hint_div_check(scale_i, (unsigned int) (sum * 255ULL)))
		return false;
	memset(t, 0, sizeof(*t));
	for (int s = 0; s < 4; s++) {
		for (int j = 0; j < 8; j++) {
			unsigned int d = 0;
			for (int b = 0; b < 4; b++) {
				const int tap = 4 * j + b - s;
				if (tap >= 0 && tap < n)
					d |= (unsigned int) coef[tap] << (8 * b);
			}
			t->coefi[8 * s + j] = d;
		}
	}
	t->scale = (float) scale_i;
	t->rscale = 1.0f / (float) scale_i;
	return true;
}

} // namespace vh
