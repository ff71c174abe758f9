// TEST INFRASTRUCTURE (CPU suite only; never part of the product): the names of
// libvips_amd/csrc/gcn.h for the host, so that a kernel body (*_body.h) runs thread by thread on
// fibers (emul.h) and its indexing and arithmetic can be compared with the oracle without a GPU.
// Each function restates the instruction's documented semantics.
#ifndef VH_GCN_H
#define VH_GCN_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "emul.h"

#define VH_DEV static inline
#ifndef __device__
#define __device__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
#endif

namespace vh {

using std::max;
using std::min;

typedef const unsigned char *gptr_in;
typedef unsigned char *gptr_out;

VH_DEV gptr_in gptr_in_of(unsigned long long v) { return (gptr_in) (uintptr_t) v; }
VH_DEV gptr_out gptr_out_of(unsigned long long v) { return (gptr_out) (uintptr_t) v; }
VH_DEV unsigned int gptr_low(gptr_out p) { return (unsigned int) (uintptr_t) p; }
VH_DEV unsigned int gload32(gptr_in base, unsigned int off)
{
	unsigned int v;
	memcpy(&v, base + off, 4);
	return v;
}
VH_DEV void gload64(gptr_in base, unsigned int off, unsigned int (&w)[2]) { memcpy(w, base + off, 8); }
VH_DEV void gload128(gptr_in base, unsigned int off, unsigned int (&w)[4]) { memcpy(w, base + off, 16); }
VH_DEV void gstore128(gptr_out p, const unsigned int (&w)[4]) { memcpy(p, w, 16); }
template <int N>
VH_DEV void gload_dwords(gptr_in base, unsigned int off, unsigned int (&w)[N]) { memcpy(w, base + off, 4 * N); }
template <int N>
VH_DEV void gstore_dwords(gptr_out p, const unsigned int (&w)[N]) { memcpy(p, w, 4 * N); }
VH_DEV unsigned char gload8(gptr_in base, unsigned int off) { return base[off]; }
VH_DEV unsigned int gload16(gptr_in base, unsigned int off)
{
	unsigned short v;
	memcpy(&v, base + off, 2);
	return v;
}
VH_DEV void gstore16(gptr_out p, unsigned short v) { memcpy(p, &v, 2); }
VH_DEV void gstore32(gptr_out p, unsigned int v) { memcpy(p, &v, 4); }
VH_DEV void gstore8(gptr_out p, unsigned char v) { *p = v; }

struct KernargWords {
	const unsigned long long *words;
	unsigned long long operator[](int i) const { return words[i]; }
};

// cross-lane moves: every thread of the block must make the same calls (the bodies do)
VH_DEV unsigned int lane_from(unsigned int v, int delta)
{
	const int t = emul::current_tid();
	return emul::exchange(v, (t & ~63) + ((t + delta) & 63));
}
VH_DEV unsigned int lane_prev(unsigned int v) { return lane_from(v, -1); }
VH_DEV unsigned int lane_next(unsigned int v) { return lane_from(v, 1); }
VH_DEV int tid() { return emul::current_tid(); }
template <typename T>
VH_DEV T uniform_load(const T *p)
{
	return *p;
}
VH_DEV int next_item(int *counter, int *slot)
{
	emul::barrier();
	if (emul::current_tid() == 0)
		*slot = __sync_fetch_and_add(counter, 1);
	emul::barrier();
	return *slot;
}
VH_DEV void barrier() { emul::barrier(); }
VH_DEV int wave_index() { return emul::current_tid() >> 6; }
VH_DEV int wave_next_item(int *counter)
{
	int v = 0;
	if ((emul::current_tid() & 63) == 0)
		v = __sync_fetch_and_add(counter, 1);
	const emul::WaveData &wd = emul::wave_share(&v, 4, true);
	int r;
	memcpy(&r, wd.data[0], 4);
	return r;
}
VH_DEV void wait_vmem0() {}
// global_load_lds_dword: lane i's dword to lds_dst + i (the copy lands at once)
VH_DEV void lds_dma_dword(gptr_in base, unsigned int voff, unsigned int *lds_dst)
{
	memcpy(lds_dst + (emul::current_tid() & 63), base + voff, 4);
}
VH_DEV void lds_dma_x4(gptr_in base, unsigned int voff, unsigned int *lds_dst)
{
	memcpy(lds_dst + 4 * (emul::current_tid() & 63), base + voff, 16);
}
// an IEEE half (denormals included) as a float
VH_DEV float half_bits_to_float(unsigned int h)
{
	const int sign = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
	float v;
	if (e == 0)
		v = ldexpf((float) m, -24);
	else if (e == 31)
		v = m ? NAN : INFINITY;
	else
		v = ldexpf((float) (m | 1024), e - 25);
	return sign ? -v : v;
}
// v_mfma_f32_32x32x16_f16 (csrc/gcn.h): a meeting of the wave's fibers; the sum in the order of the k-slots
VH_DEV void mfma_32x32x16_f16(const unsigned int (&a)[4], const unsigned int (&b)[4], float (&acc)[16])
{
	unsigned int A[64][4], Bm[64][4];
	{
		const emul::WaveData &wd = emul::wave_share(a, 16, true);
		memcpy(A, wd.data, sizeof(A));
	}
	{
		const emul::WaveData &wd = emul::wave_share(b, 16, true);
		memcpy(Bm, wd.data, sizeof(Bm));
	}
	const int lane = emul::current_tid() & 63, j = lane & 31, hf = lane >> 5;
	for (int r = 0; r < 16; r++) {
		const int i = (r & 3) + 8 * (r >> 2) + 4 * hf;
		float sum = acc[r];
		for (int half_of_wave = 0; half_of_wave < 2; half_of_wave++)
			for (int idx = 0; idx < 8; idx++) {
				const unsigned int av = (A[i + 32 * half_of_wave][idx >> 1] >> (16 * (idx & 1))) & 0xffffu;
				const unsigned int bv = (Bm[j + 32 * half_of_wave][idx >> 1] >> (16 * (idx & 1))) & 0xffffu;
				sum += half_bits_to_float(av) * half_bits_to_float(bv);
			}
		acc[r] = sum;
	}
}
VH_DEV void mfma_32x32x16_f16_first(const unsigned int (&a)[4], const unsigned int (&b)[4], float (&acc)[16])
{
	for (int r = 0; r < 16; r++)
		acc[r] = 0.0f;
	mfma_32x32x16_f16(a, b, acc);
}
// v_mfma_f32_16x16x32_f16 (csrc/gcn.h)
VH_DEV void mfma_16x16x32_f16(const unsigned int (&a)[4], const unsigned int (&b)[4], float (&acc)[4])
{
	unsigned int A[64][4], Bm[64][4];
	{
		const emul::WaveData &wd = emul::wave_share(a, 16, true);
		memcpy(A, wd.data, sizeof(A));
	}
	{
		const emul::WaveData &wd = emul::wave_share(b, 16, true);
		memcpy(Bm, wd.data, sizeof(Bm));
	}
	const int lane = emul::current_tid() & 63, j = lane & 15, g = lane >> 4;
	for (int r = 0; r < 4; r++) {
		const int i = 4 * g + r;
		float sum = acc[r];
		for (int kg = 0; kg < 4; kg++)
			for (int idx = 0; idx < 8; idx++) {
				const unsigned int av = (A[i + 16 * kg][idx >> 1] >> (16 * (idx & 1))) & 0xffffu;
				const unsigned int bv = (Bm[j + 16 * kg][idx >> 1] >> (16 * (idx & 1))) & 0xffffu;
				sum += half_bits_to_float(av) * half_bits_to_float(bv);
			}
		acc[r] = sum;
	}
}
VH_DEV void mfma_16x16x32_f16_first(const unsigned int (&a)[4], const unsigned int (&b)[4], float (&acc)[4])
{
	for (int r = 0; r < 4; r++)
		acc[r] = 0.0f;
	mfma_16x16x32_f16(a, b, acc);
}
// the wave's own LDS traffic: its fibers meet (what one lane wrote, another reads)
VH_DEV void wave_lds_fence() { (void) emul::wave_share(nullptr, 0, true); }
VH_DEV unsigned long long realtime() { return emul::clock_ticks(); }
VH_DEV void sched_fence() {}
VH_DEV void opaque(int &) {}
VH_DEV void opaque(unsigned int &) {}
VH_DEV void opaque_uniform(int &) {}

VH_DEV unsigned int perm(unsigned int hi, unsigned int lo, unsigned int sel)
{
	const unsigned long long both = ((unsigned long long) hi << 32) | lo;
	unsigned int r = 0;
	for (int k = 0; k < 4; k++) {
		const unsigned int s = (sel >> (8 * k)) & 0xffu;
		unsigned int byte;
		if (s <= 7)
			byte = (unsigned int) (both >> (8 * s)) & 0xffu;
		else if (s == 0x0c)
			byte = 0;
		else if (s >= 0x0d)
			byte = 0xff;
		else // 8..11: sign of a 16-bit half, not used by the bodies
			byte = ((both >> (16 * (s - 8) + 15)) & 1) ? 0xff : 0;
		r |= byte << (8 * k);
	}
	return r;
}
VH_DEV int dot2(unsigned int a, unsigned int b, int acc)
{
	return acc + (int) (short) (a & 0xffff) * (int) (short) (b & 0xffff) + (int) (short) (a >> 16) * (int) (short) (b >> 16);
}
VH_DEV int dot2_s(unsigned int a, unsigned int b, int acc) { return dot2(a, b, acc); }
VH_DEV unsigned int udot4(unsigned int a, unsigned int b, unsigned int acc)
{
	for (int k = 0; k < 4; k++)
		acc += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
	return acc;
}
VH_DEV int dot4(unsigned int a, unsigned int b, int acc)
{
	for (int k = 0; k < 4; k++)
		acc += (int) (signed char) (a >> (8 * k)) * (int) (signed char) (b >> (8 * k));
	return acc;
}
VH_DEV unsigned int sat_pk_u8_i16(unsigned int both)
{
	const int lo = (short) (both & 0xffff), hi = (short) (both >> 16);
	return (unsigned int) min(max(lo, 0), 255) | ((unsigned int) min(max(hi, 0), 255) << 8);
}
// csrc/kernel_stmt.h's converters (v_cvt_i32_f32 / _f64, v_cvt_u32_f32 / _f64): toward zero, saturating, NaN -> 0
template <typename F>
VH_DEV int cvt_i32(F v)
{
	if (v != v)
		return 0;
	if (v >= (F) 2147483648.0)
		return 2147483647;
	if (v <= (F) -2147483648.0)
		return -2147483647 - 1;
	return (int) v;
}
template <typename F>
VH_DEV unsigned int cvt_u32(F v)
{
	if (v != v || v <= (F) 0)
		return 0;
	if (v >= (F) 4294967296.0)
		return 0xffffffffu;
	return (unsigned int) v;
}
template <typename TOUT, typename TIN>
VH_DEV TOUT cvt_to(TIN v)
{
	if constexpr (std::is_floating_point<TOUT>::value)
		return (TOUT) v;
	else if constexpr (std::is_same<TOUT, unsigned int>::value)
		return cvt_u32(v);
	else
		return (TOUT) cvt_i32(v);
}
VH_DEV unsigned int umulhi(unsigned int a, unsigned int b) { return (unsigned int) (((unsigned long long) a * b) >> 32); }
VH_DEV float fract(float x) { return x - floorf(x); }
VH_DEV float rne(float x) { return rintf(x); }
VH_DEV unsigned int cvt_pk_u8(float v, unsigned int byte, unsigned int old)
{
	const float r = rintf(v);
	const unsigned int u = r <= 0.0f ? 0u : r >= 255.0f ? 255u : (unsigned int) r;
	return (old & ~(0xffu << (8 * byte))) | (u << (8 * byte));
}

} // namespace vh

#endif // VH_GCN_H
