// Names for the gfx950 instructions, address spaces and block-level primitives the streaming
// kernel bodies (*_body.h) are written in.  A body only uses what is declared here, so that the
// same source also runs, thread by thread on host fibers, under tests/emul/gcn.h -- the CPU suite
// checks a body's indexing and arithmetic against the oracle without a GPU (test infrastructure
// only: the product is built with THIS header).
#ifndef VH_GCN_H
#define VH_GCN_H

#include <hip/hip_runtime.h>

#define VH_DEV static __device__ __forceinline__
#define VH_CBRT_FN static __host__ __device__ __forceinline__ // (cbrt_exact.h: the host checks what the device runs)

namespace vh {

// global memory: a uniform base (SGPR pair) + a 32-bit lane offset is the saddr form of
// global_load / global_store (one VGPR per address instead of two and a 64-bit add)
typedef const unsigned char __attribute__((address_space(1))) *gptr_in;
typedef unsigned char __attribute__((address_space(1))) *gptr_out;

VH_DEV gptr_in gptr_in_of(unsigned long long v) { return (gptr_in) v; }
VH_DEV gptr_out gptr_out_of(unsigned long long v) { return (gptr_out) v; }
VH_DEV unsigned int gptr_low(gptr_out p) { return (unsigned int) (unsigned long long) p; }
VH_DEV unsigned int gload32(gptr_in base, unsigned int off)
{
	return *(const unsigned int __attribute__((address_space(1))) *) (base + off);
}
// 8 bytes at a 4-byte aligned offset: global_load_dwordx2
VH_DEV void gload64(gptr_in base, unsigned int off, unsigned int (&w)[2])
{
	typedef unsigned int gcn_uint2 __attribute__((ext_vector_type(2)));
	const gcn_uint2 v = *(const gcn_uint2 __attribute__((address_space(1), aligned(4))) *) (base + off);
	w[0] = v.x;
	w[1] = v.y;
}
// 16 bytes at a 4-byte aligned offset: global_load_dwordx4
VH_DEV void gload128(gptr_in base, unsigned int off, unsigned int (&w)[4])
{
	typedef unsigned int gcn_uint4 __attribute__((ext_vector_type(4)));
	const gcn_uint4 v = *(const gcn_uint4 __attribute__((address_space(1), aligned(4))) *) (base + off);
	w[0] = v.x;
	w[1] = v.y;
	w[2] = v.z;
	w[3] = v.w;
}
VH_DEV void gstore128(gptr_out p, const unsigned int (&w)[4])
{
	typedef unsigned int gcn_uint4 __attribute__((ext_vector_type(4)));
	const gcn_uint4 v = { w[0], w[1], w[2], w[3] };
	*(gcn_uint4 __attribute__((address_space(1), aligned(4))) *) p = v;
}
// N dwords (1..4) at a 4-byte aligned offset / address: one global_load / global_store _dword[xN]
template <int N>
VH_DEV void gload_dwords(gptr_in base, unsigned int off, unsigned int (&w)[N])
{
	if constexpr (N == 1)
		w[0] = gload32(base, off);
	else {
		typedef unsigned int gcn_uintn __attribute__((ext_vector_type(N)));
		const gcn_uintn v = *(const gcn_uintn __attribute__((address_space(1), aligned(4))) *) (base + off);
#pragma unroll
		for (int i = 0; i < N; i++)
			w[i] = v[i];
	}
}
template <int N>
VH_DEV void gstore_dwords(gptr_out p, const unsigned int (&w)[N])
{
	if constexpr (N == 1)
		*(unsigned int __attribute__((address_space(1))) *) p = w[0];
	else {
		typedef unsigned int gcn_uintn __attribute__((ext_vector_type(N)));
		gcn_uintn v;
#pragma unroll
		for (int i = 0; i < N; i++)
			v[i] = w[i];
		*(gcn_uintn __attribute__((address_space(1), aligned(4))) *) p = v;
	}
}
VH_DEV unsigned char gload8(gptr_in base, unsigned int off) { return base[off]; }
VH_DEV unsigned int gload16(gptr_in base, unsigned int off)
{
	return *(const unsigned short __attribute__((address_space(1))) *) (base + off);
}
VH_DEV void gstore16(gptr_out p, unsigned short v) { *(unsigned short __attribute__((address_space(1))) *) p = v; }
VH_DEV void gstore32(gptr_out p, unsigned int v) { *(unsigned int __attribute__((address_space(1))) *) p = v; }
VH_DEV void gstore8(gptr_out p, unsigned char v) { *p = v; }

// the image pointers of a batch where they lie in the kernarg segment (a by-value array indexed
// dynamically would be copied to scratch): 64-bit words at `offset` bytes into the segment
struct KernargWords {
	int offset;
	__device__ __forceinline__ unsigned long long operator[](int i) const
	{
		typedef const unsigned long long __attribute__((address_space(4))) *Words;
		return ((Words) ((const char __attribute__((address_space(4))) *) __builtin_amdgcn_kernarg_segment_ptr() + offset))[i];
	}
};

// the value lane - 1 / lane + 1 / lane + delta of the wave holds (lane 0 / 63: unspecified)
VH_DEV unsigned int lane_prev(unsigned int v) { return (unsigned int) __builtin_amdgcn_update_dpp(0, (int) v, 0x138, 0xf, 0xf, false); }
VH_DEV unsigned int lane_next(unsigned int v) { return (unsigned int) __builtin_amdgcn_update_dpp(0, (int) v, 0x130, 0xf, 0xf, false); }
VH_DEV unsigned int lane_from(unsigned int v, int delta)
{
	return (unsigned int) __builtin_amdgcn_ds_bpermute((int) ((threadIdx.x + delta) & 63) << 2, (int) v);
}
VH_DEV int tid() { return (int) threadIdx.x; }
// *p for a wave-uniform p into memory nobody writes while the kernel runs: scalar loads (through a
// plain pointer the compiler loads per lane and reads the first lane back)
template <typename T>
VH_DEV T uniform_load(const T *p)
{
	static_assert(sizeof(T) % 4 == 0, "whole dwords");
	typedef const unsigned int __attribute__((address_space(4))) *Words;
	const Words w = (Words) (unsigned long long) p;
	unsigned int buf[sizeof(T) / 4];
#pragma unroll
	for (unsigned int i = 0; i < sizeof(T) / 4; i++)
		buf[i] = w[i];
	T v;
	__builtin_memcpy(&v, buf, sizeof(T));
	return v;
}
// the next work item of a persistent block: one atomic per block, handed to every thread through `slot` (LDS)
VH_DEV int next_item(int *counter, int *slot)
{
	__syncthreads();
	if (threadIdx.x == 0)
		*slot = atomicAdd(counter, 1);
	__syncthreads();
	return __builtin_amdgcn_readfirstlane(*slot);
}
VH_DEV void barrier() { __syncthreads(); }
// the chip-wide 100 MHz clock
VH_DEV unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }
VH_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
VH_DEV void opaque(int &v) { asm volatile("" : "+v"(v)); }
VH_DEV void opaque(unsigned int &v) { asm volatile("" : "+v"(v)); }
VH_DEV void opaque_uniform(int &v) { asm volatile("" : "+s"(v)); } // (a wave-uniform value: stays in a scalar register)

// v_perm_b32: byte k of the result is byte sel[k] of {hi, lo} (0-3 lo, 4-7 hi, 0x0c zero)
VH_DEV unsigned int perm(unsigned int hi, unsigned int lo, unsigned int sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

typedef short gcn_short2 __attribute__((ext_vector_type(2)));
// v_dot2_i32_i16: acc + lo16(a) * lo16(b) + hi16(a) * hi16(b), signed
VH_DEV int dot2(unsigned int a, unsigned int b, int acc)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(gcn_short2, a), __builtin_bit_cast(gcn_short2, b), acc, false);
}
// the same with the coefficient pair in an SGPR and a fresh accumulator register
VH_DEV int dot2_s(unsigned int a, unsigned int b_uniform, int acc)
{
	int r;
	asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(acc));
	return r;
}
// v_dot4_u32_u8
VH_DEV unsigned int udot4(unsigned int a, unsigned int b, unsigned int acc) { return __builtin_amdgcn_udot4(a, b, acc, false); }
// v_dot4_i32_i8: acc + the four products of signed bytes
VH_DEV int dot4(unsigned int a, unsigned int b, int acc) { return __builtin_amdgcn_sdot4((int) a, (int) b, acc, false); }
// v_sat_pk_u8_i16: {0, 0, sat_u8(hi16), sat_u8(lo16)}
VH_DEV unsigned int sat_pk_u8_i16(unsigned int both)
{
	unsigned int r;
	asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(both));
	return r;
}
VH_DEV unsigned int umulhi(unsigned int a, unsigned int b) { return __umulhi(a, b); }
// x - floor(x)
VH_DEV float fract(float x) { return __builtin_amdgcn_fractf(x); }
// round to nearest even, as rintf
VH_DEV float rne(float x) { return __builtin_rintf(x); }
// v_cvt_pk_u8_f32 of an integer-valued float in 0..255 into byte `byte` of `old`
VH_DEV unsigned int cvt_pk_u8(float v, unsigned int byte, unsigned int old) { return __builtin_amdgcn_cvt_pk_u8_f32(v, byte, old); }

} // namespace vh

#include "kernel_stmt.h"

namespace vh {

// the next work item of a persistent WAVE: one atomic per wave, no barrier
VH_DEV int wave_next_item(int *counter)
{
	int v = 0;
	if ((threadIdx.x & 63) == 0)
		v = atomicAdd(counter, 1);
	return __builtin_amdgcn_readfirstlane(v);
}
// the wave's index in its block, as a scalar
VH_DEV int wave_index() { return __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)); }
// every vector memory operation of this wave has completed
VH_DEV void wait_vmem0() { VH_WAIT_VMCNT(0); }
// One dword per lane from global memory straight into LDS (global_load_lds_dword): lane i's dword, read at
// base + voff (base wave-uniform), lands at lds_dst + i (lds_dst wave-uniform, a pointer into the block's LDS).
// No register, no wait: wait_vmem0() + barrier() before anybody reads it.
VH_DEV void lds_dma_dword(gptr_in base, unsigned int voff, unsigned int *lds_dst)
{
	const unsigned int where = __builtin_amdgcn_readfirstlane((int) VH_LDS_ADDR(lds_dst));
	VH_LDS_DMA_DWORD(base, voff, where);
}
// ... 16 bytes per lane (global_load_lds_dwordx4): lane i's 16 bytes land at lds_dst + 4 i dwords; base + voff and
// lds_dst multiples of 16
VH_DEV void lds_dma_x4(gptr_in base, unsigned int voff, unsigned int *lds_dst)
{
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (base + voff),
		(__attribute__((address_space(3))) void *) VH_LDS_ADDR(lds_dst), 16, 0, 0);
}
// v_mfma_f32_32x32x16_f16: D[i][j] = C[i][j] + sum_k A[i][k] B[k][j], 32 x 32 x 16, one wave.
//   a: lane l holds A[l & 31][k] for the 8 k-slots (l >> 5, 0..7) as 8 halves (dword q = slots 2 q, 2 q + 1);
//   b: lane l holds B[k][l & 31] for the SAME 8 k-slots of its half;
//   acc: lane l holds D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31] in register r.
// (which k a slot is does not matter to a sum as long as both operands agree.)  f16 denormals are not flushed.
VH_DEV void mfma_32x32x16_f16(const unsigned int (&a)[4], const unsigned int (&b)[4], float (&acc)[16])
{
	typedef _Float16 gcn_half8 __attribute__((ext_vector_type(8)));
	typedef unsigned int gcn_uint4v __attribute__((ext_vector_type(4)));
	typedef float gcn_float16 __attribute__((ext_vector_type(16)));
	const gcn_uint4v ua = { a[0], a[1], a[2], a[3] }, ub = { b[0], b[1], b[2], b[3] };
	gcn_float16 c;
#pragma unroll
	for (int r = 0; r < 16; r++)
		c[r] = acc[r];
	c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gcn_half8, ua), __builtin_bit_cast(gcn_half8, ub), c, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 16; r++)
		acc[r] = c[r];
}

// ... with C = 0 (an inline constant: no register is cleared)
VH_DEV void mfma_32x32x16_f16_first(const unsigned int (&a)[4], const unsigned int (&b)[4], float (&acc)[16])
{
	typedef _Float16 gcn_half8 __attribute__((ext_vector_type(8)));
	typedef unsigned int gcn_uint4v __attribute__((ext_vector_type(4)));
	typedef float gcn_float16 __attribute__((ext_vector_type(16)));
	const gcn_uint4v ua = { a[0], a[1], a[2], a[3] }, ub = { b[0], b[1], b[2], b[3] };
	gcn_float16 c = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gcn_half8, ua), __builtin_bit_cast(gcn_half8, ub), c, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 16; r++)
		acc[r] = c[r];
}
// v_mfma_f32_16x16x32_f16: D[i][j] = C[i][j] + sum_k A[i][k] B[k][j], 16 x 16 x 32, one wave.
//   a: lane l holds A[l & 15][k] for the 8 k-slots (l >> 4, 0..7) as 8 halves; b: B[k][l & 15] for the SAME slots;
//   acc: lane l holds D[4 (l >> 4) + r][l & 15] in register r.  (first: C = 0)
VH_DEV void mfma_16x16x32_f16(const unsigned int (&a)[4], const unsigned int (&b)[4], float (&acc)[4])
{
	typedef _Float16 gcn_half8 __attribute__((ext_vector_type(8)));
	typedef unsigned int gcn_uint4v __attribute__((ext_vector_type(4)));
	typedef float gcn_float4 __attribute__((ext_vector_type(4)));
	const gcn_uint4v ua = { a[0], a[1], a[2], a[3] }, ub = { b[0], b[1], b[2], b[3] };
	gcn_float4 c = { acc[0], acc[1], acc[2], acc[3] };
	c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gcn_half8, ua), __builtin_bit_cast(gcn_half8, ub), c, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 4; r++)
		acc[r] = c[r];
}
VH_DEV void mfma_16x16x32_f16_first(const unsigned int (&a)[4], const unsigned int (&b)[4], float (&acc)[4])
{
	typedef _Float16 gcn_half8 __attribute__((ext_vector_type(8)));
	typedef unsigned int gcn_uint4v __attribute__((ext_vector_type(4)));
	typedef float gcn_float4 __attribute__((ext_vector_type(4)));
	const gcn_uint4v ua = { a[0], a[1], a[2], a[3] }, ub = { b[0], b[1], b[2], b[3] };
	gcn_float4 c = { 0, 0, 0, 0 };
	c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gcn_half8, ua), __builtin_bit_cast(gcn_half8, ub), c, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 4; r++)
		acc[r] = c[r];
}
// LDS written by this wave is read back by this wave only: its LDS operations complete in order, the
// compiler must not move them across this point
VH_DEV void wave_lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

} // namespace vh

#endif // VH_GCN_H
