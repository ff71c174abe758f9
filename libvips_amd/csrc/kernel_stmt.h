// The few statements with no C++ spelling that whole kernel FILES (convsep_stream.hip, convsep_f32.hip,
// reduce_u8.hip ...) are written with -- register-class hints, s_waitcnt, the LDS-DMA, dynamic LDS -- under
// names, so that the same files also compile for host fibers: tests/emul/kernel_prelude.h gives the names host
// meanings (and defines this header's guard), the product gets the instructions.
#ifndef VH_KERNEL_STMT_H
#define VH_KERNEL_STMT_H
#include <type_traits>

// register-class hints: "hold this value in a scalar / vector register here" (stops a hoist or a merge)
#define VH_SCALAR(x) asm volatile("" : "+s"(x))
#define VH_SCALAR2(x, y) asm volatile("" : "+s"(x), "+s"(y))
#define VH_VECTOR1(a) asm volatile("" : "+v"(a))
#define VH_VECTOR2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define VH_USE2(a, b) asm volatile("" ::"v"(a), "v"(b)) // (a use the compiler cannot drop)
#define VH_VECTOR5(a, b, c, d, e) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e))
// at most n vector memory operations of this wave still in flight
#define VH_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// One dword per lane from global memory straight into LDS: lane i's dword lands at LDS byte address
// lds_dst + 4 i (lds_dst wave-uniform), read from src + voff (src wave-uniform, voff per lane).  M0 carries
// the LDS address and belongs to the compiler: saved, written and restored inside the one statement.
#define VH_LDS_DMA_DWORD(src, voff, lds_dst) \
	do { \
		unsigned int vh_keep_m0; \
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0" \
					 : "=&s"(vh_keep_m0) \
					 : "v"(voff), "s"(src), "s"(lds_dst) \
					 : "memory"); \
	} while (0)
// instructions the compiler does not select by itself
#define VH_SAT_PK_U8_I16(r, both) asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(both)) // {0, 0, sat_u8(hi16), sat_u8(lo16)}
#define VH_DOT2_SCALAR_COEF(dst, pk, coef, acc) asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(dst) : "v"(pk), "s"(coef), "v"(acc))
// 24-bit multiplies (full rate; v_mul_lo_u32 / v_mul_hi_u32 issue at a quarter of it): both factors below 2^24
#define VH_MAD_U24(dst, a, b, c) asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(dst) : "v"(a), "s"(b), "v"(c)) // a * b + c, b scalar
#define VH_MUL_HI_U24(dst, a, b) asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(dst) : "v"(a), "v"(b))          // (a * b) >> 32
// hand-off between blocks of one launch without fences (MI355X_MICROARCH.md "valid forms": system-scope stores and
// loads on both sides, an atomic for the flag): a 16-byte write-through store, a dword load that bypasses the caches
#define VH_STORE4_SYS(p, v) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory")
#define VH_LOAD_SYS(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define VH_STORE_SYS(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
// which of the part's 8 XCDs (each with an L2 of its own) this wave runs on: HW_REG_XCC_ID, bits 3:0
#define VH_XCC_ID() ((int) (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u))
#define VH_STORE_BYTE(p, v) asm volatile("global_store_byte %0, %1, off" : : "v"(p), "v"(v) : "memory")
// a marker that keeps two otherwise identical arms of a switch apart (merged, their register index is dynamic)
// LDS written by this wave is read back by this wave only: its LDS operations complete in order, the compiler must
// not move them across this point
#define VH_WAVE_LDS_FENCE() \
	do { \
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
		__builtin_amdgcn_wave_barrier(); \
	} while (0)
#define VH_ASM_MARK(text) asm volatile("; " text)
// the block's dynamic LDS, and the LDS byte address of a pointer into LDS
#define VH_DYNAMIC_LDS(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#define VH_LDS_ADDR(p) ((unsigned int) (size_t) (p))

// ---- float -> integer conversions, spelled as the instruction that does them.
// A C cast of a NaN or of an out-of-range value is undefined in the language; what these kernels need is what
// the part's converters do, so every float -> int conversion of the device code is one of these (no C cast of
// a float to an integer type is left in device code: `hipcc -S` with and without
// -fno-strict-float-cast-overflow is the same text, tests/test_isa_guard.py).
//   v_cvt_i32_f32 / v_cvt_i32_f64: round toward zero, saturate at INT_MIN / INT_MAX, NaN -> 0
//   v_cvt_u32_f32 / v_cvt_u32_f64: round toward zero, saturate at 0 / UINT_MAX,      NaN -> 0
// (the x86 reference gives INT_MIN -- "integer indefinite" -- for NaN and for anything out of range; where a
// caller can see the difference it says what it does about it)
namespace vh {
static __device__ __forceinline__ int cvt_i32(float v)
{
	int r;
	asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
	return r;
}
static __device__ __forceinline__ int cvt_i32(double v)
{
	int r;
	asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
	return r;
}
static __device__ __forceinline__ unsigned int cvt_u32(float v)
{
	unsigned int r;
	asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
	return r;
}
static __device__ __forceinline__ unsigned int cvt_u32(double v)
{
	unsigned int r;
	asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
	return r;
}
// a real value to a pel format: integer formats through the converters above (narrowing an int to 8 / 16
// bits is defined: modulo 2^n), float formats by the ordinary conversion
template <typename TOUT, typename TIN>
static __device__ __forceinline__ TOUT cvt_to(TIN v)
{
	if constexpr (std::is_floating_point<TOUT>::value)
		return (TOUT) v;
	else if constexpr (std::is_same<TOUT, unsigned int>::value)
		return cvt_u32(v);
	else
		return (TOUT) cvt_i32(v);
}
} // namespace vh

#endif // VH_KERNEL_STMT_H
