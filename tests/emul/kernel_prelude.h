// TEST INFRASTRUCTURE (CPU suite only): host meanings for what a whole kernel FILE
// (libvips_amd/csrc/convsep_stream.hip) is written with, so that the file itself -- not a restatement --
// compiles for host fibers: the statements of csrc/gcn.h's last section, the few amdgcn builtins and
// CUDA-style rounding intrinsics it and colour_device.h use, threadIdx / __syncthreads / atomicAdd,
// __shared__ (one copy per block: a block's fibers share one OS thread) and the launch.
// Include after gcn.h (this directory's) and before the kernel file.
#pragma once

#include "gcn.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

// ---- csrc/kernel_stmt.h (its guard is taken here: the product's meanings never reach a host build)
#define VH_KERNEL_STMT_H
#define VH_SAT_PK_U8_I16(r, both) ((r) = vh::sat_pk_u8_i16(both))
#define VH_DOT2_SCALAR_COEF(dst, pk, coef, acc) ((dst) = vh::dot2(pk, coef, acc))
#define VH_MAD_U24(dst, a, b, c) ((dst) = ((unsigned int) (a) & 0xffffffu) * ((unsigned int) (b) & 0xffffffu) + (unsigned int) (c))
#define VH_MUL_HI_U24(dst, a, b) \
	((dst) = (unsigned int) (((unsigned long long) ((unsigned int) (a) & 0xffffffu) * ((unsigned int) (b) & 0xffffffu)) >> 32))
#define VH_XCC_ID() ((int) (blockIdx.x & 7u))
#define VH_STORE4_SYS(p, v) (*(p) = (v))
#define VH_LOAD_SYS(p) (*(p))
#define VH_STORE_SYS(p, v) (*(p) = (v))
#define VH_STORE_BYTE(p, v) (*(unsigned char *) (p) = (unsigned char) (v))
#define VH_ASM_MARK(text) ((void) 0)
#define VH_VECTOR1(a) ((void) (a))
#define VH_USE2(a, b) ((void) (a), (void) (b))
#define VH_SCALAR(x) ((void) (x))
#define VH_SCALAR2(x, y) ((void) (x), (void) (y))
#define VH_VECTOR2(a, b) ((void) (a), (void) (b))
#define VH_VECTOR5(a, b, c, d, e) ((void) (a), (void) (b), (void) (c), (void) (d), (void) (e))
#define VH_WAIT_VMCNT(n) ((void) 0) // (the emulated LDS-DMA lands at once)
// lane i's dword to LDS offset lds_dst + 4 i
static inline void emul_lds_dma_dword(const void *src, unsigned int voff, unsigned int lds_dst)
{
	const size_t at = (size_t) lds_dst + 4u * (unsigned int) (emul::current_tid() & 63);
	if (at + 4 > emul::lds_bytes()) {
		fprintf(stderr, "emul: LDS-DMA to byte %zu of %zu\n", at, emul::lds_bytes());
		abort();
	}
	memcpy(emul::lds_base() + at, (const char *) src + voff, 4);
}
#define VH_LDS_DMA_DWORD(src, voff, lds_dst) emul_lds_dma_dword(src, voff, lds_dst)
#define VH_WAVE_LDS_FENCE() ((void) emul::wave_share(nullptr, 0, true))
#define VH_DYNAMIC_LDS(T, name) T *name = reinterpret_cast<T *>(emul::lds_base())
#define VH_LDS_ADDR(p) ((unsigned int) (reinterpret_cast<unsigned char *>(p) - emul::lds_base()))

using std::isnan;
using std::isinf;
using std::rint;
using std::rintf;

// ---- language
#undef __global__
#define __global__
#undef __shared__
#define __shared__ static thread_local
#undef __launch_bounds__
#define __launch_bounds__(...)
#undef __restrict__
#define __restrict__
#define address_space(n) // (inside __attribute__(( )): every pointer is a host pointer)

namespace emul {
struct Geometry {
	dim3 grid, block, block_idx;
	const void *kernarg;
};
inline Geometry &geometry()
{
	static thread_local Geometry g;
	return g;
}
struct ThreadIdx {
	struct X {
		operator unsigned int() const { return (unsigned int) current_tid() % geometry().block.x; }
	} x;
	struct Y {
		operator unsigned int() const { return (unsigned int) current_tid() / geometry().block.x % geometry().block.y; }
	} y;
	struct Z {
		operator unsigned int() const { return (unsigned int) current_tid() / (geometry().block.x * geometry().block.y); }
	} z;
};
} // namespace emul
#define threadIdx (emul::ThreadIdx())
#define blockIdx (emul::geometry().block_idx)
#define blockDim (emul::geometry().block)
#define gridDim (emul::geometry().grid)
#define __builtin_amdgcn_kernarg_segment_ptr() (emul::geometry().kernarg)
#define __syncthreads() emul::barrier()
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned int atomicAdd(unsigned int *p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicExch(int *p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }

// ---- amdgcn builtins (wave-uniform values are uniform by construction in the kernels emulated here)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void) 0)
#define __builtin_amdgcn_ballot_w64(p) emul::ballot(p)
#define __all(p) (emul::ballot(!(p)) == 0)
#define __any(p) (emul::ballot(p) != 0)
// packed-integer instructions: the meanings of this directory's gcn.h
template <typename V2>
static inline int emul_sdot2(V2 a, V2 b, int acc)
{
	return vh::dot2(__builtin_bit_cast(unsigned int, a), __builtin_bit_cast(unsigned int, b), acc);
}
#define __builtin_amdgcn_sdot2(a, b, acc, clamp) emul_sdot2(a, b, acc)
#define __builtin_amdgcn_udot4(a, b, acc, clamp) vh::udot4(a, b, acc)
#define __builtin_amdgcn_sdot4(a, b, acc, clamp) vh::dot4((unsigned int) (a), (unsigned int) (b), acc)
#define __builtin_amdgcn_perm(hi, lo, sel) vh::perm(hi, lo, sel)
#define __builtin_amdgcn_cvt_pk_u8_f32(v, byte, old) vh::cvt_pk_u8(v, byte, old)
#define __builtin_amdgcn_s_memrealtime() emul::clock_ticks()
// v_mov_b32 with a DPP quad_perm control: lane l of a quad reads lane (ctrl >> 2 (l & 3)) & 3 of it
static inline int emul_mov_dpp(int v, int ctrl)
{
	const emul::WaveData &wd = emul::wave_share(&v, 4, true);
	const int lane = emul::current_tid() & 63;
	if (ctrl >= 0x100) {
		fprintf(stderr, "emul: DPP control %#x is not emulated\n", ctrl);
		abort();
	}
	const int src = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
	int r;
	memcpy(&r, wd.data[src], 4);
	return r;
}
#define __builtin_amdgcn_mov_dpp(v, ctrl, row_mask, bank_mask, bound_ctrl) emul_mov_dpp((int) (v), ctrl)
// v_mfma_f32_4x4x4_16b_f16: 16 blocks of 4 lanes; in a block lane i holds row i of A (4 halves), lane j column j
// of B (4 halves) and gets column j of D = C + A B (4 floats)
template <typename H4, typename F4>
static inline F4 emul_mfma_4x4x4f16(H4 a, H4 b, F4 c)
{
	static_assert(sizeof(H4) == 8 && sizeof(F4) == 16, "operand sizes");
	const emul::WaveData &wd = emul::wave_share(&a, 8, true);
	const int base = (emul::current_tid() & 63) & ~3;
	F4 d = c;
	for (int i = 0; i < 4; i++) {
		H4 ai;
		memcpy(&ai, wd.data[base + i], 8);
		float sum = c[i];
		for (int k = 0; k < 4; k++)
			sum += (float) ai[k] * (float) b[k];
		d[i] = sum;
	}
	return d;
}
#define __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, cbsz, abid, blgp) emul_mfma_4x4x4f16(a, b, c)
// v_div_fixup_f64: IEEE division's special cases around a computed quotient
static inline double emul_div_fixup(double q, double den, double num)
{
	if (std::isfinite(num) && num != 0.0 && std::isfinite(den) && den != 0.0)
		return q;
	return num / den;
}
#define __builtin_amdgcn_div_fixup(q, d, n) emul_div_fixup(q, d, n)

// ---- rounding intrinsics: the host compiles with -ffp-contract=off, so these are the IEEE operations
#define __fmul_rn(a, b) ((float) (a) * (float) (b))
#define __fadd_rn(a, b) ((float) (a) + (float) (b))
#define __fsub_rn(a, b) ((float) (a) - (float) (b))
#define __dmul_rn(a, b) ((double) (a) * (double) (b))
#define __dadd_rn(a, b) ((double) (a) + (double) (b))
#define __dsub_rn(a, b) ((double) (a) - (double) (b))
#define __ddiv_rn(a, b) ((double) (a) / (double) (b))
#define __fdiv_rn(a, b) ((float) (a) / (float) (b))
#define __fma_rn(a, b, c) __builtin_fma((double) (a), (double) (b), (double) (c))
// v_cvt_i32_f32: toward zero, saturating, NaN -> 0
static inline int emul_float2int_rz(float v)
{
	if (v != v)
		return 0;
	if (v >= 2147483648.0f)
		return 2147483647;
	if (v <= -2147483648.0f)
		return -2147483647 - 1;
	return (int) v;
}
#define __float2int_rz(v) emul_float2int_rz(v)
// v_mul_i32_i24: the product of the operands' low 24 bits, sign-extended
#define __mul24(a, b) ((int) (((int) ((unsigned int) (a) << 8) >> 8) * (long long) ((int) ((unsigned int) (b) << 8) >> 8)))
#define __umulhi(a, b) ((unsigned int) (((unsigned long long) (unsigned int) (a) * (unsigned int) (b)) >> 32))

// ---- the launch: the grid's blocks over the host's threads, each with its own dynamic LDS; the kernarg
// segment is rebuilt for the kernels that read their arguments where they lie
#include <tuple>
namespace emul {
template <typename F>
static void launch(dim3 grid, dim3 block, size_t lds_bytes, const void *kernarg, F body)
{
	const long long blocks = (long long) grid.x * grid.y * grid.z;
	std::atomic<long long> next(0);
	auto worker = [&]() {
		// (EMUL_EXACT_LDS, the sanitizer build: exactly the bytes the launch asked for, so that a step past
		// the block's dynamic LDS is a heap overflow the sanitizer sees)
#ifdef EMUL_EXACT_LDS
		const size_t lds_alloc = (lds_bytes + 15) / 16 * 16;
		unsigned char *base = static_cast<unsigned char *>(aligned_alloc(16, lds_alloc ? lds_alloc : 16));
		struct Free {
			unsigned char *p;
			~Free() { free(p); }
		} lds_free{base};
#else
		std::vector<unsigned char> lds(lds_bytes + 64);
		unsigned char *base = lds.data();
		base += (16 - ((uintptr_t) base & 15)) & 15;
#endif
		for (;;) {
			const long long wg = next.fetch_add(1);
			if (wg >= blocks)
				break;
			Geometry &g = geometry();
			g.grid = grid;
			g.block = block;
			g.block_idx = dim3((unsigned int) (wg % grid.x), (unsigned int) (wg / grid.x % grid.y), (unsigned int) (wg / ((long long) grid.x * grid.y)));
			g.kernarg = kernarg;
			set_lds_base(base, lds_bytes);
			run_block((int) (block.x * block.y * block.z), body);
		}
	};
	long long nthreads = std::thread::hardware_concurrency();
	nthreads = nthreads < 1 ? 1 : nthreads > blocks ? blocks : nthreads;
	std::vector<std::thread> pool;
	for (long long i = 0; i < nthreads; i++)
		pool.emplace_back(worker);
	for (std::thread &t : pool)
		t.join();
}
} // namespace emul
namespace emul {
// the kernarg segment: the arguments one after the other, each at its own alignment
template <typename... A>
static std::vector<unsigned long long> pack_kernarg(const A &...args)
{
	size_t size = 0;
	((size = (size + alignof(A) - 1) / alignof(A) * alignof(A) + sizeof(A)), ...);
	std::vector<unsigned long long> buf(size / 8 + 4);
	unsigned char *p = reinterpret_cast<unsigned char *>(buf.data());
	size_t off = 0;
	((off = (off + alignof(A) - 1) / alignof(A) * alignof(A), memcpy(p + off, &args, sizeof(A)), off += sizeof(A)), ...);
	return buf;
}
} // namespace emul
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, strm, ...) \
	do { \
		(void) hipStreamSynchronize(strm); \
		auto emul_args = std::make_tuple(__VA_ARGS__); \
		const auto emul_kernarg = std::apply([](const auto &...a) { return emul::pack_kernarg(a...); }, emul_args); \
		emul::launch(dim3(grid), dim3(block), (size_t) (lds), emul_kernarg.data(), [&]() { std::apply(kernel, emul_args); }); \
	} while (0)
