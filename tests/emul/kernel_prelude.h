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
#include <thread>
#include <vector>

// ---- csrc/gcn.h, last section
#define VH_SCALAR(x) ((void) (x))
#define VH_SCALAR2(x, y) ((void) (x), (void) (y))
#define VH_VECTOR2(a, b) ((void) (a), (void) (b))
#define VH_VECTOR5(a, b, c, d, e) ((void) (a), (void) (b), (void) (c), (void) (d), (void) (e))
#define VH_WAIT_VMCNT(n) ((void) 0) // (the emulated LDS-DMA lands at once)
// lane i's dword to LDS offset lds_dst + 4 i
#define VH_LDS_DMA_DWORD(src, voff, lds_dst) \
	memcpy(emul::lds_base() + (lds_dst) + 4u * (unsigned int) (emul::current_tid() & 63), (const char *) (src) + (voff), 4)
#define VH_DYNAMIC_LDS(T, name) T *name = reinterpret_cast<T *>(emul::lds_base())
#define VH_LDS_ADDR(p) ((unsigned int) (reinterpret_cast<unsigned char *>(p) - emul::lds_base()))

using std::isnan;
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

struct EmulThreadIdx {
	struct X {
		operator int() const { return emul::current_tid(); }
	} x;
};
#define threadIdx (EmulThreadIdx())
#define __syncthreads() emul::barrier()
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

// ---- amdgcn builtins (wave-uniform values are uniform by construction in the kernels emulated here)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void) 0)
#define __builtin_amdgcn_ballot_w64(p) emul::ballot(p)
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

// ---- the launch: `grid` persistent blocks over the host's threads, each with its own dynamic LDS
namespace emul {
template <typename F>
static void launch(int grid, int block, size_t lds_bytes, F body)
{
	std::atomic<int> next(0);
	auto worker = [&]() {
		std::vector<unsigned char> lds(lds_bytes + 64);
		for (;;) {
			const int wg = next.fetch_add(1);
			if (wg >= grid)
				break;
			unsigned char *base = lds.data();
			base += (16 - ((uintptr_t) base & 15)) & 15;
			set_lds_base(base);
			run_block(block, body);
		}
	};
	unsigned int nthreads = std::thread::hardware_concurrency();
	nthreads = nthreads < 1 ? 1 : nthreads > (unsigned int) grid ? (unsigned int) grid : nthreads;
	std::vector<std::thread> pool;
	for (unsigned int i = 0; i < nthreads; i++)
		pool.emplace_back(worker);
	for (std::thread &t : pool)
		t.join();
}
} // namespace emul
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, strm, ...) \
	do { \
		(void) hipStreamSynchronize(strm); \
		emul::launch((int) (grid).x, (int) (block).x, (size_t) (lds), [&]() { kernel(__VA_ARGS__); }); \
	} while (0)
