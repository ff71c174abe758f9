// The fused resize + sharpen kernel of the batched thumbnail pipeline (BASELINE config 4): the
// __global__ wrapper and the launch; the kernel's body is resize_sharpen_body.h, its host side
// resize_sharpen_host.h (both shared with the CPU emulation of tests/emul).
#include "resize_sharpen_body.h"

namespace vh {

// 512 threads, at most 128 registers: two blocks per CU (the second __launch_bounds__ argument is
// waves per SIMD on this compiler)
template <int VS, int NP>
__global__ void __launch_bounds__(RSH_NT, 4)
resize_sharpen_u8(RshArgs a, RshPtrs ptrs_by_value)
{
	extern __shared__ __attribute__((aligned(16))) unsigned int rsh_lds[];
	(void) ptrs_by_value;
	static_assert(sizeof(RshArgs) % 8 == 0, "kernarg layout");
	const KernargWords kp = { (int) sizeof(RshArgs) };
	resize_sharpen_body<VS, NP>(a, kp, (int) blockIdx.x, rsh_lds);
}

} // namespace vh

#include "resize_sharpen_host.h"

namespace vh {

template <int VS>
static int rsh_launch_vs(const RshArgs &a, const RshPtrs &p, unsigned int blocks, size_t lds)
{
	// more than the 64 KB a launch gets without asking
	static std::mutex mutex;
	static std::map<int, size_t> allowed; // per device
	{
		std::lock_guard<std::mutex> lock(mutex);
		size_t &have = allowed[current_device()];
		if (lds > have) {
			if (hipFuncSetAttribute(reinterpret_cast<const void *>(&resize_sharpen_u8<VS, RSH_NP>),
					hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess)
				return -1;
			have = lds;
		}
	}
	hipLaunchKernelGGL((resize_sharpen_u8<VS, RSH_NP>), dim3(blocks, 1, 1), dim3(RSH_NT, 1, 1), lds, stream(), a, p);
	return hipGetLastError() != hipSuccess ? -1 : 0;
}

static int rsh_launch(int vs, const RshArgs &a, const RshPtrs &p, unsigned int blocks, size_t lds)
{
	// (one scale for both axes and boxes of whole dwords: the vertical box is 4 or 8 like the horizontal one)
	return vs == 4 ? rsh_launch_vs<4>(a, p, blocks, lds) : rsh_launch_vs<8>(a, p, blocks, lds);
}

} // namespace vh
