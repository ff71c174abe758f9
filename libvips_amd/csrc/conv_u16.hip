// Integer convolution of ushort images on packed 16-bit lanes: the __global__ wrapper and launch of
// conv_u16_body.h (see there); host side conv_u16_host.h (both shared with tests/emul).
#include "conv_u16_body.h"

namespace vh {

template <int B, int MH, int H>
__global__ void __launch_bounds__(256)
conv_u16_2d(Cu16Args a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned int cu16_lds[];
	conv_u16_2d_block<B, MH, H>(a, cu16_lds);
}

} // namespace vh

#include "conv_u16_host.h"

namespace vh {

#define CU16_2D(B, MH, H) \
	if (bands == B && mh == MH && h == H) { \
		hipLaunchKernelGGL((conv_u16_2d<B, MH, H>), dim3(grid), dim3(CU16_NT), lds, stream(), a); \
		VH_CHECK(hipGetLastError()); \
		return 0; \
	}
#define CU16_2D_B(B) CU16_2D(B, 1, 1) CU16_2D(B, 1, 2) CU16_2D(B, 3, 1) CU16_2D(B, 3, 2) CU16_2D(B, 5, 1) CU16_2D(B, 5, 2)

static int cu16_launch(int bands, int mh, int h, const Cu16Args &a, int grid, size_t lds)
{
	CU16_2D_B(1) CU16_2D_B(3) CU16_2D_B(4)
	return 1;
}

} // namespace vh
