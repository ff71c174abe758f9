// vips_reduceh on uchar with one coefficient row, packed bytes: __global__ wrappers and launches of
// reduceh_u8_body.h; host side reduceh_u8_host.h (both shared with tests/emul).
#include "reduceh_u8_body.h"

namespace vh {

template <int B, int STEP4, int ND>
__global__ void __launch_bounds__(RH8_NT)
reduceh_u8p(Rh8Args a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned int rh8_lds[];
	reduceh_u8p_body<B, STEP4, ND>(a, (int) blockIdx.x, (int) blockIdx.y, (int) gridDim.y, rh8_lds);
}

} // namespace vh

#include "reduceh_u8_host.h"

namespace vh {

template <int B, int STEP4>
static int rh8_launch_nd(int nd, const Rh8Args &a, const dim3 &grid, size_t lds)
{
	const dim3 block(RH8_NT, 1, 1);
#define RH8_CASE(ND) \
	case ND: \
		hipLaunchKernelGGL((reduceh_u8p<B, STEP4, ND>), grid, block, lds, stream(), a); \
		break;
	switch (nd) {
		RH8_CASE(3) RH8_CASE(5) RH8_CASE(7) RH8_CASE(9) RH8_CASE(13)
	default:
		return -1;
	}
#undef RH8_CASE
	return hipGetLastError() != hipSuccess ? -1 : 0;
}

template <int B>
static int rh8_launch_b(int step4, int nd, const Rh8Args &a, const dim3 &grid, size_t lds)
{
	return step4 == 1 ? rh8_launch_nd<B, 1>(nd, a, grid, lds) : step4 == 2 ? rh8_launch_nd<B, 2>(nd, a, grid, lds) : -1;
}

static int rh8_launch(int bands, int step4, int nd, const Rh8Args &a, int gx, int gy, size_t lds)
{
	const dim3 grid(gx, gy, 1);
	switch (bands) {
	case 1: return rh8_launch_b<1>(step4, nd, a, grid, lds);
	case 2: return rh8_launch_b<2>(step4, nd, a, grid, lds);
	case 3: return rh8_launch_b<3>(step4, nd, a, grid, lds);
	case 4: return rh8_launch_b<4>(step4, nd, a, grid, lds);
	default: return -1;
	}
}

} // namespace vh
