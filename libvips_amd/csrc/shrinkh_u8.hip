// vips_shrinkh on uchar, packed bytes: __global__ wrappers and launches of shrinkh_u8_body.h; host
// side shrinkh_u8_host.h (both shared with tests/emul).
#include "shrinkh_u8_body.h"

namespace vh {

template <int B, int HS>
__global__ void __launch_bounds__(SH8_NT)
shrinkh_u8(Sh8Args a)
{
	shrinkh_u8_body<B, HS>(a, (int) blockIdx.x, (int) blockIdx.y, (int) gridDim.y);
}

} // namespace vh

#include "shrinkh_u8_host.h"

namespace vh {

template <int B>
static int sh8_launch_b(int hs, const Sh8Args &a, const dim3 &grid)
{
	const dim3 block(SH8_NT, 1, 1);
#define SH8_CASE(HS) \
	case HS: \
		hipLaunchKernelGGL((shrinkh_u8<B, HS>), grid, block, 0, stream(), a); \
		break;
	switch (hs) {
		SH8_CASE(0) SH8_CASE(2) SH8_CASE(3) SH8_CASE(4) SH8_CASE(5) SH8_CASE(6) SH8_CASE(7) SH8_CASE(8)
	default:
		return -1;
	}
#undef SH8_CASE
	return hipGetLastError() != hipSuccess ? -1 : 0;
}

static int sh8_launch(int bands, int hs_template, const Sh8Args &a, int gx, int gy)
{
	const dim3 grid(gx, gy, 1);
	switch (bands) {
	case 1: return sh8_launch_b<1>(hs_template, a, grid);
	case 2: return sh8_launch_b<2>(hs_template, a, grid);
	case 3: return sh8_launch_b<3>(hs_template, a, grid);
	case 4: return sh8_launch_b<4>(hs_template, a, grid);
	default: return -1;
	}
}

} // namespace vh
