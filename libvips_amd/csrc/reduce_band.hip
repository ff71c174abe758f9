// vips_reducev on uchar with a coefficient row per output row, as a banded matrix product on the matrix
// cores: the __global__ wrapper and launch of reduce_band_body.h (see there); host side reduce_band_host.h.
#include "reduce_band_body.h"

namespace vh {

// a wave per (strip, block of rows); blockIdx -> (group of 4 strips, block): a group's blocks of rows all land on
// one XCD (the hardware deals consecutive blocks to the 8 XCDs in turn, groups is padded to a multiple of 8), so
// the rows two neighbouring blocks share are in its L2.  (Persistent waves dealt (strip, block) items from an
// atomic counter ran 3 x slower -- 0.146 against 0.048 ms on 8192^2 x 3 by 7.3, profiles/r05j_reduce_band_ops.txt --
// and 3 waves per SIMD instead of 4 5 % slower.)
template <bool U16>
__global__ void __launch_bounds__(256, 4)
reducev_u8_band(RbArgs a, int groups)
{
	const int wv = wave_index();
	const int id = (int) blockIdx.x;
	const int g = id / groups, grp = id - g * groups;
	const int strip = 4 * grp + wv;
	if (strip < a.strips)
		reducev_band_wave<U16>(a, strip, g, rb_bottom_up(a, g));
}

template <int VS>
__global__ void __launch_bounds__(256, 3)
shrinkv_reducev_u8_band(RbArgs a, int groups)
{
	const int wv = wave_index();
	const int id = (int) blockIdx.x;
	const int g = id / groups, grp = id - g * groups;
	const int strip = 4 * grp + wv;
	if (strip < a.strips)
		reducev_box_band_wave<VS>(a, strip, g, rb_bottom_up(a, g));
}

template <int B>
__global__ void __launch_bounds__(256)
reduceh_u8_band(RbhArgs a, int groups)
{
	const int wv = wave_index();
	const int id = (int) blockIdx.x;
	const int yt = id / groups, grp = id - yt * groups;
	const int xt = 4 * grp + wv;
	if (xt < a.xtiles)
		reduceh_band_wave<B>(a, xt, yt);
}

template <int B>
__global__ void __launch_bounds__(256)
reduceh_u16_band(RbhArgs a, int groups)
{
	const int wv = wave_index();
	const int id = (int) blockIdx.x;
	const int yt = id / groups, grp = id - yt * groups;
	const int xt = 4 * grp + wv;
	if (xt < a.xtiles)
		reduceh16_band_wave<B>(a, xt, yt);
}

} // namespace vh

#include "reduce_band_host.h"

namespace vh {

static int rb_launch(const RbArgs &a, int groups, int wblocks, bool u16)
{
	const int grid = groups * wblocks;
	if (a.vs > 1) {
		switch (a.vs) {
#define RB_BOX(VS) \
	case VS: \
		hipLaunchKernelGGL(shrinkv_reducev_u8_band<VS>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups); \
		break;
			RB_BOX(2) RB_BOX(3) RB_BOX(4) RB_BOX(5) RB_BOX(6) RB_BOX(7) RB_BOX(8) RB_BOX(9) RB_BOX(10) RB_BOX(11) RB_BOX(12)
			RB_BOX(13) RB_BOX(14) RB_BOX(15) RB_BOX(16)
#undef RB_BOX
		default:
			return 1;
		}
	}
	else if (u16)
		hipLaunchKernelGGL(reducev_u8_band<true>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
	else
		hipLaunchKernelGGL(reducev_u8_band<false>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
	VH_CHECK(hipGetLastError());
	return 0;
}

static int rbh_launch(int bands, const RbhArgs &a, int grid, bool u16)
{
	const int groups = (a.xtiles + 3) / 4;
	if (u16) {
		switch (bands) {
		case 1:
			hipLaunchKernelGGL(reduceh_u16_band<1>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
			break;
		case 2:
			hipLaunchKernelGGL(reduceh_u16_band<2>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
			break;
		case 3:
			hipLaunchKernelGGL(reduceh_u16_band<3>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
			break;
		case 4:
			hipLaunchKernelGGL(reduceh_u16_band<4>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
			break;
		default:
			return 1;
		}
		VH_CHECK(hipGetLastError());
		return 0;
	}
	switch (bands) {
	case 1:
		hipLaunchKernelGGL(reduceh_u8_band<1>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
		break;
	case 2:
		hipLaunchKernelGGL(reduceh_u8_band<2>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
		break;
	case 3:
		hipLaunchKernelGGL(reduceh_u8_band<3>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
		break;
	case 4:
		hipLaunchKernelGGL(reduceh_u8_band<4>, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
		break;
	default:
		return 1;
	}
	VH_CHECK(hipGetLastError());
	return 0;
}

} // namespace vh
