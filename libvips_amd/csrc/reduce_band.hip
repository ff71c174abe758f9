// vips_reducev on uchar with a coefficient row per output row, as a banded matrix product on the matrix
// cores: the __global__ wrapper and launch of reduce_band_body.h (see there); host side reduce_band_host.h.
#include "reduce_band_body.h"

namespace vh {

__global__ void __launch_bounds__(256)
reducev_u8_band(RbArgs a, int groups)
{
	// blockIdx -> (group of 4 strips, block of rows): neighbouring blocks of rows on one XCD (they share 16 % of
	// their input rows and the hardware deals consecutive blocks to the 8 XCDs in turn)
	const int wv = wave_index();
	const int id = (int) blockIdx.x;
	const int g = id / groups, grp = id - g * groups;
	const int strip = 4 * grp + wv;
	if (strip < a.strips)
		reducev_band_wave(a, strip, g);
}

} // namespace vh

#include "reduce_band_host.h"

namespace vh {

static int rb_launch(const RbArgs &a, int grid)
{
	const int groups = (a.strips + 3) / 4;
	hipLaunchKernelGGL(reducev_u8_band, dim3(grid), dim3(RB_NT), 0, stream(), a, groups);
	VH_CHECK(hipGetLastError());
	return 0;
}

} // namespace vh
