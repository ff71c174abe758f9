// uchar fast paths -- see reduce_u8.h.  (first cut: general kernels only)
#include "reduce_u8.h"

namespace vh {

int reducev_u8_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const ReducePos *pos, const short *table)
{
	return 0;
}

int reduceh_u8_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const ReducePos *pos, const short *table)
{
	return 0;
}

int shrinkv_u8_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return 0;
}

int shrinkh_u8_try(int hshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return 0;
}

} // namespace vh

extern "C" {

int vips_hip_reduce_gen_tiled(const VipsHipReduce *reducev, const VipsHipReduce *reduceh,
	const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	return 1;
}

int vips_hip_reduce_gen(const VipsHipReduce *reducev, const VipsHipReduce *reduceh,
	const VipsHipRegion *in, const VipsHipRegion *out)
{
	return vips_hip_reduce_gen_tiled(reducev, reduceh, in, out, 0);
}

} // extern "C"
