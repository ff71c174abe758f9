// uchar fast paths: the gfx950 counterparts of the reference's Highway kernels
// (resample/reduceh_hwy.cpp:79, reducev_hwy.cpp:94, shrinkh_hwy.cpp:68,
// shrinkv_hwy.cpp:90,133).  Each *_try returns 1 when it handled the call,
// 0 when the geometry is outside what the fast kernel covers (the caller then
// uses the general kernel) and -1 on a launch error.  Only the vertical passes have one:
// after them the image is small and resample.hip's row-batched horizontal kernels are
// launch-latency bound.
#pragma once

#include "resample.h"

namespace vh {

int reducev_u8_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const ReducePos *pos, const short *table, int tile);
int shrinkv_u8_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out);

const ReducePos *reduce_device_positions(_VipsHipReduce *r, int start, int count, int tile);

} // namespace vh
