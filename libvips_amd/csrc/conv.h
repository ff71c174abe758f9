// Internal: convolution state.
#pragma once

#include "internal.h"

#include <mutex>
#include <vector>

struct _VipsHipConv {
	int precision;
	int mask_width, mask_height;
	int nnz;
	std::vector<int> coeffi;
	std::vector<double> coefff;
	std::vector<int> pos; // index into the mask, row-major
	int scale_i, rounding, offset_i;
	double scale, offset;
	// device tables
	void *d_coeff; // int[nnz] or double[nnz]
	short *d_dx, *d_dy;
	void *d_dense; // int / double [mask_width * mask_height], zeros kept (tiled kernels)
	double *d_dense8; // convf: rows zero-padded to np8 doubles (grouped kernel)
	int np8;
	std::mutex mutex;
};

namespace vh {

// convsep_f32.hip: both passes of a separable convolution of a float image in one
// streaming kernel.  Returns 1 when the case is not covered (caller runs two conv passes).
int convsep_f32_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *pass1,
	double offset2);

// approx.hip: both passes of a convasep plan through the fused kernel above; 1 when not covered.
int convasep_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConva *plan);

} // namespace vh
