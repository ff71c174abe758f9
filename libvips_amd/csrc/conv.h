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
	mutable std::atomic<int> device{ -1 }; // where the device tables live (vh::plan_device)
	// device tables
	void *d_coeff; // int[nnz] or double[nnz]
	short *d_dx, *d_dy;
	void *d_dense; // int / double [mask_width * mask_height], zeros kept (tiled kernels)
	double *d_dense8; // convf: rows zero-padded to np8 doubles (grouped kernel)
	int np8;
	// the Highway variant of convi on uchar (convi.c:925-1120): 8-bit mantissas sharing one
	// exponent; hwy_ok is false when vips_convi_intize refuses the mask (the C path then runs)
	bool hwy_ok;
	bool no_vector; // internal plans (approx.hip) always want the C path's arithmetic
	int hwy_exp;
	std::vector<int> hwy_mant, hwy_pos;
	void *d_hwy_coeff, *d_hwy_dense;
	short *d_hwy_dx, *d_hwy_dy;
	std::mutex mutex;
};

namespace vh {

// convsep_f32.hip: both passes of a separable convolution of a float image in one
// streaming kernel.  Returns 1 when the case is not covered (caller runs two conv passes).
int convsep_f32_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *pass1,
	double offset2);

// convsep_stream.hip: the float cases of the above with the vertical sums in registers, and
// (route_steps != NULL, 3-band images) the colour route that follows fused behind it: `out`
// then holds the converted image.  Returns 1 when the case is not covered.
int convsep_stream_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *pass1,
	double offset2, const int *route_steps, int n_route);

// conv_u8.hip: integer convolution of uchar images on packed bytes (v_dot4_i32_i8), streaming: both
// passes of a separable mask, or a two-dimensional mask of up to 7 x 9.  1 = not their case.
int conv_u8_sep_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *pass1, double offset2);
// ... on the matrix cores (conv_u8_mfma.hip): masks of 3 .. 33 taps, 1 .. 4 bands
int conv_u8_mfma_sep_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *pass1, double offset2);
// ... and of ushort images: their bytes as 2 x bands planes, two exact products per sample and pass
int conv_u16_mfma_sep_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *pass1, double offset2);
int conv_u8_mfma_2d_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c);
int conv_u8_2d_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c);
// conv_u16.hip: the same for ushort images and masks up to 5 x 5 (1 = not its case, 0 = done, -1 = error)
int conv_u16_2d_try(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConv *c);

// approx.hip: both passes of a convasep plan through the fused kernel above; 1 when not covered.
int convasep_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConva *plan);
// approx.hip: a conva plan's pass on a whole uchar image through conv_u8_mfma_2d; 1 when not covered.
int conva_fast_image(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConva *plan);

} // namespace vh
