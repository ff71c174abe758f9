// convi (C path) / convf generates for gfx950, plus vips_gaussmat.
//
// Arithmetic contracts (reference: libvips/convolution):
//   convi, integer input   sum(int64) of coeff[i] * p[off[i]] over the non-zero mask
//                          elements in row-major order, (sum + scale/2) / scale + offset
//                          with C (truncating) division, clip to the format
//                          (convi.c:698-716, 743-751); coeff = rint(mask), scale/offset =
//                          rint() of the mask's scale/offset (convi.c:760-762,886-894)
//   convi, float input     double sum, sum / scale + offset           (convi.c:721-741)
//   convf                  double sum seeded with offset, coeff = mask / scale,
//                          separate mul and add in mask order; integer inputs give
//                          float output                         (convf.c:163-181,300-355)
// The "convolution" is a correlation: out(x, y) = sum mask(i, j) * in(x + i - mw/2,
// y + j - mh/2), input coordinates clamped to the image, which is the
// vips_embed(VIPS_EXTEND_COPY) of convi.c:1142-1147 / convf.c:331-336.
#include "conv.h"

#include <climits>
#include <cmath>

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
	std::mutex mutex;
};

namespace vh {

struct ConvArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int in_left, in_top, im_width, im_height;
	int out_left, out_top, out_width, out_height;
	int epp;
	int nnz;
	int half_w, half_h;
	const void *coeff;
	const short *dx, *dy;
	int scale_i, rounding, offset_i;
	double offset;
};

template <typename T>
struct ConvClip;
#define CONV_CLIP(TYPE, LO, HI) \
	template <> \
	struct ConvClip<TYPE> { \
		static __device__ __forceinline__ TYPE run(long long v) \
		{ \
			v = v > (long long) (HI) ? (long long) (HI) : v; \
			v = v < (long long) (LO) ? (long long) (LO) : v; \
			return (TYPE) v; \
		} \
	};
CONV_CLIP(unsigned char, 0, UCHAR_MAX)
CONV_CLIP(signed char, SCHAR_MIN, SCHAR_MAX)
CONV_CLIP(unsigned short, 0, USHRT_MAX)
CONV_CLIP(short, SHRT_MIN, SHRT_MAX)
#undef CONV_CLIP
// CLIP_NONE: the int64 is assigned straight to the 32-bit type (convi.c:832-838)
template <>
struct ConvClip<unsigned int> {
	static __device__ __forceinline__ unsigned int run(long long v) { return (unsigned int) v; }
};
template <>
struct ConvClip<int> {
	static __device__ __forceinline__ int run(long long v) { return (int) v; }
};

// MODE 0: convi on an integer format.  MODE 1: convi on float/double.  MODE 2: convf.
template <typename TIN, typename TOUT, int MODE>
__global__ void __launch_bounds__(256)
conv_general(ConvArgs a)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.out_width * a.epp)
		return;
	const int x = e / a.epp;
	const int b = e - x * a.epp;
	const int gx = a.out_left + x - a.half_w;
	for (int y = blockIdx.y; y < a.out_height; y += gridDim.y) {
		const int gy = a.out_top + y - a.half_h;
		TOUT *dst = (TOUT *) (a.out + (long long) y * a.out_stride);
		if constexpr (MODE == 0) {
			const int *c = (const int *) a.coeff;
			long long sum = 0;
			for (int i = 0; i < a.nnz; i++) {
				const int col = min(max(gx + a.dx[i], 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + a.dy[i], 0), a.im_height - 1) - a.in_top;
				const TIN *src = (const TIN *) (a.in + row * a.in_stride);
				sum += (long long) c[i] * (long long) src[(long long) col * a.epp + b];
			}
			sum = ((sum + a.rounding) / a.scale_i) + a.offset_i;
			dst[e] = (TOUT) ConvClip<TIN>::run(sum);
		}
		else if constexpr (MODE == 1) {
			const int *c = (const int *) a.coeff;
			double sum = 0;
			for (int i = 0; i < a.nnz; i++) {
				const int col = min(max(gx + a.dx[i], 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + a.dy[i], 0), a.im_height - 1) - a.in_top;
				const TIN *src = (const TIN *) (a.in + row * a.in_stride);
				sum = __dadd_rn(sum, __dmul_rn((double) c[i], (double) src[(long long) col * a.epp + b]));
			}
			sum = __dadd_rn(__ddiv_rn(sum, (double) a.scale_i), (double) a.offset_i);
			dst[e] = (TOUT) sum;
		}
		else {
			const double *c = (const double *) a.coeff;
			double sum = a.offset;
			for (int i = 0; i < a.nnz; i++) {
				const int col = min(max(gx + a.dx[i], 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + a.dy[i], 0), a.im_height - 1) - a.in_top;
				const TIN *src = (const TIN *) (a.in + row * a.in_stride);
				sum = __dadd_rn(sum, __dmul_rn(c[i], (double) src[(long long) col * a.epp + b]));
			}
			dst[e] = (TOUT) sum;
		}
	}
}

template <typename TIN, typename TOUT, int MODE>
static int launch_conv(const ConvArgs &a, const char *gate_name)
{
	const int ne = a.out_width * a.epp;
	dim3 block(256, 1, 1);
	dim3 grid((ne + 255) / 256, a.out_height < 32768 ? a.out_height : 32768, 1);
	Gate gate(gate_name);
	hipLaunchKernelGGL((conv_general<TIN, TOUT, MODE>), grid, block, 0, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

static int conv_tables(_VipsHipConv *c)
{
	std::lock_guard<std::mutex> lock(c->mutex);
	if (c->d_coeff)
		return 0;
	std::vector<short> dx(c->nnz), dy(c->nnz);
	for (int i = 0; i < c->nnz; i++) {
		dx[i] = (short) (c->pos[i] % c->mask_width);
		dy[i] = (short) (c->pos[i] / c->mask_width);
	}
	c->d_dx = (short *) upload(dx.data(), dx.size() * sizeof(short));
	c->d_dy = (short *) upload(dy.data(), dy.size() * sizeof(short));
	if (c->precision == VIPS_HIP_PRECISION_INTEGER)
		c->d_coeff = upload(c->coeffi.data(), c->coeffi.size() * sizeof(int));
	else
		c->d_coeff = upload(c->coefff.data(), c->coefff.size() * sizeof(double));
	if (!c->d_dx || !c->d_dy || !c->d_coeff)
		return -1;
	return 0;
}

} // namespace vh

using namespace vh;

extern "C" {

VipsHipConv *vips_hip_conv_new(const double *mask, int mask_width, int mask_height, double scale,
	double offset, int precision)
{
	const char *domain = precision == VIPS_HIP_PRECISION_INTEGER ? "convi" : "convf";
	if (!mask || mask_width <= 0 || mask_height <= 0) {
		error(domain, "bad mask");
		return nullptr;
	}
	// vips_check_matrix, iofuncs/error.c:1196
	if (mask_width > 100000 || mask_height > 100000) {
		error(domain, "matrix image too large");
		return nullptr;
	}
	if (precision != VIPS_HIP_PRECISION_INTEGER && precision != VIPS_HIP_PRECISION_FLOAT) {
		error(domain, "precision 'approximate' (vips_conva) is outside the HIP path");
		return nullptr;
	}
	if (mask_width > 32767 || mask_height > 32767) {
		error(domain, "mask too large for the HIP path");
		return nullptr;
	}
	VipsHipConv *c = new VipsHipConv;
	c->precision = precision;
	c->mask_width = mask_width;
	c->mask_height = mask_height;
	c->scale = scale;
	c->offset = offset;
	c->d_coeff = nullptr;
	c->d_dx = c->d_dy = nullptr;
	c->scale_i = c->rounding = c->offset_i = 0;
	const int ne = mask_width * mask_height;
	if (precision == VIPS_HIP_PRECISION_INTEGER) {
		// vips_convi_gen reads scale/offset from the ORIGINAL mask (convi.c:760-762):
		// the scale adjustment vips__image_intize computes (:909-915) is not used by
		// the C path.  Elements are rint()ed (:886-889), zeros squeezed out (:1191-1209).
		c->scale_i = (int) rint(scale);
		c->rounding = c->scale_i / 2;
		c->offset_i = (int) rint(offset);
		if (c->scale_i == 0) {
			error(domain, "mask scale rounds to zero");
			delete c;
			return nullptr;
		}
		for (int i = 0; i < ne; i++) {
			const double v = rint(mask[i]);
			if (v) {
				c->coeffi.push_back((int) v);
				c->pos.push_back(i);
			}
		}
		if (c->coeffi.empty()) {
			c->coeffi.push_back(0);
			c->pos.push_back(0);
		}
		c->nnz = (int) c->coeffi.size();
	}
	else {
		// convf.c:300-323: bake the scale into the mask, keep the non-zero elements
		for (int i = 0; i < ne; i++) {
			const double v = mask[i] / scale;
			if (v) {
				c->coefff.push_back(v);
				c->pos.push_back(i);
			}
		}
		if (c->coefff.empty()) {
			c->coefff.push_back(0);
			c->pos.push_back(0);
		}
		c->nnz = (int) c->coefff.size();
	}
	return c;
}

void vips_hip_conv_free(VipsHipConv *c)
{
	if (!c)
		return;
	vips_hip_free(c->d_coeff);
	vips_hip_free(c->d_dx);
	vips_hip_free(c->d_dy);
	delete c;
}

int vips_hip_conv_get_nnz(const VipsHipConv *c)
{
	return c ? c->nnz : -1;
}

int vips_hip_conv_out_format(const VipsHipConv *c, int format)
{
	if (!c)
		return -1;
	// convf.c:354-355
	if (c->precision == VIPS_HIP_PRECISION_FLOAT && format_isint(format))
		return VIPS_HIP_FORMAT_FLOAT;
	return format;
}

int vips_hip_conv_gen(const VipsHipConv *conv, const VipsHipRegion *in, const VipsHipRegion *out)
{
	const char *domain = conv && conv->precision == VIPS_HIP_PRECISION_INTEGER ? "convi" : "convf";
	if (ensure_init())
		return -1;
	if (!conv) {
		error("conv", "null conv");
		return -1;
	}
	_VipsHipConv *c = const_cast<_VipsHipConv *>(conv);
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->bands != out->bands || out->format != vips_hip_conv_out_format(c, in->format)) {
		error(domain, "output region has the wrong bands or format");
		return -1;
	}
	if (in->im_width != out->im_width || in->im_height != out->im_height) {
		error(domain, "input and output images must have the same size");
		return -1;
	}
	// the window must cover out rect grown by the mask (convi.c:778-782), clipped
	const int half_w = c->mask_width / 2, half_h = c->mask_height / 2;
	{
		int x0 = out->left - half_w, x1 = out->left + out->width - 1 - half_w + c->mask_width - 1;
		int y0 = out->top - half_h, y1 = out->top + out->height - 1 - half_h + c->mask_height - 1;
		x0 = x0 < 0 ? 0 : x0;
		y0 = y0 < 0 ? 0 : y0;
		x1 = x1 > in->im_width - 1 ? in->im_width - 1 : x1;
		y1 = y1 > in->im_height - 1 ? in->im_height - 1 : y1;
		if (x0 < in->left || y0 < in->top || x1 >= in->left + in->width ||
			y1 >= in->top + in->height) {
			error(domain, "input region too small");
			return -1;
		}
	}
	if (conv_tables(c))
		return -1;

	ConvArgs a;
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_left = in->left;
	a.in_top = in->top;
	a.im_width = in->im_width;
	a.im_height = in->im_height;
	a.out_left = out->left;
	a.out_top = out->top;
	a.out_width = out->width;
	a.out_height = out->height;
	a.epp = region_elems_per_pel(in);
	a.nnz = c->nnz;
	a.half_w = half_w;
	a.half_h = half_h;
	a.coeff = c->d_coeff;
	a.dx = c->d_dx;
	a.dy = c->d_dy;
	a.scale_i = c->scale_i;
	a.rounding = c->rounding;
	a.offset_i = c->offset_i;
	a.offset = c->offset;

	const int fmt = format_real(in->format);
	if (c->precision == VIPS_HIP_PRECISION_INTEGER) {
		switch (fmt) {
		case VIPS_HIP_FORMAT_UCHAR: return launch_conv<unsigned char, unsigned char, 0>(a, "convi");
		case VIPS_HIP_FORMAT_CHAR: return launch_conv<signed char, signed char, 0>(a, "convi");
		case VIPS_HIP_FORMAT_USHORT: return launch_conv<unsigned short, unsigned short, 0>(a, "convi");
		case VIPS_HIP_FORMAT_SHORT: return launch_conv<short, short, 0>(a, "convi");
		case VIPS_HIP_FORMAT_UINT: return launch_conv<unsigned int, unsigned int, 0>(a, "convi");
		case VIPS_HIP_FORMAT_INT: return launch_conv<int, int, 0>(a, "convi");
		case VIPS_HIP_FORMAT_FLOAT: return launch_conv<float, float, 1>(a, "convi");
		case VIPS_HIP_FORMAT_DOUBLE: return launch_conv<double, double, 1>(a, "convi");
		default: break;
		}
	}
	else {
		switch (fmt) {
		case VIPS_HIP_FORMAT_UCHAR: return launch_conv<unsigned char, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_CHAR: return launch_conv<signed char, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_USHORT: return launch_conv<unsigned short, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_SHORT: return launch_conv<short, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_UINT: return launch_conv<unsigned int, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_INT: return launch_conv<int, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_FLOAT: return launch_conv<float, float, 2>(a, "convf");
		case VIPS_HIP_FORMAT_DOUBLE: return launch_conv<double, double, 2>(a, "convf");
		default: break;
		}
	}
	error(domain, "unsupported band format %d", in->format);
	return -1;
}

// vips_gaussmat_build, create/gaussmat.c:95-167
int vips_hip_gaussmat(double sigma, double min_ampl, int separable, int precision, double *mask,
	int max, double *scale)
{
	const double sig2 = 2. * sigma * sigma;
	double clipped = 8 * sigma;
	clipped = clipped > 5000 ? 5000 : clipped; // VIPS_CLIP(0, 8 * sigma, MASK_SANITY)
	clipped = clipped < 0 ? 0 : clipped;
	const int max_x = (int) clipped;
	int x, y;

	if (precision == VIPS_HIP_PRECISION_APPROXIMATE)
		precision = VIPS_HIP_PRECISION_INTEGER; // "!= FLOAT" rounds, gaussmat.c:152
	for (x = 0; x < max_x; x++) {
		const double v = exp(-((double) (x * x)) / sig2);

		if (v < min_ampl)
			break;
	}
	if (x >= 5000) {
		error("gaussmat", "mask too large");
		return -1;
	}
	const int width = 2 * ((x - 1) > 0 ? (x - 1) : 0) + 1;
	const int height = separable ? 1 : width;
	if (!mask || (long long) width * height > max) {
		if (mask) {
			error("gaussmat", "mask buffer too small (%d x %d)", width, height);
			return -1;
		}
		return width;
	}
	double sum = 0.0;
	for (y = 0; y < height; y++)
		for (x = 0; x < width; x++) {
			const int xo = x - width / 2;
			const int yo = y - height / 2;
			const double distance = xo * xo + yo * yo;
			double v = exp(-distance / sig2);

			if (precision != VIPS_HIP_PRECISION_FLOAT)
				v = rint(20 * v);
			mask[(size_t) y * width + x] = v;
			sum += v;
		}
	if (sum == 0)
		sum = 1;
	if (scale)
		*scale = sum;
	return width;
}

} // extern "C"
