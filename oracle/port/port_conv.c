/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's convolution hot path.
 *
 * Follows libvips 8.19.0 (/root/reference/libvips):
 *   convi, C path        convolution/convi.c:698-716 (CONV_INT), :721-741 (CONV_FLOAT),
 *                        :753-857 (generate), :860-923 (intize), :1123-1233 (build)
 *   convf                convolution/convf.c:163-181, :185-283, :285-369
 *   gaussmat             create/gaussmat.c:95-167
 *   sharpen generate     convolution/sharpen.c:116-168, LUT :230-257
 *   edge handling        vips_embed(VIPS_EXTEND_COPY): conversion/embed.c:226-341
 *
 * Parity status: PINNED by tests/test_oracle_conv_colour.py against golden vectors
 * made by the compiled reference (tests/golden/conv_colour.npz) and against
 * oracle/_ref directly where present.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "port.h"

static int
clampi(int v, int lo, int hi)
{
	return v < lo ? lo : (v > hi ? hi : v);
}

#define CLIPV(A, V, B) ((V) < (A) ? (A) : ((V) > (B) ? (B) : (V)))

#define CONVI_INT(TYPE, CLIP_LO, CLIP_HI, DO_CLIP) \
	for (int y = 0; y < height; y++) \
		for (int x = 0; x < width; x++) \
			for (int b = 0; b < bands; b++) { \
				int64_t sum = 0; \
				for (int i = 0; i < nnz; i++) { \
					int xx = clampi(x + pos[i] % mw - mw / 2, 0, width - 1); \
					int yy = clampi(y + pos[i] / mw - mh / 2, 0, height - 1); \
					sum += (int64_t) coeff[i] * \
						((const TYPE *) in)[((size_t) yy * width + xx) * bands + b]; \
				} \
				sum = ((sum + rounding) / iscale) + ioffset; \
				if (DO_CLIP) \
					sum = CLIPV((int64_t) CLIP_LO, sum, (int64_t) CLIP_HI); \
				((TYPE *) out)[((size_t) y * width + x) * bands + b] = sum; \
			}

#define CONVI_FLOAT(TYPE) \
	for (int y = 0; y < height; y++) \
		for (int x = 0; x < width; x++) \
			for (int b = 0; b < bands; b++) { \
				double sum = 0; \
				for (int i = 0; i < nnz; i++) { \
					int xx = clampi(x + pos[i] % mw - mw / 2, 0, width - 1); \
					int yy = clampi(y + pos[i] / mw - mh / 2, 0, height - 1); \
					sum += (double) coeff[i] * \
						((const TYPE *) in)[((size_t) yy * width + xx) * bands + b]; \
				} \
				sum = (sum / iscale) + ioffset; \
				((TYPE *) out)[((size_t) y * width + x) * bands + b] = sum; \
			}

int
port_convi(const void *in, int width, int height, int bands, int format,
	const double *mask, int mw, int mh, double scale, double offset, void *out)
{
	const int n = mw * mh;
	int *coeff = malloc(sizeof(int) * n);
	int *pos = malloc(sizeof(int) * n);
	int nnz = 0;
	/* convi.c:760-762: scale / offset of the original mask, rint()ed */
	const int iscale = rint(scale);
	const int rounding = iscale / 2;
	const int ioffset = rint(offset);

	for (int i = 0; i < n; i++) {
		const double v = rint(mask[i]);
		if (v) {
			coeff[nnz] = v;
			pos[nnz] = i;
			nnz += 1;
		}
	}
	if (nnz == 0) {
		coeff[0] = 0;
		pos[0] = 0;
		nnz = 1;
	}
	if (iscale == 0) {
		free(coeff);
		free(pos);
		return -1;
	}

	switch (format) {
	case PORT_FORMAT_UCHAR:
		CONVI_INT(unsigned char, 0, UCHAR_MAX, 1);
		break;
	case PORT_FORMAT_CHAR:
		CONVI_INT(signed char, SCHAR_MIN, SCHAR_MAX, 1);
		break;
	case PORT_FORMAT_USHORT:
		CONVI_INT(unsigned short, 0, USHRT_MAX, 1);
		break;
	case PORT_FORMAT_SHORT:
		CONVI_INT(short, SHRT_MIN, SHRT_MAX, 1);
		break;
	case PORT_FORMAT_UINT:
		CONVI_INT(unsigned int, 0, 0, 0);
		break;
	case PORT_FORMAT_INT:
		CONVI_INT(int, 0, 0, 0);
		break;
	case PORT_FORMAT_FLOAT:
		CONVI_FLOAT(float);
		break;
	case PORT_FORMAT_DOUBLE:
		CONVI_FLOAT(double);
		break;
	default:
		free(coeff);
		free(pos);
		return -1;
	}
	free(coeff);
	free(pos);
	return 0;
}

#define CONVF(ITYPE, OTYPE) \
	for (int y = 0; y < height; y++) \
		for (int x = 0; x < width; x++) \
			for (int b = 0; b < bands; b++) { \
				double sum = offset; \
				for (int i = 0; i < nnz; i++) { \
					int xx = clampi(x + pos[i] % mw - mw / 2, 0, width - 1); \
					int yy = clampi(y + pos[i] / mw - mh / 2, 0, height - 1); \
					sum += coeff[i] * \
						((const ITYPE *) in)[((size_t) yy * width + xx) * bands + b]; \
				} \
				((OTYPE *) out)[((size_t) y * width + x) * bands + b] = sum; \
			}

/* Output is float for every input format except double (convf.c:354-355). */
int
port_convf(const void *in, int width, int height, int bands, int format,
	const double *mask, int mw, int mh, double scale, double offset, void *out)
{
	const int n = mw * mh;
	double *coeff = malloc(sizeof(double) * n);
	int *pos = malloc(sizeof(int) * n);
	int nnz = 0;

	for (int i = 0; i < n; i++) {
		const double v = mask[i] / scale;
		if (v) {
			coeff[nnz] = v;
			pos[nnz] = i;
			nnz += 1;
		}
	}
	if (nnz == 0) {
		coeff[0] = 0;
		pos[0] = 0;
		nnz = 1;
	}

	switch (format) {
	case PORT_FORMAT_UCHAR:
		CONVF(unsigned char, float);
		break;
	case PORT_FORMAT_CHAR:
		CONVF(signed char, float);
		break;
	case PORT_FORMAT_USHORT:
		CONVF(unsigned short, float);
		break;
	case PORT_FORMAT_SHORT:
		CONVF(short, float);
		break;
	case PORT_FORMAT_UINT:
		CONVF(unsigned int, float);
		break;
	case PORT_FORMAT_INT:
		CONVF(int, float);
		break;
	case PORT_FORMAT_FLOAT:
		CONVF(float, float);
		break;
	case PORT_FORMAT_DOUBLE:
		CONVF(double, double);
		break;
	default:
		free(coeff);
		free(pos);
		return -1;
	}
	free(coeff);
	free(pos);
	return 0;
}

/* Returns the mask width; height is 1 when separable else == width.  With
 * mask == NULL only the width is computed.
 */
int
port_gaussmat(double sigma, double min_ampl, int separable, int integer, double *mask,
	double *scale)
{
	const double sig2 = 2. * sigma * sigma;
	const int max_x = CLIPV(0, 8 * sigma, 5000);
	int x, y;

	for (x = 0; x < max_x; x++) {
		const double v = exp(-((double) (x * x)) / sig2);

		if (v < min_ampl)
			break;
	}
	if (x >= 5000)
		return -1;
	const int width = 2 * ((x - 1) > 0 ? (x - 1) : 0) + 1;
	const int height = separable ? 1 : width;
	if (!mask)
		return width;

	double sum = 0.0;
	for (y = 0; y < height; y++)
		for (x = 0; x < width; x++) {
			const int xo = x - width / 2;
			const int yo = y - height / 2;
			const double distance = xo * xo + yo * yo;
			double v = exp(-distance / sig2);

			if (integer)
				v = rint(20 * v);
			mask[(size_t) y * width + x] = v;
			sum += v;
		}
	if (sum == 0)
		sum = 1;
	*scale = sum;
	return width;
}

/* sharpen.c:230-257 */
void
port_sharpen_lut(double x1, double y2, double y3, double m1, double m2, int *lut)
{
	for (int i = 0; i < 65536; i++) {
		double v = (i - 32767) / 327.67;
		double y;

		if (v < -x1)
			y = (v + x1) * m2 + -x1 * m1;
		else if (v < x1)
			y = v * m1;
		else
			y = (v - x1) * m2 + x1 * m1;

		if (y < -y3)
			y = -y3;
		if (y > y2)
			y = y2;

		lut[i] = rint(y * 327.67);
	}
}

/* sharpen.c:131-162 on band 0 of a LabS image; other bands copied. */
void
port_sharpen_apply(const short *in, const short *blur, int n_pixels, int bands, const int *lut,
	short *out)
{
	for (int i = 0; i < n_pixels; i++) {
		const int v1 = in[(size_t) i * bands];
		const int v2 = blur[i];
		const int diff = ((v1 & 0x7fff) - (v2 & 0x7fff));
		int o = v1 + lut[diff + 32768];

		if (o < 0)
			o = 0;
		if (o > 32767)
			o = 32767;
		out[(size_t) i * bands] = o;
		for (int b = 1; b < bands; b++)
			out[(size_t) i * bands + b] = in[(size_t) i * bands + b];
	}
}

/* ------------------------------------------------------------------ convi, Highway variant
 *
 * PARITY UNPINNED: the reference's Highway path cannot be built in this image (no libhwy), so
 * nothing below has been run against it.  It restates
 *   vips_convi_intize             convolution/convi.c:925-1120 (HAVE_HWY branch): the mask as 8-bit
 *                                 mantissas with one shared exponent, refused when out of range or
 *                                 more than 2 grey levels off on a flat image
 *   vips_convi_uchar_hwy          convolution/convi_hwy.cpp:264-273, the scalar tail that the
 *                                 vector body has to equal lane for lane: int32 sum seeded with
 *                                 1 << (exp - 1), arithmetic shift by exp, offset, clip
 * and is what a Highway-built libvips computes for uchar images with precision=integer when
 * vips_vector_isenabled().
 */
int
port_convi_hwy_intize(const double *mask, int n_point, double scale, short *mant, int *pos,
	int *nnz_out, int *exp_out)
{
	if (n_point < 1)
		return -1;
	double *scaled = malloc(sizeof(double) * n_point);
	double mx, mn;
	int shift, exp, nnz;

	for (int i = 0; i < n_point; i++)
		scaled[i] = mask[i] / scale;
	mx = mn = scaled[0];
	for (int i = 1; i < n_point; i++) {
		if (scaled[i] > mx)
			mx = scaled[i];
		if (scaled[i] < mn)
			mn = scaled[i];
	}
	(void) mn;
	/* +1 so that a max exactly on a power of two still fits signed 8 bits after the * 128 */
	const double fshift = ceil(log2(mx) + 1);
	if (!(fshift <= 6 && fshift >= -24)) { /* also catches NaN / -inf from mx <= 0 */
		free(scaled);
		return -1;
	}
	shift = fshift;
	if (ceil(log2(n_point)) > 10) {
		free(scaled);
		return -1;
	}
	exp = 7 - shift;

	nnz = 0;
	for (int i = 0; i < n_point; i++) {
		const double m = rint(128 * scaled[i] * pow(2, -shift));
		if (m < -128 || m > 127) {
			free(scaled);
			return -1;
		}
		if (m) {
			mant[nnz] = m;
			pos[nnz] = i;
			nnz += 1;
		}
	}
	if (nnz == 0) {
		mant[0] = 0;
		pos[0] = 0;
		nnz = 1;
	}

	/* accuracy on a flat image */
	double true_sum = 0.0;
	int int_sum = 0;
	for (int i = 0; i < nnz; i++) {
		true_sum += 128 * scaled[pos[i]];
		int_sum += 128 * mant[i];
	}
	const int true_value = true_sum < 0 ? 0 : (true_sum > 255 ? 255 : true_sum);
	int int_value = (int_sum + (1 << (exp - 1))) >> exp;
	int_value = int_value < 0 ? 0 : (int_value > 255 ? 255 : int_value);
	free(scaled);
	if (abs(true_value - int_value) > 2)
		return -1;

	*nnz_out = nnz;
	*exp_out = exp;
	return 0;
}

/* 0 = done, 1 = the mask is refused by the intize (the reference then runs its C path). */
int
port_convi_hwy(const unsigned char *in, int width, int height, int bands, const double *mask, int mw,
	int mh, double scale, double offset, unsigned char *out)
{
	const int n = mw * mh;
	short *mant = malloc(sizeof(short) * n);
	int *pos = malloc(sizeof(int) * n);
	int nnz, exp;
	const int ioffset = rint(offset);

	if (port_convi_hwy_intize(mask, n, scale, mant, pos, &nnz, &exp)) {
		free(mant);
		free(pos);
		return 1;
	}
	for (int y = 0; y < height; y++)
		for (int x = 0; x < width; x++)
			for (int b = 0; b < bands; b++) {
				int32_t sum = 1 << (exp - 1);
				for (int i = 0; i < nnz; i++) {
					const int xx = clampi(x + pos[i] % mw - mw / 2, 0, width - 1);
					const int yy = clampi(y + pos[i] / mw - mh / 2, 0, height - 1);
					sum += in[((size_t) yy * width + xx) * bands + b] * mant[i];
				}
				const int v = (sum >> exp) + ioffset;
				out[((size_t) y * width + x) * bands + b] = v < 0 ? 0 : (v > 255 ? 255 : v);
			}
	free(mant);
	free(pos);
	return 0;
}
