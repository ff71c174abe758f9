/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's resample hot path.
 *
 * Plain C, whole-image, one loop nest per operation, following libvips 8.19.0
 * (/root/reference/libvips/resample):
 *   coefficient tables   templates.h:346-354,396-402,453-531; reduceh.cpp:113-141,483-506
 *   reduceh              reduceh.cpp:216-335 (generate), :396-565 (build: sizes, offset, embed)
 *   reducev              reducev.cpp:418-459,517-619 (generate), :859-1075 (build)
 *   shrinkh              shrinkh.c:78-232 (loops), :357-440 (build)
 *   shrinkv              shrinkv.c:158-310 (loops), :474-620 (build)
 * Rounding: templates.h:152-157 (unsigned), :203-209 (signed).
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file (a) against the
 * committed golden vectors in tests/golden/ that were produced by the compiled
 * reference (tests/golden/make_golden.py drives oracle/_ref), including the
 * SURVEY.md 8(c) checksum 16793779256, and (b) directly against oracle/_ref
 * wherever that library is present.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into this library, and only as the checker.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "port.h"

#define TRANSFORM_SHIFT 6
#define TRANSFORM_SCALE (1 << TRANSFORM_SHIFT)
#define INTERPOLATE_SHIFT 12
#define INTERPOLATE_SCALE (1 << INTERPOLATE_SHIFT)
#define MAX_POINT 2000
#define PORT_PI 3.14159265358979323846

/* ------------------------------------------------------------------ filters */

static double
sinc(double x)
{
	if (x == 0.0)
		return 1.0;
	x = x * PORT_PI;
	return sin(x) / x;
}

static double
cubic(double x, double B, double C)
{
	const double ax = fabs(x);
	const double ax2 = ax * ax;
	const double ax3 = ax2 * ax;

	if (ax <= 1)
		return ((12 - 9 * B - 6 * C) * ax3 + (-18 + 12 * B + 6 * C) * ax2 + (6 - 2 * B)) / 6;
	if (ax <= 2)
		return ((-B - 6 * C) * ax3 + (6 * B + 30 * C) * ax2 + (-12 * B - 48 * C) * ax +
				   (8 * B + 24 * C)) /
			6;
	return 0.0;
}

static double
kernel_value(int kernel, double x)
{
	switch (kernel) {
	case PORT_KERNEL_LINEAR:
		x = fabs(x);
		return x < 1.0 ? 1.0 - x : 0.0;
	case PORT_KERNEL_CUBIC:
		return cubic(x, 0.0, 0.5);
	case PORT_KERNEL_MITCHELL:
		return cubic(x, 1.0 / 3.0, 1.0 / 3.0);
	case PORT_KERNEL_LANCZOS2:
		return (x >= -2 && x <= 2) ? sinc(x) * sinc(x / 2) : 0.0;
	case PORT_KERNEL_LANCZOS3:
		return (x >= -3 && x <= 3) ? sinc(x) * sinc(x / 3) : 0.0;
	case PORT_KERNEL_MKS2013:
		x = fabs(x);
		if (x >= 2.5)
			return 0.0;
		if (x >= 1.5)
			return (x - 5.0 / 2.0) * (x - 5.0 / 2.0) / -8.0;
		if (x >= 0.5)
			return (4.0 * x * x - 11.0 * x + 7.0) / 4.0;
		return 17.0 / 16.0 - 7.0 * x * x / 4.0;
	case PORT_KERNEL_MKS2021:
		x = fabs(x);
		if (x >= 4.5)
			return 0.0;
		if (x >= 3.5)
			return (4.0 * x * x - 36.0 * x + 81.0) / -1152.0;
		if (x >= 2.5)
			return (4.0 * x * x - 27.0 * x + 45.0) / 144.0;
		if (x >= 1.5)
			return (24.0 * x * x - 113.0 * x + 130.0) / -144.0;
		if (x >= 0.5)
			return (140.0 * x * x - 379.0 * x + 239.0) / 144.0;
		return 577.0 / 576.0 - 239.0 * x * x / 144.0;
	default:
		return 0.0;
	}
}

int
port_reduce_get_points(int kernel, double shrink)
{
	switch (kernel) {
	case PORT_KERNEL_NEAREST:
		return 1;
	case PORT_KERNEL_LINEAR:
		return 2 * rint(shrink) + 1;
	case PORT_KERNEL_CUBIC:
	case PORT_KERNEL_MITCHELL:
	case PORT_KERNEL_LANCZOS2:
		return 2 * rint(2 * shrink) + 1;
	case PORT_KERNEL_LANCZOS3:
	case PORT_KERNEL_MKS2013:
		return 2 * rint(3 * shrink) + 1;
	case PORT_KERNEL_MKS2021:
		return 2 * rint(5 * shrink) + 1;
	default:
		return 0;
	}
}

void
port_reduce_make_mask(double *c, int kernel, int n_points, double shrink, double x)
{
	if (kernel == PORT_KERNEL_NEAREST) {
		c[0] = 1.0;
		return;
	}
	const double half = x + n_points / 2.0 - 1;
	const double scale = 1.0 / shrink;
	double sum = 0.0;
	for (int i = 0; i < n_points; i++) {
		const double xp = (i - half) * scale;
		const double l = kernel_value(kernel, xp);
		c[i] = l;
		sum += l;
	}
	for (int i = 0; i < n_points; i++)
		c[i] /= sum;
}

/* ------------------------------------------------------------ reduce state */

typedef struct {
	int n_point;
	double shrink;
	double offset;
	int embed; /* pixels of EXTEND_COPY border before the image */
	double *matrixf;
	short *matrixs;
} Reduce;

static int
reduce_init(Reduce *r, int kernel, double shrink, int in_size, int out_size, double extra_pixels)
{
	r->shrink = shrink;
	r->n_point = port_reduce_get_points(kernel, shrink);
	if (r->n_point <= 0 || r->n_point > MAX_POINT)
		return -1;
	r->offset = (1 + extra_pixels) / 2.0 - 1;
	r->embed = (int) (ceil(r->n_point / 2.0) - 1);
	r->matrixf = malloc(sizeof(double) * (TRANSFORM_SCALE + 1) * r->n_point);
	r->matrixs = malloc(sizeof(short) * (TRANSFORM_SCALE + 1) * r->n_point);
	for (int x = 0; x < TRANSFORM_SCALE + 1; x++) {
		double *cf = r->matrixf + (size_t) x * r->n_point;
		short *cs = r->matrixs + (size_t) x * r->n_point;
		port_reduce_make_mask(cf, kernel, r->n_point, shrink, (float) x / TRANSFORM_SCALE);
		for (int i = 0; i < r->n_point; i++)
			cs[i] = (short) (cf[i] * INTERPOLATE_SCALE);
	}
	return 0;
}

static void
reduce_free(Reduce *r)
{
	free(r->matrixf);
	free(r->matrixs);
}

static int
clampi(int v, int lo, int hi)
{
	return v < lo ? lo : (v > hi ? hi : v);
}

#define CLIP(A, V, B) ((V) < (A) ? (A) : ((V) > (B) ? (B) : (V)))

/* One output element: n taps `step` elements apart starting at tap index
 * `first` (embedded coordinates already removed), clamped to [0, limit).
 */
#define REDUCE_UNSIGNED(TYPE, ACC, MAXV) \
	{ \
		ACC sum = 0; \
		for (int i = 0; i < n; i++) { \
			int k = clampi(first + i, 0, limit - 1); \
			sum += (ACC) cs[i] * ((const TYPE *) base)[(size_t) k * step]; \
		} \
		sum = (sum + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT; \
		*((TYPE *) q) = (TYPE) CLIP(0, sum, (ACC) MAXV); \
	}

#define REDUCE_SIGNED(TYPE, ACC, MINV, MAXV) \
	{ \
		ACC sum = 0; \
		for (int i = 0; i < n; i++) { \
			int k = clampi(first + i, 0, limit - 1); \
			sum += (ACC) cs[i] * ((const TYPE *) base)[(size_t) k * step]; \
		} \
		const int sign_of_v = 2 * (sum >= 0) - 1; \
		const int round_by = sign_of_v * (INTERPOLATE_SCALE >> 1); \
		sum = (sum + round_by) >> INTERPOLATE_SHIFT; \
		*((TYPE *) q) = (TYPE) CLIP((ACC) MINV, sum, (ACC) MAXV); \
	}

static void
reduce_element(int format, const void *base, size_t step, int first, int limit, int n,
	const short *cs, const double *cf, void *q)
{
	switch (format) {
	case PORT_FORMAT_UCHAR:
		REDUCE_UNSIGNED(unsigned char, int32_t, UCHAR_MAX);
		break;
	case PORT_FORMAT_CHAR:
		REDUCE_SIGNED(signed char, int32_t, SCHAR_MIN, SCHAR_MAX);
		break;
	case PORT_FORMAT_USHORT:
		REDUCE_UNSIGNED(unsigned short, int32_t, USHRT_MAX);
		break;
	case PORT_FORMAT_SHORT:
		REDUCE_SIGNED(short, int32_t, SHRT_MIN, SHRT_MAX);
		break;
	case PORT_FORMAT_UINT:
		REDUCE_UNSIGNED(unsigned int, int64_t, UINT_MAX);
		break;
	case PORT_FORMAT_INT:
		REDUCE_SIGNED(int, int64_t, INT_MIN, INT_MAX);
		break;
	case PORT_FORMAT_FLOAT: {
		double sum = 0;
		for (int i = 0; i < n; i++) {
			int k = clampi(first + i, 0, limit - 1);
			sum += cf[i] * ((const float *) base)[(size_t) k * step];
		}
		*((float *) q) = sum;
		break;
	}
	default:
		break;
	}
}

static int
format_size(int format)
{
	static const int sizes[] = { 1, 1, 2, 2, 4, 4, 4, 8, 8, 16 };
	return format >= 0 && format < 10 ? sizes[format] : 0;
}

/* out(x) for x in [0,out_width): reduceh.cpp:254-276,326.  X is seeded at
 * r->left = 0 (the reference evaluates reduceh in full-width strips).
 */
int
port_reduceh(const void *in, int width, int height, int bands, int format,
	double hshrink, int kernel, int out_width, double extra_pixels, void *out)
{
	Reduce r;
	const int es = format_size(format);

	if (isnan(extra_pixels))
		extra_pixels = out_width * hshrink - width;
	if (reduce_init(&r, kernel, hshrink, width, out_width, extra_pixels))
		return -1;
	for (int y = 0; y < height; y++) {
		const char *line = (const char *) in + (size_t) y * width * bands * es;
		char *q = (char *) out + (size_t) y * out_width * bands * es;
		double X = (0 + 0.5) * r.shrink - 0.5 - r.offset;
		for (int x = 0; x < out_width; x++) {
			const int ix = (int) X;
			const int sx = X * TRANSFORM_SCALE * 2;
			const int six = sx & (TRANSFORM_SCALE * 2 - 1);
			const int tx = (six + 1) >> 1;
			for (int b = 0; b < bands; b++)
				reduce_element(format, line + (size_t) b * es, bands, ix - r.embed, width,
					r.n_point, r.matrixs + (size_t) tx * r.n_point,
					r.matrixf + (size_t) tx * r.n_point, q + ((size_t) x * bands + b) * es);
			X += r.shrink;
		}
	}
	reduce_free(&r);
	return 0;
}

/* reducev.cpp:548-560,611: Y re-seeded at the top of every `tile`-row strip,
 * the way the reference's sink hands out FATSTRIP tiles (thread.c:301-325).
 */
int
port_reducev(const void *in, int width, int height, int bands, int format,
	double vshrink, int kernel, int out_height, double extra_pixels, int tile, void *out)
{
	Reduce r;
	const int es = format_size(format);
	const size_t ne = (size_t) width * bands;

	if (isnan(extra_pixels))
		extra_pixels = out_height * vshrink - height;
	if (reduce_init(&r, kernel, vshrink, height, out_height, extra_pixels))
		return -1;
	if (tile <= 0)
		tile = out_height;
	for (int top = 0; top < out_height; top += tile) {
		const int rows = out_height - top < tile ? out_height - top : tile;
		double Y = (top + 0.5) * r.shrink - 0.5 - r.offset;
		for (int y = 0; y < rows; y++) {
			const int py = (int) Y;
			const int sy = Y * TRANSFORM_SCALE * 2;
			const int siy = sy & (TRANSFORM_SCALE * 2 - 1);
			const int ty = (siy + 1) >> 1;
			char *q = (char *) out + (size_t) (top + y) * ne * es;
			for (size_t e = 0; e < ne; e++)
				reduce_element(format, (const char *) in + e * es, ne, py - r.embed, height,
					r.n_point, r.matrixs + (size_t) ty * r.n_point,
					r.matrixf + (size_t) ty * r.n_point, q + e * es);
			Y += r.shrink;
		}
	}
	reduce_free(&r);
	return 0;
}

/* ------------------------------------------------------------------ shrink */

#define SHRINK_LOOP(TYPE, ACC, SEED, FIN) \
	{ \
		ACC sum = SEED; \
		for (int i = 0; i < shrink; i++) { \
			int k = first + i; \
			if (k > limit - 1) \
				k = limit - 1; \
			sum += ((const TYPE *) base)[(size_t) k * step]; \
		} \
		*((TYPE *) q) = FIN; \
	}

static void
shrink_element(int format, const void *base, size_t step, int first, int limit, int shrink,
	void *q)
{
	const int amend = shrink / 2;
	const unsigned int multiplier = (1LL << 32) / ((1 << 8) * shrink);
	const uint64_t ushort_multiplier = ((1ULL << 32) + shrink - 1) / shrink;
	const double inv = 1.0 / shrink;

	switch (format) {
	case PORT_FORMAT_UCHAR:
		SHRINK_LOOP(unsigned char, int, amend, (sum * multiplier) >> 24);
		break;
	case PORT_FORMAT_CHAR:
		SHRINK_LOOP(signed char, int, amend, sum / shrink);
		break;
	case PORT_FORMAT_USHORT:
		SHRINK_LOOP(unsigned short, int, amend, ((int64_t) sum * ushort_multiplier) >> 32);
		break;
	case PORT_FORMAT_SHORT:
		SHRINK_LOOP(short, int, amend, sum / shrink);
		break;
	case PORT_FORMAT_UINT:
		SHRINK_LOOP(unsigned int, int64_t, amend, sum / shrink);
		break;
	case PORT_FORMAT_INT:
		SHRINK_LOOP(int, int64_t, amend, sum / shrink);
		break;
	case PORT_FORMAT_FLOAT:
		SHRINK_LOOP(float, double, 0.0, sum * inv);
		break;
	case PORT_FORMAT_DOUBLE:
		SHRINK_LOOP(double, double, 0.0, sum * inv);
		break;
	default:
		break;
	}
}

int
port_shrink_out_size(int in_size, int shrink, int ceil_mode)
{
	if (ceil_mode)
		return (int) ceil((double) in_size / shrink);
	return (int) ((double) in_size / shrink + 0.5);
}

int
port_shrinkh(const void *in, int width, int height, int bands, int format, int hshrink,
	int ceil_mode, void *out)
{
	const int es = format_size(format);
	const int out_width = port_shrink_out_size(width, hshrink, ceil_mode);

	for (int y = 0; y < height; y++) {
		const char *line = (const char *) in + (size_t) y * width * bands * es;
		char *q = (char *) out + (size_t) y * out_width * bands * es;
		for (int x = 0; x < out_width; x++)
			for (int b = 0; b < bands; b++)
				shrink_element(format, line + (size_t) b * es, bands, x * hshrink, width, hshrink,
					q + ((size_t) x * bands + b) * es);
	}
	return 0;
}

int
port_shrinkv(const void *in, int width, int height, int bands, int format, int vshrink,
	int ceil_mode, void *out)
{
	const int es = format_size(format);
	const int out_height = port_shrink_out_size(height, vshrink, ceil_mode);
	const size_t ne = (size_t) width * bands;

	for (int y = 0; y < out_height; y++) {
		char *q = (char *) out + (size_t) y * ne * es;
		for (size_t e = 0; e < ne; e++)
			shrink_element(format, (const char *) in + e * es, ne, y * vshrink, height, vshrink,
				q + e * es);
	}
	return 0;
}
