// Host side of reduceh/reducev: what vips_reduceh_build() / vips_reducev_build()
// compute once per operation (resample/reduceh.cpp:396-565, reducev.cpp:859-1075)
// -- n_point, h/v offset, the 65-phase coefficient tables -- and what each
// generate call computes before its pixel loop: the double position accumulator
// (reduceh.cpp:254-276,326; reducev.cpp:548-560,611).  All of it is double
// arithmetic on the host, done here with the same libm and the same operation
// order so the tables and phase indices are bit-identical to the reference's;
// the device never evaluates a transcendental.
#include "resample.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace vh {

static const double PI = 3.14159265358979323846; // VIPS_PI, include/vips/basic.h

// resample/templates.h:330-346 (cubic_filter), :346-354 (sinc_filter),
// :359-451 (filter<K>)
static double cubic_filter(double x, double B, double C)
{
	const double ax = fabs(x);
	const double ax2 = ax * ax;
	const double ax3 = ax2 * ax;

	if (ax <= 1)
		return ((12 - 9 * B - 6 * C) * ax3 +
				   (-18 + 12 * B + 6 * C) * ax2 +
				   (6 - 2 * B)) /
			6;

	if (ax <= 2)
		return ((-B - 6 * C) * ax3 +
				   (6 * B + 30 * C) * ax2 +
				   (-12 * B - 48 * C) * ax +
				   (8 * B + 24 * C)) /
			6;

	return 0.0;
}

static double sinc_filter(double x)
{
	if (x == 0.0)
		return 1.0;
	x = x * PI;
	return sin(x) / x;
}

static double filter_value(int kernel, double x)
{
	switch (kernel) {
	case VIPS_HIP_KERNEL_LINEAR:
		x = fabs(x);
		return x < 1.0 ? 1.0 - x : 0.0;
	case VIPS_HIP_KERNEL_CUBIC:
		return cubic_filter(x, 0.0, 0.5);
	case VIPS_HIP_KERNEL_MITCHELL:
		return cubic_filter(x, 1.0 / 3.0, 1.0 / 3.0);
	case VIPS_HIP_KERNEL_LANCZOS2:
		if (x >= -2 && x <= 2)
			return sinc_filter(x) * sinc_filter(x / 2);
		return 0.0;
	case VIPS_HIP_KERNEL_LANCZOS3:
		if (x >= -3 && x <= 3)
			return sinc_filter(x) * sinc_filter(x / 3);
		return 0.0;
	case VIPS_HIP_KERNEL_MKS2013:
		x = fabs(x);
		if (x >= 2.5)
			return 0.0;
		if (x >= 1.5)
			return (x - 5.0 / 2.0) * (x - 5.0 / 2.0) / -8.0;
		if (x >= 0.5)
			return (4.0 * x * x - 11.0 * x + 7.0) / 4.0;
		return 17.0 / 16.0 - 7.0 * x * x / 4.0;
	case VIPS_HIP_KERNEL_MKS2021:
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

// vips_reduce_make_mask<T> -> calculate_coefficients<T> (templates.h:453-531): the filter is
// evaluated in double whatever T is; the sum and the normalising division are T's.
template <typename T>
static void make_mask(T *c, int kernel, int n_points, double shrink, double x)
{
	if (kernel == VIPS_HIP_KERNEL_NEAREST) {
		c[0] = 1.0;
		return;
	}

	const double half = x + n_points / 2.0 - 1;
	const double scale = 1.0 / shrink;
	T sum = 0.0;
	for (int i = 0; i < n_points; i++) {
		const double xp = (i - half) * scale;
		double l = filter_value(kernel, xp);
		c[i] = l;
		sum += l;
	}
	for (int i = 0; i < n_points; i++)
		c[i] /= sum;
}

void reduce_make_mask(double *c, int kernel, int n_points, double shrink, double x)
{
	make_mask<double>(c, kernel, n_points, shrink, x);
}

// Double images take the reference's "ultra-high-quality" path (reduceh.cpp:196-213,
// reducev.cpp:497-515): no coefficient table -- a mask made for the exact fractional position of
// every output column (row), in LONG DOUBLE (LongT<double>, templates.h:556-560), and a long
// double sum.  On x86-64 that is the x87 extended format: 64 bits of mantissa.  The masks are made
// here, on the host, with the same arithmetic (this file is built by the same compiler for the
// same ABI), and handed to the device as (sign, exponent, 64-bit mantissa) triples; the device
// adds and multiplies them with integer instructions, rounding to 64 bits of mantissa after every
// operation exactly as the x87 does (resample.hip x80_*).
#if !defined(__HIP_DEVICE_COMPILE__) // (this file also passes through the device compiler, whose long double is a double)
static_assert(sizeof(long double) == 16 && __LDBL_MANT_DIG__ == 64, "the host's long double is not the x87 extended format");
#endif

void reduce_notab_masks(const _VipsHipReduce *r, int start, int count, int tile, std::vector<ReducePos> &pos,
	std::vector<ReduceTap80> &taps)
{
	const int n = r->n_point;
	pos.resize(count);
	taps.resize((size_t) count * n);
	if (tile <= 0)
		tile = count;
	std::vector<long double> cx(n);
	for (int t0 = 0; t0 < count; t0 += tile) {
		const int m = count - t0 < tile ? count - t0 : tile;
		double X = (start + t0 + 0.5) * r->shrink - 0.5 - r->offset;
		for (int k = 0; k < m; k++) {
			const int ix = (int) X;
			pos[t0 + k].first = ix - r->embed;
			pos[t0 + k].phase = 0;
			make_mask<long double>(cx.data(), r->kernel, n, r->shrink, X - ix);
			for (int i = 0; i < n; i++) {
				unsigned char raw[16];
				memcpy(raw, &cx[i], sizeof(long double));
				unsigned long long mant;
				unsigned short se;
				memcpy(&mant, raw, 8);
				memcpy(&se, raw + 8, 2);
				ReduceTap80 &tap = taps[(size_t) (t0 + k) * n + i];
				tap.mant = mant;
				tap.sign = se >> 15;
				tap.exp = (int) (se & 0x7fff) - 16383;
				if (mant && !(mant >> 63)) { // a denormal extended value: normalise
					const int lz = __builtin_clzll(mant);
					tap.mant = mant << lz;
					tap.exp = -16382 - lz;
				}
			}
			X += r->shrink;
		}
	}
}

// One generate call's position walk, reduceh.cpp:254-276,326 (and the same
// code in reducev.cpp:548-560,611): seeded from the rect origin, advanced by
// repeated addition.
void reduce_positions(const _VipsHipReduce *r, int start, int count, int tile,
	std::vector<ReducePos> &pos)
{
	pos.resize(count);
	if (tile <= 0)
		tile = count;
	for (int t0 = 0; t0 < count; t0 += tile) {
		int n = count - t0 < tile ? count - t0 : tile;
		double X = (start + t0 + 0.5) * r->shrink - 0.5 - r->offset;
		for (int k = 0; k < n; k++) {
			const int ix = (int) X;
			const int sx = X * TRANSFORM_SCALE * 2;
			const int six = sx & (TRANSFORM_SCALE * 2 - 1);
			const int tx = (six + 1) >> 1;
			// ix indexes the embedded image; translate to the un-embedded one.
			pos[t0 + k].first = ix - r->embed;
			pos[t0 + k].phase = tx;
			X += r->shrink;
		}
	}
}

} // namespace vh

using namespace vh;

extern "C" {

// resample/reduceh.cpp:113-141
int vips_hip_reduce_get_points(int kernel, double shrink)
{
	switch (kernel) {
	case VIPS_HIP_KERNEL_NEAREST:
		return 1;
	case VIPS_HIP_KERNEL_LINEAR:
		return 2 * rint(shrink) + 1;
	case VIPS_HIP_KERNEL_CUBIC:
	case VIPS_HIP_KERNEL_MITCHELL:
		return 2 * rint(2 * shrink) + 1;
	case VIPS_HIP_KERNEL_LANCZOS2:
		return 2 * rint(2 * shrink) + 1;
	case VIPS_HIP_KERNEL_LANCZOS3:
		return 2 * rint(3 * shrink) + 1;
	case VIPS_HIP_KERNEL_MKS2013:
		return 2 * rint(3 * shrink) + 1;
	case VIPS_HIP_KERNEL_MKS2021:
		return 2 * rint(5 * shrink) + 1;
	default:
		return 0;
	}
}

VipsHipReduce *vips_hip_reduce_new(int kernel, double shrink, int in_size, int out_size,
	double extra_pixels)
{
	const char *domain = "reduce";

	if (kernel < VIPS_HIP_KERNEL_NEAREST || kernel > VIPS_HIP_KERNEL_MKS2021) {
		error(domain, "unknown kernel %d", kernel);
		return nullptr;
	}
	if (shrink < 1.0) {
		error(domain, "reduce factor should be >= 1.0");
		return nullptr;
	}
	if (out_size <= 0) {
		error(domain, "image has shrunk to nothing");
		return nullptr;
	}
	int n_point = vips_hip_reduce_get_points(kernel, shrink);
	if (n_point > MAX_POINT) {
		error(domain, "reduce factor too large");
		return nullptr;
	}

	VipsHipReduce *r = new VipsHipReduce;
	r->kernel = kernel;
	r->shrink = shrink;
	r->in_size = in_size;
	r->out_size = out_size;
	r->n_point = n_point;
	if (std::isnan(extra_pixels))
		extra_pixels = out_size * shrink - in_size;
	// reduceh.cpp:480, reducev.cpp:941
	r->offset = (1 + extra_pixels) / 2.0 - 1;
	// reduceh.cpp:515-520: vips_embed(x = ceil(n_point / 2.0) - 1)
	r->embed = (int) (ceil(n_point / 2.0) - 1);

	r->matrixf.resize((size_t) (TRANSFORM_SCALE + 1) * n_point);
	r->matrixs.resize((size_t) (TRANSFORM_SCALE + 1) * n_point);
	for (int x = 0; x < TRANSFORM_SCALE + 1; x++) {
		double *cf = &r->matrixf[(size_t) x * n_point];
		short *cs = &r->matrixs[(size_t) x * n_point];
		// reduceh.cpp:493-495: the phase is computed in float
		reduce_make_mask(cf, kernel, n_point, shrink, (float) x / TRANSFORM_SCALE);
		for (int i = 0; i < n_point; i++)
			cs[i] = (short) (cf[i] * INTERPOLATE_SCALE);
	}
	r->d_matrixf = nullptr;
	r->d_matrixs = nullptr;
	return r;
}

void vips_hip_reduce_free(VipsHipReduce *r)
{
	if (!r)
		return;
	vips_hip_free(r->d_matrixf);
	vips_hip_free(r->d_matrixs);
	for (auto &kv : r->pos_cache)
		vips_hip_free(kv.second);
	delete r;
}

int vips_hip_reduce_get_n_point(const VipsHipReduce *r) { return r ? r->n_point : -1; }
int vips_hip_reduce_get_out_size(const VipsHipReduce *r) { return r ? r->out_size : -1; }
double vips_hip_reduce_get_offset(const VipsHipReduce *r) { return r ? r->offset : 0.0; }

int vips_hip_reduce_get_matrixs(const VipsHipReduce *r, int phase, short *out)
{
	if (!r || !out || phase < 0 || phase > TRANSFORM_SCALE)
		return -1;
	memcpy(out, &r->matrixs[(size_t) phase * r->n_point], sizeof(short) * r->n_point);
	return 0;
}

int vips_hip_reduce_get_matrixf(const VipsHipReduce *r, int phase, double *out)
{
	if (!r || !out || phase < 0 || phase > TRANSFORM_SCALE)
		return -1;
	memcpy(out, &r->matrixf[(size_t) phase * r->n_point], sizeof(double) * r->n_point);
	return 0;
}

static void reduce_need(const VipsHipReduce *r, int start, int count, int *in_start, int *in_count)
{
	if (!r || !in_start || !in_count) {
		if (in_start)
			*in_start = 0;
		if (in_count)
			*in_count = 0;
		return;
	}
	// reduceh.cpp:237-240 in embedded coordinates...
	int s0 = (int) (start * r->shrink - r->offset);
	int sn = (int) (count * r->shrink + r->n_point);
	// ...translated to the un-embedded input and clipped (what vips_embed's
	// generate asks of its own input, conversion/embed.c:300-341).
	int lo = s0 - r->embed;
	int hi = lo + sn;
	if (lo < 0)
		lo = 0;
	if (hi > r->in_size)
		hi = r->in_size;
	if (lo > r->in_size - 1)
		lo = r->in_size - 1;
	if (hi < lo + 1)
		hi = lo + 1;
	*in_start = lo;
	*in_count = hi - lo;
}

void vips_hip_reduceh_need(const VipsHipReduce *r, int left, int width, int *in_left, int *in_width)
{
	reduce_need(r, left, width, in_left, in_width);
}

void vips_hip_reducev_need(const VipsHipReduce *r, int top, int height, int *in_top, int *in_height)
{
	reduce_need(r, top, height, in_top, in_height);
}

// shrinkh.c:414-416, shrinkv.c:566-568
int vips_hip_shrink_out_size(int in_size, int shrink, int ceil_mode)
{
	if (shrink < 1)
		return -1;
	if (ceil_mode)
		return (int) ceil((double) in_size / shrink);
	return (int) ((double) in_size / shrink + 0.5); // VIPS_ROUND_UINT
}

} // extern "C"
