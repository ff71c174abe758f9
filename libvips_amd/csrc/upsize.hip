// Upsizing half of vips_resize (resample/resize.c:230-300) for gfx950: vips_affine restricted
// to a pure scale (b = c = 0) with the nearest / bilinear / bicubic interpolators, and vips_zoom.
//
//   transform   resample/transform.c:40-75, :180-252; build resample/affine.c:420-620 (embed by
//               window_offset + 1 with EXTEND_COPY -- never materialised: coordinates clamp)
//   generate    resample/affine.c:230-397.  Per generate rect the input coordinate of the
//               first pixel is computed from scratch and then ACCUMULATED (`ix += ddx`) along
//               the row, so a pixel's x coordinate depends on where its rect starts; the
//               host replays that accumulation into a table of one double per output column
//               (rects start at multiples of `tile_width`, 0 = whole rows: the FATSTRIP
//               geometry a scale-only affine asks for).  y is computed fresh per row.
//   nearest     resample/interpolate.c:336-352
//   bilinear    resample/interpolate.c:432-484: 12-bit fixed point for 8 / 16 bit formats,
//               double for uint / int / float
//   bicubic     resample/bicubic.cpp:482-600, tables :620-633, arithmetic
//               resample/templates.h:152-290: fixed point for (u)char, double with clip for the
//               16 / 32-bit integers, double rounded to float per row for float
// One thread per output pixel column position; all float arithmetic in the reference's
// order with separately rounded operations (the library is built with -ffp-contract=off).
#include "resample.h"
#include "kernel_stmt.h"

#include <climits>
#include <cmath>
#include <cstring>
#include <memory>
#include <type_traits>
#include <list>
#include <mutex>
#include <vector>

namespace vh {

// TRANSFORM_SCALE (64), INTERPOLATE_SHIFT (12), INTERPOLATE_SCALE: resample.h

struct BicubicTables {
	int mi[TRANSFORM_SCALE + 1][4];
	double mf[TRANSFORM_SCALE + 1][4];
};

struct UpsizeArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int in_left, in_top;       // origin of the input window
	int im_width, im_height;   // the whole input image
	int out_left, out_top, out_width, out_height; // the rect being generated
	int bands;
	int window_offset;
	double id, tidy;           // y = id * oy - tidy + window_offset
	const double *tabx;        // x coordinate (embedded space) of output column out_left + i
	const BicubicTables *tables;
};

template <typename T>
static __device__ __forceinline__ T fetch(const UpsizeArgs &a, int ex, int ey, int z)
{
	// the embedded image (affine.c:520-532): original pixel (px, py) sits at (px + off, py + off)
	const int off = a.window_offset + 1;
	const int px = min(max(ex - off, 0), a.im_width - 1) - a.in_left;
	const int py = min(max(ey - off, 0), a.im_height - 1) - a.in_top;
	return ((const T *) (a.in + (long long) py * a.in_stride))[(long long) px * a.bands + z];
}

static __device__ __forceinline__ int unsigned_fixed_round(int v)
{
	return (v + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT;
}

static __device__ __forceinline__ int signed_fixed_round(int v)
{
	const int sign_of_v = 2 * (v >= 0) - 1;
	const int round_by = sign_of_v * (INTERPOLATE_SCALE >> 1);
	return (v + round_by) >> INTERPOLATE_SHIFT;
}

template <typename T>
struct UpTraits; // INT_PATH: fixed-point bilinear / bicubic; LO / HI: clip of the double bicubic
#define UP_TRAITS(TYPE, FIXED, SIGNED_, LO_, HI_) \
	template <> \
	struct UpTraits<TYPE> { \
		static constexpr bool fixed_bilinear = FIXED; \
		static constexpr bool is_signed = SIGNED_; \
		static __device__ __forceinline__ double lo() { return (double) (LO_); } \
		static __device__ __forceinline__ double hi() { return (double) (HI_); } \
		static constexpr int ilo = (int) (LO_); \
		static constexpr int ihi = (int) (HI_); \
	};
UP_TRAITS(unsigned char, true, false, 0, UCHAR_MAX)
UP_TRAITS(signed char, true, true, SCHAR_MIN, SCHAR_MAX)
UP_TRAITS(unsigned short, true, false, 0, USHRT_MAX)
UP_TRAITS(short, true, true, SHRT_MIN, SHRT_MAX)
UP_TRAITS(unsigned int, false, false, 0, INT_MAX)
UP_TRAITS(int, false, true, INT_MIN, INT_MAX)
UP_TRAITS(float, false, true, 0, 0)
UP_TRAITS(double, false, true, 0, 0)
#undef UP_TRAITS

// calculate_coefficients_catmull (templates.h:296-320), every operation rounded: what the
// no-table bicubic of double images evaluates per output pixel (bicubic.cpp:419-480)
static __device__ __forceinline__ void catmull_device(double c[4], const double x)
{
	const double cr1 = __dsub_rn(1.0, x);
	const double cr2 = __dmul_rn(-0.5, x);
	const double cr3 = __dmul_rn(cr1, cr2);
	const double cone = __dmul_rn(cr1, cr3);
	const double cfou = __dmul_rn(x, cr3);
	const double cr4 = __dsub_rn(cfou, cone);
	const double ctwo = __dadd_rn(__dsub_rn(cr1, cone), cr4);
	const double cthr = __dsub_rn(__dsub_rn(x, cfou), cr4);
	c[0] = cone;
	c[3] = cfou;
	c[1] = ctwo;
	c[2] = cthr;
}

// a * b + c * d + e * f + g * h, left to right, every operation rounded (cubic_float)
static __device__ __forceinline__ double dot4(double c0, double v0, double c1, double v1, double c2,
	double v2, double c3, double v3)
{
	double s = __dmul_rn(c0, v0);
	s = __dadd_rn(s, __dmul_rn(c1, v1));
	s = __dadd_rn(s, __dmul_rn(c2, v2));
	s = __dadd_rn(s, __dmul_rn(c3, v3));
	return s;
}

template <typename T, int INTERP>
__global__ void __launch_bounds__(256)
upsize_kernel(UpsizeArgs a)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.out_width)
		return;
	const double x = a.tabx[i];
	const int wo = a.window_offset;
	// affine.c:330-336: the clip rectangle, embedded coordinates
	const int ile = wo, ito = wo, iri = wo + a.im_width, ibo = wo + a.im_height;
	const int fx = vh::cvt_i32(floor(x));
	const int ix = vh::cvt_i32(x);

	for (int yy = blockIdx.y; yy < a.out_height; yy += gridDim.y) {
		// affine.c:343-365 with ib = ic = -0: y = id * oy, -= idy, += window_offset
		double y = __dmul_rn(a.id, (double) (a.out_top + yy));
		y = __dsub_rn(y, a.tidy);
		y = __dadd_rn(y, (double) wo);
		const int fy = vh::cvt_i32(floor(y));
		const int iy = vh::cvt_i32(y);
		T *q = (T *) (a.out + (long long) yy * a.out_stride) + (long long) i * a.bands;

		if (!(fx >= ile && fx <= iri && fy >= ito && fy <= ibo)) {
			for (int z = 0; z < a.bands; z++)
				q[z] = (T) 0;
			continue;
		}
		if (INTERP == 0) {
			for (int z = 0; z < a.bands; z++)
				q[z] = fetch<T>(a, ix, iy, z);
		}
		else if (INTERP == 1) {
			if (UpTraits<T>::fixed_bilinear) {
				const int X = vh::cvt_i32(__dmul_rn(__dsub_rn(x, (double) ix), (double) INTERPOLATE_SCALE));
				const int Y = vh::cvt_i32(__dmul_rn(__dsub_rn(y, (double) iy), (double) INTERPOLATE_SCALE));
				const int Yd = INTERPOLATE_SCALE - Y;
				const int c4 = (Y * X) >> INTERPOLATE_SHIFT;
				const int c2 = (Yd * X) >> INTERPOLATE_SHIFT;
				const int c3 = Y - c4;
				const int c1 = Yd - c2;
				for (int z = 0; z < a.bands; z++)
					q[z] = (T) ((c1 * (int) fetch<T>(a, ix, iy, z) + c2 * (int) fetch<T>(a, ix + 1, iy, z) +
									c3 * (int) fetch<T>(a, ix, iy + 1, z) + c4 * (int) fetch<T>(a, ix + 1, iy + 1, z) +
									(1 << INTERPOLATE_SHIFT) / 2) >>
						INTERPOLATE_SHIFT);
			}
			else {
				const double X = __dsub_rn(x, (double) ix);
				const double Y = __dsub_rn(y, (double) iy);
				const double Yd = __dsub_rn(1.0, Y);
				const double c4 = __dmul_rn(Y, X);
				const double c2 = __dmul_rn(Yd, X);
				const double c3 = __dsub_rn(Y, c4);
				const double c1 = __dsub_rn(Yd, c2);
				for (int z = 0; z < a.bands; z++)
					q[z] = vh::cvt_to<T>(dot4(c1, (double) fetch<T>(a, ix, iy, z), c2, (double) fetch<T>(a, ix + 1, iy, z), c3,
						(double) fetch<T>(a, ix, iy + 1, z), c4, (double) fetch<T>(a, ix + 1, iy + 1, z)));
			}
		}
		else {
			// bicubic.cpp:488-502: table index with round to nearest
			const int sx = vh::cvt_i32(__dmul_rn(__dmul_rn(x, (double) TRANSFORM_SCALE), 2.0));
			const int sy = vh::cvt_i32(__dmul_rn(__dmul_rn(y, (double) TRANSFORM_SCALE), 2.0));
			const int tx = ((sx & (TRANSFORM_SCALE * 2 - 1)) + 1) >> 1;
			const int ty = ((sy & (TRANSFORM_SCALE * 2 - 1)) + 1) >> 1;
			if (sizeof(T) == 1) {
				const int *cx = a.tables->mi[tx];
				const int *cy = a.tables->mi[ty];
				for (int z = 0; z < a.bands; z++) {
					int r[4];
#pragma unroll
					for (int j = 0; j < 4; j++) {
						const int s = cx[0] * (int) fetch<T>(a, ix - 1, iy - 1 + j, z) +
							cx[1] * (int) fetch<T>(a, ix, iy - 1 + j, z) +
							cx[2] * (int) fetch<T>(a, ix + 1, iy - 1 + j, z) +
							cx[3] * (int) fetch<T>(a, ix + 2, iy - 1 + j, z);
						r[j] = UpTraits<T>::is_signed ? signed_fixed_round(s) : unsigned_fixed_round(s);
					}
					const int s = cy[0] * r[0] + cy[1] * r[1] + cy[2] * r[2] + cy[3] * r[3];
					int v = UpTraits<T>::is_signed ? signed_fixed_round(s) : unsigned_fixed_round(s);
					v = min(max(v, UpTraits<T>::ilo), UpTraits<T>::ihi);
					q[z] = (T) v;
				}
			}
			else {
				const double *cx = a.tables->mf[tx];
				const double *cy = a.tables->mf[ty];
				double nx[4], ny[4];
				if (std::is_same<T, double>::value) {
					// double images: no table, the coefficients of the exact offsets (bicubic_notab)
					catmull_device(nx, __dsub_rn(x, (double) ix));
					catmull_device(ny, __dsub_rn(y, (double) iy));
					cx = nx;
					cy = ny;
				}
				for (int z = 0; z < a.bands; z++) {
					double r[4];
#pragma unroll
					for (int j = 0; j < 4; j++) {
						r[j] = dot4(cx[0], (double) fetch<T>(a, ix - 1, iy - 1 + j, z), cx[1],
							(double) fetch<T>(a, ix, iy - 1 + j, z), cx[2], (double) fetch<T>(a, ix + 1, iy - 1 + j, z),
							cx[3], (double) fetch<T>(a, ix + 2, iy - 1 + j, z));
						if (std::is_same<T, float>::value)
							r[j] = (double) (float) r[j]; // cubic_float<float> returns a float
					}
					double v = dot4(cy[0], r[0], cy[1], r[1], cy[2], r[2], cy[3], r[3]);
					if (!std::is_floating_point<T>::value) {
						// VIPS_CLIP(lo, v, hi), then the C conversion
						v = v < UpTraits<T>::lo() ? UpTraits<T>::lo() : (v > UpTraits<T>::hi() ? UpTraits<T>::hi() : v);
					}
					q[z] = vh::cvt_to<T>(v);
				}
			}
		}
	}
}

// Bicubic on uchar, the way a column is walked (round 6): upsize_kernel makes every output pixel from scratch --
// sixteen clamped byte fetches and four horizontal sums per band and pixel, 1.54 ms for 3276^2 x 3 -> 8190^2 (1.9 %
// of 8 TB/s).  But the four horizontally interpolated rows an output needs (bicubic.cpp:482-600: the horizontal sums
// rounded to the pel type, then the vertical one) are its upper neighbour's, moved down by at most one input row
// when the image is enlarged: a thread owns an output column, walks down a segment of rows and keeps the four
// rounded horizontal sums per band in registers; an input row is interpolated once per segment and column, the
// clamped column offsets and the horizontal coefficients once per thread.  The same integer arithmetic in the same
// order.
template <int B>
__global__ void __launch_bounds__(256)
upsize_bicubic_u8_walk(UpsizeArgs a, int seg)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.out_width)
		return;
	const double x = a.tabx[i];
	const int wo = a.window_offset;
	const int ile = wo, ito = wo, iri = wo + a.im_width, ibo = wo + a.im_height;
	const int fx = vh::cvt_i32(floor(x));
	const int ix = vh::cvt_i32(x);
	const bool x_in = fx >= ile && fx <= iri;
	const int sx = vh::cvt_i32(__dmul_rn(__dmul_rn(x, (double) TRANSFORM_SCALE), 2.0));
	const int tx = ((sx & (TRANSFORM_SCALE * 2 - 1)) + 1) >> 1;
	const int cx0 = a.tables->mi[tx][0], cx1 = a.tables->mi[tx][1], cx2 = a.tables->mi[tx][2], cx3 = a.tables->mi[tx][3];
	const int off = wo + 1; // embedded -> original coordinates (fetch())
	int col[4];
#pragma unroll
	for (int k = 0; k < 4; k++)
		col[k] = (min(max(ix - 1 + k - off, 0), a.im_width - 1) - a.in_left) * B;
	// the horizontal sums of embedded row ey, rounded (bicubic.cpp: bicubic_unsigned_int_tab's first stage)
	auto hrow = [&](int ey, int (&h)[B]) {
		const int py = min(max(ey - off, 0), a.im_height - 1) - a.in_top;
		const unsigned char *row = a.in + (long long) py * a.in_stride;
#pragma unroll
		for (int z = 0; z < B; z++)
			h[z] = unsigned_fixed_round(cx0 * (int) row[col[0] + z] + cx1 * (int) row[col[1] + z] + cx2 * (int) row[col[2] + z] +
				cx3 * (int) row[col[3] + z]);
	};
	const int y_first = blockIdx.y * seg, y_end = min(y_first + seg, a.out_height);
	int hr[4][B];       // hr[j] = row have + j
	int have = INT_MIN; // the first embedded row of the window (iy - 1), INT_MIN: nothing yet
	for (int yy = y_first; yy < y_end; yy++) {
		double y = __dmul_rn(a.id, (double) (a.out_top + yy));
		y = __dsub_rn(y, a.tidy);
		y = __dadd_rn(y, (double) wo);
		const int fy = vh::cvt_i32(floor(y));
		const int iy = vh::cvt_i32(y);
		unsigned char *q = a.out + (long long) yy * a.out_stride + (long long) i * B;
		if (!(x_in && fy >= ito && fy <= ibo)) {
#pragma unroll
			for (int z = 0; z < B; z++)
				q[z] = 0;
			continue;
		}
		// the window to rows iy - 1 .. iy + 2 (iy is the same for every thread of the launch's row: no divergence)
		const int want = iy - 1;
		if (have == INT_MIN || want < have || want > have + 3) {
#pragma unroll
			for (int j = 0; j < 4; j++)
				hrow(want + j, hr[j]);
		}
		else {
			for (int step = have; step < want; step++) { // (at most one turn when enlarging)
#pragma unroll
				for (int j = 0; j < 3; j++)
#pragma unroll
					for (int z = 0; z < B; z++)
						hr[j][z] = hr[j + 1][z];
				hrow(step + 4, hr[3]);
			}
		}
		have = want;
		const int sy = vh::cvt_i32(__dmul_rn(__dmul_rn(y, (double) TRANSFORM_SCALE), 2.0));
		const int ty = ((sy & (TRANSFORM_SCALE * 2 - 1)) + 1) >> 1;
		const int *cy = a.tables->mi[ty];
		const int cy0 = cy[0], cy1 = cy[1], cy2 = cy[2], cy3 = cy[3];
#pragma unroll
		for (int z = 0; z < B; z++) {
			int v = unsigned_fixed_round(cy0 * hr[0][z] + cy1 * hr[1][z] + cy2 * hr[2][z] + cy3 * hr[3][z]);
			v = min(max(v, 0), 255);
			q[z] = (unsigned char) v;
		}
	}
}

template <typename T>
__global__ void __launch_bounds__(256)
zoom_kernel(UpsizeArgs a, int xfac, int yfac)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.out_width)
		return;
	const int px = (a.out_left + i) / xfac - a.in_left;
	for (int yy = blockIdx.y; yy < a.out_height; yy += gridDim.y) {
		const int py = (a.out_top + yy) / yfac - a.in_top;
		const T *p = (const T *) (a.in + (long long) py * a.in_stride) + (long long) px * a.bands;
		T *q = (T *) (a.out + (long long) yy * a.out_stride) + (long long) i * a.bands;
		for (int z = 0; z < a.bands; z++)
			q[z] = p[z];
	}
}

// vips_subsample (conversion/subsample.c): out(x, y) = in(x * xfac, y * yfac)
template <typename T>
__global__ void __launch_bounds__(256)
subsample_kernel(UpsizeArgs a, int xfac, int yfac)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.out_width)
		return;
	const long long px = (long long) (a.out_left + i) * xfac - a.in_left;
	for (int yy = blockIdx.y; yy < a.out_height; yy += gridDim.y) {
		const long long py = (long long) (a.out_top + yy) * yfac - a.in_top;
		const T *p = (const T *) (a.in + py * a.in_stride) + px * a.bands;
		T *q = (T *) (a.out + (long long) yy * a.out_stride) + (long long) i * a.bands;
		for (int z = 0; z < a.bands; z++)
			q[z] = p[z];
	}
}

// ---------------------------------------------------------------------- host side

// templates.h:296-320
static void coefficients_catmull(double c[4], const double x)
{
	const double cr1 = 1. - x;
	const double cr2 = -.5 * x;
	const double cr3 = cr1 * cr2;
	const double cone = cr1 * cr3;
	const double cfou = x * cr3;
	const double cr4 = cfou - cone;
	const double ctwo = cr1 - cone + cr4;
	const double cthr = x - cfou - cr4;
	c[0] = cone;
	c[3] = cfou;
	c[1] = ctwo;
	c[2] = cthr;
}

static std::mutex &g_up_mutex = *new std::mutex;
static const BicubicTables *g_bicubic_by_device[64]; // device memory: one per device

static const BicubicTables *bicubic_tables()
{
	std::lock_guard<std::mutex> lock(g_up_mutex);
	const BicubicTables *&g_bicubic = g_bicubic_by_device[current_device() < 0 ? 0 : current_device() & 63];
	if (!g_bicubic) {
		BicubicTables t;
		// bicubic.cpp:624-633
		for (int x = 0; x < TRANSFORM_SCALE + 1; x++) {
			coefficients_catmull(t.mf[x], (float) x / TRANSFORM_SCALE);
			for (int i = 0; i < 4; i++)
				t.mi[x][i] = t.mf[x][i] * INTERPOLATE_SCALE;
		}
		g_bicubic = (const BicubicTables *) upload(&t, sizeof(t));
	}
	return g_bicubic;
}

// the per-column x coordinates, cached (a table upload synchronises the stream)
struct TabKey {
	int device;
	int out_left, out_width, window_offset, tile_width;
	double ia, tidx;
	bool operator==(const TabKey &o) const
	{
		return device == o.device && out_left == o.out_left && out_width == o.out_width && window_offset == o.window_offset &&
			tile_width == o.tile_width && memcmp(&ia, &o.ia, sizeof(double)) == 0 &&
			memcmp(&tidx, &o.tidx, sizeof(double)) == 0;
	}
};
typedef std::shared_ptr<double> TabPtr;
static std::list<std::pair<TabKey, TabPtr>> &g_tabs = *new std::list<std::pair<TabKey, TabPtr>>;

static TabPtr column_table(const TabKey &key, int full_width)
{
	{
		std::lock_guard<std::mutex> lock(g_up_mutex);
		for (auto it = g_tabs.begin(); it != g_tabs.end(); ++it)
			if (it->first == key) {
				g_tabs.splice(g_tabs.begin(), g_tabs, it);
				return g_tabs.front().second;
			}
	}
	// affine.c:343-392, ib * oy = -0: x = ia * le, -= idx, += window_offset, then += ia per pixel
	std::vector<double> tab(key.out_width);
	const int tw = key.tile_width > 0 ? key.tile_width : full_width;
	int col = (key.out_left / tw) * tw;
	while (col < key.out_left + key.out_width) {
		const int end = col + tw;
		double x = key.ia * (double) col;
		x -= key.tidx;
		x += key.window_offset;
		for (int xx = col; xx < end && xx < key.out_left + key.out_width; xx++) {
			if (xx >= key.out_left)
				tab[xx - key.out_left] = x;
			x += key.ia;
		}
		col = end;
	}
	double *d = (double *) upload(tab.data(), tab.size() * sizeof(double));
	if (!d)
		return TabPtr();
	TabPtr p(d, [](double *q) { vips_hip_free(q); });
	std::lock_guard<std::mutex> lock(g_up_mutex);
	g_tabs.emplace_front(key, p);
	while (g_tabs.size() > 32)
		g_tabs.pop_back();
	return p;
}

static int rows_grid(int gx, int height)
{
	int gy = 16384 / (gx > 0 ? gx : 1);
	gy = gy < 1 ? 1 : gy;
	return height < gy ? height : gy;
}

template <typename T>
static int launch_upsize(const UpsizeArgs &a, int interpolate)
{
	dim3 block(256, 1, 1);
	const int gx = (a.out_width + 255) / 256;
	dim3 grid(gx, rows_grid(gx, a.out_height), 1);
	if (interpolate == VIPS_HIP_INTERPOLATE_BICUBIC && std::is_same<T, unsigned char>::value && a.bands >= 1 && a.bands <= 4 &&
		!getenv("VIPS_HIP_NO_UPSIZE_WALK")) {
		// a column walked down segments of rows: enough blocks to fill the part, segments long enough for the four
		// rows a segment starts with to be a small share
		int segs = 2048 / gx;
		segs = segs < 1 ? 1 : segs;
		int seg = (a.out_height + segs - 1) / segs;
		seg = seg < 32 ? 32 : seg;
		dim3 wgrid(gx, (a.out_height + seg - 1) / seg, 1);
		Gate gate("upsize_bicubic_u8_walk");
		switch (a.bands) {
		case 1: hipLaunchKernelGGL(upsize_bicubic_u8_walk<1>, wgrid, block, 0, stream(), a, seg); break;
		case 2: hipLaunchKernelGGL(upsize_bicubic_u8_walk<2>, wgrid, block, 0, stream(), a, seg); break;
		case 3: hipLaunchKernelGGL(upsize_bicubic_u8_walk<3>, wgrid, block, 0, stream(), a, seg); break;
		default: hipLaunchKernelGGL(upsize_bicubic_u8_walk<4>, wgrid, block, 0, stream(), a, seg); break;
		}
		VH_CHECK(hipGetLastError());
		return 0;
	}
	Gate gate("upsize");
	if (interpolate == VIPS_HIP_INTERPOLATE_NEAREST)
		hipLaunchKernelGGL((upsize_kernel<T, 0>), grid, block, 0, stream(), a);
	else if (interpolate == VIPS_HIP_INTERPOLATE_BILINEAR)
		hipLaunchKernelGGL((upsize_kernel<T, 1>), grid, block, 0, stream(), a);
	else
		hipLaunchKernelGGL((upsize_kernel<T, 2>), grid, block, 0, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

template <typename T>
static int launch_subsample(const UpsizeArgs &a, int xfac, int yfac)
{
	dim3 block(256, 1, 1);
	const int gx = (a.out_width + 255) / 256;
	dim3 grid(gx, rows_grid(gx, a.out_height), 1);
	Gate gate("subsample");
	hipLaunchKernelGGL((subsample_kernel<T>), grid, block, 0, stream(), a, xfac, yfac);
	VH_CHECK(hipGetLastError());
	return 0;
}

template <typename T>
static int launch_zoom(const UpsizeArgs &a, int xfac, int yfac)
{
	dim3 block(256, 1, 1);
	const int gx = (a.out_width + 255) / 256;
	dim3 grid(gx, rows_grid(gx, a.out_height), 1);
	Gate gate("zoom");
	hipLaunchKernelGGL((zoom_kernel<T>), grid, block, 0, stream(), a, xfac, yfac);
	VH_CHECK(hipGetLastError());
	return 0;
}

static int fill_args(const char *domain, const VipsHipRegion *in, const VipsHipRegion *out, UpsizeArgs *a)
{
	if (ensure_init())
		return -1;
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->bands != out->bands || in->format != out->format) {
		error(domain, "input and output must have the same bands and format");
		return -1;
	}
	a->in = (const unsigned char *) in->data;
	a->out = (unsigned char *) out->data;
	a->in_stride = (long long) in->stride;
	a->out_stride = (long long) out->stride;
	a->in_left = in->left;
	a->in_top = in->top;
	a->im_width = in->im_width;
	a->im_height = in->im_height;
	a->out_left = out->left;
	a->out_top = out->top;
	a->out_width = out->width;
	a->out_height = out->height;
	a->bands = region_elems_per_pel(in);
	return 0;
}

#define UP_DISPATCH(FMT, CALL) \
	switch (FMT) { \
	case VIPS_HIP_FORMAT_UCHAR: return CALL(unsigned char); \
	case VIPS_HIP_FORMAT_CHAR: return CALL(signed char); \
	case VIPS_HIP_FORMAT_USHORT: return CALL(unsigned short); \
	case VIPS_HIP_FORMAT_SHORT: return CALL(short); \
	case VIPS_HIP_FORMAT_UINT: return CALL(unsigned int); \
	case VIPS_HIP_FORMAT_INT: return CALL(int); \
	case VIPS_HIP_FORMAT_FLOAT: \
	case VIPS_HIP_FORMAT_COMPLEX: return CALL(float); \
	case VIPS_HIP_FORMAT_DOUBLE: \
	case VIPS_HIP_FORMAT_DPCOMPLEX: return CALL(double); \
	default: break; \
	}

} // namespace vh

using namespace vh;

extern "C" {

int vips_hip_affine_out_size(int in_size, double scale)
{
	// transform.c:220-231: the corners 0 and scale * in_size, VIPS_ROUND_INT
	const double right = scale * in_size;
	return (int) (right > 0 ? right + 0.5 : right - 0.5);
}

int vips_hip_upsize_gen(const VipsHipRegion *in, const VipsHipRegion *out, double hscale, double vscale,
	double idx, double idy, int interpolate, int tile_width)
{
	const char *domain = "affine";
	UpsizeArgs a;
	if (fill_args(domain, in, out, &a))
		return -1;
	if (!(hscale > 0.0) || !(vscale > 0.0)) {
		error(domain, "scale factors should be > 0");
		return -1;
	}
	if (interpolate < VIPS_HIP_INTERPOLATE_NEAREST || interpolate > VIPS_HIP_INTERPOLATE_BICUBIC) {
		error(domain, "interpolator %d is outside the HIP path (nearest, bilinear, bicubic)", interpolate);
		return -1;
	}
	const int window_size = interpolate == VIPS_HIP_INTERPOLATE_NEAREST ? 1
		: interpolate == VIPS_HIP_INTERPOLATE_BILINEAR               ? 2
																	 : 4;
	// interpolate.c:150-167
	const int window_offset = window_size / 2 - 1 > 0 ? window_size / 2 - 1 : 0;
	// the window must hold every pixel the stencils of this rect touch, after clamping (with a
	// one-pixel margin for the accumulated rounding of the column coordinates)
	{
		const double tmp0 = 1.0 / (hscale * vscale);
		const double iax = tmp0 * vscale, iay = tmp0 * hscale;
		const int s_left = interpolate == VIPS_HIP_INTERPOLATE_BICUBIC ? 1 : 0;
		const int s_right = window_size - 1 - s_left;
		const int shift = window_offset + 1; // embedded -> original coordinates
		const int x_lo = (int) floor(iax * out->left - (idx - 1) + window_offset) - s_left - shift - 1;
		const int x_hi = (int) floor(iax * (out->left + out->width - 1) - (idx - 1) + window_offset) + s_right - shift + 1;
		const int y_lo = (int) floor(iay * out->top - (idy - 1) + window_offset) - s_left - shift - 1;
		const int y_hi = (int) floor(iay * (out->top + out->height - 1) - (idy - 1) + window_offset) + s_right - shift + 1;
		const int nx0 = x_lo < 0 ? 0 : (x_lo > in->im_width - 1 ? in->im_width - 1 : x_lo);
		const int ny0 = y_lo < 0 ? 0 : (y_lo > in->im_height - 1 ? in->im_height - 1 : y_lo);
		const int nx1 = x_hi > in->im_width - 1 ? in->im_width - 1 : (x_hi < 0 ? 0 : x_hi);
		const int ny1 = y_hi > in->im_height - 1 ? in->im_height - 1 : (y_hi < 0 ? 0 : y_hi);
		if (nx0 < in->left || ny0 < in->top || nx1 >= in->left + in->width || ny1 >= in->top + in->height) {
			error(domain, "input region too small");
			return -1;
		}
	}
	// transform.c:40-75 with b = c = 0
	const double det = hscale * vscale - 0.0 * 0.0;
	const double tmp = 1.0 / det;
	const double ia = tmp * vscale;
	a.id = tmp * hscale;
	a.tidy = idy - 1; // affine.c:538-539
	a.window_offset = window_offset;
	TabKey key = { current_device(), out->left, out->width, window_offset, tile_width, ia, idx - 1 };
	TabPtr tab = column_table(key, out->im_width);
	if (!tab)
		return -1;
	a.tabx = tab.get();
	a.tables = bicubic_tables();
	if (!a.tables)
		return -1;
#define CALL(T) launch_upsize<T>(a, interpolate)
	UP_DISPATCH(in->format, CALL)
#undef CALL
	error(domain, "band format %d is outside the HIP path", in->format);
	return -1;
}

int vips_hip_zoom_gen(const VipsHipRegion *in, const VipsHipRegion *out, int xfac, int yfac)
{
	const char *domain = "zoom";
	UpsizeArgs a;
	if (fill_args(domain, in, out, &a))
		return -1;
	if (xfac < 1 || yfac < 1) {
		error(domain, "zoom factors should be >= 1");
		return -1;
	}
	if (out->left / xfac < in->left || out->top / yfac < in->top ||
		(out->left + out->width - 1) / xfac >= in->left + in->width ||
		(out->top + out->height - 1) / yfac >= in->top + in->height) {
		error(domain, "input region too small");
		return -1;
	}
	a.window_offset = 0;
	a.id = a.tidy = 0.0;
	a.tabx = nullptr;
	a.tables = nullptr;
	// pixel replication is format-blind: move whole pels of 1, 2, 4 bytes per band
	const int es = format_sizeof(format_real(in->format));
	switch (es) {
	case 1: return launch_zoom<unsigned char>(a, xfac, yfac);
	case 2: return launch_zoom<unsigned short>(a, xfac, yfac);
	case 4: return launch_zoom<unsigned int>(a, xfac, yfac);
	case 8: return launch_zoom<unsigned long long>(a, xfac, yfac);
	default: break;
	}
	error(domain, "unsupported band format %d", in->format);
	return -1;
}

int vips_hip_subsample_gen(const VipsHipRegion *in, const VipsHipRegion *out, int xfac, int yfac)
{
	const char *domain = "subsample";
	UpsizeArgs a;
	if (fill_args(domain, in, out, &a))
		return -1;
	if (xfac < 1 || yfac < 1) {
		error(domain, "factors should be positive");
		return -1;
	}
	if ((long long) out->left * xfac < in->left || (long long) out->top * yfac < in->top ||
		(long long) (out->left + out->width - 1) * xfac >= in->left + in->width ||
		(long long) (out->top + out->height - 1) * yfac >= in->top + in->height) {
		error(domain, "input region too small");
		return -1;
	}
	a.window_offset = 0;
	a.id = a.tidy = 0.0;
	a.tabx = nullptr;
	a.tables = nullptr;
	const int es = format_sizeof(format_real(in->format));
	switch (es) {
	case 1: return launch_subsample<unsigned char>(a, xfac, yfac);
	case 2: return launch_subsample<unsigned short>(a, xfac, yfac);
	case 4: return launch_subsample<unsigned int>(a, xfac, yfac);
	case 8: return launch_subsample<unsigned long long>(a, xfac, yfac);
	default: break;
	}
	error(domain, "unsupported band format %d", in->format);
	return -1;
}

} // extern "C"
