// reduceh / reducev / shrinkh / shrinkv generates for gfx950.
//
// Two tiers:
//   * general kernels: every band format, any band count, any shrink -- one
//     thread per output element, taps fetched with edge clamping.  They define
//     correctness for the whole format matrix of reduceh.cpp:278-322 /
//     reducev.cpp:563-609 / shrinkh.c:175-232 / shrinkv.c:181-310.
//   * uchar fast kernels (reduce_u8.hip): LDS-staged, wave-coalesced kernels for
//     the 8-bit paths the reference vectorises with Highway
//     (reduceh_hwy.cpp:79, reducev_hwy.cpp:94, shrinkh_hwy.cpp:68,
//     shrinkv_hwy.cpp:90,133), selected here when the geometry allows.
//
// Arithmetic contracts (bit-exact for the integer formats):
//   unsigned ints  (sum + 2048) >> 12, clip           templates.h:152-157
//   signed ints    (sum + sign(sum)*2048) >> 12, clip templates.h:203-209
//   32-bit ints    the same in int64                   templates.h:537-545
//   float          double coefficients, double sum, separate mul and add in
//                  tap order, then (float)              templates.h:550-578
#include "resample.h"
#include "x80.h"
#include "reduce_u8.h"

#include <climits>
#include <cmath>

namespace vh {

// --------------------------------------------------------------- finalisers

template <typename T>
struct ReduceTraits;

template <>
struct ReduceTraits<unsigned char> {
	typedef int acc_t;
	typedef short coef_t;
	static __device__ __forceinline__ unsigned char fin(int s)
	{
		s = (s + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT;
		return (unsigned char) min(max(s, 0), 255);
	}
};

template <>
struct ReduceTraits<signed char> {
	typedef int acc_t;
	typedef short coef_t;
	static __device__ __forceinline__ signed char fin(int s)
	{
		const int round_by = (s >= 0 ? 1 : -1) * (INTERPOLATE_SCALE >> 1);
		s = (s + round_by) >> INTERPOLATE_SHIFT;
		return (signed char) min(max(s, (int) SCHAR_MIN), (int) SCHAR_MAX);
	}
};

template <>
struct ReduceTraits<unsigned short> {
	typedef int acc_t;
	typedef short coef_t;
	static __device__ __forceinline__ unsigned short fin(int s)
	{
		s = (s + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT;
		return (unsigned short) min(max(s, 0), (int) USHRT_MAX);
	}
};

template <>
struct ReduceTraits<short> {
	typedef int acc_t;
	typedef short coef_t;
	static __device__ __forceinline__ short fin(int s)
	{
		const int round_by = (s >= 0 ? 1 : -1) * (INTERPOLATE_SCALE >> 1);
		s = (s + round_by) >> INTERPOLATE_SHIFT;
		return (short) min(max(s, (int) SHRT_MIN), (int) SHRT_MAX);
	}
};

template <>
struct ReduceTraits<unsigned int> {
	typedef long long acc_t;
	typedef short coef_t;
	static __device__ __forceinline__ unsigned int fin(long long s)
	{
		s = (s + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT;
		return (unsigned int) min(max(s, 0ll), (long long) UINT_MAX);
	}
};

template <>
struct ReduceTraits<int> {
	typedef long long acc_t;
	typedef short coef_t;
	static __device__ __forceinline__ int fin(long long s)
	{
		const int round_by = (s >= 0 ? 1 : -1) * (INTERPOLATE_SCALE >> 1);
		s = (s + round_by) >> INTERPOLATE_SHIFT;
		return (int) min(max(s, (long long) INT_MIN), (long long) INT_MAX);
	}
};

template <>
struct ReduceTraits<float> {
	typedef double acc_t;
	typedef double coef_t;
	static __device__ __forceinline__ float fin(double s) { return (float) s; }
};

// Accumulate one tap.  For the float path the multiply and the add must stay
// two IEEE operations (the reference runs on baseline x86-64: no FMA).
template <typename ACC, typename COEF, typename T>
static __device__ __forceinline__ ACC mac(ACC sum, COEF c, T v)
{
	return sum + (ACC) c * (ACC) v;
}

template <>
__device__ __forceinline__ double mac<double, double, float>(double sum, double c, float v)
{
	return __dadd_rn(sum, __dmul_rn(c, (double) v));
}

static __device__ __forceinline__ int clampi(int v, int lo, int hi)
{
	return min(max(v, lo), hi);
}

// --------------------------------------------------------- general reducev

struct RegionArgs {
	const unsigned char *data;
	long long stride;
	int left, top, width, height, im_width, im_height;
};

static RegionArgs region_args(const VipsHipRegion *r)
{
	RegionArgs a;
	a.data = (const unsigned char *) r->data;
	a.stride = (long long) r->stride;
	a.left = r->left;
	a.top = r->top;
	a.width = r->width;
	a.height = r->height;
	a.im_width = r->im_width;
	a.im_height = r->im_height;
	return a;
}

// One thread per output element; blockIdx.y walks output rows.  All lanes of a
// block share the row, so the coefficient row and the clamped source rows are
// wave-uniform (scalar loads / scalar address math).
template <typename T>
__global__ void __launch_bounds__(256)
reducev_general(RegionArgs in, RegionArgs out, int ne, int epp /* elements per pel */,
	int n_point, const ReducePos *__restrict__ pos,
	const typename ReduceTraits<T>::coef_t *__restrict__ table, int gx, int band)
{
	typedef typename ReduceTraits<T>::acc_t ACC;
	// Neighbouring output rows share most of their input rows and an L2 is per XCD: block b
	// runs on XCD b % 8, so each XCD takes one contiguous band of output rows (1-D grid of
	// 8 * band * gx blocks; a row-major grid would make all 8 L2s fetch every input row).
	const int local = blockIdx.x / 8;
	const int yb = local / gx;
	const int e = (local - yb * gx) * blockDim.x + threadIdx.x;
	const int y = (blockIdx.x % 8) * band + yb;
	if (e >= ne || y >= out.height)
		return;
	// out.left is also the column in the input (reducev.cpp:536)
	const long long col = (long long) (out.left - in.left) * epp + e;
	{
		const ReducePos p = pos[y];
		const typename ReduceTraits<T>::coef_t *c = table + (size_t) p.phase * n_point;
		ACC sum = 0;
		for (int i = 0; i < n_point; i++) {
			const int row = clampi(p.first + i, 0, in.im_height - 1) - in.top;
			const T *src = (const T *) (in.data + row * in.stride);
			sum = mac<ACC>(sum, c[i], src[col]);
		}
		T *dst = (T *) (out.data + (long long) y * out.stride);
		dst[e] = ReduceTraits<T>::fin(sum);
	}
}

constexpr int HRU = 4; // rows per loop trip of the horizontal kernels

// grid.y for a kernel that takes HRU rows per trip
static inline int hru_grid(int gx, int height)
{
	const int groups = (height + HRU - 1) / HRU;
	int gy = 16384 / (gx > 0 ? gx : 1);
	gy = gy < 1 ? 1 : gy;
	return groups < gy ? groups : gy;
}

// One thread per output element (x, band).
template <typename T>
__global__ void __launch_bounds__(256)
reduceh_general(RegionArgs in, RegionArgs out, int epp, int n_point,
	const ReducePos *__restrict__ pos,
	const typename ReduceTraits<T>::coef_t *__restrict__ table)
{
	typedef typename ReduceTraits<T>::acc_t ACC;
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= out.width * epp)
		return;
	const int x = e / epp;
	const int b = e - x * epp;
	const ReducePos p = pos[x];
	const typename ReduceTraits<T>::coef_t *c = table + (size_t) p.phase * n_point;
	// HRU rows per trip: position, phase and the coefficient of a tap are shared by the rows,
	// and the rows' loads of one tap are independent (the tap loop is a dependent chain of
	// n_point loads per output otherwise: latency, not bandwidth, bounds these small images)
	for (int y0 = blockIdx.y * HRU; y0 < out.height; y0 += gridDim.y * HRU) {
		const T *src[HRU];
		ACC sum[HRU];
#pragma unroll
		for (int r = 0; r < HRU; r++) {
			// same row of the input (reduceh.cpp:238)
			const int y = min(y0 + r, out.height - 1);
			src[r] = (const T *) (in.data + (long long) (out.top + y - in.top) * in.stride);
			sum[r] = 0;
		}
		for (int i = 0; i < n_point; i++) {
			const int colx = clampi(p.first + i, 0, in.im_width - 1) - in.left;
			const long long off = (long long) colx * epp + b;
			const typename ReduceTraits<T>::coef_t ci = c[i];
#pragma unroll
			for (int r = 0; r < HRU; r++)
				sum[r] = mac<ACC>(sum[r], ci, src[r][off]);
		}
#pragma unroll
		for (int r = 0; r < HRU; r++)
			if (y0 + r < out.height)
				((T *) (out.data + (long long) (y0 + r) * out.stride))[e] = ReduceTraits<T>::fin(sum[r]);
	}
}

// uchar horizontal reduce through LDS: a block makes HB_PX output pixels of HB_ROWS rows.  The
// input bytes those pixels' taps touch are staged with coalesced dword loads (a strided gather
// straight from memory touches ~5 cache lines per wave instruction, and there is one per tap),
// then thread (pixel, band) sums its taps from LDS.  Same i32 arithmetic as reduceh_general.
constexpr int HB_PX = 64;     // output pixels per block (x 4 band lanes = 256 threads)
constexpr int HB_ROWS = 4;    // rows per block trip
constexpr int HB_SPAN = 4096; // staged bytes per row, at most

struct ReducehLdsArgs {
	const unsigned char *in; // row 0 of the rect in the window, byte 0 of the WINDOW's first pixel
	unsigned char *out;
	long long in_stride, out_stride;
	int in_left, im_width;   // window origin (pixels), image width
	int out_width, out_height, bands, n_point;
};

__global__ void __launch_bounds__(256)
reduceh_u8_lds(ReducehLdsArgs a, const ReducePos *__restrict__ pos, const short *__restrict__ table)
{
	__shared__ __attribute__((aligned(16))) unsigned int stage[HB_ROWS][HB_SPAN / 4 + 2];
	const int t = threadIdx.x;
	const int x0 = blockIdx.x * HB_PX;
	const int nx = min(HB_PX, a.out_width - x0);
	const int B = a.bands;
	// the pixel range the block's taps touch, clamped to the image (vips_embed COPY)
	const int p_lo = min(max(pos[x0].first, 0), a.im_width - 1);
	const int p_hi = min(max(pos[x0 + nx - 1].first + a.n_point - 1, 0), a.im_width - 1);
	const long long byte_lo = (long long) (p_lo - a.in_left) * B;
	const long long byte_hi = (long long) (p_hi - a.in_left + 1) * B; // exclusive
	const long long start_al = byte_lo & ~3LL;
	const int nfull = (int) ((byte_hi - start_al) >> 2); // whole dwords; the tail goes bytewise
	const int ntail = (int) ((byte_hi - start_al) & 3);  // (never read past the last needed byte)
	const int skew = (int) (byte_lo - start_al);

	const int px = t >> 2, band = t & 3;
	const bool mine = px < nx && band < B;
	ReducePos p = { 0, 0 };
	if (mine)
		p = pos[x0 + px];
	const short *c = table + (size_t) p.phase * a.n_point;

	for (int y0 = blockIdx.y * HB_ROWS; y0 < a.out_height; y0 += gridDim.y * HB_ROWS) {
		__syncthreads();
#pragma unroll
		for (int r = 0; r < HB_ROWS; r++) {
			const int y = min(y0 + r, a.out_height - 1);
			const unsigned int *src = reinterpret_cast<const unsigned int *>(a.in + (long long) y * a.in_stride + start_al);
			for (int i = t; i < nfull; i += 256)
				stage[r][i] = src[i];
			if (t < ntail)
				reinterpret_cast<unsigned char *>(stage[r])[4 * nfull + t] =
					reinterpret_cast<const unsigned char *>(src)[4 * nfull + t];
		}
		__syncthreads();
		if (mine) {
			int sum[HB_ROWS];
#pragma unroll
			for (int r = 0; r < HB_ROWS; r++)
				sum[r] = 0;
			for (int i = 0; i < a.n_point; i++) {
				const int s = min(max(p.first + i, 0), a.im_width - 1);
				const int off = (s - p_lo) * B + band + skew;
				const int ci = c[i];
#pragma unroll
				for (int r = 0; r < HB_ROWS; r++)
					sum[r] += ci * (int) reinterpret_cast<const unsigned char *>(stage[r])[off];
			}
#pragma unroll
			for (int r = 0; r < HB_ROWS; r++)
				if (y0 + r < a.out_height)
					a.out[(long long) (y0 + r) * a.out_stride + (long long) (x0 + px) * B + band] =
						ReduceTraits<unsigned char>::fin(sum[r]);
		}
	}
}

// 1 = handled, 0 = not this kernel's case (the general kernel runs), -1 = error
static int reduceh_u8_lds_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const ReducePos *pos, const short *table)
{
	if (getenv("VIPS_HIP_NO_REDUCEH_LDS"))
		return 0;
	const int B = in->bands;
	if (B < 1 || B > 4 || out->width < 1)
		return 0;
	// dword-aligned window rows; the staged span of a block must fit
	if (((uintptr_t) in->data & 3) || (in->stride & 3))
		return 0;
	if (((double) HB_PX * r->shrink + r->n_point + 2) * B + 8 > HB_SPAN)
		return 0;
	ReducehLdsArgs a;
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_left = in->left;
	a.im_width = in->im_width;
	a.out_width = out->width;
	a.out_height = out->height;
	a.bands = B;
	a.n_point = r->n_point;
	const int gx = (out->width + HB_PX - 1) / HB_PX;
	int gy = 16384 / gx;
	gy = gy < 1 ? 1 : gy;
	const int groups = (out->height + HB_ROWS - 1) / HB_ROWS;
	gy = groups < gy ? groups : gy;
	Gate gate("reduceh_u8_lds");
	hipLaunchKernelGGL(reduceh_u8_lds, dim3(gx, gy, 1), dim3(256, 1, 1), 0, stream(), a, pos, table);
	VH_CHECK(hipGetLastError());
	return 1;
}

template <typename T>
static int launch_reducev(const _VipsHipReduce *r, const VipsHipRegion *in,
	const VipsHipRegion *out, const ReducePos *pos, const void *table)
{
	const int epp = region_elems_per_pel(out);
	const int ne = out->width * epp;
	dim3 block(256, 1, 1);
	const int gx = (ne + 255) / 256;
	const int band = (out->height + 7) / 8;
	if ((long long) gx * band * 8 > 0x7fffffffLL) {
		error("reducev", "region too large for one launch");
		return -1;
	}
	dim3 grid(gx * band * 8, 1, 1);
	Gate gate("reducev_general");
	hipLaunchKernelGGL(reducev_general<T>, grid, block, 0, stream(),
		region_args(in), region_args(out), ne, epp, r->n_point, pos,
		(const typename ReduceTraits<T>::coef_t *) table, gx, band);
	VH_CHECK(hipGetLastError());
	return 0;
}

template <typename T>
static int launch_reduceh(const _VipsHipReduce *r, const VipsHipRegion *in,
	const VipsHipRegion *out, const ReducePos *pos, const void *table)
{
	const int epp = region_elems_per_pel(out);
	const int ne = out->width * epp;
	dim3 block(256, 1, 1);
	dim3 grid((ne + 255) / 256, hru_grid((ne + 255) / 256, out->height), 1);
	Gate gate("reduceh_general");
	hipLaunchKernelGGL(reduceh_general<T>, grid, block, 0, stream(),
		region_args(in), region_args(out), epp, r->n_point, pos,
		(const typename ReduceTraits<T>::coef_t *) table);
	VH_CHECK(hipGetLastError());
	return 0;
}

// Device-resident tables / position arrays, created on first use.
int reduce_tables(_VipsHipReduce *r, bool want_float, const void **table)
{
	std::lock_guard<std::mutex> lock(r->mutex);
	if (want_float) {
		if (!r->d_matrixf) {
			r->d_matrixf = (double *) upload(r->matrixf.data(),
				r->matrixf.size() * sizeof(double));
			if (!r->d_matrixf)
				return -1;
		}
		*table = r->d_matrixf;
	}
	else {
		if (!r->d_matrixs) {
			r->d_matrixs = (short *) upload(r->matrixs.data(),
				r->matrixs.size() * sizeof(short));
			if (!r->d_matrixs)
				return -1;
		}
		*table = r->d_matrixs;
	}
	return 0;
}

const ReducePos *reduce_device_positions(_VipsHipReduce *r, int start, int count, int tile)
{
	std::lock_guard<std::mutex> lock(r->mutex);
	auto key = std::make_tuple(start, count, tile);
	auto it = r->pos_cache.find(key);
	if (it != r->pos_cache.end())
		return it->second;
	std::vector<ReducePos> pos;
	reduce_positions(r, start, count, tile, pos);
	ReducePos *d = (ReducePos *) upload(pos.data(), pos.size() * sizeof(ReducePos));
	if (!d)
		return nullptr;
	// Bound the cache: tiled callers walk many distinct rects.
	// (only the position arrays -- keys with start >= 0 -- go: the map also holds the plan's long-lived blobs under
	// negative tags (the matrix-core operand tables, reduce_band's coefficient blocks of up to 64 MB, the ushort
	// schedules with their host records in blob_info), which other threads hold raw pointers to: ADVICE r5)
	if (r->pos_cache.size() > 256) {
		for (auto it2 = r->pos_cache.begin(); it2 != r->pos_cache.end();) {
			if (std::get<0>(it2->first) >= 0) {
				vips_hip_free(it2->second);
				it2 = r->pos_cache.erase(it2);
			}
			else
				++it2;
		}
	}
	r->pos_cache[key] = d;
	return d;
}

// ------------------------------------------------------------ double images
//
// reduceh_notab / reducev_notab (reduceh.cpp:196-213, reducev.cpp:497-515; reduce_sum<T, IT>,
// templates.h:533-554): sum += c[i] * in[i] in LONG DOUBLE, the coefficients a long double mask
// made for the output's exact position (resample_host.cpp reduce_notab_masks), the result cast
// to double.  The x87 extended format on the device: a value is (sign, exponent, 64-bit mantissa
// with its leading one explicit); a product is the 128-bit integer product rounded to 64 bits, a
// sum is aligned in 128 bits with the shifted-out bits jammed into the last one, renormalised and
// rounded to 64 bits -- round to nearest even, once per operation, like fmul / faddp.  The
// extended exponent range (15 bits) cannot be left by sums of doubles times coefficients of
// magnitude <= 1.2, so no overflow / underflow handling is needed in between.
// One thread per output element.  VERTICAL: the mask belongs to the output row and the taps walk
// down a column; else it belongs to the output column and the taps walk along the row.
template <bool VERTICAL>
__global__ void __launch_bounds__(256)
reduce_notab_f64(RegionArgs in, RegionArgs out, int epp, int n_point, const ReducePos *__restrict__ pos,
	const ReduceTap80 *__restrict__ taps)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;
	if (e >= out.width * epp || y >= out.height)
		return;
	const int x = e / epp;
	const int b = e - x * epp;
	const int which = VERTICAL ? y : x;
	const ReducePos p = pos[which];
	const ReduceTap80 *c = taps + (size_t) which * n_point;
	X80 sum = { 0, 0, 0 };
	double plain = 0.0; // the sum in double: what an inf or NaN operand turns the result into
	bool finite = true;
	for (int i = 0; i < n_point; i++) {
		double v;
		if (VERTICAL) {
			const int row = clampi(p.first + i, 0, in.im_height - 1) - in.top;
			v = ((const double *) (in.data + (long long) row * in.stride))[(long long) (out.left - in.left) * epp + e];
		}
		else {
			const int colx = clampi(p.first + i, 0, in.im_width - 1) - in.left;
			v = ((const double *) (in.data + (long long) (out.top + y - in.top) * in.stride))[(long long) colx * epp + b];
		}
		X80 ci = { c[i].mant, c[i].exp, c[i].sign };
		if (!(fabs(v) <= 1.7976931348623157e308)) // inf or NaN
			finite = false;
		if (finite)
			sum = x80_add(sum, x80_mul(ci, x80_from_double(v)));
		plain += x80_to_double(ci) * v;
	}
	((double *) (out.data + (long long) y * out.stride))[e] = finite ? x80_to_double(sum) : plain;
}

static int launch_reduce_notab(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile,
	bool vertical)
{
	if (out->height > 65535) {
		error(vertical ? "reducev" : "reduceh", "double images taller than 65535 rows per call are not supported");
		return -1;
	}
	std::vector<ReducePos> pos;
	std::vector<ReduceTap80> taps;
	reduce_notab_masks(r, vertical ? out->top : out->left, vertical ? out->height : out->width, tile, pos, taps);
	ReducePos *d_pos = (ReducePos *) upload(pos.data(), pos.size() * sizeof(ReducePos));
	ReduceTap80 *d_taps = (ReduceTap80 *) upload(taps.data(), taps.size() * sizeof(ReduceTap80));
	int result = -1;
	if (d_pos && d_taps) {
		const int epp = region_elems_per_pel(out);
		const int ne = out->width * epp;
		dim3 block(256, 1, 1), grid((ne + 255) / 256, out->height, 1);
		Gate gate(vertical ? "reducev_notab_f64" : "reduceh_notab_f64");
		if (vertical)
			hipLaunchKernelGGL(reduce_notab_f64<true>, grid, block, 0, stream(), region_args(in), region_args(out), epp,
				r->n_point, d_pos, d_taps);
		else
			hipLaunchKernelGGL(reduce_notab_f64<false>, grid, block, 0, stream(), region_args(in), region_args(out), epp,
				r->n_point, d_pos, d_taps);
		result = hipGetLastError() == hipSuccess ? 0 : hip_failed(hipErrorUnknown, "reduce_notab_f64 launch");
	}
	// (the pool orders reuse of these blocks behind the kernel: same thread, same stream)
	vips_hip_free(d_pos);
	vips_hip_free(d_taps);
	return result;
}

static int reduce_check(const char *domain, const _VipsHipReduce *r, const VipsHipRegion *in,
	const VipsHipRegion *out, bool vertical)
{
	if (!r) {
		error(domain, "null reduce");
		return -1;
	}
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->bands != out->bands || in->format != out->format) {
		error(domain, "input and output must have the same bands and format");
		return -1;
	}
	int in_size = vertical ? in->im_height : in->im_width;
	int out_size = vertical ? out->im_height : out->im_width;
	if (in_size != r->in_size || out_size != r->out_size) {
		error(domain, "region does not belong to an image of the size this reduce was built for");
		return -1;
	}
	// Does the window cover what the generate needs?
	int need0, needn;
	if (vertical) {
		vips_hip_reducev_need(r, out->top, out->height, &need0, &needn);
		if (need0 < in->top || need0 + needn > in->top + in->height ||
			out->left < in->left || out->left + out->width > in->left + in->width) {
			error(domain, "input region too small: need rows %d..%d", need0, need0 + needn);
			return -1;
		}
	}
	else {
		vips_hip_reduceh_need(r, out->left, out->width, &need0, &needn);
		if (need0 < in->left || need0 + needn > in->left + in->width ||
			out->top < in->top || out->top + out->height > in->top + in->height) {
			error(domain, "input region too small: need columns %d..%d", need0, need0 + needn);
			return -1;
		}
	}
	return 0;
}

static int reduce_gen(const char *domain, const VipsHipReduce *reduce, const VipsHipRegion *in,
	const VipsHipRegion *out, int tile, bool vertical)
{
	if (ensure_init())
		return -1;
	_VipsHipReduce *r = const_cast<_VipsHipReduce *>(reduce);
	if (reduce_check(domain, r, in, out, vertical))
		return -1;
	if (plan_device(domain, &r->device))
		return -1;

	const int fmt = format_real(out->format);
	const bool want_float = fmt == VIPS_HIP_FORMAT_FLOAT;
	if (fmt == VIPS_HIP_FORMAT_DOUBLE) // reduceh_notab / reducev_notab: no table, long double
		return launch_reduce_notab(r, in, out, tile, vertical);
	const void *table;
	if (reduce_tables(r, want_float, &table))
		return -1;
	const ReducePos *pos = vertical
		? reduce_device_positions(r, out->top, out->height, tile)
		: reduce_device_positions(r, out->left, out->width, tile);
	if (!pos)
		return -1;

	if (fmt == VIPS_HIP_FORMAT_UCHAR) {
		int done = 0;
		if (vertical)
			done = reducev_u8_try(r, in, out, pos, (const short *) table, tile);
		else {
			// for an integer factor of 4 / 8, packed bytes on the vector ALU (reduceh_u8.hip: coalesced staging -- on a
			// 201 MB input 0.062 ms against 0.075 for the matrix-core kernel, whose lanes read a row each); any other
			// factor: the matrix cores (reduce_band.hip).  $VIPS_HIP_REDUCEH_FIRST=band for the other order
			const char *first = getenv("VIPS_HIP_REDUCEH_FIRST");
			const bool packed_first = !(first && !strcmp(first, "band"));
			// three bands, a factor of 8: the fused reduce's horizontal walk on rows staged with whole-line loads (round 6)
			done = reduceh_u8x3_try(r, in, out, tile);
			if (!done && packed_first)
				done = reduceh_u8p_try(r, in, out, tile);
			if (!done)
				done = reduceh_band_try(r, in, out, tile);
			if (!done && !packed_first)
				done = reduceh_u8p_try(r, in, out, tile);
		}
		if (!done && !vertical)
			done = reduceh_u8_lds_try(r, in, out, pos, (const short *) table);
		if (done < 0)
			return -1;
		if (done > 0)
			return 0;
	}

	if (fmt == VIPS_HIP_FORMAT_USHORT && !format_iscomplex(out->format)) {
		// the matrix cores first (reduce_band.hip: the low and the high bytes as two products)
		int done = vertical ? reducev_band_try(r, in, out, tile) : reduceh_band_try(r, in, out, tile);
		if (!done)
			done = vertical ? reducev16_stream_try(r, in, out, tile) : reduceh16_stream_try(r, in, out, pos, (const short *) table);
		if (done < 0)
			return -1;
		if (done > 0)
			return 0;
	}

	if (fmt == VIPS_HIP_FORMAT_FLOAT && !format_iscomplex(out->format)) {
		// one phase and an integer step: the streaming / staged kernels (resample_f32.hip)
		std::vector<ReducePos> host_pos;
		if (vertical)
			reduce_positions(r, out->top, out->height, tile, host_pos);
		else
			reduce_positions(r, out->left, out->width, tile, host_pos);
		const int done = vertical ? reducev_f32_stream_try(r, in, out, host_pos, (const double *) table)
								  : reduceh_f32_lds_try(r, in, out, host_pos, (const double *) table);
		if (done < 0)
			return -1;
		if (done > 0)
			return 0;
	}

#define DISPATCH(FN) \
	switch (fmt) { \
	case VIPS_HIP_FORMAT_UCHAR: return FN<unsigned char>(r, in, out, pos, table); \
	case VIPS_HIP_FORMAT_CHAR: return FN<signed char>(r, in, out, pos, table); \
	case VIPS_HIP_FORMAT_USHORT: return FN<unsigned short>(r, in, out, pos, table); \
	case VIPS_HIP_FORMAT_SHORT: return FN<short>(r, in, out, pos, table); \
	case VIPS_HIP_FORMAT_UINT: return FN<unsigned int>(r, in, out, pos, table); \
	case VIPS_HIP_FORMAT_INT: return FN<int>(r, in, out, pos, table); \
	case VIPS_HIP_FORMAT_FLOAT: return FN<float>(r, in, out, pos, table); \
	default: break; \
	}
	if (vertical) {
		DISPATCH(launch_reducev)
	}
	else {
		DISPATCH(launch_reduceh)
	}
#undef DISPATCH
	error(domain, "unsupported band format %d", out->format);
	return -1;
}

// ------------------------------------------------------------------ shrink

template <typename T>
struct ShrinkTraits;

// shrinkh.c:78-92 / shrinkv.c:218-228: ((sum + amend) * multiplier) >> 24 in
// unsigned 32-bit arithmetic.
template <>
struct ShrinkTraits<unsigned char> {
	typedef int acc_t;
	static __device__ __forceinline__ unsigned char fin(int sum, int shrink, unsigned int mult8,
		unsigned long long mult16, double inv)
	{
		return (unsigned char) (((unsigned int) sum * mult8) >> 24);
	}
};
// shrinkh.c:98-112 / shrinkv.c:233-244
template <>
struct ShrinkTraits<unsigned short> {
	typedef int acc_t;
	static __device__ __forceinline__ unsigned short fin(int sum, int shrink, unsigned int mult8,
		unsigned long long mult16, double inv)
	{
		return (unsigned short) (((unsigned long long) (long long) sum * mult16) >> 32);
	}
};
// shrinkh.c:117-130 / shrinkv.c:248-257: C division, truncating toward zero
#define SHRINK_INT_TRAITS(TYPE, ACC) \
	template <> \
	struct ShrinkTraits<TYPE> { \
		typedef ACC acc_t; \
		static __device__ __forceinline__ TYPE fin(ACC sum, int shrink, unsigned int mult8, \
			unsigned long long mult16, double inv) \
		{ \
			return (TYPE) (sum / shrink); \
		} \
	};
SHRINK_INT_TRAITS(signed char, int)
SHRINK_INT_TRAITS(short, int)
SHRINK_INT_TRAITS(unsigned int, long long)
SHRINK_INT_TRAITS(int, long long)
// shrinkh.c:136-151 / shrinkv.c:261-268: double sum * (1.0 / shrink)
#define SHRINK_FLOAT_TRAITS(TYPE) \
	template <> \
	struct ShrinkTraits<TYPE> { \
		typedef double acc_t; \
		static __device__ __forceinline__ TYPE fin(double sum, int shrink, unsigned int mult8, \
			unsigned long long mult16, double inv) \
		{ \
			return (TYPE) __dmul_rn(sum, inv); \
		} \
	};
SHRINK_FLOAT_TRAITS(float)
SHRINK_FLOAT_TRAITS(double)

template <typename ACC, typename T>
static __device__ __forceinline__ ACC shrink_add(ACC sum, T v)
{
	return sum + (ACC) v;
}
template <>
__device__ __forceinline__ double shrink_add<double, float>(double sum, float v)
{
	return __dadd_rn(sum, (double) v);
}
template <>
__device__ __forceinline__ double shrink_add<double, double>(double sum, double v)
{
	return __dadd_rn(sum, v);
}

template <typename T>
static __device__ __forceinline__ typename ShrinkTraits<T>::acc_t shrink_seed(int amend)
{
	return (typename ShrinkTraits<T>::acc_t) amend;
}
template <>
__device__ __forceinline__ double shrink_seed<float>(int amend) { return 0.0; }
template <>
__device__ __forceinline__ double shrink_seed<double>(int amend) { return 0.0; }

template <typename T>
__global__ void __launch_bounds__(256)
shrinkh_general(RegionArgs in, RegionArgs out, int epp, int hshrink, unsigned int mult8,
	unsigned long long mult16, double inv)
{
	typedef typename ShrinkTraits<T>::acc_t ACC;
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= out.width * epp)
		return;
	const int x = e / epp;
	const int b = e - x * epp;
	const int x0 = (out.left + x) * hshrink;
	for (int y0 = blockIdx.y * HRU; y0 < out.height; y0 += gridDim.y * HRU) {
		const T *src[HRU];
		ACC sum[HRU];
#pragma unroll
		for (int r = 0; r < HRU; r++) {
			const int y = min(y0 + r, out.height - 1);
			src[r] = (const T *) (in.data + (long long) (out.top + y - in.top) * in.stride);
			sum[r] = shrink_seed<T>(hshrink / 2);
		}
		for (int i = 0; i < hshrink; i++) {
			const int colx = min(x0 + i, in.im_width - 1) - in.left;
			const long long off = (long long) colx * epp + b;
#pragma unroll
			for (int r = 0; r < HRU; r++)
				sum[r] = shrink_add<ACC, T>(sum[r], src[r][off]);
		}
#pragma unroll
		for (int r = 0; r < HRU; r++)
			if (y0 + r < out.height)
				((T *) (out.data + (long long) (y0 + r) * out.stride))[e] =
					ShrinkTraits<T>::fin(sum[r], hshrink, mult8, mult16, inv);
	}
}

template <typename T>
__global__ void __launch_bounds__(256)
shrinkv_general(RegionArgs in, RegionArgs out, int ne, int epp, int vshrink, unsigned int mult8,
	unsigned long long mult16, double inv)
{
	typedef typename ShrinkTraits<T>::acc_t ACC;
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= ne)
		return;
	const long long col = (long long) (out.left - in.left) * epp + e;
	for (int y = blockIdx.y; y < out.height; y += gridDim.y) {
		const int y0 = (out.top + y) * vshrink;
		// shrinkv.c:170: sums start at 0 and `amend` is added at write time
		// (:218-257); for the integer formats that is the same number.
		ACC sum = shrink_seed<T>(vshrink / 2);
		for (int i = 0; i < vshrink; i++) {
			const int row = min(y0 + i, in.im_height - 1) - in.top;
			const T *src = (const T *) (in.data + row * in.stride);
			sum = shrink_add<ACC, T>(sum, src[col]);
		}
		T *dst = (T *) (out.data + (long long) y * out.stride);
		dst[e] = ShrinkTraits<T>::fin(sum, vshrink, mult8, mult16, inv);
	}
}

template <typename T>
static int launch_shrinkh(int hshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	const int epp = region_elems_per_pel(out);
	const int ne = out->width * epp;
	const unsigned int mult8 = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) hshrink));
	const unsigned long long mult16 = ((1ULL << 32) + hshrink - 1) / hshrink;
	dim3 block(256, 1, 1);
	dim3 grid((ne + 255) / 256, hru_grid((ne + 255) / 256, out->height), 1);
	Gate gate("shrinkh_general");
	hipLaunchKernelGGL(shrinkh_general<T>, grid, block, 0, stream(), region_args(in),
		region_args(out), epp, hshrink, mult8, mult16, 1.0 / hshrink);
	VH_CHECK(hipGetLastError());
	return 0;
}

template <typename T>
static int launch_shrinkv(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	const int epp = region_elems_per_pel(out);
	const int ne = out->width * epp;
	const unsigned int mult8 = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) vshrink));
	const unsigned long long mult16 = ((1ULL << 32) + vshrink - 1) / vshrink;
	dim3 block(256, 1, 1);
	dim3 grid((ne + 255) / 256, out->height < 32768 ? out->height : 32768, 1);
	Gate gate("shrinkv_general");
	hipLaunchKernelGGL(shrinkv_general<T>, grid, block, 0, stream(), region_args(in),
		region_args(out), ne, epp, vshrink, mult8, mult16, 1.0 / vshrink);
	VH_CHECK(hipGetLastError());
	return 0;
}

static int shrink_gen(const char *domain, int shrink, const VipsHipRegion *in,
	const VipsHipRegion *out, bool vertical)
{
	if (ensure_init())
		return -1;
	if (shrink < 1) {
		error(domain, "shrink factors should be >= 1");
		return -1;
	}
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->bands != out->bands || in->format != out->format) {
		error(domain, "input and output must have the same bands and format");
		return -1;
	}
	// window check: shrinkh.c:262-268 / shrinkv.c:347-358, clipped to the image
	if (vertical) {
		long long lo = (long long) out->top * shrink;
		long long hi = (long long) (out->top + out->height) * shrink;
		if (hi > in->im_height)
			hi = in->im_height;
		if (lo > in->im_height - 1)
			lo = in->im_height - 1;
		if (lo < in->top || hi > in->top + in->height || out->left < in->left ||
			out->left + out->width > in->left + in->width) {
			error(domain, "input region too small");
			return -1;
		}
	}
	else {
		long long lo = (long long) out->left * shrink;
		long long hi = (long long) (out->left + out->width) * shrink;
		if (hi > in->im_width)
			hi = in->im_width;
		if (lo > in->im_width - 1)
			lo = in->im_width - 1;
		if (lo < in->left || hi > in->left + in->width || out->top < in->top ||
			out->top + out->height > in->top + in->height) {
			error(domain, "input region too small");
			return -1;
		}
	}

	const int fmt = format_real(out->format);
	if (fmt == VIPS_HIP_FORMAT_UCHAR && vertical) {
		int done = shrinkv_u8_try(shrink, in, out);
		if (done < 0)
			return -1;
		if (done > 0)
			return 0;
	}
	if (fmt == VIPS_HIP_FORMAT_UCHAR && !vertical) {
		int done = shrinkh_u8_stream_try(shrink, in, out);
		if (done < 0)
			return -1;
		if (done > 0)
			return 0;
	}
	if (fmt == VIPS_HIP_FORMAT_FLOAT && vertical && !format_iscomplex(out->format)) { // (resample_f32.hip)
		const int done = shrinkv_f32_stream_try(shrink, in, out);
		if (done < 0)
			return -1;
		if (done > 0)
			return 0;
	}
	if (fmt == VIPS_HIP_FORMAT_USHORT) {
		int done = vertical ? shrinkv16_stream_try(shrink, in, out) : shrinkh16_stream_try(shrink, in, out);
		if (done < 0)
			return -1;
		if (done > 0)
			return 0;
	}
#define DISPATCH(FN) \
	switch (fmt) { \
	case VIPS_HIP_FORMAT_UCHAR: return FN<unsigned char>(shrink, in, out); \
	case VIPS_HIP_FORMAT_CHAR: return FN<signed char>(shrink, in, out); \
	case VIPS_HIP_FORMAT_USHORT: return FN<unsigned short>(shrink, in, out); \
	case VIPS_HIP_FORMAT_SHORT: return FN<short>(shrink, in, out); \
	case VIPS_HIP_FORMAT_UINT: return FN<unsigned int>(shrink, in, out); \
	case VIPS_HIP_FORMAT_INT: return FN<int>(shrink, in, out); \
	case VIPS_HIP_FORMAT_FLOAT: return FN<float>(shrink, in, out); \
	case VIPS_HIP_FORMAT_DOUBLE: return FN<double>(shrink, in, out); \
	default: break; \
	}
	if (vertical) {
		DISPATCH(launch_shrinkv)
	}
	else {
		DISPATCH(launch_shrinkh)
	}
#undef DISPATCH
	error(domain, "unsupported band format %d", out->format);
	return -1;
}

} // namespace vh

using namespace vh;

extern "C" {

int vips_hip_reduceh_gen(const VipsHipReduce *reduce, const VipsHipRegion *in,
	const VipsHipRegion *out)
{
	return reduce_gen("reduceh", reduce, in, out, 0, false);
}

int vips_hip_reducev_gen(const VipsHipReduce *reduce, const VipsHipRegion *in,
	const VipsHipRegion *out)
{
	return reduce_gen("reducev", reduce, in, out, 0, true);
}

int vips_hip_reduceh_gen_tiled(const VipsHipReduce *reduce, const VipsHipRegion *in,
	const VipsHipRegion *out, int tile)
{
	return reduce_gen("reduceh", reduce, in, out, tile, false);
}

int vips_hip_reducev_gen_tiled(const VipsHipReduce *reduce, const VipsHipRegion *in,
	const VipsHipRegion *out, int tile)
{
	return reduce_gen("reducev", reduce, in, out, tile, true);
}

int vips_hip_shrinkh_gen(int hshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return shrink_gen("shrinkh", hshrink, in, out, false);
}

int vips_hip_shrinkv_gen(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return shrink_gen("shrinkv", vshrink, in, out, true);
}

} // extern "C"
