// vips_reducev and vips_shrinkv of FLOAT images as streams (round 6; north_star: "within 1 ULP for float reduce").
//
// The reference sums float pixels in double, one rounded multiply and one rounded add per tap, in tap order
// (resample/templates.h:183-194 reduce_sum<float, double>: sum += c[i] * in[i]; reducev.cpp:418-459); the box shrink
// adds the rows of a box in double, in row order, and multiplies by 1.0 / vshrink (shrinkv.c:261-268).  The general
// kernels (resample.hip) do exactly that with one thread per output element and one 4-byte load per tap: every input
// row is read n / shrink times through the L2 and a wave's load is 256 bytes -- 8192 x 8192 x 3 float reduced by 8:
// 1.06 ms for the vertical pass (9.5 % of 8 TB/s), the box shrink by 4 0.29 ms (34 %).
//
// reducev_f32_stream<S, N>: an integer shrink S with ONE coefficient phase (what vips_reduce makes of a size the
// factor divides; N = n_point taps, D = ceil(N / S) output rows in flight).  A lane owns two neighbouring floats of
// the row (a wave reads 512 contiguous bytes) and walks down a segment of output rows; input row S G + i ("group" G,
// row i) is tap S d + i of output row G - d, d = 0 .. D - 1: D accumulator pairs per lane rotate statically through
// an unrolled body of D groups, so an output's taps arrive -- and are added -- in the reference's order, every row
// is read once per segment (the D - 1 groups a segment shares with the next one twice), and the coefficients are
// broadcast reads of a small LDS table.  Same operations, same order, same bits.
//
// shrinkv_f32_stream: a lane owns four floats (one 16-byte load per row), the rows of a box one after the other.
#include "resample.h"
#include "kernel_stmt.h"

#include <cstdlib>

namespace vh {

struct RfArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int im_height, in_top; // rows clamp to the image (vips_embed COPY); the window's first row
	int lanes;             // lanes that hold data in a row (elements / 2 for the reduce, / 4 for the shrink)
	int first0;            // input row of tap 0 of the region's output row 0 (may be negative)
	int out_height, seg;   // output rows of the region, per segment
	int shrink;
	double inv;
	const double *coef; // device: the coefficient row (n_point doubles)
};

template <int S, int N>
__global__ void __launch_bounds__(256)
reducev_f32_stream(RfArgs a)
{
	constexpr int D = (N + S - 1) / S;
	constexpr int LAST = N - 1 - S * (D - 1); // the row of a group an output's last tap is
	// the coefficients in LDS: every lane reads the same one (a broadcast); as scalar operands from the kernarg
	// segment the compiler kept all N of them live and spilled 379 scalar registers into vector lanes
	__shared__ double s_c[N];
	if (threadIdx.x < N)
		s_c[threadIdx.x] = a.coef[threadIdx.x];
	__syncthreads();
	const int lane = blockIdx.x * 256 + threadIdx.x;
	if (lane >= a.lanes)
		return;
	const int j0 = blockIdx.y * a.seg, j1 = min(j0 + a.seg, a.out_height);
	if (j0 >= j1)
		return;
	const unsigned char *col = a.in + 8LL * lane;
	unsigned char *dst = a.out + 8LL * lane;
	auto load = [&](int G, int i) -> float2 {
		const int row = min(max(a.first0 + S * G + i, 0), a.im_height - 1) - a.in_top;
		return *reinterpret_cast<const float2 *>(col + (long long) row * a.in_stride);
	};
	double acc[D][2];
#pragma unroll
	for (int s = 0; s < D; s++)
		acc[s][0] = acc[s][1] = 0.0;
	float2 cur[S];
#pragma unroll
	for (int i = 0; i < S; i++)
		cur[i] = load(j0, i);
	const int gend = j1 + D - 1; // groups j0 .. gend - 1
	// (the outputs before j0 that a segment's first groups "contribute" to are zeroed where they would retire,
	// one group before the slot's real owner starts; those from j1 on never retire)
	for (int gb = j0; gb < gend; gb += D) {
#pragma unroll
		for (int r = 0; r < D; r++) {
			const int G = gb + r;
			if (G >= gend)
				break;
			float2 nxt[S];
			const int Gn = G + 1 < gend ? G + 1 : G; // (the last group loads itself again: a load on every path)
#pragma unroll
			for (int i = 0; i < S; i++)
				nxt[i] = load(Gn, i);
#pragma unroll
			for (int i = 0; i < S; i++) {
				const double x = (double) cur[i].x, y = (double) cur[i].y;
#pragma unroll
				for (int d = 0; d < D; d++) {
					if (S * d + i < N) {
						const int slot = (r - d + D) % D;
						// (an index the compiler cannot see through: it would otherwise keep all N coefficients in
						// registers across the loop -- 98 of them for 49 taps, one wave a SIMD)
						int kk = S * d + i;
						VH_VECTOR1(kk);
						const double c = s_c[kk];
						acc[slot][0] = __dadd_rn(acc[slot][0], __dmul_rn(c, x));
						acc[slot][1] = __dadd_rn(acc[slot][1], __dmul_rn(c, y));
					}
				}
				if (i == LAST) {
					const int slot = (r + 1) % D; // = (r - (D - 1)) mod D
					const int j = G - (D - 1);
					if (j >= j0)
						*reinterpret_cast<float2 *>(dst + (long long) j * a.out_stride) =
							make_float2((float) acc[slot][0], (float) acc[slot][1]);
					acc[slot][0] = acc[slot][1] = 0.0;
				}
			}
#pragma unroll
			for (int i = 0; i < S; i++)
				cur[i] = nxt[i];
		}
	}
}

__global__ void __launch_bounds__(256)
shrinkv_f32_stream(RfArgs a)
{
	const int lane = blockIdx.x * 256 + threadIdx.x;
	if (lane >= a.lanes)
		return;
	const unsigned char *col = a.in + 16LL * lane;
	unsigned char *dst = a.out + 16LL * lane;
	for (int y = blockIdx.y; y < a.out_height; y += gridDim.y) {
		const int r0 = a.first0 + y * a.shrink;
		double s[4] = { 0.0, 0.0, 0.0, 0.0 };
		for (int i0 = 0; i0 < a.shrink; i0 += 4) {
			float4 v[4];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int row = min(r0 + min(i0 + k, a.shrink - 1), a.im_height - 1) - a.in_top;
				v[k] = *reinterpret_cast<const float4 *>(col + (long long) row * a.in_stride);
			}
#pragma unroll
			for (int k = 0; k < 4; k++)
				if (i0 + k < a.shrink) {
					s[0] = __dadd_rn(s[0], (double) v[k].x);
					s[1] = __dadd_rn(s[1], (double) v[k].y);
					s[2] = __dadd_rn(s[2], (double) v[k].z);
					s[3] = __dadd_rn(s[3], (double) v[k].w);
				}
		}
		*reinterpret_cast<float4 *>(dst + (long long) y * a.out_stride) = make_float4((float) __dmul_rn(s[0], a.inv),
			(float) __dmul_rn(s[1], a.inv), (float) __dmul_rn(s[2], a.inv), (float) __dmul_rn(s[3], a.inv));
	}
}

// reduceh_f32_lds<B>: the horizontal reduce with one phase and an integer step.  The general kernel's lanes gather
// their taps straight from memory, S x B floats apart (a cache line per lane and tap: 0.19 ms for the 100 MB the
// vertical pass leaves of 8192^2 x 3); here a block stages the columns its 256 / B output pixels touch, for four rows,
// with coalesced loads (columns beyond the image: its edge column, vips_embed COPY), and a lane sums its element's
// taps from LDS -- in tap order, in double, multiply and add rounded separately (templates.h:183-194).
struct RhArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int im_width, in_left;  // columns clamp to the image; the window's first column
	int out_width, out_height;
	int first0, step, n;    // first tap (image column) of the region's output column 0, columns per output, taps
	int opb;                // output pixels per block
	const double *coef;
};

template <int B>
__global__ void __launch_bounds__(256)
reduceh_f32_lds(RhArgs a)
{
	constexpr int ROWS = 4;
	VH_DYNAMIC_LDS(unsigned char, rh_lds);
	double *s_c = reinterpret_cast<double *>(rh_lds);                       // n coefficients (64 slots)
	float *s_in = reinterpret_cast<float *>(rh_lds + 64 * sizeof(double));  // ROWS x span floats
	const int t = threadIdx.x;
	if (t < a.n)
		s_c[t] = a.coef[t];
	const int x0 = blockIdx.x * a.opb, nx = min(a.opb, a.out_width - x0);
	const int span_px = (nx - 1) * a.step + a.n; // input pixels the block's outputs touch
	const int span = span_px * B;
	const int c0 = a.first0 + x0 * a.step;       // image column of the span's first pixel
	const int xo = t / B, b = t - xo * B;        // this lane's output pixel of the block, band
	for (int y0 = blockIdx.y * ROWS; y0 < a.out_height; y0 += gridDim.y * ROWS) {
		__syncthreads(); // (the coefficients; the last rows' readers)
		for (int e = t; e < span; e += 256) {
			const int j = e / B, bb = e - j * B;
			const int col = min(max(c0 + j, 0), a.im_width - 1) - a.in_left;
#pragma unroll
			for (int r = 0; r < ROWS; r++) {
				const int y = min(y0 + r, a.out_height - 1);
				s_in[r * span + e] = reinterpret_cast<const float *>(a.in + (long long) y * a.in_stride)[col * B + bb];
			}
		}
		__syncthreads();
		if (xo < nx) {
			double sum[ROWS];
#pragma unroll
			for (int r = 0; r < ROWS; r++)
				sum[r] = 0.0;
			const float *p = s_in + xo * a.step * B + b;
			for (int k = 0; k < a.n; k++) {
				const double c = s_c[k];
#pragma unroll
				for (int r = 0; r < ROWS; r++)
					sum[r] = __dadd_rn(sum[r], __dmul_rn(c, (double) p[r * span + k * B]));
			}
#pragma unroll
			for (int r = 0; r < ROWS; r++)
				if (y0 + r < a.out_height)
					reinterpret_cast<float *>(a.out + (long long) (y0 + r) * a.out_stride)[(x0 + xo) * B + b] = (float) sum[r];
		}
	}
}

static bool rf_common(const VipsHipRegion *in, const VipsHipRegion *out, int per_lane, RfArgs *a)
{
	if (in->format != VIPS_HIP_FORMAT_FLOAT || out->format != VIPS_HIP_FORMAT_FLOAT || in->bands != out->bands)
		return false;
	const long long ne = (long long) out->width * out->bands;
	const long long col0 = (long long) (out->left - in->left) * in->bands * 4; // bytes into the window's row
	const unsigned int al = per_lane * 4 - 1;
	if (ne % per_lane || ne / per_lane > 0x7fffffff || ne < 1024)
		return false;
	const unsigned char *base = (const unsigned char *) in->data + col0;
	if (((uintptr_t) base & al) || (in->stride & al) || ((uintptr_t) out->data & al) || (out->stride & al))
		return false;
	a->in = base;
	a->out = (unsigned char *) out->data;
	a->in_stride = (long long) in->stride;
	a->out_stride = (long long) out->stride;
	a->im_height = in->im_height;
	a->in_top = in->top;
	a->lanes = (int) (ne / per_lane);
	a->out_height = out->height;
	return true;
}

// 1 launched, 0 not this kernel's case, -1 error.  pos: the region's positions (host), coef: the plan's double table
int reducev_f32_stream_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const std::vector<ReducePos> &pos, const double *coef)
{
	if (getenv("VIPS_HIP_NO_F32_STREAM") || pos.empty())
		return 0;
	RfArgs a;
	if (!rf_common(in, out, 2, &a))
		return 0;
	// one phase, one integer step
	int step = 0;
	for (size_t k = 1; k < pos.size(); k++) {
		const int d = pos[k].first - pos[k - 1].first;
		if (pos[k].phase != pos[0].phase || (step && d != step))
			return 0;
		step = d;
	}
	if (pos.size() == 1)
		step = (int) (r->shrink + 0.5);
	const int N = r->n_point;
	if (N > 56 || (size_t) (pos[0].phase + 1) * N > r->matrixf.size())
		return 0;
	a.first0 = pos[0].first;
	a.coef = coef + (size_t) pos[0].phase * N;
	a.shrink = step;
	a.inv = 0.0;
	const int gx = (a.lanes + 255) / 256;
	int segs = 1536 / gx;
	segs = segs < 1 ? 1 : segs;
	int seg = (out->height + segs - 1) / segs;
	seg = seg < 16 ? 16 : seg;
	a.seg = seg;
	dim3 grid(gx, (out->height + seg - 1) / seg, 1);
	Gate gate("reducev_f32_stream");
#define RF_CASE(SS, NN) \
	if (step == SS && N == NN) { \
		hipLaunchKernelGGL((reducev_f32_stream<SS, NN>), grid, dim3(256), 0, stream(), a); \
		VH_CHECK(hipGetLastError()); \
		return 1; \
	}
	// lanczos3 / mks2013 at integer shrinks: n = 6 S + 1 rounded as reduceh.cpp:113-141 does (odd)
	RF_CASE(2, 13)
	RF_CASE(3, 19)
	RF_CASE(4, 25)
	RF_CASE(5, 31)
	RF_CASE(6, 37)
	RF_CASE(8, 49)
	// lanczos2 / cubic / mitchell: 4 S + 1
	RF_CASE(2, 9)
	RF_CASE(4, 17)
	RF_CASE(8, 33)
#undef RF_CASE
	return 0;
}

// the region's positions (host): out column k reads image columns pos[k].first ...; 1 launched, 0 not this kernel's case
int reduceh_f32_lds_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const std::vector<ReducePos> &pos, const double *coef)
{
	if (getenv("VIPS_HIP_NO_F32_STREAM") || pos.empty())
		return 0;
	if (in->format != VIPS_HIP_FORMAT_FLOAT || out->format != VIPS_HIP_FORMAT_FLOAT || in->bands != out->bands ||
		in->bands < 1 || in->bands > 4 || out->width < 64)
		return 0;
	int step = 0;
	for (size_t k = 1; k < pos.size(); k++) {
		const int d = pos[k].first - pos[k - 1].first;
		if (pos[k].phase != pos[0].phase || (step && d != step))
			return 0;
		step = d;
	}
	const int N = r->n_point, B = in->bands;
	if (step < 1 || N > 64 || (size_t) (pos[0].phase + 1) * N > r->matrixf.size())
		return 0;
	RhArgs a;
	// rows of the output region are rows out->top ... of the window (reduceh.cpp:237-240: same rows in and out)
	a.in = (const unsigned char *) in->data + (long long) (out->top - in->top) * in->stride;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.im_width = in->im_width;
	a.in_left = in->left;
	a.out_width = out->width;
	a.out_height = out->height;
	a.first0 = pos[0].first;
	a.step = step;
	a.n = N;
	a.opb = 256 / B;
	a.coef = coef + (size_t) pos[0].phase * N;
	const size_t span = (size_t) ((a.opb - 1) * step + N) * B;
	const size_t lds = 64 * sizeof(double) + 4 * span * sizeof(float);
	if (lds > 60 * 1024)
		return 0;
	const int gx = (out->width + a.opb - 1) / a.opb;
	int gy = 4096 / gx;
	gy = gy < 1 ? 1 : gy;
	const int groups = (out->height + 3) / 4;
	gy = gy > groups ? groups : gy;
	Gate gate("reduceh_f32_lds");
	switch (B) {
	case 1: hipLaunchKernelGGL(reduceh_f32_lds<1>, dim3(gx, gy, 1), dim3(256), lds, stream(), a); break;
	case 2: hipLaunchKernelGGL(reduceh_f32_lds<2>, dim3(gx, gy, 1), dim3(256), lds, stream(), a); break;
	case 3: hipLaunchKernelGGL(reduceh_f32_lds<3>, dim3(gx, gy, 1), dim3(256), lds, stream(), a); break;
	default: hipLaunchKernelGGL(reduceh_f32_lds<4>, dim3(gx, gy, 1), dim3(256), lds, stream(), a); break;
	}
	VH_CHECK(hipGetLastError());
	return 1;
}

int shrinkv_f32_stream_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	if (getenv("VIPS_HIP_NO_F32_STREAM") || vshrink < 2 || vshrink > 64)
		return 0;
	RfArgs a;
	if (!rf_common(in, out, 4, &a))
		return 0;
	a.first0 = out->top * vshrink;
	a.shrink = vshrink;
	a.inv = 1.0 / vshrink;
	a.seg = 0;
	const int gx = (a.lanes + 255) / 256;
	int gy = 4096 / gx;
	gy = gy < 1 ? 1 : gy;
	gy = gy > out->height ? out->height : gy;
	Gate gate("shrinkv_f32_stream");
	hipLaunchKernelGGL(shrinkv_f32_stream, dim3(gx, gy, 1), dim3(256), 0, stream(), a);
	VH_CHECK(hipGetLastError());
	return 1;
}

} // namespace vh
