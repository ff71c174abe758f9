// Colour path for gfx950: the per-scanline process_line functions of colour/
// (colour.c:119-156 drives them one scanline at a time) as ONE kernel per region.
//
// A colourspace conversion in the reference is a chain of images, each step
// rounding its result to float (or uchar / short) in memory:
//     sRGB -> LAB = cast(uchar), sRGB2scRGB, scRGB2XYZ, XYZ2Lab   (colourspace.c:362)
// Every step is per-pixel, so the chain is evaluated here in registers with the same
// intermediate types and the same operation order: one read and one write per pixel
// instead of four of each, identical bits.  Transcendentals never run on the device:
// the LUTs the reference builds with powf()/cbrtf() (LabQ2sRGB.c:130-160,
// XYZ2Lab.c:92-106) are built on the host by the same libm calls and uploaded.
//
// Float arithmetic: the reference is compiled for baseline x86-64 (SSE2, no FMA), so
// every multiply and add below is a separate IEEE operation (__fmul_rn/__fadd_rn,
// __dmul_rn/...; the file is also built with -ffp-contract=off).
#include "colour_device.h"
#define VH_CBRT_FN static __host__ __device__ __forceinline__
#include "cbrt_exact.h"
#include "cbrt_quad.h"

#include <climits>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <type_traits>

namespace vh {

// The pointwise kernels below give a thread one element column and RU rows per loop trip,
// all RU loads issued before the first is used: with one load in flight per thread these
// kernels sit at ~1 TB/s (bytes in flight, not bandwidth, bound them); four gets 3-4x.
constexpr int RU = 4;

// grid.y for a row-looping kernel: enough blocks to fill the part, few enough that per-block
// setup is amortised over many rows
static inline int rows_grid(int gx, int height)
{
	const int groups = (height + RU - 1) / RU;
	int gy = 16384 / (gx > 0 ? gx : 1);
	gy = gy < 1 ? 1 : gy;
	return groups < gy ? groups : gy;
}

// ------------------------------------------------------------------ tables


static std::mutex &g_tables_mutex = *new std::mutex;
// one set per device (the tables are device memory)
static ColourTables g_tables_by_device[64];
#define g_tables (g_tables_by_device[current_device() < 0 ? 0 : current_device() & 63])

// calcul_tables(), LabQ2sRGB.c:130-160
static void calcul_tables(int range, std::vector<int> &Y2v, std::vector<float> &v2Y)
{
	Y2v.resize(range + 1);
	v2Y.resize(range);
	for (int i = 0; i < range; i++) {
		float f = (float) i / (range - 1);
		float v;

		if (f <= 0.0031308)
			v = 12.92F * f;
		else
			v = (1.0F + 0.055F) * powf(f, 1.0F / 2.4F) - 0.055F;

		Y2v[i] = rintf((range - 1) * v);
	}
	Y2v[range] = Y2v[range - 1];

	for (int i = 0; i < range; i++) {
		float f = (float) i / (range - 1);

		if (f <= 0.04045)
			v2Y[i] = f / 12.92F;
		else
			v2Y[i] = powf((f + 0.055F) / (1 + 0.055F), 2.4F);
	}
}

static void cbrt_table(std::vector<float> &cb)
{
	// table_init(), XYZ2Lab.c:92-106
	const int QUANT_ELEMENTS = 100000;
	cb.resize(QUANT_ELEMENTS);
	for (int i = 0; i < QUANT_ELEMENTS; i++) {
		float Y = (double) i / QUANT_ELEMENTS;

		if (Y < 0.008856)
			cb[i] = 7.787F * Y + (16.0F / 116.0F);
		else
			cb[i] = cbrtf(Y);
	}
}

void colour_tables_host(std::vector<float> &v2Y_8, std::vector<int> &Y2v_8, std::vector<float> &cbrt)
{
	calcul_tables(256, Y2v_8, v2Y_8);
	cbrt_table(cbrt);
}

// The tables of cbrt_exact.h for the calling thread's device: made from the cube-root table this
// host's libm produces, every entry checked to come out of cbrt_pair() bit for bit.  nullptr: the
// scheme does not fit this host's cbrtf (callers then read the table itself), or an upload failed.
const CbrtExact *cbrt_exact_tables()
{
	static std::mutex mutex;
	static CbrtExact by_device[64];
	static int state[64]; // 0 not tried, 1 ready, -1 unusable
	std::lock_guard<std::mutex> lock(mutex);
	const int dev = current_device() < 0 ? 0 : current_device() & 63;
	if (state[dev])
		return state[dev] > 0 ? &by_device[dev] : nullptr;
	state[dev] = -1;
	std::vector<float> cb;
	cbrt_table(cb);
	std::vector<CbrtBlockD> bd(CBRT_BLOCKS + 1);
	std::vector<CbrtBlockI> bi(CBRT_BLOCKS + 1);
	std::vector<unsigned int> res(CBRT_RES_WORDS, 0x55555555u); // residual 0 everywhere
	for (int k = 0; k <= CBRT_BLOCKS; k++) {
		bi[k].i0 = 0;
		bi[k].count = 0;
	}
	for (int i = 1; i < CBRT_N + 4096; i++) { // (one block past the table's last)
		const int k = (int) (cbrt_bits((float) i) >> 18) - CBRT_KEY0;
		if (k < 0 || k > CBRT_BLOCKS)
			continue;
		if (!bi[k].count)
			bi[k].i0 = i;
		bi[k].count++;
	}
	for (int k = 0; k <= CBRT_BLOCKS; k++) {
		if (!bi[k].count)
			return nullptr;
		bd[k].c0 = cbrt((double) bi[k].i0 / CBRT_N);
		bd[k].inv = 1.0 / bi[k].i0;
	}
	for (int i = CBRT_LINEAR; i < CBRT_N; i++) {
		const int k = (int) (cbrt_bits((float) i) >> 18) - CBRT_KEY0;
		if (k < 0 || k >= CBRT_BLOCKS)
			return nullptr;
		const long long r = (long long) cbrt_bits(cb[i]) - (long long) cbrt_predict(bd[k], bi[k], i);
		if (r < -1 || r > 1)
			return nullptr;
		res[i >> 4] = (res[i >> 4] & ~(3u << (2 * (i & 15)))) | ((unsigned int) (r + 1) << (2 * (i & 15)));
	}
	// every pair a kernel can ask for, against the table the reference would read
	{
		for (int pass = 0; pass < 2; pass++) {
		CbrtExact host = { bd.data(), bi.data(), res.data(), pass ? cb.data() : nullptr };
		for (int i = 0; i + 1 < CBRT_N; i++) {
			float t0, dt;
			cbrt_pair(host, i, &t0, &dt);
			const float want_dt = cb[i + 1] - cb[i];
			if (memcmp(&t0, &cb[i], 4) || memcmp(&dt, &want_dt, 4))
				return nullptr;
		}
		}
	}
	// the single-precision form, under the same two checks; left out (kernels take the double form)
	// when this host's cbrtf does not fit it
	std::vector<CbrtBlockF> bf(CBRT_BLOCKS + 1);
	std::vector<unsigned int> res32(CBRT_RES_WORDS, 0x55555555u);
	bool have32 = !getenv("VIPS_HIP_CBRT_F64");
	for (int k = 0; k <= CBRT_BLOCKS; k++) {
		bf[k].c0 = (float) bd[k].c0;
		bf[k].inv = (float) bd[k].inv;
	}
	for (int i = CBRT_LINEAR; i < CBRT_N && have32; i++) {
		const int k = (int) (cbrt_bits((float) i) >> 18) - CBRT_KEY0;
		const long long r = (long long) cbrt_bits(cb[i]) - (long long) cbrt_predict32(bf[k], i - bi[k].i0);
		if (r < -1 || r > 1)
			have32 = false;
		else
			res32[i >> 4] = (res32[i >> 4] & ~(3u << (2 * (i & 15)))) | ((unsigned int) (r + 1) << (2 * (i & 15)));
	}
	// (the first entry of a block must be the block's c0 itself: cbrt_pair32 reads it for the pair that straddles)
	for (int k = 1; k <= CBRT_BLOCKS && have32; k++)
		if (bi[k].i0 < CBRT_N && cbrt_predict32(bf[k], 0) != cbrt_bits(bf[k].c0))
			have32 = false;
	for (int pass = 0; pass < 2 && have32; pass++) {
		CbrtExact host = { bd.data(), bi.data(), res.data(), pass ? cb.data() : nullptr, bf.data(), res32.data() };
		for (int i = 0; i + 1 < CBRT_N && have32; i++) {
			float t0, dt;
			cbrt_pair32(host, i, &t0, &dt);
			const float want_dt = cb[i + 1] - cb[i];
			if (memcmp(&t0, &cb[i], 4) || memcmp(&dt, &want_dt, 4))
				have32 = false;
		}
	}
	if (getenv("VIPS_HIP_DEBUG_CBRT"))
		fprintf(stderr, "cbrt_exact_tables: the %s form\n", have32 ? "single-precision" : "double");
	CbrtExact &t = by_device[dev];
	t.bf = nullptr;
	t.res32 = nullptr;
	if (have32) {
		t.bf = (const CbrtBlockF *) upload(bf.data(), bf.size() * sizeof(CbrtBlockF));
		t.res32 = (const unsigned int *) upload(res32.data(), res32.size() * sizeof(unsigned int));
		if (!t.bf || !t.res32) {
			t.bf = nullptr;
			t.res32 = nullptr;
			vips_hip_error_clear();
		}
	}
	t.bd = (const CbrtBlockD *) upload(bd.data(), bd.size() * sizeof(CbrtBlockD));
	t.bi = (const CbrtBlockI *) upload(bi.data(), bi.size() * sizeof(CbrtBlockI));
	t.res = (const unsigned int *) upload(res.data(), res.size() * sizeof(unsigned int));
	t.lin = (const float *) upload(cb.data(), CBRT_LINEAR * sizeof(float));
	if (!t.bd || !t.bi || !t.res || !t.lin)
		return nullptr;
	state[dev] = 1;
	return &t;
}

// The tables of cbrt_quad.h for the calling thread's device, made from the cube-root table this host's libm
// produces: per block a least-squares quadratic through the exact cube roots, then the residuals and every pair
// checked by running cbq_pair() itself.  nullptr: the scheme does not fit this host's cbrtf, or an upload failed.
static const CbrtQuad *cbq_refused(int where)
{
	if (getenv("VIPS_HIP_DEBUG_CBRT"))
		fprintf(stderr, "cbrt_quad_tables: this host's cbrtf does not fit the scheme (check %d)\n", where);
	return nullptr;
}

const CbrtQuad *cbrt_quad_tables()
{
	static std::mutex mutex;
	static CbrtQuad by_device[64];
	static int state[64]; // 0 not tried, 1 ready, -1 unusable
	std::lock_guard<std::mutex> lock(mutex);
	const int dev = current_device() < 0 ? 0 : current_device() & 63;
	if (state[dev])
		return state[dev] > 0 ? &by_device[dev] : nullptr;
	state[dev] = -1;
	if (getenv("VIPS_HIP_NO_CBRT_QUAD"))
		return cbq_refused(1);
	std::vector<float> cb;
	cbrt_table(cb);
	std::vector<CbqBlock> blk(CBQ_BLOCKS);
	std::vector<int> first(CBQ_BLOCKS, -1), count(CBQ_BLOCKS, 0);
	for (int i = 0; i < CBQ_N; i++) {
		const int k = cbq_key((float) (i + CBQ_SHIFT));
		if (k < 0 || k >= CBQ_BLOCKS)
			return cbq_refused(2);
		if (first[k] < 0)
			first[k] = i;
		count[k]++;
	}
	for (int k = 0; k < CBQ_BLOCKS; k++) {
		CbqBlock &q = blk[k];
		q.c0 = q.c1 = q.c2 = q.next = 0.0f;
		if (first[k] < 0)
			continue; // (a key no integer has)
		const int i0 = first[k], n = count[k];
		q.c0 = cb[i0];
		q.next = i0 + n < CBQ_N ? cb[i0 + n] : cb[CBQ_N - 1];
		if (n >= 2) {
			// least squares of T[i0 + j] - c0 = c1 j + c2 j^2 over the block (c2 = 0 for two entries)
			double s11 = 0, s12 = 0, s22 = 0, r1 = 0, r2 = 0;
			for (int j = 1; j < n; j++) {
				const double y = (double) cb[i0 + j] - (double) q.c0, x = j, x2 = x * x;
				s11 += x * x;
				s12 += x * x2;
				s22 += x2 * x2;
				r1 += x * y;
				r2 += x2 * y;
			}
			if (n == 2) {
				q.c1 = (float) (r1 / s11);
			}
			else {
				const double det = s11 * s22 - s12 * s12;
				q.c1 = (float) ((r1 * s22 - r2 * s12) / det);
				q.c2 = (float) ((r2 * s11 - r1 * s12) / det);
			}
		}
	}
	// residuals: a signed 2-bit field + the block's bias (the lowest bit of c2, which takes part in the
	// prediction: set first, then measured)
	std::vector<unsigned int> res(CBQ_RES_WORDS, 0u);
	for (int k = 0; k < CBQ_BLOCKS; k++) {
		if (first[k] < 0)
			continue;
		const int i0 = first[k], n = count[k];
		bool done = false;
		for (unsigned int bias = 0; bias < 2 && !done; bias++) {
			CbqBlock q = blk[k];
			q.c2 = cbq_float((cbq_bits(q.c2) & ~1u) | bias);
			bool ok = true;
			for (int j = 0; j < n && ok; j++) {
				const long long r = (long long) cbq_bits(cb[i0 + j]) - (long long) cbq_bits(cbq_predict(q, (float) j)) - (long long) bias;
				ok = r >= -2 && r <= 1 && (j > 0 || r + (long long) bias == 0);
			}
			if (!ok)
				continue;
			blk[k] = q;
			for (int j = 0; j < n; j++) {
				const int i = i0 + j;
				const long long r = (long long) cbq_bits(cb[i]) - (long long) cbq_bits(cbq_predict(q, (float) j)) - (long long) bias;
				res[i >> 4] |= ((unsigned int) r & 3u) << (2 * (i & 15));
			}
			done = true;
		}
		if (!done)
			return cbq_refused(100 + k);
	}
	// every pair a kernel can ask for, against the table the reference would read
	for (int i = 0; i + 1 < CBQ_N; i++) {
		float t0, dt;
		cbq_pair(blk.data(), res.data(), i, (float) i, &t0, &dt);
		const float want_dt = cb[i + 1] - cb[i];
		if (memcmp(&t0, &cb[i], 4) || memcmp(&dt, &want_dt, 4))
			return cbq_refused(4);
	}
	if (getenv("VIPS_HIP_DEBUG_CBRT"))
		fprintf(stderr, "cbrt_quad_tables: every pair checked\n");
	CbrtQuad &t = by_device[dev];
	t.blk = (const CbqBlock *) upload(blk.data(), blk.size() * sizeof(CbqBlock));
	t.res = (const unsigned int *) upload(res.data(), res.size() * sizeof(unsigned int));
	if (!t.blk || !t.res) {
		vips_hip_error_clear();
		return cbq_refused(5);
	}
	state[dev] = 1;
	return &t;
}

static int ensure_tables()
{
	std::lock_guard<std::mutex> lock(g_tables_mutex);
	if (g_tables.cbrt)
		return 0;
	std::vector<int> Y2v;
	std::vector<float> v2Y;
	calcul_tables(256, Y2v, v2Y);
	g_tables.Y2v_8 = (int *) upload(Y2v.data(), Y2v.size() * sizeof(int));
	g_tables.v2Y_8 = (float *) upload(v2Y.data(), v2Y.size() * sizeof(float));
	calcul_tables(65536, Y2v, v2Y);
	g_tables.Y2v_16 = (int *) upload(Y2v.data(), Y2v.size() * sizeof(int));
	g_tables.v2Y_16 = (float *) upload(v2Y.data(), v2Y.size() * sizeof(float));
	std::vector<float> cb;
	cbrt_table(cb);
	float *cbrt = (float *) upload(cb.data(), cb.size() * sizeof(float));
	if (!g_tables.Y2v_8 || !g_tables.v2Y_8 || !g_tables.Y2v_16 || !g_tables.v2Y_16 || !cbrt)
		return -1;
	g_tables.cbrt = cbrt;
	return 0;
}

// One thread per pixel.  TIN/TOUT are the in-memory band formats.
template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256)
colour_route_kernel(RouteArgs a)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
		TIN i0[RU], i1[RU], i2[RU];
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = min(y0 + r, a.height - 1);
			const TIN *p = (const TIN *) (a.in + (long long) y * a.in_stride) + (long long) x * a.in_bands;
			i0[r] = p[0];
			i1[r] = p[1];
			i2[r] = p[2];
		}
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = y0 + r;
			if (y < a.height) {
				const TIN *p = (const TIN *) (a.in + (long long) y * a.in_stride) + (long long) x * a.in_bands;
				TOUT *q = (TOUT *) (a.out + (long long) y * a.out_stride) + (long long) x * a.out_bands;
				route_pixel<TIN, TOUT>(a, a.tables.v2Y_8, a.tables.Y2v_8, i0[r], i1[r], i2[r], q[0], q[1], q[2]);
				for (int e = 0; e < a.extra_bands; e++)
					q[3 + e] = Carry<TIN, TOUT>::run(p[3 + e], a.alpha_scale);
			}
		}
	}
}

// 3 bands -> 3 bands, 4 pixels per thread: the 12 input elements arrive as three aligned
// vector loads and the 12 outputs leave as three aligned vector stores (dword / dwordx2 /
// dwordx4 for 1 / 2 / 4-byte elements).  One element per store instruction, as the scalar
// kernel does for interleaved bands, makes every wave store touch all of the wave's cache
// lines (3x the L1->L2 write requests); this is the shape of every BASELINE colour config
// (C3: float -> float, thumbnails: uchar -> float / short and back).
template <typename T>
struct __attribute__((aligned(4 * sizeof(T)))) Vec4 {
	T v[4];
};

template <typename TIN, typename TOUT, int ROUTE>
__global__ void __launch_bounds__(256)
colour_route_x4_kernel(RouteArgs a)
{
	// the two 8-bit tables in LDS: up to six of a pixel's table reads stay off the vector memory path
	__shared__ float s_v2Y[256];
	__shared__ int s_Y2v[260];
	s_v2Y[threadIdx.x] = a.tables.v2Y_8[threadIdx.x];
	s_Y2v[threadIdx.x] = a.tables.Y2v_8[threadIdx.x];
	if (threadIdx.x == 0)
		s_Y2v[256] = a.tables.Y2v_8[256];
	__syncthreads();
	const int x4 = blockIdx.x * blockDim.x + threadIdx.x;
	if (x4 * 4 >= a.width)
		return;
	for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
		const Vec4<TIN> *p = (const Vec4<TIN> *) (a.in + (long long) y * a.in_stride) + (long long) x4 * 3;
		Vec4<TOUT> *q = (Vec4<TOUT> *) (a.out + (long long) y * a.out_stride) + (long long) x4 * 3;
		const Vec4<TIN> v0 = p[0], v1 = p[1], v2 = p[2];
		Vec4<TOUT> r0, r1, r2;
		route_pixel<TIN, TOUT, ROUTE>(a, s_v2Y, s_Y2v, v0.v[0], v0.v[1], v0.v[2], r0.v[0], r0.v[1], r0.v[2]);
		route_pixel<TIN, TOUT, ROUTE>(a, s_v2Y, s_Y2v, v0.v[3], v1.v[0], v1.v[1], r0.v[3], r1.v[0], r1.v[1]);
		route_pixel<TIN, TOUT, ROUTE>(a, s_v2Y, s_Y2v, v1.v[2], v1.v[3], v2.v[0], r1.v[2], r1.v[3], r2.v[0]);
		route_pixel<TIN, TOUT, ROUTE>(a, s_v2Y, s_Y2v, v2.v[1], v2.v[2], v2.v[3], r2.v[1], r2.v[2], r2.v[3]);
		q[0] = r0;
		q[1] = r1;
		q[2] = r2;
	}
}

// sRGB -> Lab / LabS of 3-band uchar or float images with XYZ2Lab's cube-root table in LDS
// (cbrt_exact.h: every entry bit for bit from 34 KB): a lane that gathers from the 400 KB table in
// global memory pulls a 128-byte line through its CU's L1 fill path per channel (~146 cycles of that
// path per wave instruction, tools/gather_probe.hip), which bounded colour_route_x4_kernel on these
// routes; here a pixel reads nothing but its own bytes.  Persistent blocks (the tables are copied
// once per block), 4 pixels per thread, rows dealt round the grid.
// F32: the table's single-precision form (cbrt_exact.h), when the host found that its cbrtf fits it
template <typename TIN, bool LABS, bool F32>
__global__ void __launch_bounds__(256)
colour_lab_lds_kernel(RouteArgs a, CbrtExact cx)
{
	__shared__ float s_v2Y[256];
	__shared__ unsigned int s_res[CBRT_RES_WORDS];
	__shared__ CbrtBlockD s_bd[F32 ? 1 : CBRT_BLOCKS + 1];
	__shared__ CbrtBlockF s_bf[F32 ? CBRT_BLOCKS + 1 : 1];
	__shared__ CbrtBlockI s_bi[CBRT_BLOCKS + 1];
	__shared__ float s_lin[CBRT_LINEAR];
	const int t = threadIdx.x;
	s_v2Y[t] = a.tables.v2Y_8[t];
	for (int i = t; i < CBRT_RES_WORDS; i += 256)
		s_res[i] = F32 ? cx.res32[i] : cx.res[i];
	for (int i = t; i < CBRT_LINEAR; i += 256)
		s_lin[i] = cx.lin[i];
	if (t <= CBRT_BLOCKS) {
		if constexpr (F32)
			s_bf[t] = cx.bf[t];
		else
			s_bd[t] = cx.bd[t];
		s_bi[t] = cx.bi[t];
	}
	__syncthreads();
	const CbrtExact lds = { s_bd, s_bi, s_res, s_lin, s_bf, s_res };
	const int x4 = blockIdx.x * blockDim.x + t;
	if (x4 * 4 >= a.width)
		return;
	typedef typename std::conditional<LABS, short, float>::type TOUT;
	for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
		const Vec4<TIN> *p = (const Vec4<TIN> *) (a.in + (long long) y * a.in_stride) + (long long) x4 * 3;
		Vec4<TOUT> *q = (Vec4<TOUT> *) (a.out + (long long) y * a.out_stride) + (long long) x4 * 3;
		const Vec4<TIN> v0 = p[0], v1 = p[1], v2 = p[2];
		const TIN in[12] = { v0.v[0], v0.v[1], v0.v[2], v0.v[3], v1.v[0], v1.v[1], v1.v[2], v1.v[3], v2.v[0], v2.v[1], v2.v[2],
			v2.v[3] };
		TOUT out[12];
#pragma unroll
		for (int m = 0; m < 4; m++) {
			// sRGB2scRGB.c:72-90 (vips_colour_code_build casts to uchar), scRGB2XYZ.c:58-82
			Px v;
			v.a = s_v2Y[load_as_uchar_like<TIN>(in[3 * m], 255)];
			v.b = s_v2Y[load_as_uchar_like<TIN>(in[3 * m + 1], 255)];
			v.c = s_v2Y[load_as_uchar_like<TIN>(in[3 * m + 2], 255)];
			v = step_scRGB2XYZ(v);
			// XYZ2Lab.c:109-138 on small finite values
			const float n0 = quant_div_finite<0>(__fmul_rn(100000.0f, v.a));
			const float n1 = quant_div_finite<1>(__fmul_rn(100000.0f, v.b));
			const float n2 = quant_div_finite<2>(__fmul_rn(100000.0f, v.c));
			const int i0 = min(max(vh::cvt_i32(n0), 0), CBRT_N - 2);
			const int i1 = min(max(vh::cvt_i32(n1), 0), CBRT_N - 2);
			const int i2 = min(max(vh::cvt_i32(n2), 0), CBRT_N - 2);
			float t0, dt;
			if constexpr (F32)
				cbrt_pair32(lds, i0, &t0, &dt);
			else
				cbrt_pair(lds, i0, &t0, &dt);
			const float cbx = __fadd_rn(t0, __fmul_rn(__fsub_rn(n0, (float) i0), dt));
			if constexpr (F32)
				cbrt_pair32(lds, i1, &t0, &dt);
			else
				cbrt_pair(lds, i1, &t0, &dt);
			const float cby = __fadd_rn(t0, __fmul_rn(__fsub_rn(n1, (float) i1), dt));
			if constexpr (F32)
				cbrt_pair32(lds, i2, &t0, &dt);
			else
				cbrt_pair(lds, i2, &t0, &dt);
			const float cbz = __fadd_rn(t0, __fmul_rn(__fsub_rn(n2, (float) i2), dt));
			const float L = __fsub_rn(__fmul_rn(116.0F, cby), 16.0F);
			const float A = __fmul_rn(500.0F, __fsub_rn(cbx, cby));
			const float B = __fmul_rn(200.0F, __fsub_rn(cby, cbz));
			if constexpr (LABS) {
				out[3 * m] = lab2labs_finite(L, 32767.0 / 100.0, 0.0);
				out[3 * m + 1] = lab2labs_finite(A, 32768.0 / 128.0, -32768.0);
				out[3 * m + 2] = lab2labs_finite(B, 32768.0 / 128.0, -32768.0);
			}
			else {
				out[3 * m] = L;
				out[3 * m + 1] = A;
				out[3 * m + 2] = B;
			}
		}
		Vec4<TOUT> r0 = { { out[0], out[1], out[2], out[3] } }, r1 = { { out[4], out[5], out[6], out[7] } },
				   r2 = { { out[8], out[9], out[10], out[11] } };
		q[0] = r0;
		q[1] = r1;
		q[2] = r2;
	}
}

// ... with the table in cbrt_quad.h's form: 27 vector instructions and two LDS reads per channel instead of ~45
// and seven (one 16-byte block record, the pair's residual bits), 57 KB of LDS: two blocks per CU
// (1024 threads: ONE block's tables serve a CU's 16 waves -- 256-thread blocks, two per CU by their LDS, left two
// waves per SIMD to hide the LDS reads behind and ran no faster than the older form)
template <typename TIN, bool LABS>
__global__ void __launch_bounds__(1024)
colour_lab_quad_kernel(RouteArgs a, CbrtQuad cq)
{
	__shared__ float s_v2Y[256];
	__shared__ unsigned int s_res[CBQ_RES_WORDS];
	__shared__ __attribute__((aligned(16))) CbqBlock s_blk[CBQ_BLOCKS];
	const int t = threadIdx.x;
	if (t < 256)
		s_v2Y[t] = a.tables.v2Y_8[t];
	for (int i = t; i < CBQ_RES_WORDS; i += 1024)
		s_res[i] = cq.res[i];
	for (int i = t; i < CBQ_BLOCKS; i += 1024)
		s_blk[i] = cq.blk[i];
	__syncthreads();
	const int x4 = blockIdx.x * blockDim.x + t;
	if (x4 * 4 >= a.width)
		return;
	typedef typename std::conditional<LABS, short, float>::type TOUT;
	// (the next row's pixels travel while this row's are converted: with 600 instructions between a load and its
	// store a wave otherwise sits out the whole memory latency once per 4 pixels)
	Vec4<TIN> n0, n1, n2;
	{
		const Vec4<TIN> *p = (const Vec4<TIN> *) (a.in + (long long) blockIdx.y * a.in_stride) + (long long) x4 * 3;
		n0 = p[0];
		n1 = p[1];
		n2 = p[2];
	}
	for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
		Vec4<TOUT> *q = (Vec4<TOUT> *) (a.out + (long long) y * a.out_stride) + (long long) x4 * 3;
		const Vec4<TIN> v0 = n0, v1 = n1, v2 = n2;
		if (y + (int) gridDim.y < a.height) {
			const Vec4<TIN> *p = (const Vec4<TIN> *) (a.in + (long long) (y + (int) gridDim.y) * a.in_stride) + (long long) x4 * 3;
			n0 = p[0];
			n1 = p[1];
			n2 = p[2];
		}
		const TIN in[12] = { v0.v[0], v0.v[1], v0.v[2], v0.v[3], v1.v[0], v1.v[1], v1.v[2], v1.v[3], v2.v[0], v2.v[1], v2.v[2],
			v2.v[3] };
		TOUT out[12];
#pragma unroll
		for (int m = 0; m < 4; m++) {
			// sRGB2scRGB.c:72-90 (vips_colour_code_build casts to uchar), scRGB2XYZ.c:58-82
			Px v;
			v.a = s_v2Y[load_as_uchar_like<TIN>(in[3 * m], 255)];
			v.b = s_v2Y[load_as_uchar_like<TIN>(in[3 * m + 1], 255)];
			v.c = s_v2Y[load_as_uchar_like<TIN>(in[3 * m + 2], 255)];
			v = step_scRGB2XYZ(v);
			// XYZ2Lab.c:109-138 on small finite values
			const float n[3] = { quant_div_finite<0>(__fmul_rn(100000.0f, v.a)), quant_div_finite<1>(__fmul_rn(100000.0f, v.b)),
				quant_div_finite<2>(__fmul_rn(100000.0f, v.c)) };
			float cb[3];
#pragma unroll
			for (int c = 0; c < 3; c++) {
				const int i = min(max(vh::cvt_i32(n[c]), 0), CBRT_N - 2);
				const float fi = (float) i;
				float t0, dt;
				cbq_pair(s_blk, s_res, i, fi, &t0, &dt);
				cb[c] = __fadd_rn(t0, __fmul_rn(__fsub_rn(n[c], fi), dt));
			}
			const float L = __fsub_rn(__fmul_rn(116.0F, cb[1]), 16.0F);
			const float A = __fmul_rn(500.0F, __fsub_rn(cb[0], cb[1]));
			const float B = __fmul_rn(200.0F, __fsub_rn(cb[1], cb[2]));
			if constexpr (LABS) {
				out[3 * m] = lab2labs_finite(L, 32767.0 / 100.0, 0.0);
				out[3 * m + 1] = lab2labs_finite(A, 32768.0 / 128.0, -32768.0);
				out[3 * m + 2] = lab2labs_finite(B, 32768.0 / 128.0, -32768.0);
			}
			else {
				out[3 * m] = L;
				out[3 * m + 1] = A;
				out[3 * m + 2] = B;
			}
		}
		Vec4<TOUT> r0 = { { out[0], out[1], out[2], out[3] } }, r1 = { { out[4], out[5], out[6], out[7] } },
				   r2 = { { out[8], out[9], out[10], out[11] } };
		q[0] = r0;
		q[1] = r1;
		q[2] = r2;
	}
}

// ------------------------------------------------------------------- cast

template <typename TIN, typename TOUT>
struct CastOne;

// CAST_INT_INT: through int (<= 16 bit targets) or int64 (32 bit targets)
#define CAST_II(TIN, TOUT, TEMP, LO, HI) \
	template <> \
	struct CastOne<TIN, TOUT> { \
		static __device__ __forceinline__ TOUT run(TIN v) \
		{ \
			TEMP t = (TEMP) v; \
			t = t > (TEMP) (HI) ? (TEMP) (HI) : t; \
			t = t < (TEMP) (LO) ? (TEMP) (LO) : t; \
			return (TOUT) t; \
		} \
	};
#define CAST_II_ALL(TIN) \
	CAST_II(TIN, unsigned char, int, 0, UCHAR_MAX) \
	CAST_II(TIN, signed char, int, SCHAR_MIN, SCHAR_MAX) \
	CAST_II(TIN, unsigned short, int, 0, USHRT_MAX) \
	CAST_II(TIN, short, int, SHRT_MIN, SHRT_MAX) \
	CAST_II(TIN, unsigned int, long long, 0, UINT_MAX) \
	CAST_II(TIN, int, long long, INT_MIN, INT_MAX)
CAST_II_ALL(unsigned char)
CAST_II_ALL(signed char)
CAST_II_ALL(unsigned short)
CAST_II_ALL(short)
CAST_II_ALL(unsigned int)
CAST_II_ALL(int)
#undef CAST_II_ALL
#undef CAST_II

// CAST_FLOAT_INT: clip as double, then C truncation
#define CAST_FI(TIN, TOUT, LO, HI) \
	template <> \
	struct CastOne<TIN, TOUT> { \
		static __device__ __forceinline__ TOUT run(TIN v) \
		{ \
			double d = (double) v; \
			d = (double) (HI) < d ? (double) (HI) : d; \
			d = (double) (LO) > d ? (double) (LO) : d; \
			return vh::cvt_to<TOUT>(d); \
		} \
	};
#define CAST_FI_ALL(TIN) \
	CAST_FI(TIN, unsigned char, 0, UCHAR_MAX) \
	CAST_FI(TIN, signed char, SCHAR_MIN, SCHAR_MAX) \
	CAST_FI(TIN, unsigned short, 0, USHRT_MAX) \
	CAST_FI(TIN, short, SHRT_MIN, SHRT_MAX) \
	CAST_FI(TIN, unsigned int, 0, UINT_MAX) \
	CAST_FI(TIN, int, INT_MIN, INT_MAX)
CAST_FI_ALL(float)
CAST_FI_ALL(double)
#undef CAST_FI_ALL
#undef CAST_FI

// CAST_REAL_FLOAT: plain conversion
#define CAST_RF(TIN, TOUT) \
	template <> \
	struct CastOne<TIN, TOUT> { \
		static __device__ __forceinline__ TOUT run(TIN v) { return (TOUT) v; } \
	};
#define CAST_RF_ALL(TIN) \
	CAST_RF(TIN, float) \
	CAST_RF(TIN, double)
CAST_RF_ALL(unsigned char)
CAST_RF_ALL(signed char)
CAST_RF_ALL(unsigned short)
CAST_RF_ALL(short)
CAST_RF_ALL(unsigned int)
CAST_RF_ALL(int)
CAST_RF_ALL(float)
CAST_RF_ALL(double)
#undef CAST_RF_ALL
#undef CAST_RF

struct CastArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int ne; // elements per row copied
	int height;
	// band remap for extract_band / bandjoin: element e of the output row reads
	// input pel e / out_take, band in_first + e % out_take; writes band out_first + ...
	int in_bands, out_bands, in_first, out_first, take;
};

template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256)
cast_kernel(CastArgs a)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.ne)
		return;
	const int x = e / a.take;
	const int b = e - x * a.take;
	const long long ie = (long long) x * a.in_bands + a.in_first + b;
	const long long oe = (long long) x * a.out_bands + a.out_first + b;
	for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
		TIN v[RU];
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = min(y0 + r, a.height - 1);
			v[r] = ((const TIN *) (a.in + (long long) y * a.in_stride))[ie];
		}
#pragma unroll
		for (int r = 0; r < RU; r++)
			if (y0 + r < a.height)
				((TOUT *) (a.out + (long long) (y0 + r) * a.out_stride))[oe] = CastOne<TIN, TOUT>::run(v[r]);
	}
}

// A plain cast (every band, rows that start where a group of four elements does on both sides): a lane owns FOUR
// neighbouring elements -- one load, one store of 4 x sizeof (16 bytes for a float: a wave writes 1 KiB of a line
// in one instruction) -- and RU rows are in flight per lane.  vips_cast's arithmetic is CastOne's, element by
// element (conversion/cast.c:120-330); round 5's one-element-per-lane form ran at 56 % of 8 TB/s on uchar -> float.
template <typename T>
struct alignas(4 * sizeof(T)) CastQuad {
	T v[4];
};

template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256)
cast_quad_kernel(CastArgs a)
{
	const int g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= a.ne / 4)
		return;
	for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
		CastQuad<TIN> v[RU];
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = min(y0 + r, a.height - 1);
			v[r] = ((const CastQuad<TIN> *) (a.in + (long long) y * a.in_stride))[g];
		}
#pragma unroll
		for (int r = 0; r < RU; r++)
			if (y0 + r < a.height) {
				CastQuad<TOUT> o;
#pragma unroll
				for (int i = 0; i < 4; i++)
					o.v[i] = CastOne<TIN, TOUT>::run(v[r].v[i]);
				((CastQuad<TOUT> *) (a.out + (long long) (y0 + r) * a.out_stride))[g] = o;
			}
	}
}

template <typename TIN, typename TOUT>
static bool cast_quad_ok(const CastArgs &a)
{
	if (a.take != a.in_bands || a.take != a.out_bands || a.in_first != 0 || a.out_first != 0 || a.ne < 1024)
		return false;
	if (getenv("VIPS_HIP_NO_CAST_QUAD"))
		return false;
	return !(((uintptr_t) a.in | (uintptr_t) a.in_stride) % (4 * sizeof(TIN))) &&
		!(((uintptr_t) a.out | (uintptr_t) a.out_stride) % (4 * sizeof(TOUT)));
}

template <typename TIN>
static int launch_cast_out(const CastArgs &a_all, int out_format)
{
	dim3 block(256, 1, 1);
	CastArgs a = a_all;
	Gate gate("cast");
#define GO(TOUT) \
	if (cast_quad_ok<TIN, TOUT>(a)) { \
		const int groups = a.ne / 4; \
		dim3 qgrid((groups + 255) / 256, rows_grid((groups + 255) / 256, a.height), 1); \
		hipLaunchKernelGGL((cast_quad_kernel<TIN, TOUT>), qgrid, block, 0, stream(), a); \
		/* the last ne % 4 elements of every row: the element kernel on what is left */ \
		a.in += (size_t) groups * 4 * sizeof(TIN); \
		a.out += (size_t) groups * 4 * sizeof(TOUT); \
		a.ne -= groups * 4; \
		/* (a.take stays: x = e / take, b = e % take address the same elements from the moved base \
		   only when the split falls on a pixel boundary or take == bands == the whole row's period; \
		   with every band taken ie == oe == e, so any split is right) */ \
	} \
	if (a.ne > 0) { \
		dim3 grid((a.ne + 255) / 256, rows_grid((a.ne + 255) / 256, a.height), 1); \
		hipLaunchKernelGGL((cast_kernel<TIN, TOUT>), grid, block, 0, stream(), a); \
	} \
	break;
	switch (out_format) {
	case VIPS_HIP_FORMAT_UCHAR: GO(unsigned char)
	case VIPS_HIP_FORMAT_CHAR: GO(signed char)
	case VIPS_HIP_FORMAT_USHORT: GO(unsigned short)
	case VIPS_HIP_FORMAT_SHORT: GO(short)
	case VIPS_HIP_FORMAT_UINT: GO(unsigned int)
	case VIPS_HIP_FORMAT_INT: GO(int)
	case VIPS_HIP_FORMAT_FLOAT: GO(float)
	case VIPS_HIP_FORMAT_DOUBLE: GO(double)
	default:
		error("cast", "unsupported band format %d", out_format);
		return -1;
	}
#undef GO
	VH_CHECK(hipGetLastError());
	return 0;
}

int band_cast(const VipsHipRegion *in, int in_first, const VipsHipRegion *out, int out_first, int take)
{
	if (ensure_init())
		return -1;
	if (check_region("cast", in) || check_region("cast", out))
		return -1;
	if (format_iscomplex(in->format) || format_iscomplex(out->format)) {
		error("cast", "complex formats are outside the HIP path");
		return -1;
	}
	if (out->left < in->left || out->top < in->top ||
		out->left + out->width > in->left + in->width ||
		out->top + out->height > in->top + in->height) {
		error("cast", "input region too small");
		return -1;
	}
	if (in_first < 0 || take <= 0 || in_first + take > in->bands || out_first < 0 ||
		out_first + take > out->bands) {
		error("cast", "bad band range");
		return -1;
	}
	CastArgs a;
	const int ies = format_sizeof(in->format);
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride +
		(size_t) (out->left - in->left) * in->bands * ies;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.ne = out->width * take;
	a.height = out->height;
	a.in_bands = in->bands;
	a.out_bands = out->bands;
	a.in_first = in_first;
	a.out_first = out_first;
	a.take = take;
	switch (in->format) {
	case VIPS_HIP_FORMAT_UCHAR: return launch_cast_out<unsigned char>(a, out->format);
	case VIPS_HIP_FORMAT_CHAR: return launch_cast_out<signed char>(a, out->format);
	case VIPS_HIP_FORMAT_USHORT: return launch_cast_out<unsigned short>(a, out->format);
	case VIPS_HIP_FORMAT_SHORT: return launch_cast_out<short>(a, out->format);
	case VIPS_HIP_FORMAT_UINT: return launch_cast_out<unsigned int>(a, out->format);
	case VIPS_HIP_FORMAT_INT: return launch_cast_out<int>(a, out->format);
	case VIPS_HIP_FORMAT_FLOAT: return launch_cast_out<float>(a, out->format);
	case VIPS_HIP_FORMAT_DOUBLE: return launch_cast_out<double>(a, out->format);
	default:
		error("cast", "unsupported band format %d", in->format);
		return -1;
	}
}

// ------------------------------------------------------------------ sharpen

struct SharpenArgs {
	const unsigned char *in, *blur;
	unsigned char *out;
	long long in_stride, blur_stride, out_stride;
	int width, height, bands;
	const int *lut;
};

// vips_sharpen_generate, sharpen.c:116-168, on band 0 of a LabS image; the other
// bands are copied (the reference splits them off and joins them back, :274-295).
__global__ void __launch_bounds__(256)
sharpen_kernel(SharpenArgs a)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
		int v1[RU], v2[RU];
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = min(y0 + r, a.height - 1);
			v1[r] = ((const short *) (a.in + (long long) y * a.in_stride))[(long long) x * a.bands];
			v2[r] = ((const short *) (a.blur + (long long) y * a.blur_stride))[x];
		}
#pragma unroll
		for (int r = 0; r < RU; r++) {
			const int y = y0 + r;
			if (y < a.height) {
				const short *p1 = (const short *) (a.in + (long long) y * a.in_stride) + (long long) x * a.bands;
				short *q = (short *) (a.out + (long long) y * a.out_stride) + (long long) x * a.bands;
				const int diff = (v1[r] & 0x7fff) - (v2[r] & 0x7fff);
				int out = v1[r] + a.lut[diff + 32768];
				out = min(max(out, 0), 32767);
				q[0] = (short) out;
				for (int b = 1; b < a.bands; b++)
					q[b] = p1[b];
			}
		}
	}
}


// ------------------------------------------------- vips_sharpen on sRGB uchar, one kernel
//
// vips_sharpen (sharpen.c:171-302) on a 3-band uchar sRGB image is six operations on a small
// image -- colourspace(LABS), extract L, convsep of a 3..5-tap integer gaussian (two passes),
// the LUT step (sharpen.c:116-168), colourspace(sRGB) -- each a kernel that is over before the
// launch latency is (BASELINE config 4: 0.06 ms of the 0.14 ms per thumbnail).  Here a block
// owns a 64 x 32 (or 64 x 16) pixel tile: it converts the tile and its halo to LabS into LDS (route code of
// colour_device.h, image edges clamped = the embed of the convolution), runs the horizontal
// and the vertical pass on L in LDS with the convi C-path arithmetic and its rounding to short
// between the passes ((sum + scale / 2) / scale, C division: convi.c:698-716), applies the LUT
// and converts back: one read of the image, one write.
constexpr int SF_TW = 64, SF_MAXHALF = 2;
constexpr int SF_RW = SF_TW + 2 * SF_MAXHALF;

constexpr int SF_MAXB = 64; // images per launch

// the images of a launch (blockIdx.z), read where they lie in the kernarg segment
struct SharpenFusedPtrs {
	const unsigned char *in[SF_MAXB];
	unsigned char *out[SF_MAXB];
};

struct SharpenFusedArgs {
	long long in_stride, out_stride;
	int width, height;
	int n, half;          // blur taps, n / 2
	int coef[2 * SF_MAXHALF + 1];
	int scale, rounding;
	unsigned int magic;   // n / scale = (t + ((n - t) >> 1)) >> shift with t = mulhi(magic, n), 0 <= n < 2^32
	int shift;            // (Granlund & Montgomery's round-up multiplier; scale > 1)
	const int *lut;       // 65536 ints (sharpen.c:230-257)
	int zero_lo, zero_hi; // SKIP: lut[diff + 32768] == 0 for zero_lo <= diff <= zero_hi (empty: lo > hi)
	// SKIP: n / scale == ((n << sh24) * m24) >> 32 for every n a blur sum can be, both factors below 2^24 (the
	// full-rate v_mul_hi_u32_u24 instead of the quarter-rate v_mul_hi_u32; the host tried every n); sh24 < 0: no
	unsigned int m24;
	int sh24;
	// SKIP on large images (round 6): a tile whose list is longer than defer_threshold is left to the all-in-LDS
	// kernel -- its 64 x 64 tile goes on a device-side list (defer[0] the count, then one flag per such tile, then
	// the list) that sharpen_quad_u8_kernel walks afterwards; nullptr: every tile is finished here
	int *defer;
	int defer_threshold, q_tiles_x, q_tiles_y;
	int defer_raw;     // > 0: the tile is judged by its middle row's raw bytes (green of neighbours more than this apart)
	ColourTables tables;
};

static __device__ __forceinline__ int sf_convi_fin(int sum, const SharpenFusedArgs &a)
{
	// ((sum + rounding) / scale) with C (truncating) division, offset 0, clip to short.  The
	// divisor is the same for the whole launch: a multiply-high and two shifts on the magnitude
	// instead of the ~30 instructions of a 32-bit division
	const int x = sum + a.rounding;
	const unsigned int n = (unsigned int) (x < 0 ? -x : x);
	unsigned int m = n;
	if (a.scale != 1) {
		const unsigned int t = __umulhi(a.magic, n);
		m = (t + ((n - t) >> 1)) >> a.shift;
	}
	const int q = x < 0 ? -(int) m : (int) m;
	return min(max(q, -32768), 32767);
}

// the same for a sum that is not negative (the SKIP kernel: L >= 0, coefficients >= 0): n = sum + rounding < 2^31
static __device__ __forceinline__ unsigned int sf_div_nonneg(unsigned int n, const SharpenFusedArgs &a)
{
	// (the SKIP kernel is only launched with sh24 >= 0)
	unsigned int m;
	VH_MUL_HI_U24(m, n << a.sh24, a.m24);
	return min(m, 32767u);
}

// SF_TH = rows of a tile: 16 rows per pass of the block's 256 threads (4 pixels each), SF_TH / 16
// passes; the halo ring and the tables a block loads are shared by all of them
//
// SKIP (round 6): the pixels sharpen leaves alone skip the way back.  With the default parameters (m1 = 0) the
// LUT is 0 for |L - blur| < x1 = 2 L units (sharpen.c:230-257) -- every pixel of a smooth region: 95.7 % of the
// pixels of BASELINE config 4's thumbnails, most of a photograph.  Such a pixel leaves vips_sharpen as
// sRGB(LabS(pixel)) with L, a, b untouched, and sRGB -> LabS -> sRGB is the IDENTITY on all 2^24 colours: checked
// with the compiled reference on the host (tests/test_oracle_conv_colour.py) and, for the device's own tables and
// these very functions, exhaustively on the device before the first launch (sharpen_identity_kernel: the kernel is
// only selected when the count of colours that do not come back is 0).  So: L alone (not a, b: a third of the
// forward path's table reads) for the tile and its halo, the blur, the test of L - blur against the LUT's zero
// window (no gather); the tile's INPUT bytes wait in LDS as its output; the few pixels whose LUT entry is not 0
// are appended to a list (LDS atomic), taken through the full forward and backward path DENSELY (a wave of list
// entries, not a wave of tile pixels 4 % of which are live) and patched into the staged bytes.
// (NT = taps the SKIP kernel's blur passes compute: 3 or 5; the other kernel always runs five, zeros included)
// (a flag another kernel of the stream wrote: the same for every lane)
static __device__ __forceinline__ int uniform_flag(const int *p)
{
	return __builtin_amdgcn_readfirstlane(*p);
}

// Large images, before sharpen_fused_u8_kernel<32, true>: which tiles are noise or dense edges -- the all-in-LDS
// kernel's anyway -- judged from the raw bytes of ONE row: a THREAD per 64 x 32 tile walks its middle row and counts
// the horizontal neighbours whose green differs by more than `raw`; more than half: the tile's 64 x 64 parent goes on
// the device-side list (SharpenFusedArgs::defer) and both kernels know.  A heuristic for speed only: both kernels make
// the same pixels.  The list's slots are handed out a wave at a time (one atomic on the counter per wave that has any):
// round 6's first form judged inside the skip kernel, a block a tile, and what a noisy 8192 x 8192 image paid 0.19 ms
// for was 16 384 atomics on ONE counter -- measured again with a wave a tile in this kernel: the same 0.19 ms.
__global__ void __launch_bounds__(256)
sharpen_survey_kernel(SharpenFusedPtrs ptrs_by_value, int width, int height, long long in_stride, int raw, int *defer,
	int q_tiles_x, int q_tiles_y, int tiles_x, int tiles_y)
{
	__shared__ int s_base[4];
	(void) ptrs_by_value;
	typedef const unsigned long long __attribute__((address_space(4))) *KernargPtrs;
	const KernargPtrs kp = (KernargPtrs) __builtin_amdgcn_kernarg_segment_ptr();
	typedef const unsigned char __attribute__((address_space(1))) *GlobalIn;
	const GlobalIn in = (GlobalIn) kp[blockIdx.z];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int tile = (int) blockIdx.x * 256 + (int) threadIdx.x;
	const bool valid = tile < tiles_x * tiles_y;
	const int tc = valid ? tile : 0;
	const int ty = tc / tiles_x, tx = tc - ty * tiles_x;
	const int x0 = tx * SF_TW, y0 = ty * 32;
	const int ym = min(y0 + 16, height - 1);
	const GlobalIn row = in + (long long) ym * in_stride;
	int count = 0;
	int prev = row[3 * x0 + 1];
	for (int j = 1; j <= SF_TW; j++) {
		const int g = row[3 * min(x0 + j, width - 1) + 1];
		count += (x0 + j - 1 < width && abs(g - prev) > raw) ? 1 : 0;
		prev = g;
	}
	const int per = q_tiles_x * q_tiles_y;
	const int qt = (int) blockIdx.z * per + (y0 / 64) * q_tiles_x + tx;
	const int total = per * (int) gridDim.z;
	bool fresh = false;
	if (valid && count > SF_TW / 2)
		fresh = atomicExch(defer + 1 + qt, 1) == 0; // (a 64 x 64 parent has two tiles: the first one lists it)
	const unsigned long long mask = __builtin_amdgcn_ballot_w64(fresh);
	const int n = __builtin_popcountll(mask);
	const int rank = __builtin_popcountll(mask & ((1ULL << lane) - 1ULL));
	if (lane == 0 && n)
		s_base[wave] = atomicAdd(defer, n);
	__syncthreads();
	if (fresh)
		defer[1 + total + s_base[wave] + rank] = qt;
}

template <int SF_TH, bool SKIP, int NT = 2 * SF_MAXHALF + 1>
__global__ void __launch_bounds__(256)
sharpen_fused_u8_kernel(SharpenFusedPtrs ptrs_by_value, SharpenFusedArgs a)
{
	constexpr int SF_RH = SF_TH + 2 * SF_MAXHALF;
	constexpr int NCH = SKIP ? 1 : 3;
	constexpr int OUT_DW = SF_TW * 3 / 4; // dwords per staged row
	__shared__ __attribute__((aligned(16))) short s_lab[SF_RH][SF_RW][NCH];
	__shared__ __attribute__((aligned(16))) short s_h[SF_RH][SF_TW];
	// the two 8-bit colour tables in LDS: six of a pixel's table reads stay off the vector memory path
	__shared__ float s_v2Y[256];
	__shared__ int s_Y2v[260];
	__shared__ unsigned int s_out[SKIP ? SF_TH * OUT_DW : 1];  // SKIP: the tile's bytes, input until patched
	__shared__ unsigned int s_list[SKIP ? SF_TH * SF_TW : 1];  // SKIP: (LUT index << 11) | row << 6 | column
	__shared__ int s_count;
	(void) ptrs_by_value;
	typedef const unsigned long long __attribute__((address_space(4))) *KernargPtrs;
	const KernargPtrs kp = (KernargPtrs) __builtin_amdgcn_kernarg_segment_ptr();
	// (pointers made from integers are generic to the compiler: say they are global)
	typedef const unsigned char __attribute__((address_space(1))) *GlobalIn;
	typedef unsigned char __attribute__((address_space(1))) *GlobalOut;
	const GlobalIn in = (GlobalIn) kp[blockIdx.z];
	const GlobalOut out = (GlobalOut) kp[SF_MAXB + blockIdx.z];
	const int t = threadIdx.x;
	const int x0 = blockIdx.x * SF_TW, y0 = blockIdx.y * SF_TH;
	const int h = a.half;
	const int rw = SF_TW + 2 * h, rh = SF_TH + 2 * h;
	if constexpr (SKIP) {
		if (a.defer && a.defer_raw > 0) {
			// 0' (large images only): sharpen_survey_kernel has judged every tile already; one that is on the list is
			// the all-in-LDS kernel's -- leave before the block has touched anything
			const int per = a.q_tiles_x * a.q_tiles_y;
			const int qt = (int) blockIdx.z * per + (y0 / 64) * a.q_tiles_x + (int) blockIdx.x;
			if (uniform_flag(a.defer + 1 + qt))
				return;
		}
	}
	s_v2Y[threadIdx.x] = a.tables.v2Y_8[threadIdx.x];
	s_Y2v[threadIdx.x] = a.tables.Y2v_8[threadIdx.x];
	if (threadIdx.x == 0) {
		s_Y2v[256] = a.tables.Y2v_8[256];
		s_count = 0;
	}
	__syncthreads();

	// ring pixel idx -> its place in the region (the frame of h pixels around the tile)
	auto ring_place = [&](int idx, int &ry, int &rx) {
		if (idx < 2 * rw * h) {
			// rows above and below (h <= 2: at most four rows, no division)
			const int r = (idx >= rw) + (idx >= 2 * rw) + (idx >= 3 * rw);
			rx = idx - r * rw;
			ry = r < h ? r : SF_TH + r;
		}
		else {
			// columns left and right of the tile's rows (2 h = 2 or 4 = 1 << h)
			const int k = idx - 2 * rw * h;
			const int r = k >> h, c = k & (2 * h - 1);
			ry = h + r;
			rx = c < h ? c : SF_TW + c;
		}
	};
	if constexpr (SKIP) {
		if (a.defer && a.defer_raw <= 0) {
			// 0 (large images only; $VIPS_HIP_SHARPEN_DEFER_RAW=0). A tile of noise or dense edges goes to the all-in-LDS kernel anyway: find out
			// from ONE row before paying for all of them -- L of the tile's middle row and the rows either side, the
			// blur of the middle row, its pixels outside the LUT's zero window counted; more than half: the tile is
			// deferred here and now.  (A heuristic for speed only: both kernels make the same pixels.)
			const int ym = min(y0 + SF_TH / 2, a.height - 1);
			const int nrow = 2 * h + 1;
			for (int idx = t; idx < nrow * rw; idx += 256) {
				const int rr = idx / rw, rx = idx - rr * rw;
				const int x = min(max(x0 + rx - h, 0), a.width - 1);
				const int y = min(max(ym + rr - h, 0), a.height - 1);
				unsigned int off;
				VH_MAD_U24(off, (unsigned int) y, (unsigned int) a.in_stride, (unsigned int) (3 * x));
				const GlobalIn p = in + off;
				short L, A = 0, B = 0;
				srgb8_to_labs<false>(a.tables, s_v2Y, p[0], p[1], p[2], L, A, B);
				s_lab[rr][rx][0] = L;
			}
			__syncthreads();
			for (int idx = t; idx < nrow * SF_TW; idx += 256) {
				const int rr = idx >> 6, cx = idx & 63;
				unsigned int sum = (unsigned int) a.rounding;
#pragma unroll
				for (int k = 0; k < NT; k++)
					VH_MAD_U24(sum, (unsigned int) (int) s_lab[rr][cx + k][0], (unsigned int) a.coef[k], sum);
				s_h[rr][cx] = (short) sf_div_nonneg(sum, a);
			}
			__syncthreads();
			if (t < SF_TW) {
				unsigned int sum = (unsigned int) a.rounding;
#pragma unroll
				for (int k = 0; k < NT; k++)
					VH_MAD_U24(sum, (unsigned int) (int) s_h[k][t], (unsigned int) a.coef[k], sum);
				const int blur = (int) sf_div_nonneg(sum, a);
				const int v1 = s_lab[h][t + h][0];
				const int diff = (v1 & 0x7fff) - (blur & 0x7fff);
				if ((diff < a.zero_lo || diff > a.zero_hi) && x0 + t < a.width)
					atomicAdd(&s_count, 1);
			}
			__syncthreads();
			const int sampled = s_count;
			__syncthreads();
			if (sampled > SF_TW / 2) {
				if (t == 0) {
					const int per = a.q_tiles_x * a.q_tiles_y;
					const int qt = (int) blockIdx.z * per + (y0 / 64) * a.q_tiles_x + (int) blockIdx.x;
					const int total = per * (int) gridDim.z;
					if (atomicExch(a.defer + 1 + qt, 1) == 0)
						a.defer[1 + total + atomicAdd(a.defer, 1)] = qt;
				}
				return;
			}
			if (t == 0)
				s_count = 0;
			// (the barrier in front of sweep 3's LDS writes orders this; s_lab / s_h rows 0 .. 2 are written anew)
			__syncthreads();
		}
		// 1 (SKIP). L of the tile and its ring, IN THREE SWEEPS over the thread's pixels (4 per 16 rows of the tile
		// and up to two of the ring): every pixel load, then every table index and its gather, then the
		// interpolations -- two memory latencies a block instead of two per pixel group (the block's life is those
		// latencies: with the conversions one after the other the kernel ran no faster for 40 % fewer instructions)
		constexpr int NPASS = SF_TH / 16, NRING = 2, NPX = 4 * NPASS + NRING;
		const int ring = rw * rh - SF_TW * SF_TH;
		const int quad = t & 15;
		const bool words = !((((unsigned long long) in) | (unsigned long long) a.in_stride) & 3) && x0 + 4 * quad + 4 <= a.width;
		unsigned int w[NPASS][3];
		unsigned int rpx[NRING][3];
		int rry[NRING], rrx[NRING];
#pragma unroll
		for (int pass = 0; pass < NPASS; pass++) {
			const int row = (t >> 4) + 16 * pass;
			const int y = min(y0 + row, a.height - 1);
			const int x = x0 + 4 * quad;
			unsigned int off;
			VH_MAD_U24(off, (unsigned int) y, (unsigned int) a.in_stride, 0u);
			const GlobalIn line = in + off;
			if (words) {
				const unsigned int __attribute__((address_space(1))) *p4 =
					(const unsigned int __attribute__((address_space(1))) *) (line + 3 * x);
#pragma unroll
				for (int k = 0; k < 3; k++)
					w[pass][k] = p4[k];
			}
			else {
				unsigned char px[12];
#pragma unroll
				for (int m = 0; m < 4; m++) {
					const GlobalIn p = line + 3 * min(x + m, a.width - 1);
					px[3 * m] = p[0];
					px[3 * m + 1] = p[1];
					px[3 * m + 2] = p[2];
				}
#pragma unroll
				for (int k = 0; k < 3; k++)
					w[pass][k] = (unsigned) px[4 * k] | ((unsigned) px[4 * k + 1] << 8) | ((unsigned) px[4 * k + 2] << 16) |
						((unsigned) px[4 * k + 3] << 24);
			}
		}
#pragma unroll
		for (int j = 0; j < NRING; j++) {
			const int idx = min(t + 256 * j, ring - 1); // (a lane beyond the ring repeats its last pixel: same value, same place)
			ring_place(max(idx, 0), rry[j], rrx[j]);
			const int x = min(max(x0 + rrx[j] - h, 0), a.width - 1);
			const int y = min(max(y0 + rry[j] - h, 0), a.height - 1);
			unsigned int off;
			VH_MAD_U24(off, (unsigned int) y, (unsigned int) a.in_stride, (unsigned int) (3 * x));
			const GlobalIn p = in + off;
			rpx[j][0] = p[0];
			rpx[j][1] = p[1];
			rpx[j][2] = p[2];
		}
		// sweep 2: Y of every pixel, its table index and fraction, the gather
		float fr[NPX];
		float2 pair[NPX];
		auto index_of = [&](int n, unsigned int r8, unsigned int g8, unsigned int b8) {
			Px v;
			v.a = s_v2Y[r8];
			v.b = s_v2Y[g8];
			v.c = s_v2Y[b8];
			v = step_scRGB2XYZ(v);
			const int i = cbrt_index_finite<1>(v.b, fr[n]);
			__builtin_memcpy(&pair[n], a.tables.cbrt + i, sizeof(float2));
		};
#pragma unroll
		for (int pass = 0; pass < NPASS; pass++)
#pragma unroll
			for (int m = 0; m < 4; m++) {
				// byte 3 m + c of the 12: dword (3 m + c) / 4, byte (3 m + c) % 4
				auto byte_of = [&](int c) -> unsigned int { return (w[pass][(3 * m + c) >> 2] >> (8 * ((3 * m + c) & 3))) & 0xffu; };
				index_of(4 * pass + m, byte_of(0), byte_of(1), byte_of(2));
			}
#pragma unroll
		for (int j = 0; j < NRING; j++)
			index_of(4 * NPASS + j, rpx[j][0], rpx[j][1], rpx[j][2]);
		// sweep 3: interpolate, L, LabS (the operations of srgb8_to_labs<false>, in its order)
		auto finish = [&](int n) -> short {
			const float cby = cbrt_finish(pair[n], fr[n]);
			return lab2labs_finite(__fsub_rn(__fmul_rn(116.0F, cby), 16.0F), 32767.0 / 100.0, 0.0);
		};
#pragma unroll
		for (int pass = 0; pass < NPASS; pass++) {
			const int row = (t >> 4) + 16 * pass;
#pragma unroll
			for (int m = 0; m < 4; m++)
				s_lab[row + h][4 * quad + m + h][0] = finish(4 * pass + m);
#pragma unroll
			for (int k = 0; k < 3; k++)
				s_out[row * OUT_DW + 3 * quad + k] = w[pass][k];
		}
#pragma unroll
		for (int j = 0; j < NRING; j++)
			if (t + 256 * j < ring)
				s_lab[rry[j]][rrx[j]][0] = finish(4 * NPASS + j);
	}
	else {
	// 1. the tile and its halo: sRGB uchar -> LabS.
	// 1a. the tile itself, 4 pixels per thread (the mapping of step 3): 12 bytes as three dwords
	// when the row allows it, the four conversions side by side (their table reads overlap)
	for (int pass = 0; pass < SF_TH / 16; pass++) {
		const int row = (t >> 4) + 16 * pass, quad = t & 15;
		const int y = min(y0 + row, a.height - 1);
		const int x = x0 + 4 * quad;
		const GlobalIn line = in + (long long) y * a.in_stride;
		unsigned char px[12];
		if (x + 4 <= a.width && !((((unsigned long long) in) | (unsigned long long) a.in_stride) & 3)) {
			const unsigned int __attribute__((address_space(1))) *p4 =
				(const unsigned int __attribute__((address_space(1))) *) (line + 3LL * x);
#pragma unroll
			for (int w = 0; w < 3; w++) {
				const unsigned int v = p4[w];
				px[4 * w] = (unsigned char) v;
				px[4 * w + 1] = (unsigned char) (v >> 8);
				px[4 * w + 2] = (unsigned char) (v >> 16);
				px[4 * w + 3] = (unsigned char) (v >> 24);
			}
		}
		else {
#pragma unroll
			for (int m = 0; m < 4; m++) {
				const GlobalIn p = line + 3LL * min(x + m, a.width - 1);
				px[3 * m] = p[0];
				px[3 * m + 1] = p[1];
				px[3 * m + 2] = p[2];
			}
		}
#pragma unroll
		for (int m = 0; m < 4; m++) {
			short L, A = 0, B = 0;
			srgb8_to_labs<!SKIP>(a.tables, s_v2Y, px[3 * m], px[3 * m + 1], px[3 * m + 2], L, A, B);
			s_lab[row + h][4 * quad + m + h][0] = L;
			if constexpr (!SKIP) {
				s_lab[row + h][4 * quad + m + h][1] = A;
				s_lab[row + h][4 * quad + m + h][2] = B;
			}
		}
		if constexpr (SKIP) {
#pragma unroll
			for (int w = 0; w < 3; w++)
				s_out[row * OUT_DW + 3 * quad + w] = (unsigned) px[4 * w] | ((unsigned) px[4 * w + 1] << 8) |
					((unsigned) px[4 * w + 2] << 16) | ((unsigned) px[4 * w + 3] << 24);
		}
	}
	// 1b. the ring of h pixels around it (image edges clamped): only L is blurred, so only L
	{
		const int ring = rw * rh - SF_TW * SF_TH;
		for (int idx = t; idx < ring; idx += 256) {
			int ry, rx;
			ring_place(idx, ry, rx);
			const int x = min(max(x0 + rx - h, 0), a.width - 1);
			const int y = min(max(y0 + ry - h, 0), a.height - 1);
			GlobalIn p;
			if constexpr (SKIP) {
				// (the host keeps height * stride below 2^31 and both factors below 2^24 for this kernel)
				unsigned int off;
				VH_MAD_U24(off, (unsigned int) y, (unsigned int) a.in_stride, (unsigned int) (3 * x));
				p = in + off;
			}
			else
				p = in + (long long) y * a.in_stride + 3LL * x;
			short L, A = 0, B = 0;
			srgb8_to_labs<false>(a.tables, s_v2Y, p[0], p[1], p[2], L, A, B);
			s_lab[ry][rx][0] = L;
		}
	}
		}
	__syncthreads();
	// 2. horizontal pass on L (all rows of the region, the tile's columns)
	if constexpr (SKIP) {
		// four neighbouring outputs per thread from ONE read of their 8 shorts (rows are 8-byte aligned); L and the
		// coefficients are not negative here (host), so a sum is not and its division is the multiply-high alone
		for (int item = t; item < rh * (SF_TW / 4); item += 256) {
			const int ry = item >> 4, q4 = (item & 15) * 4;
			const uint2 w0 = *reinterpret_cast<const uint2 *>(&s_lab[ry][q4][0]);
			const uint2 w1 = *reinterpret_cast<const uint2 *>(&s_lab[ry][q4 + 4][0]);
			const int v[8] = { (int) (w0.x & 0xffffu), (int) (w0.x >> 16), (int) (w0.y & 0xffffu), (int) (w0.y >> 16),
				(int) (w1.x & 0xffffu), (int) (w1.x >> 16), (int) (w1.y & 0xffffu), (int) (w1.y >> 16) };
			unsigned int o[4];
#pragma unroll
			for (int m = 0; m < 4; m++) {
				unsigned int sum = (unsigned int) a.rounding;
#pragma unroll
				for (int k = 0; k < NT; k++)
					VH_MAD_U24(sum, (unsigned int) v[m + k], (unsigned int) a.coef[k], sum);
				o[m] = sf_div_nonneg(sum, a);
			}
			*reinterpret_cast<uint2 *>(&s_h[ry][q4]) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
		}
	}
	else {
		for (int idx = t; idx < rh * SF_TW; idx += 256) {
			const int ry = idx / SF_TW, cx = idx - ry * SF_TW;
			// always five taps (coefficients beyond the mask are zero, the reads stay inside the arrays)
			int sum = 0;
#pragma unroll
			for (int k = 0; k < 2 * SF_MAXHALF + 1; k++)
				sum += a.coef[k] * (int) s_lab[ry][cx + k][0];
			s_h[ry][cx] = (short) sf_convi_fin(sum, a);
		}
	}
	__syncthreads();
	// 3. vertical pass, the LUT, back to sRGB: 4 pixels per thread, three dword stores
	for (int pass = 0; pass < SF_TH / 16; pass++) {
		const int row = (t >> 4) + 16 * pass, quad = t & 15;
		const int y = y0 + row;
		if constexpr (SKIP) {
			if (y >= a.height)
				continue;
			// the thread's four columns of the 2 h + 1 rows above and below: one 8-byte read per row
			unsigned int sum[4] = { (unsigned int) a.rounding, (unsigned int) a.rounding, (unsigned int) a.rounding,
				(unsigned int) a.rounding };
#pragma unroll
			for (int k = 0; k < NT; k++) {
				const uint2 w = *reinterpret_cast<const uint2 *>(&s_h[row + k][4 * quad]);
				const unsigned int ck = (unsigned int) a.coef[k];
				VH_MAD_U24(sum[0], w.x & 0xffffu, ck, sum[0]);
				VH_MAD_U24(sum[1], w.x >> 16, ck, sum[1]);
				VH_MAD_U24(sum[2], w.y & 0xffffu, ck, sum[2]);
				VH_MAD_U24(sum[3], w.y >> 16, ck, sum[3]);
			}
			// inside the LUT's zero window: the staged input bytes are the result.  The others go on the list with
			// their LUT index (the LUT itself is read in step 4: no memory latency here), one LDS atomic per thread
			unsigned int entry[4];
			int n_mine = 0;
#pragma unroll
			for (int m = 0; m < 4; m++) {
				const int cx = 4 * quad + m;
				const int blur = (int) sf_div_nonneg(sum[m], a);
				const int v1 = s_lab[row + h][cx + h][0];
				const int diff = (v1 & 0x7fff) - (blur & 0x7fff);
				const bool live = (diff < a.zero_lo || diff > a.zero_hi) && x0 + cx < a.width;
				entry[m] = live ? ((unsigned) (diff + 32768) << 11) | (unsigned) (row * SF_TW + cx) : 0xffffffffu;
				n_mine += live ? 1 : 0;
			}
			if (n_mine > 0) {
				int at = atomicAdd(&s_count, n_mine);
#pragma unroll
				for (int m = 0; m < 4; m++)
					if (entry[m] != 0xffffffffu)
						s_list[at++] = entry[m];
			}
			continue;
		}
		if (y < a.height) {
			unsigned char o[12];
#pragma unroll
			for (int m = 0; m < 4; m++) {
				const int cx = 4 * quad + m;
				int sum = 0;
#pragma unroll
				for (int k = 0; k < 2 * SF_MAXHALF + 1; k++)
					sum += a.coef[k] * (int) s_h[row + k][cx];
				const int blur = sf_convi_fin(sum, a);
				const int v1 = s_lab[row + h][cx + h][0];
				const int diff = (v1 & 0x7fff) - (blur & 0x7fff);
				int sharp = v1 + a.lut[diff + 32768];
				sharp = min(max(sharp, 0), 32767);
				labs_to_srgb8(s_Y2v, sharp, s_lab[row + h][cx + h][1], s_lab[row + h][cx + h][2], o[3 * m],
					o[3 * m + 1], o[3 * m + 2]);
			}
			const int x = x0 + 4 * quad;
			const GlobalOut dst = out + (long long) y * a.out_stride + 3LL * x;
			if (x + 4 <= a.width && !(((uintptr_t) dst) & 3)) {
				unsigned int __attribute__((address_space(1))) *d4 = (unsigned int __attribute__((address_space(1))) *) dst;
#pragma unroll
				for (int w = 0; w < 3; w++)
					d4[w] = (unsigned) o[4 * w] | ((unsigned) o[4 * w + 1] << 8) | ((unsigned) o[4 * w + 2] << 16) |
						((unsigned) o[4 * w + 3] << 24);
			}
			else {
				for (int m = 0; m < 12 && x + m / 3 < a.width; m++)
					dst[m] = o[m];
			}
		}
	}
	if constexpr (SKIP) {
		// 4. the listed pixels, densely: forward for a, b, back from (sharpened L, a, b), patched into the stage
		__syncthreads();
		const int count = s_count;
		if (a.defer && count > a.defer_threshold) {
			// mostly edges or noise: the kernel with every table in LDS does such a tile faster (the whole 64 x 64
			// tile this one is a half of; its other half may finish here: the same pixels either way)
			if (t == 0) {
				const int per = a.q_tiles_x * a.q_tiles_y;
				const int qt = (int) blockIdx.z * per + (y0 / 64) * a.q_tiles_x + (int) blockIdx.x;
				const int total = per * (int) gridDim.z;
				if (atomicExch(a.defer + 1 + qt, 1) == 0)
					a.defer[1 + total + atomicAdd(a.defer, 1)] = qt;
			}
			return;
		}
		unsigned char *bytes = reinterpret_cast<unsigned char *>(s_out);
		for (int i = t; i < count; i += 256) {
			const unsigned int e = s_list[i];
			const int pos = (int) (e & 2047u);
			const int add = a.lut[e >> 11]; // sharpen.c:116-168: index (v1 & 0x7fff) - (blur & 0x7fff) + 32768
			if (add == 0)
				continue; // (a zero of the LUT outside the window around 0: the staged bytes stand)
			const int row = pos / SF_TW, cx = pos - row * SF_TW;
			const int sharp = min(max((int) s_lab[row + h][cx + h][0] + add, 0), 32767);
			unsigned char *p = bytes + row * (OUT_DW * 4) + 3 * cx;
			short L, A, B;
			srgb8_to_labs<true>(a.tables, s_v2Y, p[0], p[1], p[2], L, A, B);
			unsigned char r8, g8, b8;
			labs_to_srgb8(s_Y2v, sharp, A, B, r8, g8, b8);
			p[0] = r8;
			p[1] = g8;
			p[2] = b8;
		}
		__syncthreads();
		// 5. the stage leaves: 4 pixels per thread, three dword stores
		for (int pass = 0; pass < SF_TH / 16; pass++) {
			const int row = (t >> 4) + 16 * pass, quad = t & 15;
			const int y = y0 + row;
			if (y >= a.height)
				continue;
			const int x = x0 + 4 * quad;
			const GlobalOut dst = out + (long long) y * a.out_stride + 3LL * x;
			const unsigned int *src = s_out + row * OUT_DW + 3 * quad;
			if (x + 4 <= a.width && !(((uintptr_t) dst) & 3)) {
				unsigned int __attribute__((address_space(1))) *d4 = (unsigned int __attribute__((address_space(1))) *) dst;
#pragma unroll
				for (int w = 0; w < 3; w++)
					d4[w] = src[w];
			}
			else {
				const unsigned char *sb = reinterpret_cast<const unsigned char *>(src);
				for (int m = 0; m < 12 && x + m / 3 < a.width; m++)
					dst[m] = sb[m];
			}
		}
	}
}

// sRGB -> LabS -> sRGB with the functions above on every one of the 2^24 colours: the count of colours that do not
// come back.  0 is what lets sharpen_fused_u8_kernel<*, true> hand a pixel whose LUT entry is 0 through untouched.
__global__ void __launch_bounds__(256)
sharpen_identity_kernel(ColourTables tables, unsigned int *bad)
{
	__shared__ float s_v2Y[256];
	__shared__ int s_Y2v[260];
	s_v2Y[threadIdx.x] = tables.v2Y_8[threadIdx.x];
	s_Y2v[threadIdx.x] = tables.Y2v_8[threadIdx.x];
	if (threadIdx.x == 0)
		s_Y2v[256] = tables.Y2v_8[256];
	__syncthreads();
	const unsigned int c = blockIdx.x * 256u + threadIdx.x;
	const int r = (int) (c >> 16), g = (int) ((c >> 8) & 255u), b = (int) (c & 255u);
	short L, A, B;
	srgb8_to_labs<true>(tables, s_v2Y, r, g, b, L, A, B);
	unsigned char r8, g8, b8;
	// (the kernel's clip of L + 0 to 0 .. 32767 is part of what is checked)
	labs_to_srgb8(s_Y2v, min(max((int) L, 0), 32767), A, B, r8, g8, b8);
	if (r8 != r || g8 != g || b8 != b)
		atomicAdd(bad, 1u);
}

// ------------------------------------------------- ... with every table in LDS (round 5)
//
// sharpen_fused_u8 reads four tables per pixel through global memory (XYZ2Lab's 400 KB cube-root table three
// times, sharpen's 256 KB LUT once) and a wave's gather from an L2-resident table takes ~146 cycles of its CU's
// L1 fill path (tools/gather_probe.hip): 4 x 146 cycles per 64 pixels IS the kernel's 1.06 ms on 8192^2.  Here a
// block of 1024 threads copies everything it looks up into LDS once and then walks tiles of 64 x 64 pixels:
//   * the cube-root table in cbrt_quad.h's form (57 KB, ~27 instructions and two LDS reads per pair);
//   * the part of sharpen's LUT that is not constant (sharpen.c:230-257 is flat outside a few thousand entries
//     round the middle: the host finds the window; a LUT with a wider one takes the older kernel);
//   * the two 8-bit sRGB tables.
// Same steps, same roundings as the older kernel (colour_device.h); one block per CU (106 KB of LDS, 16 waves).
constexpr int SQ_NT = 1024, SQ_T = 64, SQ_R = SQ_T + 2 * SF_MAXHALF;
constexpr int SQ_MAXLUT = 6144; // entries of the LUT's window

struct SharpenQuadArgs {
	SharpenFusedArgs f;
	CbrtQuad cq;
	const short *lut_win; // lut_n entries from index lut_lo
	int lut_lo, lut_n, lut_below, lut_above;
	int tiles_x, tiles_y, n_images;
	const int *defer; // list mode (see SharpenFusedArgs::defer): the tiles sharpen_fused_u8_kernel<*, true> left; else nullptr
};

// XYZ2Lab.c:109-138 on a small finite value: the table pair from LDS
static __device__ __forceinline__ float sq_cbrt(const CbqBlock *blk, const unsigned int *res, float n)
{
	const int i = min(max(vh::cvt_i32(n), 0), CBRT_N - 2);
	const float fi = (float) i;
	float t0, dt;
	cbq_pair(blk, res, i, fi, &t0, &dt);
	return __fadd_rn(t0, __fmul_rn(__fsub_rn(n, fi), dt));
}

template <bool WANT_AB>
static __device__ __forceinline__ void sq_to_labs(const CbqBlock *blk, const unsigned int *res, const float *v2Y, int r, int g,
	int b, short &L, short &A, short &B)
{
	Px v;
	v.a = v2Y[r];
	v.b = v2Y[g];
	v.c = v2Y[b];
	v = step_scRGB2XYZ(v);
	const float cby = sq_cbrt(blk, res, quant_div_finite<1>(__fmul_rn(100000.0f, v.b)));
	L = lab2labs_finite(__fsub_rn(__fmul_rn(116.0F, cby), 16.0F), 32767.0 / 100.0, 0.0);
	if (WANT_AB) {
		const float cbx = sq_cbrt(blk, res, quant_div_finite<0>(__fmul_rn(100000.0f, v.a)));
		const float cbz = sq_cbrt(blk, res, quant_div_finite<2>(__fmul_rn(100000.0f, v.c)));
		A = lab2labs_finite(__fmul_rn(500.0F, __fsub_rn(cbx, cby)), 32768.0 / 128.0, -32768.0);
		B = lab2labs_finite(__fmul_rn(200.0F, __fsub_rn(cby, cbz)), 32768.0 / 128.0, -32768.0);
	}
}

__global__ void __launch_bounds__(SQ_NT)
sharpen_quad_u8_kernel(SharpenFusedPtrs ptrs_by_value, SharpenQuadArgs q)
{
	VH_DYNAMIC_LDS(unsigned int, sq_lds);
	(void) ptrs_by_value;
	const SharpenFusedArgs &a = q.f;
	CbqBlock *const s_blk = reinterpret_cast<CbqBlock *>(sq_lds);
	unsigned int *const s_res = sq_lds + CBQ_BLOCKS * 4;
	float *const s_v2Y = reinterpret_cast<float *>(s_res + CBQ_RES_WORDS);
	int *const s_Y2v = reinterpret_cast<int *>(s_v2Y + 256);
	short *const s_lut = reinterpret_cast<short *>(s_Y2v + 260);
	short *const s_L = s_lut + SQ_MAXLUT + 8;                                   // [SQ_R][SQ_R]
	unsigned int *const s_ab = reinterpret_cast<unsigned int *>(s_L + SQ_R * SQ_R); // [SQ_T][SQ_T]: a | b << 16
	short *const s_h = reinterpret_cast<short *>(s_ab + SQ_T * SQ_T);            // [SQ_R][SQ_T]
	const int t = threadIdx.x;
	for (int i = t; i < CBQ_BLOCKS; i += SQ_NT)
		s_blk[i] = q.cq.blk[i];
	for (int i = t; i < CBQ_RES_WORDS; i += SQ_NT)
		s_res[i] = q.cq.res[i];
	if (t < 256)
		s_v2Y[t] = a.tables.v2Y_8[t];
	if (t < 257)
		s_Y2v[t] = a.tables.Y2v_8[t];
	for (int i = t; i < q.lut_n; i += SQ_NT)
		s_lut[i] = q.lut_win[i];
	typedef const unsigned long long __attribute__((address_space(4))) *KernargPtrs;
	const KernargPtrs kp = (KernargPtrs) __builtin_amdgcn_kernarg_segment_ptr();
	typedef const unsigned char __attribute__((address_space(1))) *GlobalIn;
	typedef unsigned char __attribute__((address_space(1))) *GlobalOut;
	const int h = a.half;
	const int rw = SQ_T + 2 * h;
	const int all_tiles = q.tiles_x * q.tiles_y * q.n_images;
	const int tiles = q.defer ? q.defer[0] : all_tiles;
	const int row = t >> 4, quad = t & 15;
	for (int ti = blockIdx.x; ti < tiles; ti += gridDim.x) {
		const int tile = q.defer ? q.defer[1 + all_tiles + ti] : ti;
		const int img = tile / (q.tiles_x * q.tiles_y), tt = tile - img * (q.tiles_x * q.tiles_y);
		const int ty = tt / q.tiles_x, tx = tt - ty * q.tiles_x;
		const GlobalIn in = (GlobalIn) kp[img];
		const GlobalOut out = (GlobalOut) kp[SF_MAXB + img];
		const int x0 = tx * SQ_T, y0 = ty * SQ_T;
		__syncthreads(); // (the tables are there; the last tile's readers are done)
		// 1a. the tile: 4 pixels per thread -> LabS
		{
			const int y = min(y0 + row, a.height - 1);
			const int x = x0 + 4 * quad;
			const GlobalIn line = in + (long long) y * a.in_stride;
			unsigned char px[12];
			if (x + 4 <= a.width && !((((unsigned long long) in) | (unsigned long long) a.in_stride) & 3)) {
				const unsigned int __attribute__((address_space(1))) *p4 =
					(const unsigned int __attribute__((address_space(1))) *) (line + 3LL * x);
#pragma unroll
				for (int w = 0; w < 3; w++) {
					const unsigned int v = p4[w];
					px[4 * w] = (unsigned char) v;
					px[4 * w + 1] = (unsigned char) (v >> 8);
					px[4 * w + 2] = (unsigned char) (v >> 16);
					px[4 * w + 3] = (unsigned char) (v >> 24);
				}
			}
			else {
#pragma unroll
				for (int m = 0; m < 4; m++) {
					const GlobalIn p = line + 3LL * min(x + m, a.width - 1);
					px[3 * m] = p[0];
					px[3 * m + 1] = p[1];
					px[3 * m + 2] = p[2];
				}
			}
#pragma unroll
			for (int m = 0; m < 4; m++) {
				short L, A, B;
				sq_to_labs<true>(s_blk, s_res, s_v2Y, px[3 * m], px[3 * m + 1], px[3 * m + 2], L, A, B);
				s_L[(row + h) * SQ_R + 4 * quad + m + h] = L;
				s_ab[row * SQ_T + 4 * quad + m] = ((unsigned int) A & 0xffffu) | ((unsigned int) B << 16);
			}
		}
		// 1b. the ring of h pixels round it (image edges clamped): L only
		{
			const int ring = rw * rw - SQ_T * SQ_T;
			if (t < ring) {
				int ry, rx;
				if (t < 2 * rw * h) {
					const int r = t / rw;
					rx = t - r * rw;
					ry = r < h ? r : SQ_T + r;
				}
				else {
					const int k = t - 2 * rw * h;
					const int r = k / (2 * h), c = k - r * 2 * h;
					ry = h + r;
					rx = c < h ? c : SQ_T + c;
				}
				const int x = min(max(x0 + rx - h, 0), a.width - 1);
				const int y = min(max(y0 + ry - h, 0), a.height - 1);
				const GlobalIn p = in + (long long) y * a.in_stride + 3LL * x;
				short L, A = 0, B = 0;
				sq_to_labs<false>(s_blk, s_res, s_v2Y, p[0], p[1], p[2], L, A, B);
				s_L[ry * SQ_R + rx] = L;
			}
		}
		__syncthreads();
		// 2. horizontal pass on L (all rows of the region, the tile's columns)
		for (int idx = t; idx < rw * SQ_T; idx += SQ_NT) {
			const int ry = idx >> 6, cx = idx & 63;
			int sum = 0;
#pragma unroll
			for (int k = 0; k < 2 * SF_MAXHALF + 1; k++)
				sum += a.coef[k] * (int) s_L[ry * SQ_R + cx + k];
			s_h[ry * SQ_T + cx] = (short) sf_convi_fin(sum, a);
		}
		__syncthreads();
		// 3. vertical pass, the LUT, back to sRGB: 4 pixels per thread, three dword stores
		{
			const int y = y0 + row;
			if (y < a.height) {
				unsigned char o[12];
#pragma unroll
				for (int m = 0; m < 4; m++) {
					const int cx = 4 * quad + m;
					int sum = 0;
#pragma unroll
					for (int k = 0; k < 2 * SF_MAXHALF + 1; k++)
						sum += a.coef[k] * (int) s_h[(row + k) * SQ_T + cx];
					const int blur = sf_convi_fin(sum, a);
					const int v1 = s_L[(row + h) * SQ_R + cx + h];
					// sharpen.c:116-168: index (v1 & 0x7fff) - (blur & 0x7fff) + 32768 of the LUT
					const int d = (v1 & 0x7fff) - (blur & 0x7fff) + 32768 - q.lut_lo;
					int lv = d < 0 ? q.lut_below : q.lut_above;
					if ((unsigned int) d < (unsigned int) q.lut_n)
						lv = s_lut[d];
					const int sharp = min(max(v1 + lv, 0), 32767);
					const unsigned int ab = s_ab[row * SQ_T + cx];
					labs_to_srgb8(s_Y2v, sharp, (int) (short) (ab & 0xffffu), (int) ab >> 16, o[3 * m], o[3 * m + 1], o[3 * m + 2]);
				}
				const int x = x0 + 4 * quad;
				const GlobalOut dst = out + (long long) y * a.out_stride + 3LL * x;
				if (x + 4 <= a.width && !(((uintptr_t) dst) & 3)) {
					unsigned int __attribute__((address_space(1))) *d4 = (unsigned int __attribute__((address_space(1))) *) dst;
#pragma unroll
					for (int w = 0; w < 3; w++)
						d4[w] = (unsigned) o[4 * w] | ((unsigned) o[4 * w + 1] << 8) | ((unsigned) o[4 * w + 2] << 16) |
							((unsigned) o[4 * w + 3] << 24);
				}
				else {
					for (int m = 0; m < 12 && x + m / 3 < a.width; m++)
						dst[m] = o[m];
				}
			}
		}
	}
}

// n images of one geometry (one launch per SF_MAXB of them).  0 done, 1 not this kernel's case, -1 error
// Per device, once: does every colour come back from sRGB -> LabS -> sRGB (sharpen_identity_kernel)?  A failure
// to run the check counts as "no": the kernel that takes every pixel the whole way needs no such proof.
static bool sharpen_identity_proven(const ColourTables &tables)
{
	constexpr int MAXDEV = 64;
	static std::mutex mutex;
	static signed char state[MAXDEV]; // 0 unknown, 1 proven, -1 not
	const int dev = current_device();
	if (dev < 0 || dev >= MAXDEV)
		return false;
	std::lock_guard<std::mutex> lock(mutex);
	if (state[dev] == 0) {
		state[dev] = -1;
		unsigned int zero = 0, bad = 1;
		unsigned int *d_bad = (unsigned int *) upload(&zero, sizeof(zero));
		if (d_bad) {
			hipLaunchKernelGGL(sharpen_identity_kernel, dim3(65536), dim3(256), 0, stream(), tables, d_bad);
			if (hipGetLastError() == hipSuccess &&
				hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, stream()) == hipSuccess &&
				hipStreamSynchronize(stream()) == hipSuccess && bad == 0)
				state[dev] = 1;
			vips_hip_free(d_bad);
		}
		else
			vips_hip_error_clear();
	}
	return state[dev] == 1;
}

// sharpen_quad_u8_kernel's 106 KB of dynamic LDS need the opt-in, which is a per-device attribute of the function
static bool sharpen_quad_lds_allowed()
{
	constexpr int MAXDEV = 64;
	static std::mutex mutex;
	static signed char state[MAXDEV];
	const int dev = current_device();
	if (dev < 0 || dev >= MAXDEV)
		return false;
	std::lock_guard<std::mutex> lock(mutex);
	if (state[dev] == 0) {
		const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sharpen_quad_u8_kernel),
			hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		state[dev] = e == hipSuccess ? 1 : -1;
		if (e != hipSuccess)
			(void) hipGetLastError();
	}
	return state[dev] == 1;
}

int sharpen_fused_u8(const VipsHipRegion *const *ins, const VipsHipRegion *const *outs, int n_images,
	const int *to_steps, int n_to, const int *from_steps, int n_from, const int *coef, int n, int scale,
	const int *lut, const SharpenLutWindow *win)
{
	if (getenv("VIPS_HIP_NO_FUSED_SHARPEN") || n_images < 1)
		return 1;
	const VipsHipRegion *in = ins[0], *out = outs[0];
	for (int i = 0; i < n_images; i++) {
		const VipsHipRegion *ri = ins[i], *ro = outs[i];
		if (ri->format != VIPS_HIP_FORMAT_UCHAR || ro->format != VIPS_HIP_FORMAT_UCHAR || ri->bands != 3 ||
			ro->bands != 3)
			return 1;
		if (ri->left != 0 || ri->top != 0 || ro->left != 0 || ro->top != 0 || ri->width != ri->im_width ||
			ri->height != ri->im_height || ro->width != ri->width || ro->height != ri->height)
			return 1;
		if (ri->width != in->width || ri->height != in->height || ri->stride != in->stride ||
			ro->stride != out->stride)
			return 1;
	}
	if (n < 1 || n > 2 * SF_MAXHALF + 1 || !(n & 1) || scale <= 0)
		return 1;
	long long abs_sum = 0;
	for (int k = 0; k < n; k++)
		abs_sum += coef[k] < 0 ? -(long long) coef[k] : coef[k];
	if (abs_sum * 32768 + scale >= (1LL << 31)) // 32-bit sums
		return 1;
	// the kernel runs the two chains vips_sharpen uses on sRGB, spelled out
	static const int want_to[4] = { VIPS_HIP_COLOUR_sRGB2scRGB, VIPS_HIP_COLOUR_scRGB2XYZ, VIPS_HIP_COLOUR_XYZ2Lab,
		VIPS_HIP_COLOUR_Lab2LabS };
	static const int want_from[4] = { VIPS_HIP_COLOUR_LabS2Lab, VIPS_HIP_COLOUR_Lab2XYZ, VIPS_HIP_COLOUR_XYZ2scRGB,
		VIPS_HIP_COLOUR_scRGB2sRGB };
	if (n_to != 4 || n_from != 4 || memcmp(to_steps, want_to, sizeof(want_to)) ||
		memcmp(from_steps, want_from, sizeof(want_from)))
		return 1;
	SharpenFusedArgs a;
	RouteArgs to_labs;
	if (colour_route_prepare(to_steps, n_to, &to_labs))
		return -1;
	a.tables = to_labs.tables;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.height = in->height;
	a.n = n;
	a.half = n / 2;
	for (int k = 0; k < 2 * SF_MAXHALF + 1; k++)
		a.coef[k] = k < n ? coef[k] : 0;
	a.scale = scale;
	a.rounding = scale / 2;
	a.magic = 0;
	a.shift = 0;
	if (scale > 1) {
		int l = 0;
		while ((1LL << l) < scale)
			l++;
		a.magic = (unsigned int) ((((1ULL << 32) * ((1ULL << l) - (unsigned long long) scale)) / (unsigned long long) scale) + 1);
		a.shift = l - 1;
	}
	a.lut = lut;
	// the blur's division as a 24-bit multiply-high: tried on every sum the mask can make (under a million)
	// (once per (scale, largest sum): the answer is kept)
	a.sh24 = -1;
	a.m24 = 0;
	{
		const long long n_max = abs_sum * 32767 + a.rounding;
		static std::mutex mutex;
		static std::map<std::pair<int, long long>, std::pair<int, unsigned int>> proven;
		std::lock_guard<std::mutex> lock(mutex);
		auto it = proven.find(std::make_pair(scale, n_max));
		if (it == proven.end()) {
			std::pair<int, unsigned int> answer(-1, 0u);
			int sh = 0;
			while (sh < 24 && (n_max << (sh + 1)) < (1LL << 24))
				sh++;
			const unsigned long long m = ((1ULL << (32 - sh)) + (unsigned long long) scale - 1) / (unsigned long long) scale;
			if (n_max < (1LL << 24) && (n_max << sh) < (1LL << 24) && m < (1ULL << 24)) {
				bool ok = true;
				for (long long v = 0; v <= n_max && ok; v++)
					ok = (((unsigned long long) (v << sh) * m) >> 32) == (unsigned long long) (v / scale);
				if (ok)
					answer = std::make_pair(sh, (unsigned int) m);
			}
			if (proven.size() > 64)
				proven.clear();
			it = proven.emplace(std::make_pair(scale, n_max), answer).first;
		}
		a.sh24 = it->second.first;
		a.m24 = it->second.second;
	}
	a.zero_lo = win ? win->zero_lo : 1;
	a.zero_hi = win ? win->zero_hi : 0;
	// every table in LDS (sharpen_quad_u8_kernel) when the LUT's window and this host's cbrtf allow it -- for large
	// images.  Measured (profiles/r05p_sharpen_quad.txt): 8192^2 of noise 1.08 -> 0.83 ms, but a batch of 1024^2
	// thumbnails on its 64-CU partition 0.033 -> 0.049 ms per image: a thumbnail's values are few and near each
	// other, the older kernel's table gathers hit its L1, and 106 KB of LDS per block leave that partition one block
	// of 16 waves per CU to hide behind.  $VIPS_HIP_SHARPEN_QUAD=0 / 1 forces one or the other.
	// the pixels the LUT leaves alone skip the way back (sharpen_fused_u8_kernel<*, true>) when the LUT has a zero
	// window at all and sRGB -> LabS -> sRGB is the identity with this device's tables.  $VIPS_HIP_SHARPEN_SKIP=0:
	// every pixel the whole way.
	const char *skip_env = getenv("VIPS_HIP_SHARPEN_SKIP");
	bool nonneg = true; // (a gaussian's integer mask: what lets the skip kernel divide without a sign)
	for (int k = 0; k < n; k++)
		nonneg = nonneg && coef[k] >= 0;
	const bool small = (long long) a.height * a.in_stride < (1LL << 31) && a.in_stride < (1LL << 24) && a.height < (1 << 24) &&
		a.in_stride >= 0;
	const bool skip = nonneg && small && a.sh24 >= 0 && a.zero_lo <= 0 && a.zero_hi >= 0 && !(skip_env && atoi(skip_env) == 0) &&
		sharpen_identity_proven(a.tables);
	a.defer = nullptr;
	a.defer_threshold = 0;
	a.defer_raw = 0;
	a.q_tiles_x = a.q_tiles_y = 0;
	const char *quad_env = getenv("VIPS_HIP_SHARPEN_QUAD");
	const bool want_quad = quad_env ? atoi(quad_env) != 0 : (long long) a.width * a.height >= 2048LL * 2048;
	// (the LDS opt-in is per DEVICE: asked for on every device this code reaches; a device that refuses it keeps
	// the kernel that needs none)
	if (win && win->lut_win && win->n <= SQ_MAXLUT && want_quad && !getenv("VIPS_HIP_NO_SHARPEN_QUAD") &&
		sharpen_quad_lds_allowed()) {
		const CbrtQuad *cq = cbrt_quad_tables();
		if (cq) {
			SharpenQuadArgs q;
			q.f = a;
			q.cq = *cq;
			q.lut_win = win->lut_win;
			q.lut_lo = win->lo;
			q.lut_n = win->n;
			q.lut_below = win->below;
			q.lut_above = win->above;
			q.tiles_x = (a.width + SQ_T - 1) / SQ_T;
			q.tiles_y = (a.height + SQ_T - 1) / SQ_T;
			const size_t lds = (size_t) (CBQ_BLOCKS * 4 + CBQ_RES_WORDS + 256 + 260) * 4 + (size_t) (SQ_MAXLUT + 8) * 2 +
				(size_t) SQ_R * SQ_R * 2 + (size_t) SQ_T * SQ_T * 4 + (size_t) SQ_R * SQ_T * 2;
			// Large images (round 6), ADAPTIVELY: what a photograph mostly is -- neighbouring pixels near each other,
			// the LUT's flat centre -- is what the skip kernel is fast at (8192^2 smooth: 0.79 -> ~0.3 ms), noise and
			// dense edges what this one is (0.83 against ~1.2).  So the skip kernel goes first and LEAVES the tiles
			// whose list is long (more than defer_threshold of a 64 x 32 tile's 2 048 pixels) on a device-side list;
			// this kernel then walks that list -- no host round trip, the count is read on the device.
			// $VIPS_HIP_SHARPEN_ADAPTIVE=0: this kernel alone, every tile.
			const char *ad = getenv("VIPS_HIP_SHARPEN_ADAPTIVE");
			const bool adaptive = skip && a.n <= 5 && a.height > 16 && !(ad && atoi(ad) == 0);
			q.defer = nullptr;
			for (int base = 0; base < n_images; base += SF_MAXB) {
				const int count = n_images - base < SF_MAXB ? n_images - base : SF_MAXB;
				SharpenFusedPtrs p;
				memset(&p, 0, sizeof(p));
				for (int i = 0; i < count; i++) {
					p.in[i] = (const unsigned char *) ins[base + i]->data;
					p.out[i] = (unsigned char *) outs[base + i]->data;
				}
				q.n_images = count;
				const int tiles = q.tiles_x * q.tiles_y * count;
				int *defer = nullptr;
				if (adaptive) {
					defer = (int *) vips_hip_malloc((size_t) (1 + 2 * tiles) * sizeof(int));
					if (!defer)
						return -1;
					if (hipMemsetAsync(defer, 0, (size_t) (1 + tiles) * sizeof(int), stream()) != hipSuccess) {
						vips_hip_free(defer);
						error("sharpen", "memset failed");
						return -1;
					}
					SharpenFusedArgs a1 = a;
					a1.defer = defer;
					a1.defer_threshold = getenv("VIPS_HIP_SHARPEN_DEFER") ? atoi(getenv("VIPS_HIP_SHARPEN_DEFER")) : 640;
					a1.defer_raw = getenv("VIPS_HIP_SHARPEN_DEFER_RAW") ? atoi(getenv("VIPS_HIP_SHARPEN_DEFER_RAW")) : 24; // (12 sends a fifth of a smooth image's tiles the slow way: profiles/r06n_sharpen_raw.txt)
					a1.q_tiles_x = q.tiles_x;
					a1.q_tiles_y = q.tiles_y;
					dim3 grid1((a.width + SF_TW - 1) / SF_TW, (a.height + 31) / 32, count);
					if (a1.defer_raw > 0) {
						Gate gate0("sharpen_survey");
						const int stx = (int) grid1.x, sty = (int) grid1.y;
						hipLaunchKernelGGL(sharpen_survey_kernel, dim3((stx * sty + 255) / 256, 1, count), dim3(256, 1, 1), 0, stream(), p,
							a.width, a.height, (long long) a.in_stride, a1.defer_raw, defer, q.tiles_x, q.tiles_y, stx, sty);
					}
					Gate gate1("sharpen_skip_u8");
					if (a.n <= 3)
						hipLaunchKernelGGL((sharpen_fused_u8_kernel<32, true, 3>), grid1, dim3(256, 1, 1), 0, stream(), p, a1);
					else
						hipLaunchKernelGGL((sharpen_fused_u8_kernel<32, true, 5>), grid1, dim3(256, 1, 1), 0, stream(), p, a1);
					if (hipGetLastError() != hipSuccess) {
						vips_hip_free(defer);
						error("sharpen", "kernel launch failed");
						return -1;
					}
				}
				q.defer = defer;
				const char *e = getenv("VIPS_HIP_SHARPEN_QUAD_GRID");
				int grid = e && atoi(e) > 0 ? atoi(e) : 256;
				grid = grid > tiles ? tiles : grid;
				{
					Gate gate("sharpen_quad_u8");
					hipLaunchKernelGGL(sharpen_quad_u8_kernel, dim3(grid), dim3(SQ_NT), lds, stream(), p, q);
				}
				const bool ok = hipGetLastError() == hipSuccess;
				if (defer)
					vips_hip_free(defer); // (reuse is ordered on this thread's stream)
				if (!ok) {
					error("sharpen", "kernel launch failed");
					return -1;
				}
			}
			return 0;
		}
		vips_hip_error_clear();
	}
	Gate gate(skip ? "sharpen_skip_u8" : "sharpen_fused_u8");
	for (int base = 0; base < n_images; base += SF_MAXB) {
		const int count = n_images - base < SF_MAXB ? n_images - base : SF_MAXB;
		SharpenFusedPtrs p;
		memset(&p, 0, sizeof(p));
		for (int i = 0; i < count; i++) {
			p.in[i] = (const unsigned char *) ins[base + i]->data;
			p.out[i] = (unsigned char *) outs[base + i]->data;
		}
		// 32-row tiles: the halo ring and the block's table loads are shared by twice the pixels
		// (2 % faster than 16-row tiles in the C4 batch, 64-row tiles slower again);
		// $VIPS_HIP_SHARPEN_TH=16 for the other
		const int th = getenv("VIPS_HIP_SHARPEN_TH") ? atoi(getenv("VIPS_HIP_SHARPEN_TH")) : 32;
		if (th != 16 && a.height > 16) {
			dim3 grid((a.width + SF_TW - 1) / SF_TW, (a.height + 31) / 32, count);
			if (skip && a.n <= 3)
				hipLaunchKernelGGL((sharpen_fused_u8_kernel<32, true, 3>), grid, dim3(256, 1, 1), 0, stream(), p, a);
			else if (skip)
				hipLaunchKernelGGL((sharpen_fused_u8_kernel<32, true, 5>), grid, dim3(256, 1, 1), 0, stream(), p, a);
			else
				hipLaunchKernelGGL((sharpen_fused_u8_kernel<32, false>), grid, dim3(256, 1, 1), 0, stream(), p, a);
		}
		else {
			dim3 grid((a.width + SF_TW - 1) / SF_TW, (a.height + 15) / 16, count);
			if (skip && a.n <= 3)
				hipLaunchKernelGGL((sharpen_fused_u8_kernel<16, true, 3>), grid, dim3(256, 1, 1), 0, stream(), p, a);
			else if (skip)
				hipLaunchKernelGGL((sharpen_fused_u8_kernel<16, true, 5>), grid, dim3(256, 1, 1), 0, stream(), p, a);
			else
				hipLaunchKernelGGL((sharpen_fused_u8_kernel<16, false>), grid, dim3(256, 1, 1), 0, stream(), p, a);
		}
		VH_CHECK(hipGetLastError());
	}
	return 0;
}

// ------------------------------------------------- premultiply / unpremultiply

struct PremulArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int width, height, bands;
	double max_alpha;
	int scale[256]; // the uchar fast path's table (premultiply.c:252-258, unpremultiply.c:316-323)
};

// fast uchar -> uchar path: premultiply.c:163-176, unpremultiply.c:222-235
__global__ void __launch_bounds__(256)
premul_u8_kernel(PremulArgs a)
{
	__shared__ int scale[256];
	scale[threadIdx.x] = a.scale[threadIdx.x];
	__syncthreads();
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	if (a.bands == 4) {
		for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
			unsigned int vv[RU];
#pragma unroll
			for (int r = 0; r < RU; r++) {
				const int y = min(y0 + r, a.height - 1);
				vv[r] = *reinterpret_cast<const unsigned int *>(a.in + (long long) y * a.in_stride + (long long) x * 4);
			}
#pragma unroll
			for (int r = 0; r < RU; r++) {
				if (y0 + r >= a.height)
					break;
				const unsigned int v = vv[r];
				const unsigned int alpha = v >> 24;
				const int s = scale[alpha];
				const unsigned int cr = ((int) (v & 0xff) * s + 128) >> 8;
				const unsigned int cg = ((int) ((v >> 8) & 0xff) * s + 128) >> 8;
				const unsigned int cb = ((int) ((v >> 16) & 0xff) * s + 128) >> 8;
				*reinterpret_cast<unsigned int *>(a.out + (long long) (y0 + r) * a.out_stride + (long long) x * 4) =
					(cr & 0xff) | ((cg & 0xff) << 8) | ((cb & 0xff) << 16) | (alpha << 24);
			}
		}
		return;
	}
	for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
		const unsigned char *p = a.in + (long long) y * a.in_stride + (long long) x * a.bands;
		unsigned char *q = a.out + (long long) y * a.out_stride + (long long) x * a.bands;
		{
			const int alpha = p[a.bands - 1];
			const int s = scale[alpha];
			for (int i = 0; i < a.bands - 1; i++)
				q[i] = (unsigned char) ((p[i] * s + 128) >> 8);
			q[a.bands - 1] = (unsigned char) alpha;
		}
	}
}

// PRE_MANY / PRE_RGBA (premultiply.c:78-128) and UNPRE / FUNPRE (unpremultiply.c:85-186),
// float output.
// one pixel of NB bands (alpha last); ptr-free so that the vector path stays in registers
template <typename TIN, bool INVERSE, int NB>
static __device__ __forceinline__ void premul_pixel(const TIN (&p)[NB], float (&q)[NB], double max_alpha)
{
	constexpr int ab = NB - 1;
	const TIN alpha = p[ab];
	// VIPS_CLIP(0, alpha, max_alpha) is evaluated in double
	double clip = (double) alpha;
	clip = max_alpha < clip ? max_alpha : clip;
	clip = 0.0 > clip ? 0.0 : clip;
	if (!INVERSE) {
		// IN clip_alpha = CLIP(...); OUT nalpha = (OUT) clip_alpha / max_alpha
		const TIN clip_alpha = vh::cvt_to<TIN>(clip);
		const float nalpha = (float) __ddiv_rn((double) (float) clip_alpha, max_alpha);
#pragma unroll
		for (int i = 0; i < ab; i++)
			q[i] = __fmul_rn((float) p[i], nalpha);
		q[ab] = (float) alpha;
	}
	else {
		float factor;
		if (sizeof(TIN) == 4 && ((TIN) 0.5f != (TIN) 0)) // float input: FUNPRE
			factor = fabs((double) alpha) < 0.01 ? 0.0f : (float) __ddiv_rn(max_alpha, (double) alpha);
		else
			factor = alpha == (TIN) 0 ? 0.0f : (float) __ddiv_rn(max_alpha, (double) alpha);
#pragma unroll
		for (int i = 0; i < ab; i++)
			q[i] = __fmul_rn(factor, (float) p[i]);
		q[ab] = (float) clip;
	}
}

// RGBA with aligned rows: one vector load and one 16-byte store per pixel (a wave writes 1 KiB of a line per
// instruction), RU rows in flight per lane.  8-bit alpha: what premul_pixel makes of an alpha value -- the factor
// (a double division) and the alpha it writes -- is a 256-entry table the block fills with premul_pixel itself
// (1.0f * factor == factor), so a pixel is three multiplies.  Round 5's form (a row per turn, the division per
// pixel) ran at 62 % of 8 TB/s.
template <typename TIN, bool INVERSE>
__global__ void __launch_bounds__(256)
premul_rgba_kernel(PremulArgs a)
{
	constexpr bool TABLE = sizeof(TIN) == 1;
	__shared__ float2 s_tab[TABLE ? 256 : 1];
	if constexpr (TABLE) {
		const TIN one[4] = { (TIN) 1, (TIN) 1, (TIN) 1, (TIN) (unsigned char) threadIdx.x };
		float q[4];
		premul_pixel<TIN, INVERSE, 4>(one, q, a.max_alpha);
		s_tab[threadIdx.x] = make_float2(q[0], q[3]);
		__syncthreads();
	}
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	for (int y0 = blockIdx.y * RU; y0 < a.height; y0 += gridDim.y * RU) {
		Vec4<TIN> pv[RU];
#pragma unroll
		for (int r = 0; r < RU; r++)
			pv[r] = ((const Vec4<TIN> *) (a.in + (long long) min(y0 + r, a.height - 1) * a.in_stride))[x];
#pragma unroll
		for (int r = 0; r < RU; r++) {
			if (y0 + r >= a.height)
				break;
			float qv[4];
			if constexpr (TABLE) {
				const float2 e = s_tab[(unsigned char) pv[r].v[3]];
#pragma unroll
				for (int i = 0; i < 3; i++)
					qv[i] = INVERSE ? __fmul_rn(e.x, (float) pv[r].v[i]) : __fmul_rn((float) pv[r].v[i], e.x);
				qv[3] = e.y;
			}
			else
				premul_pixel<TIN, INVERSE, 4>(pv[r].v, qv, a.max_alpha);
			((float4 *) (a.out + (long long) (y0 + r) * a.out_stride))[x] = make_float4(qv[0], qv[1], qv[2], qv[3]);
		}
	}
}

template <typename TIN, bool INVERSE>
__global__ void __launch_bounds__(256)
premul_float_kernel(PremulArgs a)
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= a.width)
		return;
	const int ab = a.bands - 1;
	for (int y = blockIdx.y; y < a.height; y += gridDim.y) {
		const TIN *p = (const TIN *) (a.in + (long long) y * a.in_stride) + (long long) x * a.bands;
		float *q = (float *) (a.out + (long long) y * a.out_stride) + (long long) x * a.bands;
		const TIN alpha = p[ab];
		// VIPS_CLIP(0, alpha, max_alpha) is evaluated in double
		double clip = (double) alpha;
		clip = a.max_alpha < clip ? a.max_alpha : clip;
		clip = 0.0 > clip ? 0.0 : clip;
		if (!INVERSE) {
			// IN clip_alpha = CLIP(...); OUT nalpha = (OUT) clip_alpha / max_alpha
			const TIN clip_alpha = vh::cvt_to<TIN>(clip);
			const float nalpha = (float) __ddiv_rn((double) (float) clip_alpha, a.max_alpha);
			for (int i = 0; i < ab; i++)
				q[i] = __fmul_rn((float) p[i], nalpha);
			q[ab] = (float) alpha;
		}
		else {
			float factor;
			if (sizeof(TIN) == 4 && ((TIN) 0.5f != (TIN) 0)) // float input: FUNPRE
				factor = fabs((double) alpha) < 0.01 ? 0.0f : (float) __ddiv_rn(a.max_alpha, (double) alpha);
			else
				factor = alpha == (TIN) 0 ? 0.0f : (float) __ddiv_rn(a.max_alpha, (double) alpha);
			for (int i = 0; i < ab; i++)
				q[i] = __fmul_rn(factor, (float) p[i]);
			q[ab] = (float) clip;
		}
	}
}

int premultiply_region(const VipsHipRegion *in, const VipsHipRegion *out, double max_alpha, int uchar,
	int inverse)
{
	const char *domain = inverse ? "unpremultiply" : "premultiply";
	if (ensure_init())
		return -1;
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->bands != out->bands || in->bands < 2) {
		error(domain, "need the same number of bands (at least 2) in and out");
		return -1;
	}
	if (out->left < in->left || out->top < in->top ||
		out->left + out->width > in->left + in->width ||
		out->top + out->height > in->top + in->height) {
		error(domain, "input region too small");
		return -1;
	}
	const bool fast = uchar && in->format == VIPS_HIP_FORMAT_UCHAR;
	if (out->format != (fast ? VIPS_HIP_FORMAT_UCHAR : VIPS_HIP_FORMAT_FLOAT)) {
		error(domain, "output region has the wrong format");
		return -1;
	}
	PremulArgs a;
	const int ies = format_sizeof(in->format);
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride +
		(size_t) (out->left - in->left) * in->bands * ies;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = out->width;
	a.height = out->height;
	a.bands = in->bands;
	a.max_alpha = max_alpha;
	for (int i = 0; i < 256; i++) {
		double clip = (double) i;
		clip = max_alpha < clip ? max_alpha : clip;
		clip = 0.0 > clip ? 0.0 : clip;
		if (inverse)
			a.scale[i] = clip == 0 ? 0 : (int) (256 * max_alpha / clip);
		else
			a.scale[i] = (int) (256 * clip / max_alpha);
	}
	dim3 block(256, 1, 1);
	dim3 grid((a.width + 255) / 256, rows_grid((a.width + 255) / 256, a.height), 1);
	Gate gate(inverse ? "unpremultiply" : "premultiply");
	if (fast) {
		if (a.bands == 4 && (((uintptr_t) a.in | (uintptr_t) a.out | a.in_stride | a.out_stride) & 3)) {
			error(domain, "RGBA rows must be 4-byte aligned");
			return -1;
		}
		hipLaunchKernelGGL(premul_u8_kernel, grid, block, 0, stream(), a);
	}
	else {
#define GO(TIN) \
	if (a.bands == 4 && !(((uintptr_t) a.in | (uintptr_t) a.in_stride) % (4 * sizeof(TIN))) && \
		!(((uintptr_t) a.out | (uintptr_t) a.out_stride) & 15) && !getenv("VIPS_HIP_NO_PREMUL_RGBA")) { \
		if (inverse) \
			hipLaunchKernelGGL((premul_rgba_kernel<TIN, true>), grid, block, 0, stream(), a); \
		else \
			hipLaunchKernelGGL((premul_rgba_kernel<TIN, false>), grid, block, 0, stream(), a); \
	} \
	else if (inverse) \
		hipLaunchKernelGGL((premul_float_kernel<TIN, true>), grid, block, 0, stream(), a); \
	else \
		hipLaunchKernelGGL((premul_float_kernel<TIN, false>), grid, block, 0, stream(), a); \
	break;
		switch (in->format) {
		case VIPS_HIP_FORMAT_UCHAR: GO(unsigned char)
		case VIPS_HIP_FORMAT_CHAR: GO(signed char)
		case VIPS_HIP_FORMAT_USHORT: GO(unsigned short)
		case VIPS_HIP_FORMAT_SHORT: GO(short)
		case VIPS_HIP_FORMAT_UINT: GO(unsigned int)
		case VIPS_HIP_FORMAT_INT: GO(int)
		case VIPS_HIP_FORMAT_FLOAT: GO(float)
		default:
			error(domain, "band format %d is outside the HIP path", in->format);
			return -1;
		}
#undef GO
	}
	VH_CHECK(hipGetLastError());
	return 0;
}

// ----------------------------------------------------------- route launcher

int colour_route_prepare(const int *steps, int n_steps, RouteArgs *a)
{
	if (n_steps < 1 || n_steps > 8 || ensure_init() || ensure_tables())
		return -1;
	memset(a, 0, sizeof(*a));
	a->n_steps = n_steps;
	for (int s = 0; s < 8; s++)
		a->steps[s] = s < n_steps ? steps[s] : -1;
	a->alpha_scale = 1.0;
	a->in_bands = a->out_bands = 3;
	a->tables = g_tables;
	return 0;
}

int colour_route(const int *steps, int n_steps, double alpha_scale, const VipsHipRegion *in,
	const VipsHipRegion *out)
{
	const char *domain = "colour";
	if (ensure_init())
		return -1;
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (n_steps < 1 || n_steps > 8) {
		error(domain, "bad route length %d", n_steps);
		return -1;
	}
	if (in->bands < 3) {
		error(domain, "image must have at least 3 bands"); // vips_check_bands_atleast
		return -1;
	}
	if (out->bands != in->bands) {
		error(domain, "output must have as many bands as the input");
		return -1;
	}
	if (out->left < in->left || out->top < in->top ||
		out->left + out->width > in->left + in->width ||
		out->top + out->height > in->top + in->height) {
		error(domain, "input region too small");
		return -1;
	}
	// what the last step stores
	const int last = steps[n_steps - 1];
	int want_out;
	if (last == VIPS_HIP_COLOUR_scRGB2sRGB)
		want_out = VIPS_HIP_FORMAT_UCHAR;
	else if (last == VIPS_HIP_COLOUR_scRGB2sRGB16)
		want_out = VIPS_HIP_FORMAT_USHORT;
	else if (last == VIPS_HIP_COLOUR_Lab2LabS)
		want_out = VIPS_HIP_FORMAT_SHORT;
	else
		want_out = VIPS_HIP_FORMAT_FLOAT;
	if (out->format != want_out) {
		error(domain, "output region has format %d, this conversion writes %d", out->format, want_out);
		return -1;
	}
	for (int s = 0; s < n_steps; s++) {
		const int st = steps[s];
		if (st < 0 || st >= VIPS_HIP_COLOUR_LAST) {
			error(domain, "unknown colour step %d", st);
			return -1;
		}
		const bool decoder = st == VIPS_HIP_COLOUR_sRGB2scRGB || st == VIPS_HIP_COLOUR_sRGB2scRGB16 ||
			st == VIPS_HIP_COLOUR_LabS2Lab;
		const bool encoder = st == VIPS_HIP_COLOUR_scRGB2sRGB || st == VIPS_HIP_COLOUR_scRGB2sRGB16 ||
			st == VIPS_HIP_COLOUR_Lab2LabS;
		if ((decoder && s != 0) || (encoder && s != n_steps - 1)) {
			error(domain, "colour step %d cannot sit at position %d of a fused route", st, s);
			return -1;
		}
	}
	if (ensure_tables())
		return -1;

	RouteArgs a;
	const int ies = format_sizeof(in->format);
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride +
		(size_t) (out->left - in->left) * in->bands * ies;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = out->width;
	a.height = out->height;
	a.in_bands = in->bands;
	a.out_bands = out->bands;
	a.n_steps = n_steps;
	for (int s = 0; s < 8; s++)
		a.steps[s] = s < n_steps ? steps[s] : -1;
	a.extra_bands = in->bands - 3;
	a.alpha_scale = alpha_scale;
	a.tables = g_tables;

	dim3 block(256, 1, 1);
	dim3 grid((a.width + 255) / 256, rows_grid((a.width + 255) / 256, a.height), 1);
	Gate gate("colour_route");
	// 3 bands in and out, rows that start and advance on 4-element boundaries: 4 pixels per thread
	const int oes = format_sizeof(want_out);
	const bool x4 = in->bands == 3 && out->bands == 3 && !(a.width & 3) &&
		!((uintptr_t) a.in % (4 * ies)) && !((uintptr_t) a.out % (4 * oes)) &&
		!(a.in_stride % (4 * ies)) && !(a.out_stride % (4 * oes));
	// (one row per block: the table-gathering route kernels measured faster with many short blocks)
	const dim3 grid4((a.width / 4 + 255) / 256, a.height < 32768 ? a.height : 32768, 1);
	// the routes with their steps compiled in (colour_device.h kStaticRoutes), for the formats they
	// are met with: sRGB -> Lab / LabS from uchar and float, LabS / Lab -> sRGB uchar
	int route_id = 0;
	for (int r = 1; r < kStaticRouteCount && x4; r++)
		if (n_steps == kStaticRoutes[r][0] && !memcmp(steps, &kStaticRoutes[r][1], sizeof(int) * n_steps))
			route_id = r;
	bool launched = false;
	// sRGB -> Lab / LabS from uchar or float: the cube-root table from LDS (colour_lab_lds_kernel)
	if ((route_id == 1 || route_id == 2) && !getenv("VIPS_HIP_NO_LAB_LDS") &&
		(in->format == VIPS_HIP_FORMAT_UCHAR || in->format == VIPS_HIP_FORMAT_FLOAT)) {
		const CbrtQuad *cq = cbrt_quad_tables();
		const CbrtExact *cx = cq ? nullptr : cbrt_exact_tables();
		if (cq) {
			const int gx = (a.width / 4 + 1023) / 1024;
			int gy = (256 + gx - 1) / gx; // one block of 1024 per CU, each walking its share of the rows
			const char *e = getenv("VIPS_HIP_LAB_QUAD_GRID");
			if (e && atoi(e) > 0)
				gy = (atoi(e) + gx - 1) / gx;
			gy = gy > a.height ? a.height : gy;
			const dim3 gridp(gx, gy, 1), blockq(1024, 1, 1);
			if (route_id == 1 && in->format == VIPS_HIP_FORMAT_UCHAR)
				hipLaunchKernelGGL((colour_lab_quad_kernel<unsigned char, false>), gridp, blockq, 0, stream(), a, *cq);
			else if (route_id == 1)
				hipLaunchKernelGGL((colour_lab_quad_kernel<float, false>), gridp, blockq, 0, stream(), a, *cq);
			else if (in->format == VIPS_HIP_FORMAT_UCHAR)
				hipLaunchKernelGGL((colour_lab_quad_kernel<unsigned char, true>), gridp, blockq, 0, stream(), a, *cq);
			else
				hipLaunchKernelGGL((colour_lab_quad_kernel<float, true>), gridp, blockq, 0, stream(), a, *cq);
			launched = true;
		}
		else if (cx) {
			const int gx = (a.width / 4 + 255) / 256;
			int gy = (256 * 4 + gx - 1) / gx; // four 34 KB blocks per CU, each walking its share of the rows
			gy = gy > a.height ? a.height : gy;
			const dim3 gridp(gx, gy, 1);
			const bool f32 = cx->bf != nullptr;
#define LAB_LDS(TIN, LABS) \
	do { \
		if (f32) \
			hipLaunchKernelGGL((colour_lab_lds_kernel<TIN, LABS, true>), gridp, block, 0, stream(), a, *cx); \
		else \
			hipLaunchKernelGGL((colour_lab_lds_kernel<TIN, LABS, false>), gridp, block, 0, stream(), a, *cx); \
	} while (0)
			if (route_id == 1 && in->format == VIPS_HIP_FORMAT_UCHAR)
				LAB_LDS(unsigned char, false);
			else if (route_id == 1)
				LAB_LDS(float, false);
			else if (in->format == VIPS_HIP_FORMAT_UCHAR)
				LAB_LDS(unsigned char, true);
			else
				LAB_LDS(float, true);
#undef LAB_LDS
			launched = true;
		}
		else
			vips_hip_error_clear();
	}
#define STATIC_ROUTE(R, TIN, FIN, TOUT, FOUT) \
	if (!launched && route_id == R && in->format == FIN && want_out == FOUT) { \
		hipLaunchKernelGGL((colour_route_x4_kernel<TIN, TOUT, R>), grid4, block, 0, stream(), a); \
		launched = true; \
	}
	STATIC_ROUTE(1, float, VIPS_HIP_FORMAT_FLOAT, float, VIPS_HIP_FORMAT_FLOAT)
	STATIC_ROUTE(1, unsigned char, VIPS_HIP_FORMAT_UCHAR, float, VIPS_HIP_FORMAT_FLOAT)
	STATIC_ROUTE(2, float, VIPS_HIP_FORMAT_FLOAT, short, VIPS_HIP_FORMAT_SHORT)
	STATIC_ROUTE(2, unsigned char, VIPS_HIP_FORMAT_UCHAR, short, VIPS_HIP_FORMAT_SHORT)
	STATIC_ROUTE(3, short, VIPS_HIP_FORMAT_SHORT, unsigned char, VIPS_HIP_FORMAT_UCHAR)
	STATIC_ROUTE(4, float, VIPS_HIP_FORMAT_FLOAT, unsigned char, VIPS_HIP_FORMAT_UCHAR)
#undef STATIC_ROUTE
#define GO(TIN, TOUT) \
	if (launched) \
		; \
	else if (x4) \
		hipLaunchKernelGGL((colour_route_x4_kernel<TIN, TOUT, 0>), grid4, block, 0, stream(), a); \
	else \
		hipLaunchKernelGGL((colour_route_kernel<TIN, TOUT>), grid, block, 0, stream(), a)
#define GO_IN(TOUT) \
	switch (in->format) { \
	case VIPS_HIP_FORMAT_UCHAR: GO(unsigned char, TOUT); break; \
	case VIPS_HIP_FORMAT_USHORT: GO(unsigned short, TOUT); break; \
	case VIPS_HIP_FORMAT_SHORT: GO(short, TOUT); break; \
	case VIPS_HIP_FORMAT_FLOAT: GO(float, TOUT); break; \
	default: \
		error(domain, "input band format %d is outside the HIP colour path", in->format); \
		return -1; \
	}
	switch (want_out) {
	case VIPS_HIP_FORMAT_UCHAR: GO_IN(unsigned char) break;
	case VIPS_HIP_FORMAT_USHORT: GO_IN(unsigned short) break;
	case VIPS_HIP_FORMAT_SHORT: GO_IN(short) break;
	default: GO_IN(float) break;
	}
#undef GO_IN
#undef GO
	VH_CHECK(hipGetLastError());
	return 0;
}

} // namespace vh

using namespace vh;

extern "C" {

int vips_hip_colour_gen(int step, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return colour_route(&step, 1, 1.0, in, out);
}

int vips_hip_colour_route_gen(const int *steps, int n_steps, double alpha_scale,
	const VipsHipRegion *in, const VipsHipRegion *out)
{
	return colour_route(steps, n_steps, alpha_scale, in, out);
}

int vips_hip_premultiply_gen(const VipsHipRegion *in, const VipsHipRegion *out, double max_alpha,
	int uchar, int inverse)
{
	return premultiply_region(in, out, max_alpha, uchar, inverse);
}

int vips_hip_cast_gen(const VipsHipRegion *in, const VipsHipRegion *out)
{
	if (in && out && in->bands != out->bands) {
		error("cast", "input and output must have the same number of bands");
		return -1;
	}
	return band_cast(in, 0, out, 0, in ? in->bands : 0);
}

int vips_hip_sharpen_gen(const int *lut_device, const VipsHipRegion *in,
	const VipsHipRegion *blurred_l, const VipsHipRegion *out)
{
	const char *domain = "sharpen";
	if (ensure_init())
		return -1;
	if (check_region(domain, in) || check_region(domain, blurred_l) || check_region(domain, out))
		return -1;
	if (in->format != VIPS_HIP_FORMAT_SHORT || out->format != VIPS_HIP_FORMAT_SHORT ||
		blurred_l->format != VIPS_HIP_FORMAT_SHORT || blurred_l->bands != 1 ||
		in->bands != out->bands) {
		error(domain, "need LabS short images (and a 1-band blurred L)");
		return -1;
	}
	if (!lut_device) {
		error(domain, "null lut");
		return -1;
	}
	for (const VipsHipRegion *r : { in, blurred_l })
		if (out->left < r->left || out->top < r->top || out->left + out->width > r->left + r->width ||
			out->top + out->height > r->top + r->height) {
			error(domain, "input region too small");
			return -1;
		}
	SharpenArgs a;
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride +
		(size_t) (out->left - in->left) * in->bands * 2;
	a.blur = (const unsigned char *) blurred_l->data +
		(size_t) (out->top - blurred_l->top) * blurred_l->stride +
		(size_t) (out->left - blurred_l->left) * 2;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.blur_stride = (long long) blurred_l->stride;
	a.out_stride = (long long) out->stride;
	a.width = out->width;
	a.height = out->height;
	a.bands = in->bands;
	a.lut = lut_device;
	dim3 block(256, 1, 1);
	dim3 grid((a.width + 255) / 256, rows_grid((a.width + 255) / 256, a.height), 1);
	Gate gate("sharpen");
	hipLaunchKernelGGL(sharpen_kernel, grid, block, 0, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

} // extern "C"
