// Probe: v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 on gfx950 -- operand layout, issue
// rate, summation order (is D = ((((C + a0 b0) + a1 b1) + a2 b2) + a3 b3) with fused
// multiply-adds, k ascending?), and whether the matrix pipe overlaps with v_fma_f64 on the VALU.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef double double4v __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void layout16(const double *a, const double *b, const double *c, double *d)
{
	const int l = threadIdx.x;
	double4v acc;
	for (int r = 0; r < 4; r++)
		acc[r] = c[l * 4 + r];
	acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
	for (int r = 0; r < 4; r++)
		d[l * 4 + r] = acc[r];
}

__global__ void layout4(const double *a, const double *b, const double *c, double *d)
{
	const int l = threadIdx.x;
	double acc = c[l];
	acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], acc, 0, 0, 0);
	d[l] = acc;
}

// NACC independent accumulators; MODE 0: mfma 16x16x4 only, 1: v_fma_f64 only (one per mfma slot
// x VPER), 2: both interleaved, 3: mfma 4x4x4 only
template <int MODE, int NACC, int VPER>
__global__ void rate(double *out, int n, double seed)
{
	const int l = threadIdx.x;
	double a = seed + l, b = seed * 0.5 + (l & 7);
	double4v acc[NACC];
	double sacc[NACC];
	double v[NACC * VPER > 0 ? NACC * VPER : 1];
	for (int i = 0; i < NACC; i++) {
		acc[i] = (double4v){ 0, 0, 0, 0 };
		sacc[i] = 0;
	}
	for (int i = 0; i < NACC * VPER; i++)
		v[i] = i;
	for (int it = 0; it < n; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++) {
			if (MODE == 0 || MODE == 2)
				acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
			if (MODE == 3)
				sacc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, sacc[i], 0, 0, 0);
			if (MODE == 1 || MODE == 2) {
#pragma unroll
				for (int k = 0; k < VPER; k++)
					v[i * VPER + k] = __builtin_fma(a, b, v[i * VPER + k]);
			}
		}
	}
	double s = 0;
	for (int i = 0; i < NACC; i++)
		s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + sacc[i];
	for (int i = 0; i < NACC * VPER; i++)
		s += v[i];
	out[blockIdx.x * blockDim.x + l] = s;
}

template <typename F>
static double time_ms(F f, int reps)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	f();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	for (int i = 0; i < reps; i++)
		f();
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms;
	CHECK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

static double lcg(unsigned &s)
{
	s = s * 1664525u + 1013904223u;
	return (double) (s >> 8) / 16777216.0;
}

int main()
{
	double *da, *db, *dc, *dd;
	CHECK(hipMalloc(&da, 64 * 8));
	CHECK(hipMalloc(&db, 64 * 8));
	CHECK(hipMalloc(&dc, 256 * 8));
	CHECK(hipMalloc(&dd, 256 * 8));
	{
		// layout: A[i][k] = 100 i + k + 1 (lane i + 16 k), B[k][j] = asymmetric
		double ha[64], hb[64], hc[256], hd[256];
		for (int l = 0; l < 64; l++) {
			const int i = l & 15, k = l >> 4;
			ha[l] = 100.0 * i + k + 1;
			hb[l] = (k + 1) * 1000.0 + (l & 15) * 3 + 7; // B[k][j]
		}
		memset(hc, 0, sizeof(hc));
		CHECK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice));
		CHECK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
		CHECK(hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice));
		layout16<<<1, 64>>>(da, db, dc, dd);
		CHECK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
		int bad = 0;
		for (int l = 0; l < 64; l++)
			for (int r = 0; r < 4; r++) {
				const int col = l & 15, row = (l >> 4) + 4 * r;
				double want = 0;
				for (int k = 0; k < 4; k++)
					want += (100.0 * row + k + 1) * ((k + 1) * 1000.0 + col * 3 + 7);
				if (hd[l * 4 + r] != want)
					bad++;
			}
		printf("16x16x4 f64 layout (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15): %s (%d bad)\n",
			bad ? "MISMATCH" : "ok", bad);

		// 4x4x4 4-block: hypothesis block = l >> 4?? or l / 4?  test both
		for (int l = 0; l < 64; l++) {
			ha[l] = l + 1;
			hb[l] = 1000.0 + 3 * l;
			hc[l] = 0;
		}
		CHECK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice));
		CHECK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
		CHECK(hipMemcpy(dc, hc, 64 * 8, hipMemcpyHostToDevice));
		layout4<<<1, 64>>>(da, db, dc, dd);
		CHECK(hipMemcpy(hd, dd, 64 * 8, hipMemcpyDeviceToHost));
		// hypothesis H1: lane l: block = l>>4, within block lane m = l&15: A[i=m&3][k=m>>2], B[k=m>>2][j=m&3], D[i=m>>2][j=m&3]
		int bad1 = 0, bad2 = 0;
		for (int l = 0; l < 64; l++) {
			const int blk = l >> 4, m = l & 15;
			double w1 = 0, w2 = 0;
			for (int k = 0; k < 4; k++) {
				// H1: D[i = m >> 2][j = m & 3]
				w1 += ha[blk * 16 + (m >> 2) + 4 * k] * hb[blk * 16 + (m & 3) + 4 * k];
				// H2: D[i = m & 3][j = m >> 2] with A[i][k] lane i + 4k, B[k][j] lane j + 4 k
				w2 += ha[blk * 16 + (m & 3) + 4 * k] * hb[blk * 16 + (m >> 2) + 4 * k];
			}
			if (hd[l] != w1) bad1++;
			if (hd[l] != w2) bad2++;
		}
		printf("4x4x4 f64 layout: H1 (D[m>>2][m&3]) %d bad, H2 (D[m&3][m>>2]) %d bad; d[0..7] =", bad1, bad2);
		for (int l = 0; l < 8; l++)
			printf(" %.0f", hd[l]);
		printf("\n");
	}
	{
		// summation order / fusedness on random mantissas: compare with candidate orders
		double ha[64], hb[64], hc[256], hd[256];
		unsigned s = 12345;
		int m_seq = 0, m_rev = 0, m_mul_add = 0, m_pair = 0, m_exact = 0, total = 0;
		for (int trial = 0; trial < 64; trial++) {
			for (int l = 0; l < 64; l++) {
				ha[l] = (lcg(s) - 0.5) * 40.0;
				hb[l] = lcg(s) * 255.0 * (trial & 1 ? 1.0 : 1e-3 * (1 + (l & 3) * 1000));
			}
			for (int l = 0; l < 256; l++)
				hc[l] = (lcg(s) - 0.5) * 1000.0;
			CHECK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice));
			CHECK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
			CHECK(hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice));
			layout16<<<1, 64>>>(da, db, dc, dd);
			CHECK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
			for (int l = 0; l < 64; l++)
				for (int r = 0; r < 4; r++) {
					const int col = l & 15, row = (l >> 4) + 4 * r;
					double A[4], B[4];
					for (int k = 0; k < 4; k++) {
						A[k] = ha[row + 16 * k];
						B[k] = hb[col + 16 * k];
					}
					const double c0 = hc[l * 4 + r];
					double seq = c0, rev = c0, ma = c0;
					for (int k = 0; k < 4; k++)
						seq = fma(A[k], B[k], seq);
					for (int k = 3; k >= 0; k--)
						rev = fma(A[k], B[k], rev);
					for (int k = 0; k < 4; k++) {
						volatile double p = A[k] * B[k];
						ma = ma + p;
					}
					const double pair = fma(A[3], B[3], fma(A[2], B[2], 0.0)) + fma(A[1], B[1], fma(A[0], B[0], c0));
					long double ex = c0;
					for (int k = 0; k < 4; k++)
						ex += (long double) A[k] * (long double) B[k];
					const double got = hd[l * 4 + r];
					total++;
					m_seq += got == seq;
					m_rev += got == rev;
					m_mul_add += got == ma;
					m_pair += got == pair;
					m_exact += got == (double) ex;
				}
		}
		printf("summation (of %d): == fma chain k asc %d | k desc %d | mul+add chain %d | pairwise %d | long-double-rounded %d\n",
			total, m_seq, m_rev, m_mul_add, m_pair, m_exact);
	}
	double *dout;
	CHECK(hipMalloc(&dout, 4096 * 1024 * 8));
	const int n = 4000;
	for (int wps : { 1, 2, 4 }) { // waves per SIMD
		const int threads = 256, blocks = 256 * wps;
		double t0 = time_ms([&] { rate<0, 4, 0><<<blocks, threads>>>(dout, n, 1.25); }, 3);
		double t3 = time_ms([&] { rate<3, 4, 0><<<blocks, threads>>>(dout, n, 1.25); }, 3);
		double t1 = time_ms([&] { rate<1, 4, 4><<<blocks, threads>>>(dout, n, 1.25); }, 3);
		double t2 = time_ms([&] { rate<2, 4, 4><<<blocks, threads>>>(dout, n, 1.25); }, 3);
		double t2b = time_ms([&] { rate<2, 4, 8><<<blocks, threads>>>(dout, n, 1.25); }, 3);
		double t1b = time_ms([&] { rate<1, 4, 8><<<blocks, threads>>>(dout, n, 1.25); }, 3);
		const double instr = (double) n * 4 * wps; // mfma per SIMD
		printf("%d wave/SIMD: mfma16x16x4 %.3f ms = %.1f cyc/instr/SIMD (%.1f TFLOP/s) | mfma4x4x4 %.3f ms = %.1f cyc | "
			   "fma x4 %.3f ms = %.2f cyc/fma | mfma + 4 fma %.3f ms | fma x8 %.3f ms | mfma + 8 fma %.3f ms\n",
			wps, t0, t0 * 1e-3 * 2.4e9 / instr, 256.0 * 4 * instr * 2048 / (t0 * 1e-3) / 1e12, t3,
			t3 * 1e-3 * 2.4e9 / instr, t1, t1 * 1e-3 * 2.4e9 / (instr * 4), t2, t1b, t2b);
	}
	return 0;
}
