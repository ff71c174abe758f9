// Probe: operand layout and issue rate of v_mfma_f32_4x4x4_16b_f16 on gfx950.
// Hypothesis (16 independent 4x4 blocks, block = lane / 4):
//   A[i][k]: lane (i + 4*block), half k of its v4f16      B[k][j]: lane (j + 4*block), half k
//   D[i][j]: lane (j + 4*block), register i
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const _Float16 *a, const _Float16 *b, float *d)
{
	const int l = threadIdx.x;
	half4 av, bv;
	for (int k = 0; k < 4; k++) {
		av[k] = a[l * 4 + k];
		bv[k] = b[l * 4 + k];
	}
	float4v c = { 0, 0, 0, 0 };
	c = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, c, 0, 0, 0);
	for (int i = 0; i < 4; i++)
		d[l * 4 + i] = c[i];
}

__global__ void rate_kernel(float *out, int n)
{
	const int l = threadIdx.x;
	half4 av = { (_Float16) (l & 3), (_Float16) 1, (_Float16) 2, (_Float16) 3 };
	half4 bv = { (_Float16) 1, (_Float16) (l & 7), (_Float16) 1, (_Float16) 2 };
	float4v c[8];
	for (int i = 0; i < 8; i++)
		c[i] = (float4v){ 0, 0, 0, 0 };
	for (int it = 0; it < n; it++) {
#pragma unroll
		for (int i = 0; i < 8; i++)
			c[i] = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, c[i], 0, 0, 0);
	}
	float s = 0;
	for (int i = 0; i < 8; i++)
		s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
	out[blockIdx.x * blockDim.x + l] = s;
}

// Are f16 denormal B operands (bit pattern 0x00pp = p * 2^-24) honoured, exactly?
__global__ void denorm_kernel(const short *coef, const unsigned char *pix, float *d)
{
	const int l = threadIdx.x;
	half4 av, bv;
	for (int k = 0; k < 4; k++) {
		av[k] = (_Float16) (float) coef[l * 4 + k];
		const unsigned short bits = pix[l * 4 + k];
		bv[k] = __builtin_bit_cast(_Float16, bits);
	}
	float4v c = { 0, 0, 0, 0 };
	c = __builtin_amdgcn_mfma_f32_4x4x4f16(av, bv, c, 0, 0, 0);
	for (int i = 0; i < 4; i++)
		d[l * 4 + i] = c[i] * 16777216.0f;
	// v_cvt_pk_u8_f32 semantics on a few values
	if (l < 8) {
		const float tv[8] = { -3.7f, -0.4f, 0.49f, 0.5f, 1.5f, 254.6f, 255.5f, 300.0f };
		d[256 + l] = (float) __builtin_amdgcn_cvt_pk_u8_f32(tv[l], 0, 0);
	}
}

int main()
{
	{
		short hc[256];
		unsigned char hp[256];
		float hd[264];
		for (int l = 0; l < 64; l++)
			for (int k = 0; k < 4; k++) {
				hc[l * 4 + k] = (short) (((l * 37 + k * 101) % 4001) - 2000);
				hp[l * 4 + k] = (unsigned char) ((l * 13 + k * 59 + 7) & 255);
			}
		hp[0] = 0; hp[1] = 255; hp[2] = 1;
		short *dc; unsigned char *dp; float *dd;
		hipMalloc(&dc, sizeof(hc)); hipMalloc(&dp, sizeof(hp)); hipMalloc(&dd, sizeof(hd));
		hipMemcpy(dc, hc, sizeof(hc), hipMemcpyHostToDevice);
		hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice);
		denorm_kernel<<<1, 64>>>(dc, dp, dd);
		hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
		int bad = 0;
		for (int l = 0; l < 64; l++) {
			const int blk = l / 4, j = l % 4;
			for (int i = 0; i < 4; i++) {
				long want = 0;
				for (int k = 0; k < 4; k++)
					want += (long) hc[(blk * 4 + i) * 4 + k] * hp[(blk * 4 + j) * 4 + k];
				if ((float) want != hd[l * 4 + i])
					bad++;
			}
		}
		printf("f16 denormal B operands: %s (%d mismatches of 256; d[0..3] = %g %g %g %g)\n",
			bad ? "NOT exact" : "exact", bad, hd[0], hd[1], hd[2], hd[3]);
		printf("cvt_pk_u8_f32(-3.7 -0.4 0.49 0.5 1.5 254.6 255.5 300) = %g %g %g %g %g %g %g %g\n", hd[256], hd[257],
			hd[258], hd[259], hd[260], hd[261], hd[262], hd[263]);
	}
	_Float16 ha[256], hb[256];
	float hd[256];
	for (int l = 0; l < 64; l++)
		for (int k = 0; k < 4; k++) {
			ha[l * 4 + k] = (_Float16) (float) ((l * 7 + k * 3) % 11 + 1);
			hb[l * 4 + k] = (_Float16) (float) ((l * 5 + k * 2) % 13 + 1);
		}
	_Float16 *da, *db;
	float *dd;
	hipMalloc(&da, sizeof(ha));
	hipMalloc(&db, sizeof(hb));
	hipMalloc(&dd, sizeof(hd));
	hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
	hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
	layout_kernel<<<1, 64>>>(da, db, dd);
	hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
	int bad = 0;
	for (int l = 0; l < 64; l++) {
		const int blk = l / 4, j = l % 4;
		for (int i = 0; i < 4; i++) {
			float want = 0;
			for (int k = 0; k < 4; k++)
				want += (float) ha[(blk * 4 + i) * 4 + k] * (float) hb[(blk * 4 + j) * 4 + k];
			if (want != hd[l * 4 + i])
				bad++;
		}
	}
	printf("layout hypothesis: %s (%d mismatches of 256)\n", bad ? "WRONG" : "confirmed", bad);
	if (bad) {
		for (int l = 0; l < 8; l++)
			printf(" lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
	}

	float *out;
	hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
	const int n = 4096, blocks = 256 * 8;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	rate_kernel<<<blocks, 256>>>(out, n);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	rate_kernel<<<blocks, 256>>>(out, n);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	double wi = (double) blocks * 4 / 1024 * n * 8;
	printf("mfma_f32_4x4x4f16: %.3f ms -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz), 8 independent accumulators, 8 waves/SIMD\n",
		ms, ms * 1e6 / wi, ms * 1e6 / wi * 2.4);
	return 0;
}
