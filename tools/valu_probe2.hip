// Issue rates of the instruction classes the C3 kernel (convsep_stream.hip) spends its non-FMA
// issue slots on, one inline-asm instruction at a time: cycles per wave64 instruction per SIMD with
// 8 waves per SIMD (issue-bound) and with ONE wave per SIMD (dependent-issue latency visible).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe2.hip -o /tmp/valu_probe2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

// 8 independent chains per lane, 8 x 8 instructions per turn of the loop
template <int OP>
__global__ void k(double *out, int n, double seed)
{
	double a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	float f0 = (float) a0, f1 = (float) a1, f2 = (float) a2, f3 = (float) a3, f4 = (float) a4, f5 = (float) a5, f6 = (float) a6,
		  f7 = (float) a7;
	int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
	const double y = 0.999 + seed * 1e-9, x = 1e-3 * seed;
	int sacc = 0;
	__shared__ float lds[4096];
	for (int i = threadIdx.x; i < 4096; i += blockDim.x)
		lds[i] = (float) i;
	__syncthreads();
	const unsigned int la = (threadIdx.x * 16) & 16383;
	typedef float vf4 __attribute__((ext_vector_type(4)));
	vf4 q0 = { 0, 0, 0, 0 };
	for (int it = 0; it < n; it++) {
#pragma unroll
	for (int rep = 0; rep < 8; rep++) {
		if (OP == 0) { // v_fma_f64
#define I(A) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(A) : "v"(y), "v"(x));
			I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)
#undef I
		}
		else if (OP == 1) { // v_add_f64
#define I(A) asm volatile("v_add_f64 %0, %0, %1" : "+v"(A) : "v"(x));
			I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)
#undef I
		}
		else if (OP == 2) { // v_mul_f64
#define I(A) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(A) : "v"(y));
			I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)
#undef I
		}
		else if (OP == 3) { // v_cvt_f64_f32
#define I(A, F) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(A) : "v"(F));
			I(a0, f0) I(a1, f1) I(a2, f2) I(a3, f3) I(a4, f4) I(a5, f5) I(a6, f6) I(a7, f7)
#undef I
		}
		else if (OP == 4) { // v_cvt_f32_f64
#define I(A, F) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(F) : "v"(A));
			I(a0, f0) I(a1, f1) I(a2, f2) I(a3, f3) I(a4, f4) I(a5, f5) I(a6, f6) I(a7, f7)
#undef I
		}
		else if (OP == 5) { // v_div_fixup_f64
#define I(A) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(A) : "v"(y), "v"(x));
			I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)
#undef I
		}
		else if (OP == 6) { // v_cmp_u_f64 (NaN test) into vcc
#define I(A) asm volatile("v_cmp_u_f64 vcc, %0, %0" : : "v"(A) : "vcc");
			I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a7)
#undef I
		}
		else if (OP == 7) { // v_mul_lo_u32
#define I(A) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(A) : "v"(i7 | 1));
			I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
		}
		else if (OP == 8) { // v_readlane_b32
			int s;
#define I(A) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(A)); sacc += s;
			I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i7)
#undef I
		}
		else if (OP == 9) { // v_fma_f32
#define I(A) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "v"(f7), "v"(f6));
			I(f0) I(f1) I(f2) I(f3) I(f4) I(f5) I(f0) I(f1)
#undef I
		}
		else if (OP == 10) { // v_cvt_i32_f32
#define I(A, F) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(A) : "v"(F));
			I(i0, f0) I(i1, f1) I(i2, f2) I(i3, f3) I(i4, f4) I(i5, f5) I(i6, f6) I(i7, f7)
#undef I
		}
		else if (OP == 11) { // ds_read_b128, consecutive 16-byte lanes
#define I() asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q0) : "v"(la) : "memory"); f0 += q0[0];
			I() I() I() I() I() I() I() I()
#undef I
		}
		else if (OP == 12) { // v_fmac_f64 with an SGPR operand (the kernel's tap form)
#define I(A) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(A) : "s"(y), "v"(a7));
			I(a0) I(a1) I(a2) I(a3) I(a4) I(a5) I(a6) I(a0)
#undef I
		}
		else if (OP == 13) { // v_pk_mul_f32
			typedef float vf2 __attribute__((ext_vector_type(2)));
			vf2 p0 = { f0, f1 }, p1 = { f2, f3 };
#define I(A) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(A) : "v"(p1));
			I(p0) I(p0) I(p0) I(p0) I(p0) I(p0) I(p0) I(p0)
#undef I
			f0 = p0[0];
		}
		else if (OP == 14) { // v_med3_i32
#define I(A) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(A) : "v"(i7));
			I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
		}
	}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] =
		a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + sacc;
}

template <int OP>
void run(const char *name)
{
	double *out;
	hipMalloc(&out, 256 * 1024 * 8 * sizeof(double));
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	const int n = 256;
	double cyc[2];
	for (int mode = 0; mode < 2; mode++) {
		// mode 0: 8 waves per SIMD (2048 blocks of 256 over 256 CUs); mode 1: one wave per SIMD
		const int blocks = mode == 0 ? 256 * 8 : 256;
		k<OP><<<blocks, 256>>>(out, n, 1);
		hipDeviceSynchronize();
		hipEventRecord(a);
		k<OP><<<blocks, 256>>>(out, n, 2);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms;
		hipEventElapsedTime(&ms, a, b);
		const double wi = (double) blocks * 4 / 1024 * n * 64; // wave-instructions per SIMD
		cyc[mode] = ms * 1e6 / wi * 2.4;
	}
	printf("%-28s %6.2f cycles per wave-instr per SIMD (8 waves), %6.2f (1 wave)\n", name, cyc[0], cyc[1]);
	hipFree(out);
}

int main()
{
	run<0>("v_fma_f64");
	run<12>("v_fmac_f64 sgpr");
	run<1>("v_add_f64");
	run<2>("v_mul_f64");
	run<3>("v_cvt_f64_f32");
	run<4>("v_cvt_f32_f64");
	run<5>("v_div_fixup_f64");
	run<6>("v_cmp_u_f64");
	run<7>("v_mul_lo_u32");
	run<8>("v_readlane_b32");
	run<9>("v_fma_f32");
	run<13>("v_pk_mul_f32");
	run<10>("v_cvt_i32_f32");
	run<14>("v_med3_i32");
	run<11>("ds_read_b128 + wait");
	return 0;
}
