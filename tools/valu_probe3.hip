// Issue rates of the instructions the integer horizontal pass of convsep_stream
// (libvips_amd/csrc/convsep_int_body.h) is made of -- and of the ones that could replace them:
// nanoseconds per wave64 instruction per SIMD with 8 waves per SIMD and with ONE wave per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe3.hip -o tools/valu_probe3
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void k(float *out, int n, int seed)
{
	unsigned int i0 = threadIdx.x + seed, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
	float f0 = (float) i0, f1 = (float) i1, f2 = (float) i2, f3 = (float) i3, f4 = (float) i4, f5 = (float) i5, f6 = (float) i6,
		  f7 = (float) i7;
	const unsigned int c = 0x01020304u * seed, addr = ((threadIdx.x + 3) & 63) << 2;
	typedef float vf2 __attribute__((ext_vector_type(2)));
	vf2 p0 = { f0, f1 }, p1 = { f2, f3 }, p2 = { f4, f5 }, p3 = { f6, f7 };
	for (int it = 0; it < n; it++) {
#pragma unroll
		for (int rep = 0; rep < 8; rep++) {
			if (OP == 0) {
#define I(A) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(A) : "v"(c), "v"(i7));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
			}
			else if (OP == 1) {
#define I(A, F) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(A) : "v"(F));
				I(i0, f0) I(i1, f1) I(i2, f2) I(i3, f3) I(i4, f4) I(i5, f5) I(i6, f6) I(i7, f7)
#undef I
			}
			else if (OP == 2) {
#define I(A, F) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(F) : "v"(A));
				I(i0, f0) I(i1, f1) I(i2, f2) I(i3, f3) I(i4, f4) I(i5, f5) I(i6, f6) I(i7, f7)
#undef I
			}
			else if (OP == 3) {
#define I(A) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A) : "v"(c));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i7)
#undef I
			}
			else if (OP == 4) {
#define I(A) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(A) : "v"(c), "v"(i7));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
			}
			else if (OP == 5) {
#define I(A) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(A) : "v"(c), "v"(i7));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
			}
			else if (OP == 6) {
#define I(A) asm volatile("v_add_f32 %0, %0, %1" : "+v"(A) : "v"(f7));
				I(f0) I(f1) I(f2) I(f3) I(f4) I(f5) I(f6) I(f0)
#undef I
			}
			else if (OP == 7) {
#define I(A) asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(A)); 
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i7)
#undef I
			}
			else if (OP == 8) {
#define I(A) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(A) : "v"(addr) : "memory");
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i7)
#undef I
			}
			else if (OP == 9) {
#define I(A) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(A) : "v"(c), "v"(i7));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
			}
			else if (OP == 10) {
#define I(A) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xbe" : "+v"(A) : "v"(c), "v"(i7));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
			}
			else if (OP == 11) {
#define I(A) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(A) : "v"(c), "v"(i7));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
			}
			else if (OP == 12) {
#define I(A) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(A) : "v"(c), "v"(i7));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
			}
			else if (OP == 13) {
#define I(A) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(A) : "v"(p3), "v"(p3));
				I(p0) I(p1) I(p2) I(p0) I(p1) I(p2) I(p0) I(p1)
#undef I
			}
			else if (OP == 14) {
#define I(A) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(A) : "v"(p3));
				I(p0) I(p1) I(p2) I(p0) I(p1) I(p2) I(p0) I(p1)
#undef I
			}
			else if (OP == 15) {
#define I(A, F) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(F) : "v"(A));
				I(i0, f0) I(i1, f1) I(i2, f2) I(i3, f3) I(i4, f4) I(i5, f5) I(i6, f6) I(i7, f7)
#undef I
			}
			else if (OP == 16) {
#define I(A) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(A), "v"(c) : "vcc");
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i7)
#undef I
			}
			else if (OP == 17) {
#define I(A) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(A) : "v"(c), "v"(i7));
				I(i0) I(i1) I(i2) I(i3) I(i4) I(i5) I(i6) I(i0)
#undef I
			}
			else if (OP == 18) {
#define I(A) asm volatile("v_fract_f32 %0, %0" : "+v"(A));
				I(f0) I(f1) I(f2) I(f3) I(f4) I(f5) I(f6) I(f7)
#undef I
			}
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] +
		(float) (i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
}

template <int OP>
void run(const char *name)
{
	float *out;
	hipMalloc(&out, 256 * 1024 * 8 * sizeof(float));
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	const int n = 256;
	double ns[2];
	for (int mode = 0; mode < 2; mode++) {
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
		ns[mode] = ms * 1e6 / wi;
	}
	printf("%-22s %6.2f ns per wave-instr per SIMD (8 waves), %6.2f (1 wave)\n", name, ns[0], ns[1]);
	hipFree(out);
}

int main()
{
	run<6>("v_add_f32");
	run<0>("v_dot4_u32_u8");
	run<17>("v_dot4_i32_i8");
	run<12>("v_dot2_i32_i16");
	run<9>("v_mad_u32_u24");
	run<1>("v_cvt_pk_u8_f32");
	run<2>("v_cvt_f32_ubyte1");
	run<15>("v_cvt_f32_u32");
	run<18>("v_fract_f32");
	run<3>("v_xor_b32");
	run<4>("v_or3_b32");
	run<10>("v_bitop3_b32");
	run<11>("v_min3_u32");
	run<16>("v_cmp_ne_u32");
	run<5>("v_perm_b32");
	run<13>("v_pk_fma_f32");
	run<14>("v_pk_add_f32");
	run<7>("v_mov_b32 dpp wave_shl");
	run<8>("ds_bpermute + wait");
	return 0;
}
