// How much does a divergent gather cost on this part?  Every lane of a wave reads 8 bytes at a
// random index of a table: in global memory (L2-resident, 800 KB as XYZ2Lab's cube-root table, or
// L1-resident, 16 KB) or in LDS (64 KB), against the same number of coalesced reads and against
// plain VALU work.  Prints lane-reads per cycle per CU (2.4 GHz nominal).
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct P { float x, y; };

template <int MODE>
__global__ void __launch_bounds__(512) probe(const P *table, unsigned int mask, int iters, float *out)
{
	__shared__ P lds[8192];
	const int t = threadIdx.x;
	for (int i = t; i < 8192; i += 512)
		lds[i] = table[i];
	__syncthreads();
	unsigned int s = (blockIdx.x * 512 + t) * 2654435761u + 12345u;
	float acc = 0.0f;
	// U independent reads in flight per lane
	constexpr int U = 8;
	for (int i = 0; i < iters; i += U) {
		unsigned int idx[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			s = s * 1664525u + 1013904223u;
			if (MODE == 0 || MODE == 2 || MODE == 4)
				idx[u] = (s >> 8) & mask; // random
			else
				idx[u] = ((s >> 8) & mask & ~63u) + (t & 63); // a wave reads 64 consecutive entries
		}
		if (MODE == 0 || MODE == 1) {
			P v[U];
#pragma unroll
			for (int u = 0; u < U; u++)
				v[u] = table[idx[u]];
#pragma unroll
			for (int u = 0; u < U; u++)
				acc += v[u].x * v[u].y;
		}
		else if (MODE == 2 || MODE == 3) {
			P v[U];
#pragma unroll
			for (int u = 0; u < U; u++)
				v[u] = lds[idx[u] & 8191];
#pragma unroll
			for (int u = 0; u < U; u++)
				acc += v[u].x * v[u].y;
		}
		else {
			// no memory: 8 dependent fmas in place of each read
#pragma unroll
			for (int u = 0; u < U; u++) {
				float v = __uint_as_float(0x3f800000u | (idx[u] & 0xffff));
#pragma unroll
				for (int k = 0; k < 8; k++)
					v = __fmaf_rn(v, 1.0001f, 0.5f);
				acc += v;
			}
		}
	}
	out[blockIdx.x * 512 + t] = acc;
}

int main()
{
	const int N = 1 << 17;
	std::vector<P> h(N);
	for (int i = 0; i < N; i++) { h[i].x = (float) i; h[i].y = 1.0f / (i + 1); }
	P *d; float *o;
	hipMalloc(&d, N * sizeof(P));
	hipMemcpy(d, h.data(), N * sizeof(P), hipMemcpyHostToDevice);
	const int blocks = 256 * 2, iters = 4096;
	hipMalloc(&o, blocks * 512 * sizeof(float));
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	struct { const char *name; int mode; unsigned int mask; } runs[] = {
		{ "global random, 1 MB table (L2)", 0, 0x1ffff }, { "global random, 256 KB table", 0, 0x7fff }, { "global random, 64 KB table", 0, 8191 }, { "global random, 16 KB table (L1)", 0, 2047 },
		{ "global coalesced, 1 MB", 1, 0x1ffff }, { "LDS random, 64 KB", 2, 8191 }, { "LDS consecutive", 3, 8191 },
		{ "8 dependent v_fma_f32, no read", 4, 0xffff } };
	for (auto &r : runs) {
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			switch (r.mode) {
			case 0: probe<0><<<blocks, 512>>>(d, r.mask, iters, o); break;
			case 1: probe<1><<<blocks, 512>>>(d, r.mask, iters, o); break;
			case 2: probe<2><<<blocks, 512>>>(d, r.mask, iters, o); break;
			case 3: probe<3><<<blocks, 512>>>(d, r.mask, iters, o); break;
			default: probe<4><<<blocks, 512>>>(d, r.mask, iters, o); break;
			}
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			if (rep == 1) {
				const double lane_reads = (double) blocks * 512 * iters;
				const double cycles = ms * 1e-3 * 2.4e9;
				printf("%-36s %8.3f ms  %6.2f lane-reads/cycle/CU  = %6.1f cycles per wave instruction per CU\n", r.name, ms,
					lane_reads / cycles / 256.0, 64.0 / (lane_reads / cycles / 256.0));
			}
		}
	}
	return 0;
}
