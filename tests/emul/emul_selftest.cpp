// TEST INFRASTRUCTURE: the fiber emulator checked against itself -- the wave vote's exec-mask meaning, the quad
// DPP move and v_mfma_f32_4x4x4_16b_f16 (kernel_prelude.h) against their definitions.  Prints "OK".
#include "kernel_prelude.h"

#include <cstdio>
#include <vector>

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4x __attribute__((ext_vector_type(4)));

static int failures = 0;
#define CHECK(c) \
	do { \
		if (!(c)) { \
			failures++; \
			fprintf(stderr, "FAIL line %d: %s\n", __LINE__, #c); \
		} \
	} while (0)

int main()
{
	const int threads = 192; // three waves
	std::vector<unsigned long long> seen(threads, 0), seen2(threads, 0);
	std::vector<int> dpp(threads, -1);
	std::vector<float> d(threads * 4, 0.0f);
	std::vector<half4> A(threads), B(threads);
	for (int t = 0; t < threads; t++)
		for (int k = 0; k < 4; k++) {
			A[t][k] = (_Float16) (float) ((t * 7 + k * 3) % 11 - 5);
			B[t][k] = (_Float16) (float) ((t * 5 + k) % 13 - 6);
		}
	emul::run_block(threads, [&]() {
		const int t = emul::current_tid();
		// lanes 3, 6, 9 ... of every wave are masked off for the first vote (they go straight to the barrier)
		if (t % 3 != 0)
			seen[t] = __builtin_amdgcn_ballot_w64((t & 1) != 0);
		emul::barrier();
		// two votes in a row without a barrier between them; wave 1 does not vote at all
		if (t / 64 != 1) {
			const unsigned long long m1 = __builtin_amdgcn_ballot_w64(t % 5 == 0);
			const unsigned long long m2 = __builtin_amdgcn_ballot_w64(t % 7 == 0);
			seen2[t] = m1 ^ (m2 << 1);
		}
		emul::barrier();
		dpp[t] = __builtin_amdgcn_mov_dpp(t * 10, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
		float4x c = { 1.0f, 2.0f, 3.0f, 4.0f };
		const float4x r = __builtin_amdgcn_mfma_f32_4x4x4f16(A[t], B[t], c, 0, 0, 0);
		for (int i = 0; i < 4; i++)
			d[t * 4 + i] = r[i];
	});
	for (int t = 0; t < threads; t++) {
		const int w0 = t & ~63;
		unsigned long long want = 0, w1 = 0, w2 = 0;
		for (int l = 0; l < 64; l++) {
			const int u = w0 + l;
			if (u % 3 != 0 && (u & 1))
				want |= 1ull << l;
			if (u % 5 == 0)
				w1 |= 1ull << l;
			if (u % 7 == 0)
				w2 |= 1ull << l;
		}
		CHECK(seen[t] == (t % 3 != 0 ? want : 0ull));
		CHECK(seen2[t] == (t / 64 != 1 ? (w1 ^ (w2 << 1)) : 0ull));
		CHECK(dpp[t] == ((t & ~3) | ((t & 3) ^ 1)) * 10);
		// D[i][j] of lane j's block = C[i] + sum_k A(lane i)[k] * B(lane j)[k]
		const int base = t & ~3;
		for (int i = 0; i < 4; i++) {
			float want_d = (float) (i + 1);
			for (int k = 0; k < 4; k++)
				want_d += (float) A[base + i][k] * (float) B[t][k];
			CHECK(d[t * 4 + i] == want_d);
		}
	}
	if (failures)
		return 1;
	printf("OK\n");
	return 0;
}
