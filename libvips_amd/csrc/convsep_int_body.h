// The horizontal pass of convsep_stream (convsep_stream.hip) for windows that hold nothing but the
// integers 0 .. 255 -- a float image that was cast from uchar, BASELINE config 3's input and most
// real float sRGB images.  convi on float input (convi.c:721-741) sums (double) coefficient * pixel in
// double and stores (float) (sum / scale + offset): with integer pixels and integer coefficients every
// partial sum is an integer below 2^53, so the double sum IS the integer sum whatever the order, and
// the pass can run on packed bytes:
//
//   * a thread's window of 36 floats becomes 9 dwords of bytes (v_cvt_pk_u8_f32); every byte is
//     converted back (v_cvt_f32_ubyteN) and compared bit for bit with the float it came from -- for
//     ANY semantics of the conversion, a float that is not one of the integers 0 .. 255 (a fraction,
//     a negative or larger number, -0, inf, NaN) differs from what comes back.  One lane whose window
//     fails sends its whole wave down the double path (a wave-uniform branch): the result never
//     depends on which path ran;
//   * output k of the thread's 8 is sum_t c[t] * w[k + t]: 8 v_dot4_u32_u8 over the dwords k / 4 ..
//     k / 4 + 7 with the coefficients shifted by k mod 4 bytes (four sets of 8 dwords made on the host,
//     read from LDS as broadcasts; n <= 29 taps so that 3 + n <= 32);
//   * (float) ((double) S / scale) in three single-precision operations (q0 = S r, e = fma(-scale,
//     q0, S), q = fma(e, r, q0)): the host compares that sequence with the double division for EVERY
//     sum the mask can make (0 .. 255 sum(c), < 2^24) before it enables the path (hint_div_check).
//
// Written against gcn.h so that the CPU suite runs it on the host (tests/test_convsep_int.py).
#pragma once

#include "gcn.h"

namespace vh {

constexpr int HINT_MAXN = 29;  // taps
constexpr int HINT_SETS = 32;  // coefficient dwords: 4 byte shifts x 8 dwords

VH_DEV unsigned int hint_bits(float v) { return __builtin_bit_cast(unsigned int, v); }

// four floats -> one dword of bytes; `bad` collects the bits in which a float differs from its byte
VH_DEV unsigned int hint_pack4(float x0, float x1, float x2, float x3, unsigned int &bad)
{
	unsigned int pk = cvt_pk_u8(x0, 0, 0);
	pk = cvt_pk_u8(x1, 1, pk);
	pk = cvt_pk_u8(x2, 2, pk);
	pk = cvt_pk_u8(x3, 3, pk);
	bad |= hint_bits(x0) ^ hint_bits((float) (pk & 0xffu));
	bad |= hint_bits(x1) ^ hint_bits((float) ((pk >> 8) & 0xffu));
	bad |= hint_bits(x2) ^ hint_bits((float) ((pk >> 16) & 0xffu));
	bad |= hint_bits(x3) ^ hint_bits((float) (pk >> 24));
	// (kept as a word in a vector register: left alone the compiler turns the test into 36 lane
	// masks in scalar registers it does not have)
	opaque(bad);
	return pk;
}

// (float) ((double) S / scale + 0.0) for an integer 0 <= S < 2^24 (see the head of the file; a
// scale of 1 goes through the same operations: q0 = S, e = 0)
VH_DEV float hint_fin(unsigned int S, float scale, float rscale)
{
	const float s = (float) S;
	const float q0 = s * rscale;
	const float e = __builtin_fmaf(-scale, q0, s);
	return __builtin_fmaf(e, rscale, q0);
}

// The 8 outputs of a thread from its 9 window dwords.  kci: the 32 coefficient dwords (LDS).
// out[k * out_step] = output k.
VH_DEV void hint_outputs(const unsigned int (&w)[9], const unsigned int *kci, float scale, float rscale, float *out,
	int out_step)
{
#pragma unroll
	for (int s = 0; s < 4; s++) {
		unsigned int c[8];
#pragma unroll
		for (int j = 0; j < 8; j++)
			c[j] = kci[8 * s + j];
		unsigned int s0 = 0, s1 = 0;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			s0 = udot4(w[j], c[j], s0);
			s1 = udot4(w[j + 1], c[j], s1);
		}
		out[s * out_step] = hint_fin(s0, scale, rscale);
		out[(s + 4) * out_step] = hint_fin(s1, scale, rscale);
	}
}

} // namespace vh
