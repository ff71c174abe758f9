// XYZ2Lab's cube-root table (XYZ2Lab.c:92-106: 100000 floats made with the host's cbrtf) without the table,
// second form (round 5): every entry AND its successor, bit for bit, from 56 KB of LDS in ~27 vector
// instructions and two LDS reads per pair -- cbrt_exact.h's degree-4 block polynomial takes ~45 and seven.
//
//   * entry i falls in block key(i) = bits((float) (i + 2)) >> 16: the integers i + 2 that share an exponent and
//     their top 7 mantissa bits (every entry below 254 is its own block; blocks of 2^(e - 7) above; the + 2 puts a
//     block boundary at entry 886, where the table's linear arm ends -- the two arms do not meet within a unit);
//     the block's first entry is (float) (i + 2) with its low 16 bits cleared, j the exact float difference;
//   * in a block T[i0 + j] ~ fma(fma(c2, j, c1), j, c0) with c0 = T[i0] itself and c1, c2 a least-squares
//     quadratic (j / i0 < 1 / 128: what the quadratic leaves out is below a tenth of a unit in the last place);
//   * the float this makes is within one unit of the table's entry almost everywhere (cbrtf is not correctly
//     rounded: 2 entries of 100 000 sit two units away on glibc 2.35); the difference r[i] is stored as a signed
//     2-bit field, -2 .. 1, plus a per-block bias of 0 or 1 (the lowest bit of the block's c2), so a block holds
//     -2 .. 1 or -1 .. 2 (j = 0 gives c0 exactly: a block's first entry has r = 0);
//   * the pair's second entry comes from the same block (j + 1) -- or, when i is its block's last entry, is the
//     next block's first, kept in the block record itself.
// The host makes c1, c2, r[] against its own table by running THIS function (same operations, same order) and
// checks every pair a kernel can ask for; a host whose cbrtf does not fit gets no tables and the callers keep
// the older forms.
#pragma once

#include <cstdint>
#include <cstring>

#ifndef VH_CBRT_FN
#define VH_CBRT_FN static inline
#endif

namespace vh {

constexpr int CBQ_N = 100000;                         // QUANT_ELEMENTS
constexpr int CBQ_SHIFT = 2;                          // blocks are cut at i + 2: entry 886, where the table's linear arm
                                                      // ends and its cube roots begin, then starts a block
constexpr int CBQ_KEY0 = 0x40000000 >> 16;            // key of entry 0: bits(2.0f) >> 16
constexpr int CBQ_BLOCKS = (0x47C35080 >> 16) - CBQ_KEY0 + 1; // ... through key(99999 + 2): 1988 records
constexpr int CBQ_RES_WORDS = CBQ_N / 16 + 2;         // 2 bits per entry, a spare word for the pair read

struct CbqBlock {
	float c0, c1, c2, next; // T[i0 + j] ~ c0 + j (c1 + j c2); next = the entry after the block's last
};

struct CbrtQuad {
	const CbqBlock *blk;
	const unsigned int *res;
};

VH_CBRT_FN unsigned int cbq_bits(float f)
{
	unsigned int u;
	memcpy(&u, &f, 4);
	return u;
}
VH_CBRT_FN float cbq_float(unsigned int u)
{
	float f;
	memcpy(&f, &u, 4);
	return f;
}
// fs = (float) (i + 2)
VH_CBRT_FN int cbq_key(float fs)
{
	return (int) (cbq_bits(fs) >> 16) - CBQ_KEY0;
}
// entry i0 + j of a block before its residual
VH_CBRT_FN float cbq_predict(const CbqBlock &q, float jf)
{
	return __builtin_fmaf(__builtin_fmaf(q.c2, jf, q.c1), jf, q.c0);
}

// table[i] and table[i + 1] - table[i] for 0 <= i <= 99998; fi = (float) i
VH_CBRT_FN void cbq_pair(const CbqBlock *blk, const unsigned int *res, int i, float fi, float *t0, float *dt)
{
	const float fs = fi + (float) CBQ_SHIFT;
	const unsigned int b = cbq_bits(fs);
	const CbqBlock q = blk[cbq_key(fs)];
	const float jf = fs - cbq_float(b & 0xffff0000u);
	const float p0 = cbq_predict(q, jf);
	float p1 = cbq_predict(q, jf + 1.0f);
	p1 = (cbq_bits(fs + 1.0f) >> 16) != (b >> 16) ? q.next : p1;
	const unsigned int w = (unsigned int) i >> 4, sh = 2u * ((unsigned int) i & 15u);
	const unsigned long long both = ((unsigned long long) res[w + 1] << 32) | res[w];
	const unsigned int rr = (unsigned int) (both >> sh);
	// residual = the 2-bit field, sign-extended, + the block's bias (the lowest bit of c2): -2 .. 1 or -1 .. 2
	const int bias = (int) (cbq_bits(q.c2) & 1u);
	const int r0 = ((int) (rr << 30) >> 30) + bias;
	// (the second entry's bias is its own block's: the next block's first entry has residual 0 by construction)
	const int r1 = (cbq_bits(fs + 1.0f) >> 16) != (b >> 16) ? 0 : ((int) (rr << 28) >> 30) + bias;
	const float v0 = cbq_float(cbq_bits(p0) + (unsigned int) r0);
	const float v1 = cbq_float(cbq_bits(p1) + (unsigned int) r1);
	*t0 = v0;
	*dt = v1 - v0;
}

} // namespace vh
