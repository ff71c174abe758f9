// vips_shrinkh on uchar images, on packed bytes: the kernel body, written against gcn.h (product) /
// tests/emul/gcn.h (host fibers, CPU suite).
//
// The general kernel of resample.hip gives a thread one output BYTE and lets it walk its box with
// byte loads `bands` apart (31 % of HBM on 8192^2 x 3).  Here a lane owns 4 neighbouring output
// pixels of a row: their boxes are 4 hshrink pixels = hshrink B dwords that lie one after the other
// in memory, and so do the spans of neighbouring lanes, so a wave reads 256 hshrink B contiguous
// bytes of the row with whole-dword loads and writes 256 B contiguous bytes.  A group of 4 pixels (B
// dwords) is turned into B planar dwords (4 pixels of one band each, v_perm: conv_u8_body.h
// cu8_planar) and a box sum is v_dot4_u32_u8 of a planar dword with a byte mask of ones -- the box
// (or the two boxes) the group's pixels belong to is known at compile time (template HS), or the
// boxes are whole groups (HS = 0: any multiple of 4).  shrinkh.c:78-92:
//     q = ((hshrink / 2 + sum of the box) * ((1 << 32) / (256 hshrink))) >> 24
// in 32 bits: the product is below 2^32 (the sum is at most 255.5 hshrink), its top byte is the
// pixel; the four top bytes of a band are gathered with v_perm and the planar dwords interleaved
// back (cu8_interleave).  Lanes at the right edge (a quad that runs over the output's or the
// input's last pixel: vips_embed COPY, shrinkh.c:383-386) take the byte-by-byte form.
#pragma once

#include "conv_u8_body.h"

namespace vh {

constexpr int SH8_NT = 256;
constexpr int SH8_ROWS = 2; // rows in flight per lane

struct Sh8Args {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int in_width, out_width, height;
	int hshrink;
	unsigned int mult; // (1 << 32) / (256 hshrink)
	int quads;         // lanes across a row: ceil(out_width / 4)
};

// the four box sums of one band, as products whose top byte is the pixel -> one planar dword
VH_DEV unsigned int sh8_pack(const unsigned int (&p)[4])
{
	const unsigned int lo = perm(p[1], p[0], 0x0c0c0703u);
	const unsigned int hi = perm(p[3], p[2], 0x0c0c0703u);
	return perm(hi, lo, 0x05040100u);
}

template <int B, int HS>
VH_DEV void shrinkh_u8_body(const Sh8Args &a, int bx, int by, int gy)
{
	const int q = bx * SH8_NT + tid();
	if (q >= a.quads)
		return;
	const int hs = HS ? HS : a.hshrink;
	const int x0 = 4 * q;
	const unsigned int amend = (unsigned int) (hs >> 1);
	// the lane's span is whole boxes inside the image
	const bool whole = x0 + 4 <= a.out_width && (long long) (x0 + 4) * hs <= a.in_width;
	const unsigned int span0 = (unsigned int) x0 * (unsigned int) hs * B; // byte offset of the span in a row
	for (int y0 = by * SH8_ROWS; y0 < a.height; y0 += gy * SH8_ROWS) {
		if (whole) {
			unsigned int acc[SH8_ROWS][4][B];
			gptr_in line[SH8_ROWS];
#pragma unroll
			for (int r = 0; r < SH8_ROWS; r++) {
				const int y = min(y0 + r, a.height - 1);
				line[r] = gptr_in_of((unsigned long long) a.in) + (long long) y * a.in_stride;
#pragma unroll
				for (int k = 0; k < 4; k++)
#pragma unroll
					for (int b = 0; b < B; b++)
						acc[r][k][b] = amend;
			}
			if constexpr (HS != 0) {
				// group g = pixels 4 g .. 4 g + 3 of the span; pixel p lies in box p / HS
#pragma unroll
				for (int g = 0; g < HS; g++) {
#pragma unroll
					for (int r = 0; r < SH8_ROWS; r++) {
						unsigned int w[B], P[B];
						gload_dwords<B>(line[r], span0 + (unsigned int) (4 * g * B), w);
						cu8_planar<B>(w, P);
						constexpr int first = 0;
						(void) first;
#pragma unroll
						for (int k = 0; k < 4; k++) {
							// bytes of the group that fall in box k
							unsigned int mask = 0;
#pragma unroll
							for (int j = 0; j < 4; j++)
								if ((4 * g + j) / HS == k)
									mask |= 1u << (8 * j);
							if (mask) {
#pragma unroll
								for (int b = 0; b < B; b++)
									acc[r][k][b] = udot4(P[b], mask, acc[r][k][b]);
							}
						}
					}
				}
			}
			else {
				// boxes of whole groups
				const int groups = hs >> 2;
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const unsigned int box0 = span0 + (unsigned int) (k * hs * B);
					for (int g = 0; g < groups; g++) {
#pragma unroll
						for (int r = 0; r < SH8_ROWS; r++) {
							unsigned int w[B], P[B];
							gload_dwords<B>(line[r], box0 + (unsigned int) (4 * g * B), w);
							cu8_planar<B>(w, P);
#pragma unroll
							for (int b = 0; b < B; b++)
								acc[r][k][b] = udot4(P[b], 0x01010101u, acc[r][k][b]);
						}
					}
				}
			}
#pragma unroll
			for (int r = 0; r < SH8_ROWS; r++) {
				if (y0 + r < a.height) {
					unsigned int O[B], w[B];
#pragma unroll
					for (int b = 0; b < B; b++) {
						const unsigned int p[4] = { acc[r][0][b] * a.mult, acc[r][1][b] * a.mult, acc[r][2][b] * a.mult,
							acc[r][3][b] * a.mult };
						O[b] = sh8_pack(p);
					}
					cu8_interleave<B>(O, w);
					const gptr_out dst = gptr_out_of((unsigned long long) a.out) + (long long) (y0 + r) * a.out_stride + (long long) x0 * B;
					gstore_dwords<B>(dst, w);
				}
			}
		}
		else {
			// the right edge: byte by byte, input columns clamped to the image
			for (int r = 0; r < SH8_ROWS; r++) {
				const int y = y0 + r;
				if (y >= a.height)
					break;
				const gptr_in src = gptr_in_of((unsigned long long) a.in) + (long long) y * a.in_stride;
				const gptr_out dst = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride;
				for (int x = x0; x < min(x0 + 4, a.out_width); x++)
					for (int b = 0; b < B; b++) {
						unsigned int sum = amend;
						for (int i = 0; i < hs; i++) {
							const int px = min(x * hs + i, a.in_width - 1);
							sum += gload8(src, (unsigned int) (px * B + b));
						}
						gstore8(dst + x * B + b, (unsigned char) ((sum * a.mult) >> 24));
					}
			}
		}
	}
}

} // namespace vh
