// vips_reduceh on uchar images with ONE coefficient row for the whole line (an integer shrink that
// is a multiple of 4: every output's first tap is `step` pixels after its neighbour's and the phase
// is constant), on packed bytes: the kernel body, written against gcn.h (product) /
// tests/emul/gcn.h (host fibers, CPU suite).
//
// reduceh_u8_lds (resample.hip) gives a thread one output BYTE and reads its taps from LDS one byte
// at a time, 49 conflicting ds_read_u8 per output for vips_reduce(8): 18 % of HBM.  Here
//
//   staging   a block owns 256 output pixels x 4 rows.  The pixels their taps touch are loaded as
//             groups of 4 pixels (B dwords, coalesced), turned into B PLANAR dwords (4 pixels of one
//             band each: conv_u8_body.h cu8_planar), biased to signed bytes (p ^ 0x80 = p - 128)
//             and stored to one LDS plane per (row, band); columns left and right of the image are
//             the edge pixel (vips_embed COPY, reduceh.cpp:515-520) -- clamped once, here;
//   taps      a thread owns 4 neighbouring output pixels of one row.  Their windows start `step`
//             pixels = step / 4 dwords apart in a plane, so the thread reads the union once
//             (ds_read_b128) and output k is the dot product of dwords k step / 4 .. + ND - 1 with
//             the coefficient row, cut into dwords at the window's byte offset (first tap mod 4: the
//             same for every output) on the host.  Coefficients are 16-bit (reduceh.cpp:483-506),
//             v_dot4_i32_i8 takes bytes: c = 128 ch + cl with cl = c & 127, ch = c >> 7, so
//                 sum c p = 128 dot4(p', ch) + dot4(p', cl) + 128 sum c        (p' = p - 128)
//             in exact 32-bit integers, (n + 3) / 2 instructions per output instead of n
//             multiply-adds and n LDS reads; the coefficient dwords are scalar operands;
//   output    (sum + 2048) >> 12 clipped to 0 .. 255 (templates.h:152-157), the four pixels'
//             bands interleaved back (cu8_interleave) and stored as B whole dwords.
#pragma once

#include "conv_u8_body.h"

namespace vh {

constexpr int RH8_NT = 256;
constexpr int RH8_QUADS = 64; // quads of output pixels per block
constexpr int RH8_ROWS = 4;   // rows per trip: thread -> (row, quad)
constexpr int RH8_MAXND = 13; // window dwords of one output: masks up to 49 taps

struct Rh8Args {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int in_width; // of the image: columns are clamped to it
	int out_width, height;
	int f_al;     // first tap of output pixel 0, rounded down to a multiple of 4 (may be negative)
	int step;     // pixels between the first taps of neighbouring outputs: 4 STEP4
	int plane_dw; // dwords per staged plane: 255 STEP4 + ND, rounded up to a multiple of 4, + 4 (the last quad's wide read)
	int kconst;   // 128 * sum(c) + 2048
	unsigned int chi[RH8_MAXND], clo[RH8_MAXND];
};

template <int B, int STEP4, int ND>
VH_DEV void reduceh_u8p_body(const Rh8Args &a, int bx, int by, int gy, unsigned int *lds)
{
	constexpr int WN = 3 * STEP4 + ND;   // window dwords of a quad
	constexpr int WR = (WN + 3) / 4;     // ... as 16-byte reads
	constexpr int G = 255 * STEP4 + ND;  // groups of 4 pixels a block stages per row
	const int t = tid();
	const int xb = bx * (4 * RH8_QUADS);  // the block's first output pixel
	const int p_al = a.f_al + xb * a.step; // first staged pixel: a multiple of 4
	const int r = t / RH8_QUADS, q = t % RH8_QUADS;
	const int x0 = xb + 4 * q;
	// groups of the span that some output of this block reads
	const int nq = min(RH8_QUADS, (a.out_width - xb + 3) / 4);
	const int groups = min(G, (4 * nq - 1) * STEP4 + ND);
	for (int y0 = by * RH8_ROWS; y0 < a.height; y0 += gy * RH8_ROWS) {
		barrier();
		for (int rr = 0; rr < RH8_ROWS; rr++) {
			const int y = min(y0 + rr, a.height - 1);
			const gptr_in line = gptr_in_of((unsigned long long) a.in) + (long long) y * a.in_stride;
			for (int g = t; g < groups; g += RH8_NT) {
				const int p = p_al + 4 * g;
				unsigned int w[B], P[B];
				if (p >= 0 && p + 4 <= a.in_width)
					gload_dwords<B>(line, (unsigned int) (p * B), w);
				else {
					// a group over an edge of the image: pixel by pixel, clamped
					unsigned char e[4 * B];
#pragma unroll
					for (int j = 0; j < 4; j++) {
						const int pc = min(max(p + j, 0), a.in_width - 1);
#pragma unroll
						for (int b = 0; b < B; b++)
							e[j * B + b] = gload8(line, (unsigned int) (pc * B + b));
					}
#pragma unroll
					for (int d = 0; d < B; d++)
						w[d] = (unsigned int) e[4 * d] | ((unsigned int) e[4 * d + 1] << 8) | ((unsigned int) e[4 * d + 2] << 16) |
							((unsigned int) e[4 * d + 3] << 24);
				}
				cu8_planar<B>(w, P);
#pragma unroll
				for (int b = 0; b < B; b++)
					lds[(rr * B + b) * a.plane_dw + g] = P[b] ^ 0x80808080u;
			}
		}
		barrier();
		const int y = y0 + r;
		if (y < a.height && x0 < a.out_width) {
			unsigned int O[B];
#pragma unroll
			for (int b = 0; b < B; b++) {
				const unsigned int *plane = lds + (r * B + b) * a.plane_dw + q * (4 * STEP4);
				unsigned int w[4 * WR];
#pragma unroll
				for (int i = 0; i < WR; i++) {
					typedef unsigned int rh8_uint4 __attribute__((ext_vector_type(4)));
					const rh8_uint4 v = *reinterpret_cast<const rh8_uint4 *>(plane + 4 * i);
					w[4 * i] = v.x;
					w[4 * i + 1] = v.y;
					w[4 * i + 2] = v.z;
					w[4 * i + 3] = v.w;
				}
				unsigned int o = 0;
#pragma unroll
				for (int k = 0; k < 4; k++) {
					int hi = 0, lo = 0;
#pragma unroll
					for (int j = 0; j < ND; j++) {
						hi = dot4(w[k * STEP4 + j], a.chi[j], hi);
						lo = dot4(w[k * STEP4 + j], a.clo[j], lo);
					}
					int v = (hi * 128 + lo + a.kconst) >> 12;
					// (shift and clamp kept apart: fused into v_ashr_pk_u8_i32 the compiler takes bits
					// 31:16 of its result for zero, the hardware leaves the register's old contents
					// there -- NOTES 3.1 "toolchain findings"; the host fibers cannot see this)
					opaque(v);
					o |= (unsigned int) min(max(v, 0), 255) << (8 * k);
				}
				O[b] = o;
			}
			unsigned int wout[B];
			cu8_interleave<B>(O, wout);
			const gptr_out dst = gptr_out_of((unsigned long long) a.out) + (long long) y * a.out_stride + (long long) x0 * B;
			if (x0 + 4 <= a.out_width)
				gstore_dwords<B>(dst, wout);
			else {
				// the last, partial quad of a row
				for (int e = 0; e < (a.out_width - x0) * B; e++)
					gstore8(dst + e, (unsigned char) (wout[e >> 2] >> (8 * (e & 3))));
			}
		}
	}
}

} // namespace vh
