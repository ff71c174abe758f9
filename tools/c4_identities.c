/* The arithmetic identities the fused resize + sharpen kernel (libvips_amd/csrc/resize_sharpen_body.h)
 * relies on, checked exhaustively on the host (gcc -O2 -ffp-contract=off -fopenmp -lm):
 *
 *  1. (float) ((double) X / 100.0) == the correctly rounded float quotient X / 100, computed as
 *     q0 = X * RN(1/100); e = fmaf(-100, q0, X); q = fmaf(e, RN(1/100), q0)   -- every finite float
 *     of magnitude 2^-100 .. 2^100 (LabQ2sRGB.c:263-283 divides XYZ by D65_Y0 = 100.0 in double)
 *  2. (double) A / 128000.0 == (double) ((float) A / 256.0f) / 500.0 and the same for B / 51200.0,
 *     and the Markstein form (q0 = a r, e = fma(-y, q0, a), q = fma(e, r, q0)) gives exactly that
 *     quotient -- every short A (LabS2Lab.c:55-69 then Lab2XYZ.c:84-109)
 *  3. Yf - (float) (int) Yf == Yf - floorf(Yf) for Yf in [0, 255] (v_fract_f32 in scRGB2sRGB)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main(void)
{
	long bad1 = 0, n1 = 0;
	const float r = 1.0f / 100.0f;
#pragma omp parallel for reduction(+ : bad1, n1) schedule(static)
	for (uint32_t e = 27; e < 227; e++) {
		for (uint32_t m = 0; m < (1u << 23); m++) {
			for (uint32_t s = 0; s < 2; s++) {
				const float x = as_float((s << 31) | (e << 23) | m);
				const float want = (float) ((double) x / 100.0);
				const float q0 = x * r;
				const float err = fmaf(-100.0f, q0, x);
				const float got = fmaf(err, r, q0);
				n1++;
				if (memcmp(&want, &got, 4))
					bad1++;
			}
		}
	}
	/* zero and the small denormal-free range the kernel meets are inside; report */
	printf("1. float quotient X / 100: %ld values, %ld differ\n", n1, bad1);

	long bad2 = 0;
	const double ya = 128000.0, yb = 51200.0, ra = 1.0 / 128000.0, rb = 1.0 / 51200.0;
	for (int A = -32768; A <= 32767; A++) {
		const double wa = (double) ((float) A * 0.00390625f) / 500.0;
		const double wb = (double) ((float) A * 0.00390625f) / 200.0;
		double q0 = (double) A * ra;
		double ga = fma(fma(-ya, q0, (double) A), ra, q0);
		q0 = (double) A * rb;
		double gb = fma(fma(-yb, q0, (double) A), rb, q0);
		if (wa != (double) A / ya || wb != (double) A / yb || memcmp(&wa, &ga, 8) || memcmp(&wb, &gb, 8))
			bad2++;
	}
	printf("2. A / 128000 and B / 51200: 65536 values, %ld differ\n", bad2);

	long bad3 = 0, n3 = 0;
	for (uint32_t u = 0; u <= 0x437f0000u; u++) { /* 0 .. 255.0f */
		const float y = as_float(u);
		const float a = y - (float) (int) y, b = y - floorf(y);
		n3++;
		if (memcmp(&a, &b, 4))
			bad3++;
	}
	printf("3. fraction of Yf in [0, 255]: %ld values, %ld differ\n", n3, bad3);
	return bad1 || bad2 || bad3;
}
