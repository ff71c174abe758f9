/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's colour hot path.
 *
 * Each function processes n pixels of 3 bands exactly like the corresponding
 * process_line of libvips 8.19.0 (/root/reference/libvips/colour):
 *   sRGB2scRGB     sRGB2scRGB.c:72-106, tables LabQ2sRGB.c:130-160
 *   scRGB2XYZ      scRGB2XYZ.c:58-82
 *   XYZ2Lab        XYZ2Lab.c:92-138 (D65: include/vips/colour.h:58-60)
 *   Lab2XYZ        Lab2XYZ.c:84-109
 *   XYZ2scRGB      LabQ2sRGB.c:263-283 via XYZ2scRGB.c:72-94
 *   scRGB2sRGB     LabQ2sRGB.c:290-360 via scRGB2sRGB.c:84-132
 *   Lab2LabS       Lab2LabS.c:59-73       LabS2Lab   LabS2Lab.c:55-69
 *   cast           conversion/cast.c:120-330 (no shift)
 * The chains (colourspace.c:223-520) are composed by tests/helpers.py, one array per
 * step, as the reference composes one image per step.
 *
 * Parity status: PINNED by tests/test_oracle_conv_colour.py (golden vectors from the
 * compiled reference + the Lab(50,0,0) -> XYZ known answer of
 * test/test-suite/test_colour.py:53-57 + oracle/_ref directly where present).
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "port.h"

#define QUANT_ELEMENTS (100000)
#define D65_X0 (95.0470)
#define D65_Y0 (100.0)
#define D65_Z0 (108.8827)

static int Y2v_8[256 + 1];
static float v2Y_8[256];
static int Y2v_16[65536 + 1];
static float v2Y_16[65536];
static float cbrt_table[QUANT_ELEMENTS];
static int tables_made = 0;

static void
calcul_tables(int range, int *Y2v, float *v2Y)
{
	for (int i = 0; i < range; i++) {
		float f = (float) i / (range - 1);
		float v;

		if (f <= 0.0031308)
			v = 12.92F * f;
		else
			v = (1.0F + 0.055F) * powf(f, 1.0F / 2.4F) - 0.055F;

		Y2v[i] = rintf((range - 1) * v);
	}
	Y2v[range] = Y2v[range - 1];

	for (int i = 0; i < range; i++) {
		float f = (float) i / (range - 1);

		if (f <= 0.04045)
			v2Y[i] = f / 12.92F;
		else
			v2Y[i] = powf((f + 0.055F) / (1 + 0.055F), 2.4F);
	}
}

static void
make_tables(void)
{
	if (tables_made)
		return;
	calcul_tables(256, Y2v_8, v2Y_8);
	calcul_tables(65536, Y2v_16, v2Y_16);
	for (int i = 0; i < QUANT_ELEMENTS; i++) {
		float Y = (double) i / QUANT_ELEMENTS;

		if (Y < 0.008856)
			cbrt_table[i] = 7.787F * Y + (16.0F / 116.0F);
		else
			cbrt_table[i] = cbrtf(Y);
	}
	tables_made = 1;
}

#define CLIPV(A, V, B) ((V) < (A) ? (A) : ((V) > (B) ? (B) : (V)))
#define VMIN(A, B) ((A) < (B) ? (A) : (B))
#define VMAX(A, B) ((A) > (B) ? (A) : (B))
#define VCLIP(A, V, B) VMAX((A), VMIN((B), (V)))

void
port_sRGB2scRGB_8(const unsigned char *p, int n, float *q)
{
	make_tables();
	for (int i = 0; i < 3 * n; i++)
		q[i] = v2Y_8[p[i]];
}

void
port_sRGB2scRGB_16(const unsigned short *p, int n, float *q)
{
	make_tables();
	for (int i = 0; i < 3 * n; i++)
		q[i] = v2Y_16[p[i]];
}

void
port_scRGB2XYZ(const float *p, int n, float *q)
{
	for (int i = 0; i < n; i++) {
		const float R = p[0] * D65_Y0;
		const float G = p[1] * D65_Y0;
		const float B = p[2] * D65_Y0;

		q[0] = 0.4124F * R + 0.3576F * G + 0.1805F * B;
		q[1] = 0.2126F * R + 0.7152F * G + 0.0722F * B;
		q[2] = 0.0193F * R + 0.1192F * G + 0.9505F * B;
		p += 3;
		q += 3;
	}
}

void
port_XYZ2Lab(const float *p, int n, float *q)
{
	make_tables();
	for (int x = 0; x < n; x++) {
		const float X = p[0], Y = p[1], Z = p[2];
		float nX, nY, nZ, f, cbx, cby, cbz;
		int i;

		nX = QUANT_ELEMENTS * X / D65_X0;
		nY = QUANT_ELEMENTS * Y / D65_Y0;
		nZ = QUANT_ELEMENTS * Z / D65_Z0;

		i = CLIPV(0, (int) nX, QUANT_ELEMENTS - 2);
		f = nX - i;
		cbx = cbrt_table[i] + f * (cbrt_table[i + 1] - cbrt_table[i]);

		i = CLIPV(0, (int) nY, QUANT_ELEMENTS - 2);
		f = nY - i;
		cby = cbrt_table[i] + f * (cbrt_table[i + 1] - cbrt_table[i]);

		i = CLIPV(0, (int) nZ, QUANT_ELEMENTS - 2);
		f = nZ - i;
		cbz = cbrt_table[i] + f * (cbrt_table[i + 1] - cbrt_table[i]);

		q[0] = 116.0F * cby - 16.0F;
		q[1] = 500.0F * (cbx - cby);
		q[2] = 200.0F * (cby - cbz);
		p += 3;
		q += 3;
	}
}

void
port_Lab2XYZ(const float *p, int n, float *q)
{
	for (int x = 0; x < n; x++) {
		const float L = p[0], a = p[1], b = p[2];
		float X, Y, Z;
		double cby, tmp;

		if (L < 8.0) {
			Y = (L * D65_Y0) / 903.3;
			cby = 7.787 * (Y / D65_Y0) + 16.0 / 116.0;
		}
		else {
			cby = (L + 16.0) / 116.0;
			Y = D65_Y0 * cby * cby * cby;
		}

		tmp = a / 500.0 + cby;
		if (tmp < 0.2069)
			X = D65_X0 * (tmp - 0.13793) / 7.787;
		else
			X = D65_X0 * tmp * tmp * tmp;

		tmp = cby - b / 200.0;
		if (tmp < 0.2069)
			Z = D65_Z0 * (tmp - 0.13793) / 7.787;
		else
			Z = D65_Z0 * tmp * tmp * tmp;

		q[0] = X;
		q[1] = Y;
		q[2] = Z;
		p += 3;
		q += 3;
	}
}

void
port_XYZ2scRGB(const float *p, int n, float *q)
{
	for (int i = 0; i < n; i++) {
		float X = p[0], Y = p[1], Z = p[2];

		X /= D65_Y0;
		Y /= D65_Y0;
		Z /= D65_Y0;

		q[0] = 3.240625F * X + -1.537208F * Y + -0.498629F * Z;
		q[1] = -0.968931F * X + 1.875756F * Y + 0.041518F * Z;
		q[2] = 0.055710F * X + -0.204021F * Y + 1.056996F * Z;
		p += 3;
		q += 3;
	}
}

static void
scRGB2sRGB_pixel(int range, const int *lut, float R, float G, float B, int *r, int *g, int *b)
{
	const int maxval = range - 1;
	float Yf, v;
	int Yi;

	if (isnan(R) || isnan(G) || isnan(B)) {
		*r = 0;
		*g = 0;
		*b = 0;
		return;
	}
#define GAMUT(V) \
	{ \
		if ((V) < 0) \
			(V) = 0; \
		else if ((V) > maxval) \
			(V) = maxval; \
	}
	Yf = R * maxval;
	GAMUT(Yf);
	Yi = (int) Yf;
	v = lut[Yi] + (lut[Yi + 1] - lut[Yi]) * (Yf - Yi);
	*r = rintf(v);

	Yf = G * maxval;
	GAMUT(Yf);
	Yi = (int) Yf;
	v = lut[Yi] + (lut[Yi + 1] - lut[Yi]) * (Yf - Yi);
	*g = rintf(v);

	Yf = B * maxval;
	GAMUT(Yf);
	Yi = (int) Yf;
	v = lut[Yi] + (lut[Yi + 1] - lut[Yi]) * (Yf - Yi);
	*b = rintf(v);
#undef GAMUT
}

void
port_scRGB2sRGB_8(const float *p, int n, unsigned char *q)
{
	make_tables();
	for (int i = 0; i < n; i++) {
		int r, g, b;

		scRGB2sRGB_pixel(256, Y2v_8, p[0], p[1], p[2], &r, &g, &b);
		q[0] = r;
		q[1] = g;
		q[2] = b;
		p += 3;
		q += 3;
	}
}

void
port_scRGB2sRGB_16(const float *p, int n, unsigned short *q)
{
	make_tables();
	for (int i = 0; i < n; i++) {
		int r, g, b;

		scRGB2sRGB_pixel(65536, Y2v_16, p[0], p[1], p[2], &r, &g, &b);
		q[0] = r;
		q[1] = g;
		q[2] = b;
		p += 3;
		q += 3;
	}
}

void
port_Lab2LabS(const float *p, int n, short *q)
{
	for (int i = 0; i < n; i++) {
		q[0] = VCLIP(0, p[0] * (32767.0 / 100.0), SHRT_MAX);
		q[1] = VCLIP(SHRT_MIN, p[1] * (32768.0 / 128.0), SHRT_MAX);
		q[2] = VCLIP(SHRT_MIN, p[2] * (32768.0 / 128.0), SHRT_MAX);
		q += 3;
		p += 3;
	}
}

void
port_LabS2Lab(const short *p, int n, float *q)
{
	for (int i = 0; i < n; i++) {
		q[0] = p[0] / (32767.0 / 100.0);
		q[1] = p[1] / (32768.0 / 128.0);
		q[2] = p[2] / (32768.0 / 128.0);
		p += 3;
		q += 3;
	}
}

/* vips_cast without shift, real formats only. */
#define CAST_LOOP(ITYPE, EXPR_OTYPE, BODY) \
	for (size_t x = 0; x < n; x++) { \
		const ITYPE v = ((const ITYPE *) in)[x]; \
		BODY; \
	}

#define TO_INT(ITYPE, OTYPE, TEMP, LO, HI) \
	CAST_LOOP(ITYPE, OTYPE, { \
		TEMP t = (TEMP) v; \
		((OTYPE *) out)[x] = VCLIP((TEMP) (LO), t, (TEMP) (HI)); \
	})
#define F_TO_INT(ITYPE, OTYPE, LO, HI) \
	CAST_LOOP(ITYPE, OTYPE, { \
		((OTYPE *) out)[x] = VCLIP((double) (LO), (double) v, (double) (HI)); \
	})
#define TO_FLOAT(ITYPE, OTYPE) \
	CAST_LOOP(ITYPE, OTYPE, { ((OTYPE *) out)[x] = v; })

#define CAST_FROM_INT(ITYPE) \
	switch (out_format) { \
	case PORT_FORMAT_UCHAR: TO_INT(ITYPE, unsigned char, int, 0, UCHAR_MAX); break; \
	case PORT_FORMAT_CHAR: TO_INT(ITYPE, signed char, int, SCHAR_MIN, SCHAR_MAX); break; \
	case PORT_FORMAT_USHORT: TO_INT(ITYPE, unsigned short, int, 0, USHRT_MAX); break; \
	case PORT_FORMAT_SHORT: TO_INT(ITYPE, short, int, SHRT_MIN, SHRT_MAX); break; \
	case PORT_FORMAT_UINT: TO_INT(ITYPE, unsigned int, int64_t, 0, UINT_MAX); break; \
	case PORT_FORMAT_INT: TO_INT(ITYPE, int, int64_t, INT_MIN, INT_MAX); break; \
	case PORT_FORMAT_FLOAT: TO_FLOAT(ITYPE, float); break; \
	case PORT_FORMAT_DOUBLE: TO_FLOAT(ITYPE, double); break; \
	default: return -1; \
	}
#define CAST_FROM_FLOAT(ITYPE) \
	switch (out_format) { \
	case PORT_FORMAT_UCHAR: F_TO_INT(ITYPE, unsigned char, 0, UCHAR_MAX); break; \
	case PORT_FORMAT_CHAR: F_TO_INT(ITYPE, signed char, SCHAR_MIN, SCHAR_MAX); break; \
	case PORT_FORMAT_USHORT: F_TO_INT(ITYPE, unsigned short, 0, USHRT_MAX); break; \
	case PORT_FORMAT_SHORT: F_TO_INT(ITYPE, short, SHRT_MIN, SHRT_MAX); break; \
	case PORT_FORMAT_UINT: F_TO_INT(ITYPE, unsigned int, 0, UINT_MAX); break; \
	case PORT_FORMAT_INT: F_TO_INT(ITYPE, int, INT_MIN, INT_MAX); break; \
	case PORT_FORMAT_FLOAT: TO_FLOAT(ITYPE, float); break; \
	case PORT_FORMAT_DOUBLE: TO_FLOAT(ITYPE, double); break; \
	default: return -1; \
	}

int
port_cast(const void *in, size_t n, int in_format, int out_format, void *out)
{
	switch (in_format) {
	case PORT_FORMAT_UCHAR: CAST_FROM_INT(unsigned char); break;
	case PORT_FORMAT_CHAR: CAST_FROM_INT(signed char); break;
	case PORT_FORMAT_USHORT: CAST_FROM_INT(unsigned short); break;
	case PORT_FORMAT_SHORT: CAST_FROM_INT(short); break;
	case PORT_FORMAT_UINT: CAST_FROM_INT(unsigned int); break;
	case PORT_FORMAT_INT: CAST_FROM_INT(int); break;
	case PORT_FORMAT_FLOAT: CAST_FROM_FLOAT(float); break;
	case PORT_FORMAT_DOUBLE: CAST_FROM_FLOAT(double); break;
	default: return -1;
	}
	return 0;
}

/* premultiply / unpremultiply: conversion/premultiply.c:78-128,163-176,252-258 and
 * conversion/unpremultiply.c:85-186,222-235,316-323.  The last band is alpha.
 * uchar_fast: the uchar -> uchar fixed-point path; else the output is float.
 * Real, non-double formats.
 */
#define PREMUL(IN) \
	for (size_t x = 0; x < n; x++) { \
		const IN *p = (const IN *) in + x * bands; \
		float *q = (float *) out + x * bands; \
		IN alpha = p[bands - 1]; \
		if (!inverse) { \
			IN clip_alpha = VCLIP(0, alpha, max_alpha); \
			float nalpha = (float) clip_alpha / max_alpha; \
			for (int i = 0; i < bands - 1; i++) \
				q[i] = p[i] * nalpha; \
			q[bands - 1] = alpha; \
		} \
		else { \
			float factor = is_float \
				? (fabs(alpha) < 0.01 ? 0 : max_alpha / alpha) \
				: (alpha == 0 ? 0 : max_alpha / alpha); \
			for (int i = 0; i < bands - 1; i++) \
				q[i] = factor * p[i]; \
			q[bands - 1] = VCLIP(0, alpha, max_alpha); \
		} \
	}

int
port_premultiply(const void *in, size_t n, int bands, int format, double max_alpha, int uchar_fast,
	int inverse, void *out)
{
	if (uchar_fast && format == PORT_FORMAT_UCHAR) {
		int scale[256];

		for (int i = 0; i < 256; i++) {
			double clip = VCLIP(0, i, max_alpha);

			if (inverse)
				scale[i] = clip == 0 ? 0 : 256 * max_alpha / clip;
			else
				scale[i] = 256 * clip / max_alpha;
		}
		for (size_t x = 0; x < n; x++) {
			const unsigned char *p = (const unsigned char *) in + x * bands;
			unsigned char *q = (unsigned char *) out + x * bands;
			unsigned char alpha = p[bands - 1];
			int s = scale[alpha];
			int i;

			for (i = 0; i < bands - 1; i++)
				q[i] = (p[i] * s + 128) >> 8;
			q[i] = alpha;
		}
		return 0;
	}

	const int is_float = format == PORT_FORMAT_FLOAT;
	switch (format) {
	case PORT_FORMAT_UCHAR: PREMUL(unsigned char); break;
	case PORT_FORMAT_CHAR: PREMUL(signed char); break;
	case PORT_FORMAT_USHORT: PREMUL(unsigned short); break;
	case PORT_FORMAT_SHORT: PREMUL(short); break;
	case PORT_FORMAT_UINT: PREMUL(unsigned int); break;
	case PORT_FORMAT_INT: PREMUL(int); break;
	case PORT_FORMAT_FLOAT: PREMUL(float); break;
	default: return -1;
	}
	return 0;
}
