/* TEST INFRASTRUCTURE ONLY -- CPU restatement of vips_affine for a pure scale, as
 * vips_resize calls it for upsizing (resample/resize.c:230-300), with the nearest, bilinear
 * and bicubic interpolators, plus vips_zoom for integral nearest enlargement.
 *
 *   transform   resample/transform.c:40-75 (inverse), :155-168 (forward point), :180-252 (oarea)
 *   build       resample/affine.c:420-620: embed by window_offset + 1 with EXTEND_COPY, idx/idy
 *               displaced by the 1-pixel border, FATSTRIP hint for b == c == 0
 *   generate    resample/affine.c:230-397: per generate call the input coordinate of the
 *               rect's first pixel is computed from scratch and then ACCUMULATED with
 *               `ix += ddx` along the row, so a pixel's coordinate depends on where its tile
 *               starts (FATSTRIP: column 0)
 *   nearest     resample/interpolate.c:336-352
 *   bilinear    resample/interpolate.c:432-484 (fixed point for 8 / 16 bit, double for the rest)
 *   bicubic     resample/bicubic.cpp:482-600 and :620-633 (tables), resample/templates.h:152-290
 *   zoom        conversion/zoom.c (out(x, y) = in(x / xfac, y / yfac))
 */
#include "port.h"

#include <limits.h>
#include <math.h>
#include <string.h>

#define TRANSFORM_SHIFT 6
#define TRANSFORM_SCALE (1 << TRANSFORM_SHIFT)
#define INTERPOLATE_SHIFT 12
#define INTERPOLATE_SCALE (1 << INTERPOLATE_SHIFT)

#define ROUND_INT(R) ((int) ((R) > 0 ? ((R) + 0.5) : ((R) -0.5)))

static int bicubic_ready = 0;
static int bicubic_matrixi[TRANSFORM_SCALE + 1][4];
static double bicubic_matrixf[TRANSFORM_SCALE + 1][4];

/* templates.h:296-320 */
static void
coefficients_catmull(double c[4], const double x)
{
	const double cr1 = 1. - x;
	const double cr2 = -.5 * x;
	const double cr3 = cr1 * cr2;
	const double cone = cr1 * cr3;
	const double cfou = x * cr3;
	const double cr4 = cfou - cone;
	const double ctwo = cr1 - cone + cr4;
	const double cthr = x - cfou - cr4;

	c[0] = cone;
	c[3] = cfou;
	c[1] = ctwo;
	c[2] = cthr;
}

static void
bicubic_init(void)
{
	if (bicubic_ready)
		return;
	for (int x = 0; x < TRANSFORM_SCALE + 1; x++) {
		coefficients_catmull(bicubic_matrixf[x], (float) x / TRANSFORM_SCALE);
		for (int i = 0; i < 4; i++)
			bicubic_matrixi[x][i] = bicubic_matrixf[x][i] * INTERPOLATE_SCALE;
	}
	bicubic_ready = 1;
}

void
port_bicubic_tables(int *matrixi, double *matrixf)
{
	bicubic_init();
	memcpy(matrixi, bicubic_matrixi, sizeof(bicubic_matrixi));
	memcpy(matrixf, bicubic_matrixf, sizeof(bicubic_matrixf));
}

int
port_affine_out_size(int in_size, double scale)
{
	/* transform.c:220-231: the corners 0 and scale * in_size, rounded to nearest */
	const double right = scale * in_size + 0.0 * 0 + 0.0;
	return ROUND_INT(right - 0.0);
}

static int
clampi(int v, int lo, int hi)
{
	return v < lo ? lo : (v > hi ? hi : v);
}

/* the embedded image (affine.c:520-532): original pixel (px, py) sits at (px + off, py + off) */
#define E(TYPE, ex, ey, z) \
	(((const TYPE *) in)[((size_t) clampi((ey) -off, 0, height - 1) * width + \
							 clampi((ex) -off, 0, width - 1)) * \
			bands + \
		(z)])

#define UFR(v) (((v) + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT)

static int
sfr(int v)
{
	const int sign_of_v = 2 * (v >= 0) - 1;
	const int round_by = sign_of_v * (INTERPOLATE_SCALE >> 1);

	return (v + round_by) >> INTERPOLATE_SHIFT;
}

#define CLIP(lo, v, hi) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))

/* bicubic_unsigned_int_tab / bicubic_signed_int_tab */
#define BICUBIC_INT(TYPE, ROUND, LO, HI) \
	for (int z = 0; z < bands; z++) { \
		int r[4]; \
		for (int j = 0; j < 4; j++) \
			r[j] = ROUND(cxi[0] * E(TYPE, ix - 1, iy - 1 + j, z) + cxi[1] * E(TYPE, ix, iy - 1 + j, z) + \
				cxi[2] * E(TYPE, ix + 1, iy - 1 + j, z) + cxi[3] * E(TYPE, ix + 2, iy - 1 + j, z)); \
		int v = ROUND(cyi[0] * r[0] + cyi[1] * r[1] + cyi[2] * r[2] + cyi[3] * r[3]); \
		v = CLIP(LO, v, HI); \
		((TYPE *) q)[z] = v; \
	}

/* bicubic_unsigned_int32_tab / bicubic_signed_int32_tab: double intermediate, clip, C conversion */
#define BICUBIC_DBL(TYPE, LO, HI) \
	for (int z = 0; z < bands; z++) { \
		double r[4]; \
		for (int j = 0; j < 4; j++) \
			r[j] = cxf[0] * (double) E(TYPE, ix - 1, iy - 1 + j, z) + cxf[1] * (double) E(TYPE, ix, iy - 1 + j, z) + \
				cxf[2] * (double) E(TYPE, ix + 1, iy - 1 + j, z) + cxf[3] * (double) E(TYPE, ix + 2, iy - 1 + j, z); \
		double v = cyf[0] * r[0] + cyf[1] * r[1] + cyf[2] * r[2] + cyf[3] * r[3]; \
		v = CLIP(LO, v, HI); \
		((TYPE *) q)[z] = v; \
	}

/* BILINEAR_INT, interpolate.c:432-456 */
#define BILINEAR_INT(TYPE) \
	{ \
		const int X = (x - ix) * INTERPOLATE_SCALE; \
		const int Y = (y - iy) * INTERPOLATE_SCALE; \
		const int Yd = INTERPOLATE_SCALE - Y; \
		const int c4 = (Y * X) >> INTERPOLATE_SHIFT; \
		const int c2 = (Yd * X) >> INTERPOLATE_SHIFT; \
		const int c3 = Y - c4; \
		const int c1 = Yd - c2; \
		for (int z = 0; z < bands; z++) \
			((TYPE *) q)[z] = (c1 * E(TYPE, ix, iy, z) + c2 * E(TYPE, ix + 1, iy, z) + \
								  c3 * E(TYPE, ix, iy + 1, z) + c4 * E(TYPE, ix + 1, iy + 1, z) + \
								  (1 << INTERPOLATE_SHIFT) / 2) >> \
				INTERPOLATE_SHIFT; \
	}

/* BILINEAR_FLOAT, interpolate.c:462-484 */
#define BILINEAR_FLOAT(TYPE) \
	{ \
		const double X = x - ix; \
		const double Y = y - iy; \
		const double Yd = 1.0f - Y; \
		const double c4 = Y * X; \
		const double c2 = Yd * X; \
		const double c3 = Y - c4; \
		const double c1 = Yd - c2; \
		for (int z = 0; z < bands; z++) \
			((TYPE *) q)[z] = c1 * E(TYPE, ix, iy, z) + c2 * E(TYPE, ix + 1, iy, z) + \
				c3 * E(TYPE, ix, iy + 1, z) + c4 * E(TYPE, ix + 1, iy + 1, z); \
	}

static int
format_size(int format)
{
	switch (format) {
	case PORT_FORMAT_UCHAR:
	case PORT_FORMAT_CHAR:
		return 1;
	case PORT_FORMAT_USHORT:
	case PORT_FORMAT_SHORT:
		return 2;
	case PORT_FORMAT_UINT:
	case PORT_FORMAT_INT:
	case PORT_FORMAT_FLOAT:
		return 4;
	default:
		return 0;
	}
}

/* interp: 0 nearest, 1 bilinear, 2 bicubic.  tile_width: width of the generate rects (0 = whole
 * rows, what the FATSTRIP hint gives a sink).  out is out_width x out_height from
 * port_affine_out_size.  Returns 0, or -1 for formats outside the port (double, complex).
 */
int
port_affine_scale(const void *in, int width, int height, int bands, int format,
	double hscale, double vscale, double idx, double idy, int interp, int tile_width, void *out)
{
	const int es = format_size(format);
	const int window_size = interp == 0 ? 1 : (interp == 1 ? 2 : 4);
	/* interpolate.c:150-167: half the window - 1, never negative */
	const int window_offset = window_size / 2 - 1 > 0 ? window_size / 2 - 1 : 0;
	const int off = window_offset + 1;
	const int out_width = port_affine_out_size(width, hscale);
	const int out_height = port_affine_out_size(height, vscale);

	/* transform.c:40-75 */
	const double a = hscale, b = 0.0, c = 0.0, d = vscale;
	const double det = a * d - b * c;
	const double tmp = 1.0 / det;
	const double ia = tmp * d;
	const double ib = -tmp * b;
	const double ic = -tmp * c;
	const double id = tmp * a;
	const double odx = 0.0, ody = 0.0;
	/* affine.c:538-539 */
	const double tidx = idx - 1;
	const double tidy = idy - 1;

	if (es == 0)
		return -1;
	bicubic_init();
	if (tile_width <= 0)
		tile_width = out_width;

	/* affine.c:330-336: clip rectangle in the embedded image's coordinates */
	const int ile = 0 + window_offset;
	const int ito = 0 + window_offset;
	const int iri = ile + width;
	const int ibo = ito + height;

	for (int le = 0; le < out_width; le += tile_width) {
		const int ri = le + tile_width < out_width ? le + tile_width : out_width;

		for (int yy = 0; yy < out_height; yy++) {
			const double ddx = ia;
			const double ddy = ic;
			const double ox = le + 0 - odx;
			const double oy = yy + 0 - ody;
			double x, y;
			unsigned char *q = (unsigned char *) out + ((size_t) yy * out_width + le) * bands * es;

			x = ia * ox + ib * oy;
			y = ic * ox + id * oy;
			x -= tidx;
			y -= tidy;
			x += window_offset;
			y += window_offset;

			for (int xx = le; xx < ri; xx++) {
				const int fx = floor(x);
				const int fy = floor(y);

				if (fx >= ile && fx <= iri && fy >= ito && fy <= ibo) {
					const int ix = (int) x;
					const int iy = (int) y;

					if (interp == 0) {
						for (int z = 0; z < bands; z++)
							memcpy(q + z * es,
								(const unsigned char *) in +
									(((size_t) clampi(iy - off, 0, height - 1) * width +
										 clampi(ix - off, 0, width - 1)) *
											bands +
										z) *
										es,
								es);
					}
					else if (interp == 1) {
						switch (format) {
						case PORT_FORMAT_UCHAR: BILINEAR_INT(unsigned char); break;
						case PORT_FORMAT_CHAR: BILINEAR_INT(char); break;
						case PORT_FORMAT_USHORT: BILINEAR_INT(unsigned short); break;
						case PORT_FORMAT_SHORT: BILINEAR_INT(short); break;
						case PORT_FORMAT_UINT: BILINEAR_FLOAT(unsigned int); break;
						case PORT_FORMAT_INT: BILINEAR_FLOAT(int); break;
						default: BILINEAR_FLOAT(float); break;
						}
					}
					else {
						/* bicubic.cpp:488-502 */
						const int sx = x * TRANSFORM_SCALE * 2;
						const int sy = y * TRANSFORM_SCALE * 2;
						const int six = sx & (TRANSFORM_SCALE * 2 - 1);
						const int siy = sy & (TRANSFORM_SCALE * 2 - 1);
						const int tx = (six + 1) >> 1;
						const int ty = (siy + 1) >> 1;
						const int *cxi = bicubic_matrixi[tx];
						const int *cyi = bicubic_matrixi[ty];
						const double *cxf = bicubic_matrixf[tx];
						const double *cyf = bicubic_matrixf[ty];

						switch (format) {
						case PORT_FORMAT_UCHAR: BICUBIC_INT(unsigned char, UFR, 0, UCHAR_MAX); break;
						case PORT_FORMAT_CHAR: BICUBIC_INT(signed char, sfr, SCHAR_MIN, SCHAR_MAX); break;
						case PORT_FORMAT_USHORT: BICUBIC_DBL(unsigned short, 0, USHRT_MAX); break;
						case PORT_FORMAT_SHORT: BICUBIC_DBL(short, SHRT_MIN, SHRT_MAX); break;
						case PORT_FORMAT_UINT: BICUBIC_DBL(unsigned int, 0, INT_MAX); break;
						case PORT_FORMAT_INT: BICUBIC_DBL(int, INT_MIN, INT_MAX); break;
						default:
							/* bicubic_float_tab<float>: each cubic_float<float> returns a float */
							for (int z = 0; z < bands; z++) {
								float r[4];
								for (int j = 0; j < 4; j++)
									r[j] = cxf[0] * E(float, ix - 1, iy - 1 + j, z) +
										cxf[1] * E(float, ix, iy - 1 + j, z) +
										cxf[2] * E(float, ix + 1, iy - 1 + j, z) +
										cxf[3] * E(float, ix + 2, iy - 1 + j, z);
								((float *) q)[z] = cyf[0] * r[0] + cyf[1] * r[1] + cyf[2] * r[2] + cyf[3] * r[3];
							}
							break;
						}
					}
				}
				else
					memset(q, 0, (size_t) bands * es);

				x += ddx;
				y += ddy;
				q += bands * es;
			}
		}
	}

	return 0;
}

/* vips_zoom (conversion/zoom.c): integral pixel replication */
int
port_zoom(const void *in, int width, int height, int bands, int format, int xfac, int yfac, void *out)
{
	const int es = format_size(format);
	const size_t ps = (size_t) bands * es;

	if (es == 0 || xfac < 1 || yfac < 1)
		return -1;
	for (int y = 0; y < height * yfac; y++)
		for (int x = 0; x < width * xfac; x++)
			memcpy((unsigned char *) out + ((size_t) y * width * xfac + x) * ps,
				(const unsigned char *) in + ((size_t) (y / yfac) * width + x / xfac) * ps, ps);
	return 0;
}
