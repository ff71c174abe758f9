/* TEST INFRASTRUCTURE ONLY -- declarations of the CPU restatement (oracle/port).
 * See the header of each port_*.c for the reference lines it follows.
 */
#ifndef ORACLE_PORT_H
#define ORACLE_PORT_H

#include <stddef.h>

/* VipsBandFormat values (include/vips/image.h:120-133) */
enum {
	PORT_FORMAT_UCHAR = 0,
	PORT_FORMAT_CHAR = 1,
	PORT_FORMAT_USHORT = 2,
	PORT_FORMAT_SHORT = 3,
	PORT_FORMAT_UINT = 4,
	PORT_FORMAT_INT = 5,
	PORT_FORMAT_FLOAT = 6,
	PORT_FORMAT_COMPLEX = 7,
	PORT_FORMAT_DOUBLE = 8,
	PORT_FORMAT_DPCOMPLEX = 9
};

/* VipsKernel values (include/vips/resample.h:41-51) */
enum {
	PORT_KERNEL_NEAREST = 0,
	PORT_KERNEL_LINEAR,
	PORT_KERNEL_CUBIC,
	PORT_KERNEL_MITCHELL,
	PORT_KERNEL_LANCZOS2,
	PORT_KERNEL_LANCZOS3,
	PORT_KERNEL_MKS2013,
	PORT_KERNEL_MKS2021
};

/* resample */
int port_reduce_get_points(int kernel, double shrink);
void port_reduce_make_mask(double *c, int kernel, int n_points, double shrink, double x);
int port_reduceh(const void *in, int width, int height, int bands, int format,
	double hshrink, int kernel, int out_width, double extra_pixels, void *out);
int port_reducev(const void *in, int width, int height, int bands, int format,
	double vshrink, int kernel, int out_height, double extra_pixels, int tile, void *out);
int port_shrink_out_size(int in_size, int shrink, int ceil_mode);
int port_shrinkh(const void *in, int width, int height, int bands, int format, int hshrink,
	int ceil_mode, void *out);
int port_shrinkv(const void *in, int width, int height, int bands, int format, int vshrink,
	int ceil_mode, void *out);

/* upsizing (port_affine.c): vips_affine for a pure scale + interpolators, vips_zoom */
int port_affine_out_size(int in_size, double scale);
int port_affine_scale(const void *in, int width, int height, int bands, int format,
	double hscale, double vscale, double idx, double idy, int interp, int tile_width, void *out);
int port_zoom(const void *in, int width, int height, int bands, int format, int xfac, int yfac, void *out);
void port_bicubic_tables(int *matrixi, double *matrixf);

/* convolution (port_conv.c) */
int port_convi(const void *in, int width, int height, int bands, int format,
	const double *mask, int mw, int mh, double scale, double offset, void *out);
int port_convf(const void *in, int width, int height, int bands, int format,
	const double *mask, int mw, int mh, double scale, double offset, void *out);
/* the Highway variant of convi on uchar (parity unpinned: see port_conv.c) */
int port_convi_hwy_intize(const double *mask, int n_point, double scale, short *mant, int *pos,
	int *nnz_out, int *exp_out);
int port_convi_hwy(const unsigned char *in, int width, int height, int bands, const double *mask, int mw,
	int mh, double scale, double offset, unsigned char *out);
int port_gaussmat(double sigma, double min_ampl, int separable, int integer, double *mask,
	double *scale);
void port_sharpen_lut(double x1, double y2, double y3, double m1, double m2, int *lut);
void port_sharpen_apply(const short *in, const short *blur, int n_pixels, int bands,
	const int *lut, short *out);

/* approximate convolution (port_conva.c): vips_conva / vips_convasep */
int port_conva_decompose(const double *mask, int mw, int mh, double scale, double offset, int layers,
	int cluster, int *info, int *lines, int max_ints);
int port_conva(const void *in, int width, int height, int bands, int format, const double *mask,
	int mw, int mh, double scale, double offset, int layers, int cluster, void *out);
int port_convasep_decompose(const double *mask, int n, double scale, double offset, int layers,
	int *info, int *lines, int max_ints);
int port_convasep(const void *in, int width, int height, int bands, int format, const double *mask,
	int n, double scale, double offset, int layers, void *out);

/* colour (port_colour.c): n pixels of 3 bands */
void port_sRGB2scRGB_8(const unsigned char *p, int n, float *q);
void port_sRGB2scRGB_16(const unsigned short *p, int n, float *q);
void port_scRGB2XYZ(const float *p, int n, float *q);
void port_XYZ2Lab(const float *p, int n, float *q);
void port_Lab2XYZ(const float *p, int n, float *q);
void port_XYZ2scRGB(const float *p, int n, float *q);
void port_scRGB2sRGB_8(const float *p, int n, unsigned char *q);
void port_scRGB2sRGB_16(const float *p, int n, unsigned short *q);
void port_Lab2LabS(const float *p, int n, short *q);
void port_LabS2Lab(const short *p, int n, float *q);
int port_cast(const void *in, size_t n, int in_format, int out_format, void *out);
int port_premultiply(const void *in, size_t n, int bands, int format, double max_alpha,
	int uchar_fast, int inverse, void *out);

#endif
