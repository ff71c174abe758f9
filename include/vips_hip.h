/* vips_hip.h -- C ABI of libvipship.so: the MI355X (gfx950) implementation of
 * libvips' per-tile pixel pipeline (resample/, convolution/, colour/ hot loops).
 *
 * This header is the drop-in boundary.  Everything is plain C: pointers, ints,
 * doubles and two POD structs.  No HIP, torch or glib type appears in a
 * signature, so the libvips side (a C module registering `*_hip` VipsOperation
 * classes, see host/ and INTEGRATION.md) and any FFI (ctypes, cgo, JNI...) can
 * bind it directly.
 *
 * Three layers, mirroring the reference (all paths relative to the reference
 * tree, libvips 8.19.0):
 *
 *   1. runtime      device / stream / memory / error buffer
 *                   (error convention of iofuncs/error.c: 0 or -1 + message)
 *   2. region ops   "generate" replacements: fill out->valid from an input
 *                   region, exactly the contract of VipsGenerateFn
 *                   (include/vips/image.h:151-154, iofuncs/region.c:1600-1624)
 *   3. image ops    whole-image operations on device-resident images, the
 *                   analogue of the vips_reduce()/vips_conv()/... C wrappers;
 *                   they do what each class's build() does on the host (sizes,
 *                   tables, embed offsets) and then run the region ops once
 *                   over the whole output.
 *
 * Pixel layout everywhere: interleaved bands, row-major, `stride` bytes per line
 * (include/vips/image.h:382-393, include/vips/region.h:198-236).
 */
#ifndef VIPS_HIP_H
#define VIPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIPS_HIP_API __attribute__((visibility("default")))

/* Same values as VipsBandFormat, include/vips/image.h:120-133. */
typedef enum {
	VIPS_HIP_FORMAT_UCHAR = 0,
	VIPS_HIP_FORMAT_CHAR = 1,
	VIPS_HIP_FORMAT_USHORT = 2,
	VIPS_HIP_FORMAT_SHORT = 3,
	VIPS_HIP_FORMAT_UINT = 4,
	VIPS_HIP_FORMAT_INT = 5,
	VIPS_HIP_FORMAT_FLOAT = 6,
	VIPS_HIP_FORMAT_COMPLEX = 7,
	VIPS_HIP_FORMAT_DOUBLE = 8,
	VIPS_HIP_FORMAT_DPCOMPLEX = 9
} VipsHipFormat;

/* Same values as VipsKernel, include/vips/resample.h:41-51. */
typedef enum {
	VIPS_HIP_KERNEL_NEAREST = 0,
	VIPS_HIP_KERNEL_LINEAR = 1,
	VIPS_HIP_KERNEL_CUBIC = 2,
	VIPS_HIP_KERNEL_MITCHELL = 3,
	VIPS_HIP_KERNEL_LANCZOS2 = 4,
	VIPS_HIP_KERNEL_LANCZOS3 = 5,
	VIPS_HIP_KERNEL_MKS2013 = 6,
	VIPS_HIP_KERNEL_MKS2021 = 7
} VipsHipKernel;

/* The interpolators vips_resize picks for upsizing (resample/resize.c:118-133). */
typedef enum {
	VIPS_HIP_INTERPOLATE_NEAREST = 0,
	VIPS_HIP_INTERPOLATE_BILINEAR = 1,
	VIPS_HIP_INTERPOLATE_BICUBIC = 2
} VipsHipInterpolate;

/* Same values as VipsPrecision, include/vips/basic.h:106-110. */
typedef enum {
	VIPS_HIP_PRECISION_INTEGER = 0,
	VIPS_HIP_PRECISION_FLOAT = 1,
	VIPS_HIP_PRECISION_APPROXIMATE = 2
} VipsHipPrecision;

/* The subset of VipsInterpretation (include/vips/image.h:94-118) the colour
 * path routes between; same values.
 */
typedef enum {
	VIPS_HIP_INTERPRETATION_MULTIBAND = 0,
	VIPS_HIP_INTERPRETATION_B_W = 1,
	VIPS_HIP_INTERPRETATION_XYZ = 12,
	VIPS_HIP_INTERPRETATION_LAB = 13,
	VIPS_HIP_INTERPRETATION_LABS = 21,
	VIPS_HIP_INTERPRETATION_sRGB = 22,
	VIPS_HIP_INTERPRETATION_RGB16 = 25,
	VIPS_HIP_INTERPRETATION_GREY16 = 26,
	VIPS_HIP_INTERPRETATION_scRGB = 28
} VipsHipInterpretation;

/* ------------------------------------------------------------------ runtime */

/* Bind the CALLING THREAD to a device (hipSetDevice is per thread) and create the library's
 * per-device state on it: the HBM pool, the plan caches and the thread's stream are per device,
 * so one process can drive every GPU of a node from its threads -- the shape of libvips' own
 * parallelism, a worker pool inside one process (iofuncs/threadpool.c:625) -- as well as the
 * one-process-per-GPU shape of the multi-rank programs.  Safe to call repeatedly; calling it
 * with another device re-binds the thread (its external stream, if any, is dropped).
 *
 * A thread that never calls it is bound on first use: to the next entry of $VIPS_HIP_DEVICES (a
 * comma list, e.g. "0,1,2,3,4,5,6,7", dealt round-robin over such threads -- how the libvips
 * module spreads a worker pool over the GPUs), else to $VIPS_HIP_DEVICE, else to the device of
 * the process's first vips_hip_init(), else to device 0.
 *
 * Library-made images remember their device and every image-level operation runs on the device
 * its input lives on (the calling thread is re-bound if need be); plan handles
 * (vips_hip_reduce_new() ...) belong to the device of the thread that made them and fail loudly
 * on another.  Fails (-1) when no gfx950 device is visible: there is NO CPU fallback anywhere in
 * this library.
 */
VIPS_HIP_API int vips_hip_init(int device);
/* Finish and drop the calling thread's streams and cached blocks; the thread may bind again. */
VIPS_HIP_API void vips_hip_shutdown(void);
VIPS_HIP_API int vips_hip_device_count(void);
/* The device the calling thread is bound to, -1 when it has not been bound yet. */
VIPS_HIP_API int vips_hip_current_device(void);
/* The devices work is spread over: $VIPS_HIP_DEVICES in order (entries may repeat), else the
 * calling thread's device alone.  Fills devices[0 .. max) and returns how many there are,
 * -1 on a malformed list.
 */
VIPS_HIP_API int vips_hip_devices(int *devices, int max);

/* Thread-local error log, same shape as vips_error_buffer()/vips_error_clear()
 * (iofuncs/error.c): messages are "domain: text\n".
 */
VIPS_HIP_API const char *vips_hip_error_buffer(void);
VIPS_HIP_API void vips_hip_error_clear(void);

/* The stream all following calls of this thread are enqueued on.  NULL selects
 * the library's own per-thread stream (one per libvips worker thread, created in
 * start_fn and released in stop_fn by the module).  The handle is a hipStream_t
 * passed as void*, e.g. torch.cuda.current_stream().cuda_stream.
 */
VIPS_HIP_API int vips_hip_set_stream(void *stream);
VIPS_HIP_API void *vips_hip_get_stream(void);
VIPS_HIP_API int vips_hip_synchronize(void);
/* vips_vector_set_enabled / vips_vector_isenabled (iofuncs/vector.cpp:98-113).  A Highway-built
 * libvips computes convi on uchar images with 8-bit mantissas and a shared exponent
 * (convolution/convi.c:925-1120, convi_hwy.cpp) whenever vectors are enabled and the mask fits;
 * its other vector paths equal the C paths bit for bit.  Enabling this makes vips_hip_conv /
 * convsep / gaussblur with precision INTEGER on uchar compute exactly that, so that this
 * library can stand in for such a build.  Default: disabled (the C path's arithmetic).
 */
VIPS_HIP_API void vips_hip_vector_set_enabled(int enabled);
VIPS_HIP_API int vips_hip_vector_isenabled(void);

/* Float arithmetic mode.  The reference is baseline x86-64 C: a double sum is a chain of
 * separately rounded multiplies and adds (convolution/convf.c:163-181).  By default the large
 * float convolutions on integer images (masks 8 or more wide: BASELINE config 5) accumulate
 * with fused multiply-adds instead -- twice the FP64 rate, and the float result differs from the
 * reference's by at most 1 ULP (the tolerance BASELINE.json's north_star grants float paths;
 * measured: a handful of pixels per million).  Enabled (or $VIPS_HIP_EXACT_FLOAT=1 at first use)
 * every float path reproduces the reference bit for bit.  Integer paths are always exact.
 */
VIPS_HIP_API void vips_hip_set_exact_float(int enabled);
VIPS_HIP_API int vips_hip_get_exact_float(void);

/* Device memory comes from a size-bucketed caching pool (hipMalloc is far too
 * slow to sit in a per-tile path); *_host is pinned staging memory.
 */
VIPS_HIP_API void *vips_hip_malloc(size_t size);
VIPS_HIP_API void vips_hip_free(void *ptr);
VIPS_HIP_API void *vips_hip_malloc_host(size_t size);
VIPS_HIP_API void vips_hip_free_host(void *ptr);
VIPS_HIP_API int vips_hip_memcpy_h2d(void *dst, const void *src, size_t size);
VIPS_HIP_API int vips_hip_memcpy_d2h(void *dst, const void *src, size_t size);
VIPS_HIP_API int vips_hip_memcpy_d2d(void *dst, const void *src, size_t size);
/* The same copies QUEUED on the calling thread's current stream (pinned host memory --
 * vips_hip_malloc_host() -- for them to overlap anything): how the module's strip loop keeps
 * the upload of strip k + 1, the kernels of strip k and the download of strip k - 1 in flight
 * together, the way sinkdisc.c:177-220 writes one buffer behind the one being filled.
 */
VIPS_HIP_API int vips_hip_memcpy_h2d_async(void *dst, const void *src, size_t size);
VIPS_HIP_API int vips_hip_memcpy_d2h_async(void *dst, const void *src, size_t size);
/* A stream of the calling thread's device, for vips_hip_set_stream(); freed after a
 * synchronize. */
VIPS_HIP_API void *vips_hip_stream_new(void);
VIPS_HIP_API void vips_hip_stream_free(void *stream);
VIPS_HIP_API int vips_hip_memcpy2d_h2d(void *dst, size_t dpitch,
	const void *src, size_t spitch, size_t width_bytes, size_t height);
VIPS_HIP_API int vips_hip_memcpy2d_d2h(void *dst, size_t dpitch,
	const void *src, size_t spitch, size_t width_bytes, size_t height);
VIPS_HIP_API size_t vips_hip_pool_bytes(void);
VIPS_HIP_API void vips_hip_pool_trim(void);

/* HIP events on the current stream, for timing kernels where they run. */
VIPS_HIP_API void *vips_hip_event_new(void);
VIPS_HIP_API void vips_hip_event_free(void *event);
VIPS_HIP_API int vips_hip_event_record(void *event);
VIPS_HIP_API double vips_hip_event_elapsed_ms(void *start, void *stop); /* syncs on stop */
VIPS_HIP_API int vips_hip_event_synchronize(void *event); /* wait on the host for a recorded event */
VIPS_HIP_API int vips_hip_stream_wait_event(void *event); /* the current stream waits for it */

/* Per-kernel timing (the VIPS_GATE_START/STOP analogue, include/vips/gate.h):
 * when enabled every kernel launch is bracketed by events on its stream.
 */
VIPS_HIP_API void vips_hip_gate_enable(int enable);
VIPS_HIP_API void vips_hip_gate_reset(void);
/* Returns the number of launches of kernels whose gate name starts with @name
 * and their total duration in ms (synchronises the device).
 */
VIPS_HIP_API int vips_hip_gate_query(const char *name, double *total_ms);
/* Write "name launches total_ms\n" for every gate name seen into @buf (at most
 * @size bytes, NUL-terminated); returns the number of distinct names.
 */
VIPS_HIP_API int vips_hip_gate_report(char *buf, int size);

/* ---------------------------------------------------------------- regions */

/* A window onto a device-resident image: the VipsRegion of this library
 * (include/vips/region.h:96-131).  `data` points at pixel (left, top) of the
 * full image; `stride` is VIPS_REGION_LSKIP.  im_width/im_height are the
 * size of the whole image the window belongs to: coordinates that an operation
 * computes outside [0, im_width) x [0, im_height) are clamped to the edge, which
 * is what the vips_embed(VIPS_EXTEND_COPY) each reference build() inserts does
 * (conversion/embed.c:226-341).  After clamping they must fall inside the window.
 */
typedef struct {
	void *data;
	int left, top, width, height; /* valid: the window */
	int im_width, im_height;      /* the whole image */
	int bands;
	int format; /* VipsHipFormat */
	size_t stride;
} VipsHipRegion;

/* ------------------------------------------------ resample: reduceh/reducev */

/* Host-side state of one reduceh/reducev operation: what vips_reduceh_build()
 * (resample/reduceh.cpp:396-565) / vips_reducev_build() (reducev.cpp:859-1075)
 * compute once per call -- output size, n_point, h/v offset, the 65 x n_point
 * coefficient tables matrixf (double) and matrixs (short, x4096, truncated) --
 * kept resident on the device.
 */
typedef struct _VipsHipReduce VipsHipReduce;

/* @in_size: Xsize (horizontal) or Ysize (vertical) of the input image.
 * @shrink:  the *residual* shrink (after any integer pre-shrink the caller did).
 * @extra_pixels: "how many pixels we are inventing", already divided by the
 *   integer pre-shrink (reduceh.cpp:431,459); pass NAN to have it derived as
 *   out_size * shrink - in_size.
 * Errors (message as the reference's): "reduce factor should be >= 1.0",
 * "reduce factor too large", "image has shrunk to nothing".
 */
VIPS_HIP_API VipsHipReduce *vips_hip_reduce_new(int kernel, double shrink,
	int in_size, int out_size, double extra_pixels);
VIPS_HIP_API void vips_hip_reduce_free(VipsHipReduce *reduce);
VIPS_HIP_API int vips_hip_reduce_get_n_point(const VipsHipReduce *reduce);
VIPS_HIP_API int vips_hip_reduce_get_out_size(const VipsHipReduce *reduce);
VIPS_HIP_API double vips_hip_reduce_get_offset(const VipsHipReduce *reduce);
/* Copy out row @phase (0..64) of matrixs / matrixf, for tests. */
VIPS_HIP_API int vips_hip_reduce_get_matrixs(const VipsHipReduce *reduce, int phase, short *out);
VIPS_HIP_API int vips_hip_reduce_get_matrixf(const VipsHipReduce *reduce, int phase, double *out);

/* vips_reduce_get_points(), resample/reduceh.cpp:113-141. */
VIPS_HIP_API int vips_hip_reduce_get_points(int kernel, double shrink);

/* The input rectangle a generate needs for output rect (left, top, width,
 * height): reduceh.cpp:237-240 / reducev.cpp:539-542, already translated from
 * embedded to un-embedded input coordinates and clipped to the input image.
 */
VIPS_HIP_API void vips_hip_reduceh_need(const VipsHipReduce *reduce,
	int left, int width, int *in_left, int *in_width);
VIPS_HIP_API void vips_hip_reducev_need(const VipsHipReduce *reduce,
	int top, int height, int *in_top, int *in_height);

/* Fill out->valid.  Bit-exact replacements for vips_reduceh_gen
 * (reduceh.cpp:216-335) and vips_reducev_gen (reducev.cpp:517-619): the double
 * position accumulator is seeded at out->left / out->top exactly as the
 * reference seeds it per generate call.
 */
VIPS_HIP_API int vips_hip_reduceh_gen(const VipsHipReduce *reduce,
	const VipsHipRegion *in, const VipsHipRegion *out);
VIPS_HIP_API int vips_hip_reducev_gen(const VipsHipReduce *reduce,
	const VipsHipRegion *in, const VipsHipRegion *out);
/* As above, but the accumulator is re-seeded every @tile rows (columns), which
 * reproduces what the reference computes when its sink walks the output in
 * @tile-high strips (thread.c:301-325: 16 for FATSTRIP images).  tile <= 0
 * means one seed for the whole rect.
 */
VIPS_HIP_API int vips_hip_reduceh_gen_tiled(const VipsHipReduce *reduce,
	const VipsHipRegion *in, const VipsHipRegion *out, int tile);
VIPS_HIP_API int vips_hip_reducev_gen_tiled(const VipsHipReduce *reduce,
	const VipsHipRegion *in, const VipsHipRegion *out, int tile);

/* Fused reducev -> reduceh for uchar images (the vips_reduce() hot path,
 * resample/reduce.c:98-121): the vertically reduced scanlines never leave the
 * CU.  Same results as running the two gens back to back.
 */
VIPS_HIP_API int vips_hip_reduce_gen(const VipsHipReduce *reducev,
	const VipsHipReduce *reduceh,
	const VipsHipRegion *in, const VipsHipRegion *out);
/* The vertical accumulator re-seeded every @tile output rows, as above.
 * Both return 0 on success, -1 on error, and 1 when the geometry (format,
 * bands, shrink) is outside what the fused kernel covers: nothing was written
 * and the caller runs vips_hip_reducev_gen + vips_hip_reduceh_gen instead.
 */
VIPS_HIP_API int vips_hip_reduce_gen_tiled(const VipsHipReduce *reducev,
	const VipsHipReduce *reduceh,
	const VipsHipRegion *in, const VipsHipRegion *out, int tile);

/* ------------------------------------------------- resample: shrinkh/shrinkv */

/* vips_shrinkh_gen / vips_shrinkv_gen (resample/shrinkh.c:235-283,
 * shrinkv.c:318-387).  Input coordinates beyond the image edge are clamped
 * (the reference embeds, shrinkh.c:383-386, shrinkv.c:501-506).
 */
VIPS_HIP_API int vips_hip_shrinkh_gen(int hshrink,
	const VipsHipRegion *in, const VipsHipRegion *out);
VIPS_HIP_API int vips_hip_shrinkv_gen(int vshrink,
	const VipsHipRegion *in, const VipsHipRegion *out);
/* Output sizes: shrinkh.c:414-416, shrinkv.c:566-568. */
VIPS_HIP_API int vips_hip_shrink_out_size(int in_size, int shrink, int ceil_mode);

/* ------------------------------------------------------------------ upsizing */

/* vips_affine_gen (resample/affine.c:230-397) for the pure scale vips_resize asks for
 * (resize.c:268-300: matrix (hscale, 0, 0, vscale), "idx"/"idy" displacements, EXTEND_COPY,
 * premultiplied) with vips_interpolate_nearest / _bilinear (resample/interpolate.c:336-352,
 * 432-484) or _bicubic (resample/bicubic.cpp:482-600): fills @out's rect of the
 * vips_hip_affine_out_size() image from @in, whose window must hold the stencils.  The
 * reference accumulates a row's x coordinate from the first pixel of each generate rect;
 * @tile_width says where those rects start (multiples of it; 0 = whole rows, the FATSTRIP
 * geometry a scale-only affine requests, affine.c:575-579).  uchar ... int and float
 * (complex as float pairs); double images are refused (no-table bicubic).
 */
VIPS_HIP_API int vips_hip_upsize_gen(const VipsHipRegion *in, const VipsHipRegion *out,
	double hscale, double vscale, double idx, double idy, int interpolate, int tile_width);
/* Output size of the affine: VIPS_ROUND_INT(scale * in_size), resample/transform.c:220-231. */
VIPS_HIP_API int vips_hip_affine_out_size(int in_size, double scale);
/* vips_zoom (conversion/zoom.c): integral pixel replication, what vips_resize uses for
 * kernel nearest with integral scales (resize.c:257-266). */
VIPS_HIP_API int vips_hip_zoom_gen(const VipsHipRegion *in, const VipsHipRegion *out, int xfac, int yfac);
/* vips_subsample (conversion/subsample.c:148-240): out(x, y) = in(x * xfac, y * yfac), output
 * size in / fac; what vips_resize uses for the integer part of a nearest-neighbour shrink
 * (resize.c:165-203). */
VIPS_HIP_API int vips_hip_subsample_gen(const VipsHipRegion *in, const VipsHipRegion *out, int xfac, int yfac);

/* -------------------------------------------------------------- convolution */

/* Host-side state of one convi/convf: vips_convi_build (convolution/convi.c:
 * 1123-1233, C path) / vips_convf_build (convf.c:285-369): coefficients with
 * zeros squeezed out, their mask positions, scale/offset.
 */
typedef struct _VipsHipConv VipsHipConv;

/* @mask: mask_width x mask_height doubles, row-major (a VIPS matrix image).
 * precision INTEGER = convi C path, FLOAT = convf.
 */
VIPS_HIP_API VipsHipConv *vips_hip_conv_new(const double *mask,
	int mask_width, int mask_height, double scale, double offset, int precision);
VIPS_HIP_API void vips_hip_conv_free(VipsHipConv *conv);
VIPS_HIP_API int vips_hip_conv_get_nnz(const VipsHipConv *conv);
/* The Highway-variant view of an INTEGER plan (vips_convi_intize, convi.c:925-1120): the shared
 * exponent and the non-zero 8-bit mantissas with their mask positions.  Returns their number,
 * 0 when the intize refuses the mask (the C path runs even with vectors enabled), -1 on error. */
VIPS_HIP_API int vips_hip_conv_get_vector(const VipsHipConv *conv, int *exp, int *mant, int *pos, int max);
/* Output band format for input @format: convf.c:354-355; convi keeps it. */
VIPS_HIP_API int vips_hip_conv_out_format(const VipsHipConv *conv, int format);
/* Fill out->valid: vips_convi_gen (convi.c:753-857) or vips_convf_gen
 * (convf.c:185-283).  The input is the UN-embedded image; edge clamp included.
 */
VIPS_HIP_API int vips_hip_conv_gen(const VipsHipConv *conv,
	const VipsHipRegion *in, const VipsHipRegion *out);

/* precision=approximate.  Host-side state of one vips_conva (convolution/conva.c:676-767: the
 * rint()ed mask cut into `layers` slabs, every slab row a run of ones, near-identical runs
 * clustered within `cluster`, runs on consecutive rows joined into boxes) or one vips_convasep
 * (convolution/convasep.c:152-330, the 1-D form).  The decomposition fixes the approximated mask,
 * the divisor and the rounding term, so it has to be the reference's, quirk for quirk.
 */
typedef struct _VipsHipConva VipsHipConva;

VIPS_HIP_API VipsHipConva *vips_hip_conva_new(const double *mask,
	int mask_width, int mask_height, double scale, double offset, int layers, int cluster);
VIPS_HIP_API VipsHipConva *vips_hip_convasep_new(const double *mask,
	int mask_n, double scale, double offset, int layers);
VIPS_HIP_API void vips_hip_conva_free(VipsHipConva *plan);
/* The decomposition, for inspection (host only, no device needed).
 *   conva:    info = {n_runs, n_columns, divisor, rounding, offset, longest run};
 *             lines = n_runs x {start, end} then n_columns x {run, factor, first row, last row + 1}
 *             (the hlines and vlines of conva.c:140-205)
 *   convasep: info = {n_lines, divisor, rounding, offset, 0, 0}; lines = n x {start, end, factor}
 * Returns the number of ints written to @lines, or -1.
 */
VIPS_HIP_API int vips_hip_conva_get_lines(const VipsHipConva *plan, int *info, int *lines, int max_ints);
/* Fill out->valid: vips_conva_hgenerate + vips_conva_vgenerate (conva.c:876-1020, 1099-1198) in
 * one pass over the UN-embedded image; output format == input format.
 */
VIPS_HIP_API int vips_hip_conva_gen(const VipsHipConva *plan,
	const VipsHipRegion *in, const VipsHipRegion *out);
/* One pass of vips_convasep: vips_convasep_generate_horizontal (convasep.c:516-590, no offset) or
 * _vertical (:677-750, adds the offset).
 */
VIPS_HIP_API int vips_hip_convasep_gen(const VipsHipConva *plan,
	const VipsHipRegion *in, const VipsHipRegion *out, int vertical);

/* vips_gaussmat (create/gaussmat.c:95-167).  Writes at most @max doubles,
 * returns the mask width (height is 1 when separable, else == width), or -1.
 */
VIPS_HIP_API int vips_hip_gaussmat(double sigma, double min_ampl, int separable,
	int precision, double *mask, int max, double *scale);

/* ------------------------------------------------------------------- colour */

/* The per-scanline process_line functions (colour/colour.c:119-156), one call
 * per region.  in/out bands: 3 colour bands, any extra bands are copied through
 * with the format cast vips_colour_build does (colour.c:196-296).
 */
typedef enum {
	VIPS_HIP_COLOUR_sRGB2scRGB = 0, /* colour/sRGB2scRGB.c:72-106, uchar/ushort in, float out */
	VIPS_HIP_COLOUR_scRGB2XYZ,      /* scRGB2XYZ.c:58-82 */
	VIPS_HIP_COLOUR_XYZ2Lab,        /* XYZ2Lab.c:144-171 */
	VIPS_HIP_COLOUR_Lab2XYZ,        /* Lab2XYZ.c:114-143 */
	VIPS_HIP_COLOUR_XYZ2scRGB,      /* XYZ2scRGB.c:72-94 */
	VIPS_HIP_COLOUR_scRGB2sRGB,     /* scRGB2sRGB.c:84-132, float in, uchar out */
	VIPS_HIP_COLOUR_scRGB2sRGB16,   /* scRGB2sRGB.c, depth 16, ushort out */
	VIPS_HIP_COLOUR_Lab2LabS,       /* Lab2LabS.c:59-73 */
	VIPS_HIP_COLOUR_LabS2Lab,       /* LabS2Lab.c:55-69 */
	VIPS_HIP_COLOUR_sRGB2scRGB16,   /* sRGB2scRGB.c:91-105, RGB16 (ushort) in */
	VIPS_HIP_COLOUR_LAST
} VipsHipColourStep;

VIPS_HIP_API int vips_hip_colour_gen(int step,
	const VipsHipRegion *in, const VipsHipRegion *out);
/* A whole colourspace route (colourspace.c:223-520) in one pass: the steps are
 * evaluated per pixel in registers with the reference's intermediate types, so the
 * result is bit-identical to running them as separate images.  Decoding steps
 * (sRGB2scRGB*, LabS2Lab) may only come first and encoding steps (scRGB2sRGB*,
 * Lab2LabS) only last.  The stored input may be uchar, ushort, short or float: it is
 * vips_cast to what the first step wants (colour.c:343-348,428-434).  @alpha_scale is
 * max_alpha_after / max_alpha_before for the extra bands (colour.c:257-273).
 */
VIPS_HIP_API int vips_hip_colour_route_gen(const int *steps, int n_steps, double alpha_scale,
	const VipsHipRegion *in, const VipsHipRegion *out);

/* vips_cast (conversion/cast.c:120-330): clip + truncate between any two
 * non-complex band formats.
 */
VIPS_HIP_API int vips_hip_cast_gen(const VipsHipRegion *in, const VipsHipRegion *out);

/* vips_premultiply_gen (conversion/premultiply.c:134-214) / vips_unpremultiply_gen
 * (conversion/unpremultiply.c:198-262); the last band is alpha.  @uchar selects the
 * uchar -> uchar fixed-point fast path (scale table, (in * scale + 128) >> 8) that
 * vips_thumbnail uses (resample/thumbnail.c:848-904); otherwise the output is float.
 */
VIPS_HIP_API int vips_hip_premultiply_gen(const VipsHipRegion *in, const VipsHipRegion *out,
	double max_alpha, int uchar, int inverse);

/* vips_sharpen_generate (convolution/sharpen.c:116-168): LabS in, LabS out; the
 * blurred L band comes from a vips_hip_conv_gen pass the caller ran.
 */
VIPS_HIP_API int vips_hip_sharpen_gen(const int *lut_device /* 65536 ints */,
	const VipsHipRegion *in, const VipsHipRegion *blurred_l, const VipsHipRegion *out);

/* --------------------------------------------------------------- image ops */

/* A whole image resident in HBM (the VipsImage of this library). */
typedef struct _VipsHipImage VipsHipImage;

VIPS_HIP_API VipsHipImage *vips_hip_image_new(int width, int height, int bands,
	int format, int interpretation);
/* Upload from / wrap host or device memory (vips_image_new_from_memory,
 * iofuncs/image.c). */
VIPS_HIP_API VipsHipImage *vips_hip_image_new_from_memory(const void *host_data,
	int width, int height, int bands, int format, int interpretation);
VIPS_HIP_API VipsHipImage *vips_hip_image_new_from_device(void *device_data,
	int width, int height, int bands, int format, int interpretation);
VIPS_HIP_API void vips_hip_image_unref(VipsHipImage *image);
/* vips_hip_image_unref() on images[0 .. n) (NULL entries are skipped), each entry set to NULL:
 * what a caller of the batch entry points does with a batch's results (VIPS_UNREF in a loop,
 * g_object_unref, gobject/gobject.c). */
VIPS_HIP_API void vips_hip_image_unref_many(VipsHipImage **images, int n);
VIPS_HIP_API int vips_hip_image_write_to_memory(const VipsHipImage *image, void *host_data);
VIPS_HIP_API void *vips_hip_image_get_data(const VipsHipImage *image);
/* the device the pixels live on (-1 for a NULL image) */
VIPS_HIP_API int vips_hip_image_get_device(const VipsHipImage *image);
VIPS_HIP_API int vips_hip_image_get_width(const VipsHipImage *image);
VIPS_HIP_API int vips_hip_image_get_height(const VipsHipImage *image);
VIPS_HIP_API int vips_hip_image_get_bands(const VipsHipImage *image);
VIPS_HIP_API int vips_hip_image_get_format(const VipsHipImage *image);
VIPS_HIP_API int vips_hip_image_get_interpretation(const VipsHipImage *image);
VIPS_HIP_API size_t vips_hip_image_get_stride(const VipsHipImage *image);
VIPS_HIP_API void vips_hip_image_region(const VipsHipImage *image, VipsHipRegion *region);

/* The native ".v" format (doc/file-format.md; iofuncs/vips.c:283-441): 64-byte header, then
 * band-interleaved scanlines without padding, then optional XML metadata (not carried here).
 * vips_hip_vfile_read_header is host-only (vips__read_header_bytes plus the file-length check of
 * iofuncs/image.c:966-979); the loader and the saver move the pixels between the file and HBM
 * through two pinned buffers so that disc and PCIe transfers overlap (the role of the two
 * write-behind buffers of iofuncs/sinkdisc.c:195-220).  Files with big-endian pixels or LABQ /
 * RAD coding are refused.
 */
typedef struct _VipsHipVHeader {
	int width, height, bands;
	int format;         /* VipsBandFormat */
	int coding;         /* VipsCoding: 0 none, 2 LABQ, 6 RAD */
	int interpretation; /* VipsInterpretation, -1 when the file holds an unknown value */
	float xres, yres;   /* pixels per mm */
	int xoffset, yoffset;
	int msb_first;         /* pixel data is big-endian */
	long long data_offset; /* == 64 */
	long long data_size;   /* width * height * bands * sizeof(format) */
} VipsHipVHeader;

VIPS_HIP_API int vips_hip_vfile_read_header(const char *path, VipsHipVHeader *header);
VIPS_HIP_API VipsHipImage *vips_hip_image_new_from_vfile(const char *path);
VIPS_HIP_API int vips_hip_image_write_to_vfile(const VipsHipImage *image, const char *path);

/* JPEG shrink-on-load in front of the device path.  The entropy decode is
 * libjpeg's, on the host, exactly as foreign/jpeg2vips.c:517-640,800-905 drives it (scale 1/shrink,
 * output cropped to size / shrink rounded down, CMYK inverted); libjpeg.so.9 is bound with dlopen
 * at first use.  vips_hip_thumbnail is vips_thumbnail() (resample/thumbnail.c:549-676 open +
 * :678-1067 build) for JPEG and .v files: vips_thumbnail_find_jpegshrink (:488-519) picks the
 * block shrink, the pre-shrunk image is uploaded, the rest is vips_hip_thumbnail_image.  Files
 * that need auto-rotation, or ICC colour management (an embedded profile in linear mode), are
 * refused.
 */
typedef struct _VipsHipJpegHeader {
	int width, height;             /* after the shrink */
	int bands;                     /* 1 grey, 3 RGB, 4 CMYK */
	int interpretation;            /* B_W, sRGB or CMYK (15) */
	int image_width, image_height; /* of the file */
	int orientation;               /* EXIF orientation, 0 when absent */
	int has_icc;
} VipsHipJpegHeader;

VIPS_HIP_API int vips_hip_thumbnail_find_jpegshrink(int in_width, int in_height,
	int width, int height, int size, int linear, int crop);
VIPS_HIP_API int vips_hip_jpeg_read_header(const char *path, int shrink, VipsHipJpegHeader *header);
VIPS_HIP_API int vips_hip_jpeg_read_to_memory(const char *path, int shrink, void *host_data, size_t size);
VIPS_HIP_API VipsHipImage *vips_hip_image_new_from_jpeg(const char *path, int shrink);
VIPS_HIP_API int vips_hip_thumbnail(const char *path, VipsHipImage **out,
	int width, int height, int size, int linear, int crop);
/* @n files on @n_threads host threads, each with its own stream: decode, upload and device work
 * of different files overlap (the shape of BASELINE config C4 when the inputs are files).
 * Returns the number of failures; outs[i] is NULL for those and, when @errors is given
 * (n x 256 bytes), errors + 256 * i holds the message.
 */
VIPS_HIP_API int vips_hip_thumbnail_batch(const char *const *paths, int n, VipsHipImage **outs,
	char *errors, int width, int height, int size, int linear, int crop, int n_threads);

/* Emulate the reference sink's strip height when seeding the reduce position
 * accumulators (see vips_hip_reducev_gen_tiled); default 16 = vips__fatstrip_height
 * (include/vips/private.h:147-153). */
VIPS_HIP_API void vips_hip_set_fatstrip_height(int lines);

/* The operation wrappers: same names, argument meaning and error behaviour as
 * vips_reduceh() .. vips_thumbnail_image(); optional arguments are explicit.
 * On success *out is a new image the caller unrefs.
 */
VIPS_HIP_API int vips_hip_reduceh(VipsHipImage *in, VipsHipImage **out,
	double hshrink, int kernel, double gap);
VIPS_HIP_API int vips_hip_reducev(VipsHipImage *in, VipsHipImage **out,
	double vshrink, int kernel, double gap);
VIPS_HIP_API int vips_hip_reduce(VipsHipImage *in, VipsHipImage **out,
	double hshrink, double vshrink, int kernel, double gap);
VIPS_HIP_API int vips_hip_shrinkh(VipsHipImage *in, VipsHipImage **out, int hshrink, int ceil_mode);
VIPS_HIP_API int vips_hip_shrinkv(VipsHipImage *in, VipsHipImage **out, int vshrink, int ceil_mode);
VIPS_HIP_API int vips_hip_shrink(VipsHipImage *in, VipsHipImage **out,
	double hshrink, double vshrink, int ceil_mode);
/* vips_resize (resample/resize.c:135-329): integer shrink + reduce for scales < 1, vips_affine
 * with the kernel's interpolator (or vips_zoom) for scales > 1; vscale <= 0 means == scale;
 * gap < 0 selects the default 2.0 (resize.c:397).  Kernel nearest shrinks by vips_subsample
 * first (resize.c:165-203). */
VIPS_HIP_API int vips_hip_resize(VipsHipImage *in, VipsHipImage **out,
	double scale, double vscale, int kernel, double gap);
/* vips_thumbnail_image (resample/thumbnail.c:678-1067 with vips_thumbnail_calculate_shrink
 * :413-467): processing-space conversion, the shrink for the target box and fit mode,
 * vips_resize, conversion back.  @height <= 0 means == @width; @size is a VipsSize
 * (include/vips/resample.h: 0 both, 1 up, 2 down, 3 force); @linear shrinks in scRGB.
 * Images with alpha are premultiplied around the resize (thumbnail.c:848-904).  Crop,
 * auto-rotate and ICC are outside the path: images that need them are refused.
 */
VIPS_HIP_API int vips_hip_thumbnail_image(VipsHipImage *in, VipsHipImage **out,
	int width, int height, int size, int linear);
/* ... with the crop argument (a VipsInteresting, include/vips/conversion.h:97-107): the box is
 * filled instead of fitted (thumbnail.c:432-437) and the result cut to it by vips_smartcrop's
 * positional modes (conversion/smartcrop.c:359-400): 0 none, 1 centre, 4 low, 5 high, 6 all.
 * The content-driven modes (2 entropy, 3 attention) are refused.
 */
VIPS_HIP_API int vips_hip_thumbnail_image_crop(VipsHipImage *in, VipsHipImage **out,
	int width, int height, int size, int linear, int crop);
/* vips_extract_area (conversion/extract.c:137-187). */
VIPS_HIP_API int vips_hip_extract_area(VipsHipImage *in, VipsHipImage **out,
	int left, int top, int width, int height);
VIPS_HIP_API int vips_hip_conv(VipsHipImage *in, VipsHipImage **out,
	const double *mask, int mask_width, int mask_height, double scale, double offset,
	int precision);
VIPS_HIP_API int vips_hip_convsep(VipsHipImage *in, VipsHipImage **out,
	const double *mask, int mask_n, double scale, double offset, int precision);
/* vips_conva (conva.c:1231-1280) / vips_convasep (convasep.c:775-828): what vips_conv /
 * vips_convsep / vips_gaussblur run for precision APPROXIMATE (with layers 5, cluster 1). */
VIPS_HIP_API int vips_hip_conva(VipsHipImage *in, VipsHipImage **out,
	const double *mask, int mask_width, int mask_height, double scale, double offset,
	int layers, int cluster);
VIPS_HIP_API int vips_hip_convasep(VipsHipImage *in, VipsHipImage **out,
	const double *mask, int mask_n, double scale, double offset, int layers);
VIPS_HIP_API int vips_hip_gaussblur(VipsHipImage *in, VipsHipImage **out,
	double sigma, double min_ampl, int precision);
VIPS_HIP_API int vips_hip_sharpen(VipsHipImage *in, VipsHipImage **out,
	double sigma, double x1, double y2, double y3, double m1, double m2);
/* ------------------------------------------------- row strips over the devices of one process
 *
 * BASELINE config 5 (vips_conv 31 x 31 on 65536 x 65536 ushort tiled across 8 GPUs) the way
 * libvips itself is parallel: threads of ONE process (iofuncs/threadpool.c:301-373, 625), here a
 * thread per device.  An image is held as n row strips, strip k in a persistent window on
 * devices[k] with room for `halo` rows of its neighbours above and below;
 * vips_hip_strips_exchange() moves every halo row device to device (hipMemcpyPeerAsync, xGMI)
 * straight into the window that needs it; vips_hip_conv_strips() = that exchange + the region
 * operation vips_hip_conv_gen() on every window at once.  Pixels are the single-device result's
 * bit for bit.  (The one-process-per-GPU form of the same partition, halos over RCCL, is
 * libvips_amd/sharding.py.)
 */
typedef struct _VipsHipStrips VipsHipStrips;
VIPS_HIP_API VipsHipStrips *vips_hip_strips_new(int im_width, int im_height, int bands, int format,
	int n, const int *devices, int halo);
VIPS_HIP_API void vips_hip_strips_free(VipsHipStrips *strips);
VIPS_HIP_API int vips_hip_strips_count(const VipsHipStrips *strips);
/* strip k: its device, the region of its own rows and the region of its whole window (own rows +
 * halos), both in image coordinates, pointing into the window (fill `own`, read `window`) */
VIPS_HIP_API int vips_hip_strips_region(const VipsHipStrips *strips, int k, int *device,
	VipsHipRegion *own, VipsHipRegion *window);
/* the own rows must be complete (synchronised) on entry; the halos are complete on return */
VIPS_HIP_API int vips_hip_strips_exchange(VipsHipStrips *strips);
/* out[0 .. n): strip k of vips_conv(image), an image on devices[k] */
VIPS_HIP_API int vips_hip_conv_strips(VipsHipStrips *strips, VipsHipImage **out, const double *mask,
	int mask_width, int mask_height, double scale, double offset, int precision);

/* BASELINE config 4: vips_resize(scale, kernel, gap) [then vips_sharpen(sigma, x1, y2, y3, m1,
 * m2)] on n independent images, as libvips would run n pipelines over its thread pool
 * (iofuncs/threadpool.c:625).  A batch of same-sized uchar images whose resize is by 1 / (2 k)
 * runs as one launch of the whole resize chain and one of the sharpen per 64 images; any
 * other batch on n_threads host threads that take images in turn, each on its own stream.
 * sigma < 0: no sharpen.  Returns the number of images that failed (out[i] NULL), -1 when
 * the batch failed as a whole (every out[i] NULL); everything is complete on return.
 */
VIPS_HIP_API int vips_hip_resize_sharpen_batch(VipsHipImage *const *in, int n, VipsHipImage **out,
	double scale, int kernel, double gap,
	double sigma, double x1, double y2, double y3, double m1, double m2, int n_threads);
/* The same, QUEUED: a batch the library runs in batch launches on the caller's device (same-sized
 * uchar images, a 1 / (2 k) resize) returns as soon as its work is queued -- the results are
 * ordered on the calling thread's stream like those of every single-image operation (use them in
 * stream order, or vips_hip_synchronize()) and the host can prepare the next batch meanwhile: what
 * libvips' pipelines do between a sink's two buffers (iofuncs/sinkdisc.c:177-220).  Any other
 * batch, and any failure, completes before the return as above. */
VIPS_HIP_API int vips_hip_resize_sharpen_batch_queue(VipsHipImage *const *in, int n, VipsHipImage **out,
	double scale, int kernel, double gap,
	double sigma, double x1, double y2, double y3, double m1, double m2, int n_threads);
VIPS_HIP_API int vips_hip_colourspace(VipsHipImage *in, VipsHipImage **out, int space);
/* vips_gaussblur() then vips_colourspace() (convolution/gaussblur.c:71-116,
 * colour/colourspace.c:551-612) as one call: on 3-band float images both blur passes and
 * the colour route run in one streaming kernel and the blurred image never reaches HBM
 * (BASELINE config 3); other images take the two operations.  Same pixels either way.
 * The libvips module calls this when a colourspace_hip consumes a gaussblur_hip that nobody
 * has evaluated yet.
 */
VIPS_HIP_API int vips_hip_gaussblur_colourspace(VipsHipImage *in, VipsHipImage **out,
	double sigma, double min_ampl, int precision, int space);
VIPS_HIP_API int vips_hip_cast(VipsHipImage *in, VipsHipImage **out, int format);
/* vips_premultiply / vips_unpremultiply with max_alpha from the interpretation
 * (premultiply.c:246-250) and alpha = the last band. */
VIPS_HIP_API int vips_hip_premultiply(VipsHipImage *in, VipsHipImage **out, int uchar);
VIPS_HIP_API int vips_hip_unpremultiply(VipsHipImage *in, VipsHipImage **out, int uchar);

#ifdef __cplusplus
}
#endif

#endif /* VIPS_HIP_H */
