/* TEST INFRASTRUCTURE ONLY -- never linked or called by the product path.
 *
 * A thin non-variadic C shim over the *real reference* (oracle/_ref/lib/libvips.so,
 * built by oracle/build_ref.sh from /root/reference) so that pytest, the golden
 * fixture generator and bench.py's cpu_baseline leg can drive reference operations
 * through ctypes without variadic calls.
 *
 * Everything here goes through the reference's public API only:
 *   vips_operation_new()           iofuncs/operation.c:736
 *   vips_object_set_from_string()  iofuncs/object.c:2587
 *   vips_cache_operation_buildp()  iofuncs/cache.c:990
 *   vips_image_write_to_memory()   iofuncs/image.c:2901
 *
 * Compiled by oracle/Makefile against the reference headers where they lie; the
 * resulting oracle/_ref/lib/libref_shim.so travels to the GPU box.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vips/vips.h>
#include <vips/vector.h>
#include <gmodule.h>

typedef struct {
	void *data; /* interleaved pixels, no line padding */
	int width;
	int height;
	int bands;
	int format;         /* VipsBandFormat */
	int interpretation; /* VipsInterpretation, 0 = derive from bands/format */
} RefImage;

static int ref_inited = 0;

int
ref_init(int concurrency)
{
	if (!ref_inited) {
		if (VIPS_INIT("ref_shim"))
			return -1;
		/* no operation cache between timed runs. */
		vips_cache_set_max(0);
		ref_inited = 1;
	}
	if (concurrency > 0)
		vips_concurrency_set(concurrency);
	return 0;
}

int
ref_concurrency(void)
{
	return vips_concurrency_get();
}

const char *
ref_error(void)
{
	return vips_error_buffer();
}

void
ref_error_clear(void)
{
	vips_error_clear();
}

void
ref_free(void *p)
{
	g_free(p);
}

static VipsImage *
ref_wrap(const RefImage *im)
{
	VipsImage *x;

	if (!(x = vips_image_new_from_memory(im->data,
			  (size_t) im->width * im->height * im->bands *
				  vips_format_sizeof((VipsBandFormat) im->format),
			  im->width, im->height, im->bands,
			  (VipsBandFormat) im->format)))
		return NULL;
	if (im->interpretation > 0) {
		VipsImage *y;

		if (vips_copy(x, &y, "interpretation", im->interpretation, NULL)) {
			g_object_unref(x);
			return NULL;
		}
		g_object_unref(x);
		x = y;
	}
	return x;
}

/* Build operation @nick with up to three image inputs bound to the named
 * properties, the remaining arguments coming from @args ("a=1,b=2").
 * The image output named @out_name is evaluated with
 * vips_image_write_to_memory() and handed back in @out (caller ref_free()s).
 */
static int
ref_build(const char *nick,
	const char *name1, VipsImage *im1,
	const char *name2, VipsImage *im2,
	const char *args, const char *out_name, VipsImage **result)
{
	VipsOperation *op;

	if (!(op = vips_operation_new(nick)))
		return -1;
	if (name1 && im1)
		g_object_set(op, name1, im1, NULL);
	if (name2 && im2)
		g_object_set(op, name2, im2, NULL);
	if (args && args[0] &&
		vips_object_set_from_string(VIPS_OBJECT(op), args)) {
		g_object_unref(op);
		return -1;
	}
	if (vips_cache_operation_buildp(&op)) {
		vips_object_unref_outputs(VIPS_OBJECT(op));
		g_object_unref(op);
		return -1;
	}
	g_object_get(op, out_name, result, NULL);
	vips_object_unref_outputs(VIPS_OBJECT(op));
	g_object_unref(op);
	return 0;
}

static int
ref_materialise(VipsImage *im, RefImage *out)
{
	size_t size;

	out->data = vips_image_write_to_memory(im, &size);
	if (!out->data)
		return -1;
	out->width = im->Xsize;
	out->height = im->Ysize;
	out->bands = im->Bands;
	out->format = im->BandFmt;
	out->interpretation = im->Type;
	return 0;
}

/* One-input, one-output operation. */
int
ref_run(const char *nick, const RefImage *in, const char *args, RefImage *out)
{
	VipsImage *x, *y;
	int result;

	if (ref_init(0))
		return -1;
	if (!(x = ref_wrap(in)))
		return -1;
	if (ref_build(nick, "in", x, NULL, NULL, args, "out", &y)) {
		g_object_unref(x);
		return -1;
	}
	result = ref_materialise(y, out);
	g_object_unref(y);
	g_object_unref(x);
	return result;
}

/* Image + mask operation (conv, convi, convf, convsep, compass...). */
int
ref_run_mask(const char *nick, const RefImage *in, const RefImage *mask,
	double scale, double offset, const char *args, RefImage *out)
{
	VipsImage *x, *m, *y;
	int result;

	if (ref_init(0))
		return -1;
	if (!(x = ref_wrap(in)))
		return -1;
	if (!(m = ref_wrap(mask))) {
		g_object_unref(x);
		return -1;
	}
	vips_image_set_double(m, "scale", scale);
	vips_image_set_double(m, "offset", offset);
	if (ref_build(nick, "in", x, "mask", m, args, "out", &y)) {
		g_object_unref(m);
		g_object_unref(x);
		return -1;
	}
	result = ref_materialise(y, out);
	g_object_unref(y);
	g_object_unref(m);
	g_object_unref(x);
	return result;
}

/* Creator operations with no image input (gaussmat, logmat...). Returns the
 * image plus its "scale"/"offset" metadata.
 */
int
ref_create(const char *nick, const char *args, RefImage *out,
	double *scale, double *offset)
{
	VipsImage *y;
	int result;

	if (ref_init(0))
		return -1;
	if (ref_build(nick, NULL, NULL, NULL, NULL, args, "out", &y))
		return -1;
	*scale = vips_image_get_scale(y);
	*offset = vips_image_get_offset(y);
	result = ref_materialise(y, out);
	g_object_unref(y);
	return result;
}

/* A chain of one-input operations: "op1:args1;op2:args2;...". Used for the
 * BASELINE pipelines (gaussblur -> colourspace, resize -> sharpen).
 */
static int
ref_chain_build(VipsImage *x, const char *chain, VipsImage **result)
{
	char *copy = g_strdup(chain);
	char *save = NULL;
	VipsImage *cur = x;

	g_object_ref(cur);
	for (char *stage = strtok_r(copy, ";", &save); stage;
		 stage = strtok_r(NULL, ";", &save)) {
		char *colon = strchr(stage, ':');
		const char *args = "";
		VipsImage *next;

		if (colon) {
			*colon = '\0';
			args = colon + 1;
		}
		/* (vips_extract_area / vips_crop name their image argument "input") */
		if (ref_build(stage, !strcmp(stage, "extract_area") || !strcmp(stage, "crop") ? "input" : "in", cur, NULL, NULL,
				args, "out", &next)) {
			g_object_unref(cur);
			g_free(copy);
			return -1;
		}
		g_object_unref(cur);
		cur = next;
	}
	g_free(copy);
	*result = cur;
	return 0;
}

int
ref_run_chain(const char *chain, const RefImage *in, RefImage *out)
{
	VipsImage *x, *y;
	int result;

	if (ref_init(0))
		return -1;
	if (!(x = ref_wrap(in)))
		return -1;
	if (ref_chain_build(x, chain, &y)) {
		g_object_unref(x);
		return -1;
	}
	result = ref_materialise(y, out);
	g_object_unref(y);
	g_object_unref(x);
	return result;
}

/* Time a chain: graph build + full evaluation into memory, best of @repeats
 * wall-clock seconds (SURVEY.md 8(d) "CPU reference timing").  The output is
 * discarded.  Returns < 0 on error.
 */
double
ref_time_chain(const char *chain, const RefImage *in, int repeats)
{
	VipsImage *x;
	double best = -1.0;

	if (ref_init(0))
		return -1.0;
	if (!(x = ref_wrap(in)))
		return -1.0;
	for (int i = 0; i < repeats; i++) {
		VipsImage *y;
		GTimer *timer = g_timer_new();
		size_t size;
		void *data;
		double t;

		if (ref_chain_build(x, chain, &y)) {
			g_timer_destroy(timer);
			g_object_unref(x);
			return -1.0;
		}
		data = vips_image_write_to_memory(y, &size);
		t = g_timer_elapsed(timer, NULL);
		g_timer_destroy(timer);
		g_object_unref(y);
		if (!data) {
			g_object_unref(x);
			return -1.0;
		}
		g_free(data);
		if (best < 0 || t < best)
			best = t;
	}
	g_object_unref(x);
	return best;
}

/* Build operation @nick on a width x height x bands uchar vips_black() WITHOUT evaluating it:
 * the output header and the seconds build took.  What "a libvips build() moves no pixels"
 * (doc/how-it-works.md:57-80) is checked with, for the built-ins and for the *_hip module.
 * @header: Xsize, Ysize, Bands, BandFmt, Type.
 */
int
ref_build_probe(const char *nick, int width, int height, int bands, int interpretation, const char *args,
	int *header, double *seconds)
{
	VipsImage *x, *y;
	GTimer *timer;

	if (ref_init(0))
		return -1;
	if (vips_black(&x, width, height, "bands", bands, NULL))
		return -1;
	if (interpretation > 0) {
		if (vips_copy(x, &y, "interpretation", interpretation, NULL)) {
			g_object_unref(x);
			return -1;
		}
		g_object_unref(x);
		x = y;
	}
	timer = g_timer_new();
	if (ref_build(nick, "in", x, NULL, NULL, args, "out", &y)) {
		g_timer_destroy(timer);
		g_object_unref(x);
		return -1;
	}
	*seconds = g_timer_elapsed(timer, NULL);
	g_timer_destroy(timer);
	header[0] = y->Xsize;
	header[1] = y->Ysize;
	header[2] = y->Bands;
	header[3] = y->BandFmt;
	header[4] = y->Type;
	g_object_unref(y);
	g_object_unref(x);
	return 0;
}

/* Known-answer helpers from the reference's public colour API
 * (include/vips/colour.h), used to pin the colour port.
 */
void
ref_col_Lab2XYZ(float L, float a, float b, float *X, float *Y, float *Z)
{
	vips_col_Lab2XYZ(L, a, b, X, Y, Z);
}

void
ref_col_XYZ2Lab(float X, float Y, float Z, float *L, float *a, float *b)
{
	vips_col_XYZ2Lab(X, Y, Z, L, a, b);
}

/* Open a libvips module (a .so exporting g_module_check_init, libvips/module/heif.c:53-78)
 * the way vips_init() does for everything in $libdir/vips-modules-8.19
 * (iofuncs/init.c:288-330).  Used to load host/_build/vips-hip.so into the reference.
 */
int
ref_load_module(const char *path)
{
	GModule *module;

	if (ref_init(0))
		return -1;
	if (!(module = g_module_open(path, G_MODULE_BIND_LAZY))) {
		vips_error("ref_load_module", "unable to load \"%s\" -- %s", path, g_module_error());
		return -1;
	}

	return 0;
}

int
ref_vector_isenabled(void)
{
	return vips_vector_isenabled();
}

const char *
ref_version(void)
{
	return vips_version_string();
}
