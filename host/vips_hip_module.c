/* vips-hip: the libvips side of the drop-in boundary.
 *
 * A loadable libvips module (the plugin ABI of libvips/module/heif.c:53-78 and
 * iofuncs/init.c:288-330): g_module_check_init() registers VipsOperation subclasses
 *
 *     reduce_hip reduceh_hip reducev_hip shrink_hip shrinkh_hip shrinkv_hip resize_hip
 *     thumbnail_image_hip thumbnail_hip
 *     conv_hip convsep_hip gaussblur_hip sharpen_hip colourspace_hip cast_hip
 *     premultiply_hip unpremultiply_hip
 *
 * with the argument names / meaning / defaults of the originals (resample/reduce.c,
 * shrink.c, resize.c, convolution/conv.c, convsep.c, gaussblur.c, sharpen.c,
 * colour/colourspace.c, conversion/cast.c).  Built-in nicknames are not reused
 * (iofuncs/object.c:2930-2949 flags duplicates).
 *
 * Each operation keeps libvips' object model and its start / generate / stop callback
 * surface (include/vips/image.h:151-154, iofuncs/generate.c:679): build() wires
 * `out` as a partial image whose generate hands out rows of the result.  What
 * changes is where the pixels are made: the whole image lives in HBM (288 GB per
 * MI355X makes the image, not the 128x128 tile, the natural unit), the pixel work is
 * libvipship.so's hand-written HIP reached through the plain-C ABI of
 * include/vips_hip.h, and the device-resident result rides on `out` as metadata so a
 * following *_hip operation consumes it without a host round trip.
 *
 * Host code is C (the reference's language); nothing here knows about HIP.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vips/vips.h>
#include <vips/vector.h>

#include "vips_hip.h"

#define HIP_META "vips-hip-image"

static int
hip_fail(const char *domain)
{
	vips_error(domain, "%s", vips_hip_error_buffer());
	vips_hip_error_clear();

	return -1;
}

/* ------------------------------------------------------------------ device link */

/* How a *_hip operation finds the device-resident result of the *_hip operation that makes
 * its input: the producer hangs a link on its `out` image as a VipsArea.  libvips copies
 * metadata down pipelines (and into vips_image_copy_memory() descendants), so a link is only
 * honoured on the image it was made for, and it names its operation through a GWeakRef: a
 * link that outlives its operation resolves to NULL instead of to a recycled address.
 * Nothing is computed to make a link -- `device` is called when (if) a consumer evaluates.
 */
typedef VipsHipImage *(*HipDeviceFn)(GObject *producer);

typedef struct _HipLink {
	GWeakRef producer;  /* the operation that owns the pixels */
	HipDeviceFn device; /* evaluate (if need be) and return its device image, or NULL */
	VipsImage *owner;   /* the producer's `out`: the only image this link is valid on */
} HipLink;

static int
hip_link_free(void *data, void *unused)
{
	HipLink *link = (HipLink *) data;

	g_weak_ref_clear(&link->producer);
	g_free(link);

	return 0;
}

static void
hip_link_attach(VipsImage *image, GObject *producer, HipDeviceFn device)
{
	HipLink *link = g_new(HipLink, 1);

	g_weak_ref_init(&link->producer, producer);
	link->device = device;
	link->owner = image;
	vips_image_set_area(image, HIP_META, (VipsCallbackFn) hip_link_free, link);
}

/* The *_hip operation that makes @image, with a new reference, or NULL. */
static GObject *
hip_link_producer(VipsImage *image, HipDeviceFn *device)
{
	const void *data;

	if (vips_image_get_typeof(image, HIP_META) &&
		!vips_image_get_area(image, HIP_META, &data)) {
		HipLink *link = (HipLink *) data;

		if (link->owner == image) {
			GObject *producer = g_weak_ref_get(&link->producer);

			if (producer) {
				*device = link->device;
				return producer;
			}
		}
	}

	return NULL;
}

/* ------------------------------------------------------------------ base class */

/* HBM budget (bytes of input + output an operation may hold on the device at once) above
 * which strip-capable operations work through the image in row strips.
 * $VIPS_HIP_BUDGET, with an optional k / m / g suffix; default 64 GiB.
 */
static guint64
hip_budget(void)
{
	const char *env = g_getenv("VIPS_HIP_BUDGET");
	guint64 budget = (guint64) 64 << 30;

	if (env && *env) {
		char *end = NULL;
		guint64 v = g_ascii_strtoull(env, &end, 10);

		if (end && (*end == 'k' || *end == 'K'))
			v <<= 10;
		else if (end && (*end == 'm' || *end == 'M'))
			v <<= 20;
		else if (end && (*end == 'g' || *end == 'G'))
			v <<= 30;
		if (v > 0)
			budget = v;
	}

	return budget;
}

typedef struct _VipsHipOp {
	VipsOperation parent_instance;

	VipsImage *in;
	VipsImage *out;

	/* build() only records these: no pixel is touched until somebody asks for one
	 * (iofuncs/generate.c:679-728 stores the callbacks; doc/how-it-works.md:57-80). */
	VipsImage *ready;       /* `in` after vips_image_decode() */
	GObject *upstream;      /* the *_hip operation that makes `in`, if one does */
	HipDeviceFn upstream_device;

	/* Evaluation state, all under `lock`. */
	GMutex lock;
	gboolean evaluated;
	char *eval_error;       /* non-NULL: evaluation failed, with this message */
	VipsHipImage *result;   /* the result on the device (NULL after a strip-mined run) */
	VipsPel *host;          /* the result on the host, for generate */
} VipsHipOp;

typedef struct _VipsHipOpClass {
	VipsOperationClass parent_class;

	/* Run the operation on a device-resident image. */
	int (*compute)(struct _VipsHipOp *op, VipsHipImage *in, VipsHipImage **out);

	/* Optional: the region-level form, for images over the HBM budget.  strip_open makes the
	 * plan (returns 1 when this instance cannot be strip-mined, e.g. a reducing gap);
	 * strip_need maps output rows to the input rows they read; strip_run makes output region
	 * @out (rows of the whole output image) from input window @in.
	 */
	int (*strip_open)(struct _VipsHipOp *op, VipsImage *in, void **plan);
	void (*strip_need)(struct _VipsHipOp *op, void *plan, int out_top, int out_rows, int *in_top, int *in_rows);
	int (*strip_run)(struct _VipsHipOp *op, void *plan, const VipsHipRegion *in, const VipsHipRegion *out);
	void (*strip_close)(struct _VipsHipOp *op, void *plan);

	/* Optional: this operation and the not-yet-evaluated *_hip operation that makes its input
	 * as ONE device call (colourspace_hip after gaussblur_hip: BASELINE config 3 in one kernel).
	 * @up_in is the upstream operation's input on the device.  Returns 1 when the pair is not
	 * one this hook fuses. */
	int (*fuse)(struct _VipsHipOp *op, struct _VipsHipOp *up, VipsHipImage *up_in, VipsHipImage **out);
} VipsHipOpClass;

#define VIPS_TYPE_HIP_OP (vips_hip_op_get_type())
#define VIPS_HIP_OP(obj) (G_TYPE_CHECK_INSTANCE_CAST((obj), VIPS_TYPE_HIP_OP, VipsHipOp))
#define VIPS_HIP_OP_GET_CLASS(obj) (G_TYPE_INSTANCE_GET_CLASS((obj), VIPS_TYPE_HIP_OP, VipsHipOpClass))

G_DEFINE_ABSTRACT_TYPE(VipsHipOp, vips_hip_op, VIPS_TYPE_OPERATION);

/* The per-thread sequence: owns the stream this worker's copies run on. */
static void *
vips_hip_op_start(VipsImage *out, void *a, void *b)
{
	return (void *) out;
}

static int
vips_hip_op_stop(void *seq, void *a, void *b)
{
	return 0;
}

static VipsHipImage *vips_hip_op_device(GObject *producer);

static HipDeviceFn
vips_hip_op_device_fn(void)
{
	return vips_hip_op_device;
}

static void
hip_eval_fail(VipsHipOp *op, const char *domain)
{
	if (vips_hip_error_buffer()[0])
		hip_fail(domain);
	VIPS_FREE(op->eval_error);
	op->eval_error = g_strdup(vips_error_buffer());
}

/* Does the result have the header build() promised?  (The promise came from the built-in
 * operation's own build; a mismatch is a bug in this module, reported, never papered over.) */
static int
hip_check_header(VipsHipOp *op, int width, int height, int bands, int format)
{
	VipsImage *out = op->out;

	if (width != out->Xsize || height != out->Ysize || bands != out->Bands || format != (int) out->BandFmt) {
		vips_error(VIPS_OBJECT_GET_CLASS(op)->nickname,
			"device result is %dx%dx%d format %d, the operation's header says %dx%dx%d format %d",
			width, height, bands, format, out->Xsize, out->Ysize, out->Bands, (int) out->BandFmt);
		return -1;
	}

	return 0;
}

/* The image in row strips: pull the input rows a strip needs from upstream (a threaded
 * vips_sink_memory() of a vips_crop()), upload, run the region form, download into the host
 * result.  Returns 1 when this operation cannot be strip-mined.
 */
static int
hip_eval_strips(VipsHipOp *op, VipsImage *in, guint64 budget)
{
	VipsHipOpClass *hclass = VIPS_HIP_OP_GET_CLASS(op);
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;
	VipsImage *out = op->out;
	const size_t ls = VIPS_IMAGE_SIZEOF_LINE(out);
	const size_t in_ls = VIPS_IMAGE_SIZEOF_LINE(in);
	void *plan = NULL;
	VipsPel *host;
	int rows;
	int result;

	if (!hclass->strip_open || !hclass->strip_need || !hclass->strip_run)
		return 1;
	if ((result = hclass->strip_open(op, in, &plan)))
		return result;

	/* the tallest strip (a multiple of 16 lines: the reference's fat-strip height, which the
	 * vertical reduce re-seeds its position on) whose input rows + output rows fit */
	for (rows = VIPS_ROUND_UP(out->Ysize, 16); rows > 16; rows = VIPS_ROUND_UP(rows / 2, 16)) {
		int in_top, in_rows;

		hclass->strip_need(op, plan, 0, VIPS_MIN(rows, out->Ysize), &in_top, &in_rows);
		if ((guint64) in_rows * in_ls + (guint64) rows * ls <= budget)
			break;
	}

	if (!(host = (VipsPel *) g_try_malloc(ls * out->Ysize))) {
		vips_error(nick, "%s", "out of memory for the result");
		if (hclass->strip_close)
			hclass->strip_close(op, plan);
		return -1;
	}

	result = 0;
	for (int top = 0; top < out->Ysize && !result; top += rows) {
		const int n = VIPS_MIN(rows, out->Ysize - top);
		VipsImage *crop = NULL, *mem = NULL;
		VipsHipImage *dev_in = NULL, *dev_out = NULL;
		VipsHipRegion ri, ro;
		int in_top, in_rows;

		hclass->strip_need(op, plan, top, n, &in_top, &in_rows);
		if (in_top < 0) {
			in_rows += in_top;
			in_top = 0;
		}
		in_rows = VIPS_MIN(in_rows, in->Ysize - in_top);

		if (vips_image_iskilled(out)) {
			vips_error(nick, "%s", "killed");
			result = -1;
		}
		else if (vips_crop(in, &crop, 0, in_top, in->Xsize, in_rows, NULL) ||
			!(mem = vips_image_copy_memory(crop)))
			result = -1;
		else if (!(dev_in = vips_hip_image_new_from_memory(VIPS_IMAGE_ADDR(mem, 0, 0),
					   mem->Xsize, mem->Ysize, mem->Bands, mem->BandFmt, mem->Type)) ||
			!(dev_out = vips_hip_image_new(out->Xsize, n, out->Bands, out->BandFmt, out->Type)))
			result = hip_fail(nick);
		else {
			vips_hip_image_region(dev_in, &ri);
			ri.top = in_top;
			ri.im_width = in->Xsize;
			ri.im_height = in->Ysize;
			vips_hip_image_region(dev_out, &ro);
			ro.top = top;
			ro.im_width = out->Xsize;
			ro.im_height = out->Ysize;
			if (hclass->strip_run(op, plan, &ri, &ro) ||
				vips_hip_image_write_to_memory(dev_out, host + (size_t) top * ls))
				result = hip_fail(nick);
		}

		vips_hip_image_unref(dev_out);
		vips_hip_image_unref(dev_in);
		VIPS_UNREF(mem);
		VIPS_UNREF(crop);
	}

	if (hclass->strip_close)
		hclass->strip_close(op, plan);
	if (result) {
		g_free(host);
		return -1;
	}
	op->host = host;

	return 0;
}

/* The operation's whole input on the device: the upstream *_hip operation's result (evaluated
 * now if it has not been), else the image pulled from upstream (a threaded vips_sink_memory())
 * and uploaded -- then *fresh is what the caller unrefs when it is done.  NULL on failure.
 */
static VipsHipImage *
hip_input(VipsHipOp *op, VipsHipImage **fresh)
{
	VipsImage *in = op->ready;
	VipsHipImage *dev = NULL;
	VipsImage *mem;

	*fresh = NULL;
	if (op->upstream)
		dev = op->upstream_device(op->upstream);
	if (dev)
		return dev;
	if (!(mem = vips_image_copy_memory(in)))
		return NULL;
	*fresh = vips_hip_image_new_from_memory(VIPS_IMAGE_ADDR(mem, 0, 0),
		mem->Xsize, mem->Ysize, mem->Bands, mem->BandFmt, mem->Type);
	VIPS_UNREF(mem);
	if (!*fresh)
		hip_fail(VIPS_OBJECT_GET_CLASS(op)->nickname);

	return *fresh;
}

static void hip_eval(VipsHipOp *op);

/* This operation fused with the one that makes its input, when the class has a hook for the
 * pair and nobody has evaluated the upstream operation yet (if somebody asks for it later it
 * is simply evaluated then).  TRUE when the result is in place.
 */
static gboolean
hip_eval_fused(VipsHipOp *op)
{
	VipsHipOpClass *hclass = VIPS_HIP_OP_GET_CLASS(op);
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;
	gboolean done = FALSE;
	VipsHipOp *up;
	gboolean retried = FALSE;

	if (!hclass->fuse || !op->upstream || op->upstream_device != vips_hip_op_device_fn())
		return FALSE;
	up = VIPS_HIP_OP(op->upstream);

	g_mutex_lock(&up->lock); /* downstream lock, then upstream lock: the order every evaluation takes */
	if (!up->evaluated) {
		VipsHipImage *fresh = NULL;
		VipsHipImage *up_in = hip_input(up, &fresh);

		if (up_in) {
			const int r = hclass->fuse(op, up, up_in, &op->result);

			if (r == 0 && !vips_hip_synchronize())
				done = TRUE;
			else if (r != 1) {
				if (op->result) {
					vips_hip_image_unref(op->result);
					op->result = NULL;
				}
				hip_fail(nick);
				retried = TRUE;
			}
			vips_hip_image_unref(fresh);
		}
	}
	g_mutex_unlock(&up->lock);
	/* a FAILED attempt falls back to the two operations, which report for themselves: only then
	 * is the message it just logged dropped (libvips' error buffer is process-wide: clearing it
	 * on every evaluation would wipe what other pipelines logged) */
	if (retried)
		vips_error_clear();

	return done;
}

/* Evaluate, once.  Called with the lock held, from the first generate or from a downstream
 * *_hip operation that wants the device image.
 */
static void
hip_eval(VipsHipOp *op)
{
	VipsObjectClass *class = VIPS_OBJECT_GET_CLASS(op);
	VipsHipOpClass *hclass = VIPS_HIP_OP_GET_CLASS(op);
	VipsImage *in = op->ready;
	VipsImage *mem = NULL;
	VipsHipImage *dev = NULL;
	VipsHipImage *fresh = NULL;

	op->evaluated = TRUE;

	/* NULL selects the library's own per-thread stream for this (worker) thread. */
	if (vips_hip_set_stream(NULL)) {
		hip_eval_fail(op, class->nickname);
		return;
	}

	if (hip_eval_fused(op)) {
		if (hip_check_header(op, vips_hip_image_get_width(op->result), vips_hip_image_get_height(op->result),
				vips_hip_image_get_bands(op->result), vips_hip_image_get_format(op->result))) {
			vips_hip_image_unref(op->result);
			op->result = NULL;
			hip_eval_fail(op, class->nickname);
		}
		return;
	}

	/* A device-resident input (made by another *_hip op, evaluated now if it has not been)
	 * is used as it is ... */
	if (op->upstream)
		dev = op->upstream_device(op->upstream);

	if (!dev) {
		/* ... anything else is pulled from upstream -- in strips when the image is over the
		 * HBM budget and the operation has a region form, else in one piece -- and uploaded. */
		const guint64 bytes = (guint64) VIPS_IMAGE_SIZEOF_IMAGE(in) + (guint64) VIPS_IMAGE_SIZEOF_IMAGE(op->out);
		const guint64 budget = hip_budget();

		if (bytes > budget) {
			const int r = hip_eval_strips(op, in, budget);

			if (r < 0)
				hip_eval_fail(op, class->nickname);
			if (r <= 0)
				return;
			/* r == 1: no region form -- try the whole image (fails loudly if HBM runs out) */
		}
		if (!(mem = vips_image_copy_memory(in))) {
			hip_eval_fail(op, class->nickname);
			return;
		}
		if (!(fresh = vips_hip_image_new_from_memory(VIPS_IMAGE_ADDR(mem, 0, 0),
				  mem->Xsize, mem->Ysize, mem->Bands, mem->BandFmt, mem->Type))) {
			VIPS_UNREF(mem);
			hip_eval_fail(op, class->nickname);
			return;
		}
		VIPS_UNREF(mem);
		dev = fresh;
	}

	/* The Highway variant of convi on uchar (8-bit mantissas, shared exponent: convi.c:932-1120,
	 * convi_hwy.cpp) is PARITY UNPINNED -- no Highway build of the reference exists to compare
	 * with -- so it is never selected silently: only with VIPS_HIP_HWY_CONVI=1 does the module
	 * follow the host library's vector switch (iofuncs/vector.cpp:98-113).  Otherwise uchar
	 * convolutions use the C path's exact integer arithmetic (convi.c:698-716), whatever the
	 * host libvips was built with. */
	{
		const char *hwy = g_getenv("VIPS_HIP_HWY_CONVI");

		vips_hip_vector_set_enabled(hwy && atoi(hwy) == 1 && vips_vector_isenabled());
	}

	/* The kernels are queued on THIS thread's stream; other generates run on other worker
	 * threads with their own (non-blocking) streams, and a downstream *_hip op may evaluate
	 * on yet another thread: finish the work before anyone else can see the result, and
	 * before the uploaded input goes back to the pool. */
	if (hclass->compute(op, dev, &op->result) ||
		vips_hip_synchronize()) {
		if (op->result) {
			vips_hip_image_unref(op->result);
			op->result = NULL;
		}
		vips_hip_image_unref(fresh);
		hip_eval_fail(op, class->nickname);
		return;
	}
	vips_hip_image_unref(fresh);

	if (hip_check_header(op, vips_hip_image_get_width(op->result), vips_hip_image_get_height(op->result),
			vips_hip_image_get_bands(op->result), vips_hip_image_get_format(op->result))) {
		vips_hip_image_unref(op->result);
		op->result = NULL;
		hip_eval_fail(op, class->nickname);
	}
}

/* For consumers: the device image (borrowed: it lives as long as the operation), or NULL when
 * there is none (evaluation failed -- the consumer's own pull of `in` will then report it --
 * or the result was strip-mined to the host).
 */
static VipsHipImage *
vips_hip_op_device(GObject *producer)
{
	VipsHipOp *op = VIPS_HIP_OP(producer);
	VipsHipImage *result;

	g_mutex_lock(&op->lock);
	if (!op->evaluated)
		hip_eval(op);
	result = op->result;
	g_mutex_unlock(&op->lock);

	return result;
}

static int
vips_hip_op_gen(VipsRegion *out_region, void *seq, void *a, void *b, gboolean *stop)
{
	VipsHipOp *op = (VipsHipOp *) b;
	VipsRect *r = &out_region->valid;
	VipsImage *out = out_region->im;
	const size_t ps = VIPS_IMAGE_SIZEOF_PEL(out);
	const size_t ls = VIPS_IMAGE_SIZEOF_LINE(out);

	if (vips_image_iskilled(out))
		return -1;

	/* First demand: evaluate, then bring the device result to the host, once. */
	g_mutex_lock(&op->lock);
	if (!op->evaluated)
		hip_eval(op);
	if (op->eval_error) {
		vips_error(VIPS_OBJECT_GET_CLASS(op)->nickname, "%s", op->eval_error);
		g_mutex_unlock(&op->lock);
		return -1;
	}
	if (!op->host) {
		VipsPel *host = VIPS_ARRAY(NULL, ls * out->Ysize, VipsPel);

		if (!host || vips_hip_image_write_to_memory(op->result, host)) {
			g_mutex_unlock(&op->lock);
			VIPS_FREE(host);
			return hip_fail(VIPS_OBJECT_GET_CLASS(op)->nickname);
		}
		op->host = host;
	}
	g_mutex_unlock(&op->lock);

	for (int y = 0; y < r->height; y++)
		memcpy(VIPS_REGION_ADDR(out_region, r->left, r->top + y),
			op->host + (size_t) (r->top + y) * ls + (size_t) r->left * ps,
			(size_t) r->width * ps);

	return 0;
}

/* Copy every assigned input argument of @object to the same-named property of the twin. */
static void *
hip_copy_argument(VipsObject *object, GParamSpec *pspec, VipsArgumentClass *argument_class,
	VipsArgumentInstance *argument_instance, void *a, void *b)
{
	GObject *twin = G_OBJECT(a);
	const char *name = g_param_spec_get_name(pspec);

	if ((argument_class->flags & VIPS_ARGUMENT_INPUT) && argument_instance->assigned &&
		g_object_class_find_property(G_OBJECT_GET_CLASS(twin), name)) {
		GValue value = G_VALUE_INIT;

		g_value_init(&value, G_PARAM_SPEC_VALUE_TYPE(pspec));
		g_object_get_property(G_OBJECT(object), name, &value);
		g_object_set_property(twin, name, &value);
		g_value_unset(&value);
	}

	return NULL;
}

/* What will `out` look like?  A drop-in has, by definition, the header the original would
 * produce, and a libvips build() moves no pixels: so build the ORIGINAL operation (the
 * nickname less "_hip") with the same arguments, copy its output's header, drop it.  Exact by
 * construction, costs microseconds, and needs no device.
 */
static int
hip_twin_header(VipsHipOp *op, VipsImage *out)
{
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;
	const size_t len = strlen(nick);
	char base[64];
	VipsOperation *twin;
	VipsImage *twin_out = NULL;

	if (len < 5 || len >= sizeof(base) || strcmp(nick + len - 4, "_hip") != 0) {
		vips_error(nick, "%s", "not a *_hip nickname");
		return -1;
	}
	memcpy(base, nick, len - 4);
	base[len - 4] = '\0';

	if (!(twin = vips_operation_new(base)))
		return -1;
	vips_argument_map(VIPS_OBJECT(op), hip_copy_argument, twin, NULL);
	if (vips_object_build(VIPS_OBJECT(twin))) {
		vips_object_unref_outputs(VIPS_OBJECT(twin));
		g_object_unref(twin);
		return -1;
	}
	g_object_get(twin, "out", &twin_out, NULL);
	out->Xsize = twin_out->Xsize;
	out->Ysize = twin_out->Ysize;
	out->Bands = twin_out->Bands;
	out->BandFmt = twin_out->BandFmt;
	out->Coding = twin_out->Coding;
	out->Type = twin_out->Type;
	out->Xres = twin_out->Xres;
	out->Yres = twin_out->Yres;
	out->Xoffset = twin_out->Xoffset;
	out->Yoffset = twin_out->Yoffset;
	g_object_unref(twin_out);
	vips_object_unref_outputs(VIPS_OBJECT(twin));
	g_object_unref(twin);

	return 0;
}

static int
vips_hip_op_build(VipsObject *object)
{
	VipsObjectClass *class = VIPS_OBJECT_GET_CLASS(object);
	VipsHipOp *op = VIPS_HIP_OP(object);
	VipsImage **t = (VipsImage **) vips_object_local_array(object, 2);

	VipsImage *in;

	if (VIPS_OBJECT_CLASS(vips_hip_op_parent_class)->build(object))
		return -1;

	in = op->in;
	if (vips_image_decode(in, &t[0]))
		return -1;
	in = t[0];
	if (vips_band_format_iscomplex(in->BandFmt)) {
		vips_error(class->nickname, "%s", "complex images are outside the HIP path");
		return -1;
	}
	op->ready = in;
	op->upstream = hip_link_producer(op->in, &op->upstream_device);

	/* No pixel work here (iofuncs/generate.c:705-728: a partial image only stores its
	 * callbacks): the header comes from the original operation's build, the pixels are made
	 * when the first one is asked for. */
	g_object_set(object, "out", vips_image_new(), NULL);
	if (vips_image_pipelinev(op->out, VIPS_DEMAND_STYLE_ANY, in, NULL) ||
		hip_twin_header(op, op->out))
		return -1;

	if (vips_image_generate(op->out,
			vips_hip_op_start, vips_hip_op_gen, vips_hip_op_stop, in, op))
		return -1;

	/* Downstream *_hip ops reach the device copy through this. */
	hip_link_attach(op->out, G_OBJECT(op), vips_hip_op_device);

	return 0;
}

static void
vips_hip_op_dispose(GObject *gobject)
{
	VipsHipOp *op = VIPS_HIP_OP(gobject);

	VIPS_FREE(op->host);
	VIPS_FREE(op->eval_error);
	if (op->result) {
		vips_hip_image_unref(op->result);
		op->result = NULL;
	}
	VIPS_UNREF(op->upstream);

	G_OBJECT_CLASS(vips_hip_op_parent_class)->dispose(gobject);
}

static void
vips_hip_op_class_init(VipsHipOpClass *class)
{
	GObjectClass *gobject_class = G_OBJECT_CLASS(class);
	VipsObjectClass *vobject_class = VIPS_OBJECT_CLASS(class);

	gobject_class->dispose = vips_hip_op_dispose;
	gobject_class->set_property = vips_object_set_property;
	gobject_class->get_property = vips_object_get_property;

	vobject_class->nickname = "hip_op";
	vobject_class->description = "MI355X operations";
	vobject_class->build = vips_hip_op_build;

	VIPS_ARG_IMAGE(class, "in", 1, "Input", "Input image",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsHipOp, in));
	VIPS_ARG_IMAGE(class, "out", 2, "Output", "Output image",
		VIPS_ARGUMENT_REQUIRED_OUTPUT, G_STRUCT_OFFSET(VipsHipOp, out));
}

static void
vips_hip_op_init(VipsHipOp *op)
{
	g_mutex_init(&op->lock);
}

/* ------------------------------------------------------------------ subclasses */

#define HIP_SUBCLASS_FULL(TypeName, type_name, nick, desc, STRIP_HOOKS) \
	typedef VipsHipOpClass TypeName##Class; \
	G_DEFINE_TYPE(TypeName, type_name, VIPS_TYPE_HIP_OP); \
	static void type_name##_args(TypeName##Class *class); \
	static void \
	type_name##_class_init(TypeName##Class *class) \
	{ \
		GObjectClass *gobject_class = G_OBJECT_CLASS(class); \
		VipsObjectClass *vobject_class = VIPS_OBJECT_CLASS(class); \
		gobject_class->set_property = vips_object_set_property; \
		gobject_class->get_property = vips_object_get_property; \
		vobject_class->nickname = nick; \
		vobject_class->description = desc; \
		class->compute = type_name##_compute; \
		STRIP_HOOKS \
		type_name##_args(class); \
	}

#define HIP_SUBCLASS(TypeName, type_name, nick, desc) HIP_SUBCLASS_FULL(TypeName, type_name, nick, desc, )

/* For operations with a region form: images over the HBM budget go through in row strips. */
#define HIP_STRIPS(type_name) \
	class->strip_open = type_name##_strip_open; \
	class->strip_need = type_name##_strip_need; \
	class->strip_run = type_name##_strip_run; \
	class->strip_close = type_name##_strip_close;

/* reduce_hip: resample/reduce.c:98-200 */
typedef struct _VipsReduceHip {
	VipsHipOp parent_instance;
	double hshrink, vshrink, gap;
	VipsKernel kernel;
} VipsReduceHip;

static int
vips_reduce_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsReduceHip *reduce = (VipsReduceHip *) op;

	return vips_hip_reduce(in, out, reduce->hshrink, reduce->vshrink, reduce->kernel, reduce->gap);
}

/* The region form of reduce_hip: the plans the whole-image operation would make
 * (ops_resample.cpp vips_hip_reduce), run per strip through the generate replacements. */
typedef struct _ReduceStrip {
	VipsHipReduce *rv, *rh;
	int in_width, out_height;
} ReduceStrip;

static void
vips_reduce_hip_strip_close(VipsHipOp *op, void *plan)
{
	ReduceStrip *p = (ReduceStrip *) plan;

	if (p) {
		vips_hip_reduce_free(p->rv);
		vips_hip_reduce_free(p->rh);
		g_free(p);
	}
}

static int
vips_reduce_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsReduceHip *reduce = (VipsReduceHip *) op;
	ReduceStrip *p;

	/* a reducing gap puts integer pre-shrinks in front; factor 1 is a copy: whole image only */
	if (reduce->gap > 0.0 || reduce->hshrink == 1.0 || reduce->vshrink == 1.0 ||
		reduce->kernel == VIPS_KERNEL_NEAREST)
		return 1;
	p = g_new0(ReduceStrip, 1);
	p->in_width = in->Xsize;
	p->out_height = op->out->Ysize;
	p->rv = vips_hip_reduce_new(reduce->kernel, reduce->vshrink, in->Ysize, op->out->Ysize, NAN);
	p->rh = vips_hip_reduce_new(reduce->kernel, reduce->hshrink, in->Xsize, op->out->Xsize, NAN);
	if (!p->rv || !p->rh) {
		vips_reduce_hip_strip_close(op, p);
		return hip_fail("reduce_hip");
	}
	*plan = p;

	return 0;
}

static void
vips_reduce_hip_strip_need(VipsHipOp *op, void *plan, int out_top, int out_rows, int *in_top, int *in_rows)
{
	vips_hip_reducev_need(((ReduceStrip *) plan)->rv, out_top, out_rows, in_top, in_rows);
}

static int
vips_reduce_hip_strip_run(VipsHipOp *op, void *plan, const VipsHipRegion *in, const VipsHipRegion *out)
{
	ReduceStrip *p = (ReduceStrip *) plan;
	VipsHipImage *mid;
	VipsHipRegion rmid;
	int r;

	/* 16: the fat-strip height the reference's sink evaluates reducev in (thread.c:301-325),
	 * what the whole-image path uses */
	if ((r = vips_hip_reduce_gen_tiled(p->rv, p->rh, in, out, 16)) <= 0)
		return r;
	/* not the fused uchar RGBA case: the two passes, the rows between them on the device */
	if (!(mid = vips_hip_image_new(p->in_width, out->height, in->bands, in->format, 0)))
		return -1;
	vips_hip_image_region(mid, &rmid);
	rmid.top = out->top;
	rmid.im_width = p->in_width;
	rmid.im_height = p->out_height;
	r = vips_hip_reducev_gen_tiled(p->rv, in, &rmid, 16) || vips_hip_reduceh_gen(p->rh, &rmid, out);
	/* the pool orders reuse of mid's block behind these kernels (same thread, same stream) */
	vips_hip_image_unref(mid);

	return r ? -1 : 0;
}

HIP_SUBCLASS_FULL(VipsReduceHip, vips_reduce_hip, "reduce_hip", "reduce an image (MI355X)",
	HIP_STRIPS(vips_reduce_hip))

static void
vips_reduce_hip_args(VipsReduceHipClass *class)
{
	VIPS_ARG_DOUBLE(class, "hshrink", 8, "Hshrink", "Horizontal shrink factor",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsReduceHip, hshrink), 1.0, 1000000.0, 1.0);
	VIPS_ARG_DOUBLE(class, "vshrink", 9, "Vshrink", "Vertical shrink factor",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsReduceHip, vshrink), 1.0, 1000000.0, 1.0);
	VIPS_ARG_ENUM(class, "kernel", 3, "Kernel", "Resampling kernel",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsReduceHip, kernel),
		VIPS_TYPE_KERNEL, VIPS_KERNEL_LANCZOS3);
	VIPS_ARG_DOUBLE(class, "gap", 4, "Gap", "Reducing gap",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsReduceHip, gap), 0.0, 1000000.0, 0.0);
}

static void
vips_reduce_hip_init(VipsReduceHip *reduce)
{
	reduce->gap = 0.0;
	reduce->kernel = VIPS_KERNEL_LANCZOS3;
}

/* reduceh_hip / reducev_hip: resample/reduceh.cpp:567-640, reducev.cpp:1077-1150 */
typedef struct _VipsReduce1Hip {
	VipsHipOp parent_instance;
	double shrink, gap;
	VipsKernel kernel;
} VipsReduce1Hip;

typedef VipsReduce1Hip VipsReducehHip;
typedef VipsReduce1Hip VipsReducevHip;

static int
vips_reduceh_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsReduce1Hip *r = (VipsReduce1Hip *) op;

	return vips_hip_reduceh(in, out, r->shrink, r->kernel, r->gap);
}

static int
vips_reducev_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsReduce1Hip *r = (VipsReduce1Hip *) op;

	return vips_hip_reducev(in, out, r->shrink, r->kernel, r->gap);
}

HIP_SUBCLASS(VipsReducehHip, vips_reduceh_hip, "reduceh_hip", "shrink an image horizontally (MI355X)")
HIP_SUBCLASS(VipsReducevHip, vips_reducev_hip, "reducev_hip", "shrink an image vertically (MI355X)")

#define REDUCE1_ARGS(class, NAME, LONG) \
	VIPS_ARG_DOUBLE(class, NAME, 3, LONG, LONG " shrink factor", \
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsReduce1Hip, shrink), 1.0, 1000000.0, 1.0); \
	VIPS_ARG_ENUM(class, "kernel", 4, "Kernel", "Resampling kernel", \
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsReduce1Hip, kernel), \
		VIPS_TYPE_KERNEL, VIPS_KERNEL_LANCZOS3); \
	VIPS_ARG_DOUBLE(class, "gap", 5, "Gap", "Reducing gap", \
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsReduce1Hip, gap), 0.0, 1000000.0, 0.0);

static void
vips_reduceh_hip_args(VipsReducehHipClass *class)
{
	REDUCE1_ARGS(class, "hshrink", "Hshrink")
}

static void
vips_reducev_hip_args(VipsReducevHipClass *class)
{
	REDUCE1_ARGS(class, "vshrink", "Vshrink")
}

static void
vips_reduceh_hip_init(VipsReducehHip *r)
{
	r->gap = 0.0;
	r->kernel = VIPS_KERNEL_LANCZOS3;
}

static void
vips_reducev_hip_init(VipsReducevHip *r)
{
	r->gap = 0.0;
	r->kernel = VIPS_KERNEL_LANCZOS3;
}

/* shrink_hip: resample/shrink.c:77-172 */
typedef struct _VipsShrinkHip {
	VipsHipOp parent_instance;
	double hshrink, vshrink;
	gboolean ceil;
} VipsShrinkHip;

static int
vips_shrink_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsShrinkHip *shrink = (VipsShrinkHip *) op;

	return vips_hip_shrink(in, out, shrink->hshrink, shrink->vshrink, shrink->ceil);
}

HIP_SUBCLASS(VipsShrinkHip, vips_shrink_hip, "shrink_hip", "shrink an image (MI355X)")

static void
vips_shrink_hip_args(VipsShrinkHipClass *class)
{
	VIPS_ARG_DOUBLE(class, "hshrink", 8, "Hshrink", "Horizontal shrink factor",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsShrinkHip, hshrink), 1.0, 1000000.0, 1.0);
	VIPS_ARG_DOUBLE(class, "vshrink", 9, "Vshrink", "Vertical shrink factor",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsShrinkHip, vshrink), 1.0, 1000000.0, 1.0);
	VIPS_ARG_BOOL(class, "ceil", 10, "Ceil", "Round-up output dimensions",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsShrinkHip, ceil), FALSE);
}

static void
vips_shrink_hip_init(VipsShrinkHip *shrink)
{
}

/* shrinkh_hip / shrinkv_hip: resample/shrinkh.c:442-480, shrinkv.c:622-660 */
typedef struct _VipsShrink1Hip {
	VipsHipOp parent_instance;
	int shrink;
	gboolean ceil;
} VipsShrink1Hip;

typedef VipsShrink1Hip VipsShrinkhHip;
typedef VipsShrink1Hip VipsShrinkvHip;

static int
vips_shrinkh_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsShrink1Hip *s = (VipsShrink1Hip *) op;

	return vips_hip_shrinkh(in, out, s->shrink, s->ceil);
}

static int
vips_shrinkv_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsShrink1Hip *s = (VipsShrink1Hip *) op;

	return vips_hip_shrinkv(in, out, s->shrink, s->ceil);
}

HIP_SUBCLASS(VipsShrinkhHip, vips_shrinkh_hip, "shrinkh_hip", "shrink an image horizontally (MI355X)")
HIP_SUBCLASS(VipsShrinkvHip, vips_shrinkv_hip, "shrinkv_hip", "shrink an image vertically (MI355X)")

#define SHRINK1_ARGS(class, NAME, LONG) \
	VIPS_ARG_INT(class, NAME, 8, LONG, LONG " shrink factor", \
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsShrink1Hip, shrink), 1, 1000000, 1); \
	VIPS_ARG_BOOL(class, "ceil", 10, "Ceil", "Round-up output dimensions", \
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsShrink1Hip, ceil), FALSE);

static void
vips_shrinkh_hip_args(VipsShrinkhHipClass *class)
{
	SHRINK1_ARGS(class, "hshrink", "Hshrink")
}

static void
vips_shrinkv_hip_args(VipsShrinkvHipClass *class)
{
	SHRINK1_ARGS(class, "vshrink", "Vshrink")
}

static void
vips_shrinkh_hip_init(VipsShrinkhHip *s)
{
	s->shrink = 1;
}

static void
vips_shrinkv_hip_init(VipsShrinkvHip *s)
{
	s->shrink = 1;
}

/* resize_hip: resample/resize.c:331-420 (downsizing half) */
typedef struct _VipsResizeHip {
	VipsHipOp parent_instance;
	double scale, vscale, gap;
	VipsKernel kernel;
} VipsResizeHip;

static int
vips_resize_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsResizeHip *resize = (VipsResizeHip *) op;
	double vscale = vips_object_argument_isset(VIPS_OBJECT(op), "vscale") ? resize->vscale : -1.0;

	return vips_hip_resize(in, out, resize->scale, vscale, resize->kernel, resize->gap);
}

HIP_SUBCLASS(VipsResizeHip, vips_resize_hip, "resize_hip", "resize an image (MI355X)")

static void
vips_resize_hip_args(VipsResizeHipClass *class)
{
	VIPS_ARG_DOUBLE(class, "scale", 113, "Scale factor", "Scale image by this factor",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsResizeHip, scale), 0.0, 10000000.0, 0.0);
	VIPS_ARG_DOUBLE(class, "vscale", 113, "Vertical scale factor", "Vertical scale image by this factor",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsResizeHip, vscale), 0.0, 10000000.0, 0.0);
	VIPS_ARG_ENUM(class, "kernel", 3, "Kernel", "Resampling kernel",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsResizeHip, kernel),
		VIPS_TYPE_KERNEL, VIPS_KERNEL_LANCZOS3);
	VIPS_ARG_DOUBLE(class, "gap", 4, "Gap", "Reducing gap",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsResizeHip, gap), 0.0, 1000000.0, 2.0);
}

static void
vips_resize_hip_init(VipsResizeHip *resize)
{
	resize->gap = 2.0;
	resize->kernel = VIPS_KERNEL_LANCZOS3;
}

/* thumbnail_image_hip: resample/thumbnail.c:1690-1760 (vips_thumbnail_image) */
typedef struct _VipsThumbnailHip {
	VipsHipOp parent_instance;
	int width, height;
	VipsSize size;
	gboolean linear;
	VipsInteresting crop;
} VipsThumbnailHip;

static int
vips_thumbnail_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsThumbnailHip *thumbnail = (VipsThumbnailHip *) op;
	int height = vips_object_argument_isset(VIPS_OBJECT(op), "height") ? thumbnail->height : 0;

	return vips_hip_thumbnail_image_crop(in, out, thumbnail->width, height, thumbnail->size,
		thumbnail->linear, thumbnail->crop);
}

HIP_SUBCLASS(VipsThumbnailHip, vips_thumbnail_hip, "thumbnail_image_hip",
	"generate thumbnail from image (MI355X)")

static void
vips_thumbnail_hip_args(VipsThumbnailHipClass *class)
{
	VIPS_ARG_INT(class, "width", 3, "Target width", "Size to this width",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsThumbnailHip, width), 1, VIPS_MAX_COORD, 1);
	VIPS_ARG_INT(class, "height", 113, "Target height", "Size to this height",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsThumbnailHip, height), 1, VIPS_MAX_COORD, 1);
	VIPS_ARG_ENUM(class, "size", 114, "Size", "Only upsize, only downsize, or both",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsThumbnailHip, size),
		VIPS_TYPE_SIZE, VIPS_SIZE_BOTH);
	VIPS_ARG_BOOL(class, "linear", 118, "Linear", "Reduce in linear light",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsThumbnailHip, linear), FALSE);
	VIPS_ARG_ENUM(class, "crop", 116, "Crop", "Reduce to fill target rectangle, then crop",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsThumbnailHip, crop),
		VIPS_TYPE_INTERESTING, VIPS_INTERESTING_NONE);
}

static void
vips_thumbnail_hip_init(VipsThumbnailHip *thumbnail)
{
	thumbnail->width = 1;
	thumbnail->height = 1;
	thumbnail->size = VIPS_SIZE_BOTH;
	thumbnail->crop = VIPS_INTERESTING_NONE;
}

/* thumbnail_hip: vips_thumbnail() on a file (resample/thumbnail.c:1130-1330, the
 * VipsThumbnailFile class): JPEG shrink-on-load on the host, everything after it on the
 * device.  No input image, so this one is a VipsOperation of its own; it serves its result
 * the way VipsHipOp does.
 */
typedef struct _VipsThumbnailFileHip {
	VipsOperation parent_instance;

	char *filename;
	VipsImage *out;
	int width, height;
	VipsSize size;
	gboolean linear;
	VipsInteresting crop;

	VipsHipImage *result;
	VipsPel *host;
	GMutex lock;
} VipsThumbnailFileHip;

typedef VipsOperationClass VipsThumbnailFileHipClass;

G_DEFINE_TYPE(VipsThumbnailFileHip, vips_thumbnail_file_hip, VIPS_TYPE_OPERATION);

static int
vips_thumbnail_file_hip_gen(VipsRegion *out_region, void *seq, void *a, void *b, gboolean *stop)
{
	VipsThumbnailFileHip *thumbnail = (VipsThumbnailFileHip *) b;
	VipsRect *r = &out_region->valid;
	VipsImage *out = out_region->im;
	const size_t ps = VIPS_IMAGE_SIZEOF_PEL(out);
	const size_t ls = VIPS_IMAGE_SIZEOF_LINE(out);

	if (vips_image_iskilled(out))
		return -1;

	g_mutex_lock(&thumbnail->lock);
	if (!thumbnail->host) {
		VipsPel *host = VIPS_ARRAY(NULL, ls * out->Ysize, VipsPel);

		if (!host || vips_hip_image_write_to_memory(thumbnail->result, host)) {
			g_mutex_unlock(&thumbnail->lock);
			VIPS_FREE(host);
			return hip_fail("thumbnail_hip");
		}
		thumbnail->host = host;
	}
	g_mutex_unlock(&thumbnail->lock);

	for (int y = 0; y < r->height; y++)
		memcpy(VIPS_REGION_ADDR(out_region, r->left, r->top + y),
			thumbnail->host + (size_t) (r->top + y) * ls + (size_t) r->left * ps,
			(size_t) r->width * ps);

	return 0;
}

static VipsHipImage *
vips_thumbnail_file_hip_device(GObject *producer)
{
	return ((VipsThumbnailFileHip *) producer)->result;
}

static int
vips_thumbnail_file_hip_build(VipsObject *object)
{
	VipsThumbnailFileHip *thumbnail = (VipsThumbnailFileHip *) object;
	int height = vips_object_argument_isset(object, "height") ? thumbnail->height : 0;

	if (VIPS_OBJECT_CLASS(vips_thumbnail_file_hip_parent_class)->build(object))
		return -1;

	if (vips_hip_thumbnail(thumbnail->filename, &thumbnail->result, thumbnail->width, height,
			thumbnail->size, thumbnail->linear, thumbnail->crop) ||
		vips_hip_synchronize())
		return hip_fail("thumbnail_hip");

	g_object_set(object, "out", vips_image_new(), NULL);
	vips_image_init_fields(thumbnail->out,
		vips_hip_image_get_width(thumbnail->result), vips_hip_image_get_height(thumbnail->result),
		vips_hip_image_get_bands(thumbnail->result),
		(VipsBandFormat) vips_hip_image_get_format(thumbnail->result), VIPS_CODING_NONE,
		(VipsInterpretation) vips_hip_image_get_interpretation(thumbnail->result), 1.0, 1.0);
	if (vips_image_pipelinev(thumbnail->out, VIPS_DEMAND_STYLE_ANY, NULL) ||
		vips_image_generate(thumbnail->out,
			vips_hip_op_start, vips_thumbnail_file_hip_gen, vips_hip_op_stop, NULL, thumbnail))
		return -1;
	hip_link_attach(thumbnail->out, G_OBJECT(thumbnail), vips_thumbnail_file_hip_device);

	return 0;
}

static void
vips_thumbnail_file_hip_dispose(GObject *gobject)
{
	VipsThumbnailFileHip *thumbnail = (VipsThumbnailFileHip *) gobject;

	VIPS_FREE(thumbnail->host);
	if (thumbnail->result) {
		vips_hip_image_unref(thumbnail->result);
		thumbnail->result = NULL;
	}

	G_OBJECT_CLASS(vips_thumbnail_file_hip_parent_class)->dispose(gobject);
}

static void
vips_thumbnail_file_hip_class_init(VipsThumbnailFileHipClass *class)
{
	GObjectClass *gobject_class = G_OBJECT_CLASS(class);
	VipsObjectClass *vobject_class = VIPS_OBJECT_CLASS(class);

	gobject_class->dispose = vips_thumbnail_file_hip_dispose;
	gobject_class->set_property = vips_object_set_property;
	gobject_class->get_property = vips_object_get_property;

	vobject_class->nickname = "thumbnail_hip";
	vobject_class->description = "generate thumbnail from file (MI355X)";
	vobject_class->build = vips_thumbnail_file_hip_build;

	VIPS_ARG_STRING(class, "filename", 1, "Filename", "Filename to read from",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsThumbnailFileHip, filename), NULL);
	VIPS_ARG_IMAGE(class, "out", 2, "Output", "Output image",
		VIPS_ARGUMENT_REQUIRED_OUTPUT, G_STRUCT_OFFSET(VipsThumbnailFileHip, out));
	VIPS_ARG_INT(class, "width", 3, "Target width", "Size to this width",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsThumbnailFileHip, width), 1, VIPS_MAX_COORD, 1);
	VIPS_ARG_INT(class, "height", 113, "Target height", "Size to this height",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsThumbnailFileHip, height), 1, VIPS_MAX_COORD, 1);
	VIPS_ARG_ENUM(class, "size", 114, "Size", "Only upsize, only downsize, or both",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsThumbnailFileHip, size),
		VIPS_TYPE_SIZE, VIPS_SIZE_BOTH);
	VIPS_ARG_ENUM(class, "crop", 116, "Crop", "Reduce to fill target rectangle, then crop",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsThumbnailFileHip, crop),
		VIPS_TYPE_INTERESTING, VIPS_INTERESTING_NONE);
	VIPS_ARG_BOOL(class, "linear", 118, "Linear", "Reduce in linear light",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsThumbnailFileHip, linear), FALSE);
}

static void
vips_thumbnail_file_hip_init(VipsThumbnailFileHip *thumbnail)
{
	thumbnail->width = 1;
	thumbnail->height = 1;
	thumbnail->size = VIPS_SIZE_BOTH;
	thumbnail->crop = VIPS_INTERESTING_NONE;
	g_mutex_init(&thumbnail->lock);
}

/* conv_hip / convsep_hip: convolution/conv.c:120-175, convsep.c:120-170 */
typedef struct _VipsConvHip {
	VipsHipOp parent_instance;
	VipsImage *mask;
	VipsPrecision precision;
	int layers;
	int cluster;
} VipsConvHip;

typedef VipsConvHip VipsConvsepHip;

static int
vips_conv_hip_run(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out, gboolean separable)
{
	VipsConvHip *conv = (VipsConvHip *) op;
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;
	VipsImage *M;
	int result;

	if (vips_check_matrix(nick, conv->mask, &M)) {
		vips_hip_error_clear();
		return -1;
	}
	if (separable) {
		if (vips_check_separable(nick, M)) {
			g_object_unref(M);
			return -1;
		}
		/* convsep.c:81-87: approximate goes to vips_convasep with the layers argument */
		if (conv->precision == VIPS_PRECISION_APPROXIMATE)
			result = vips_hip_convasep(in, out, VIPS_MATRIX(M, 0, 0), M->Xsize * M->Ysize,
				vips_image_get_scale(M), vips_image_get_offset(M), conv->layers);
		else
			result = vips_hip_convsep(in, out, VIPS_MATRIX(M, 0, 0), M->Xsize * M->Ysize,
				vips_image_get_scale(M), vips_image_get_offset(M), conv->precision);
	}
	/* conv.c:99-107 */
	else if (conv->precision == VIPS_PRECISION_APPROXIMATE)
		result = vips_hip_conva(in, out, VIPS_MATRIX(M, 0, 0), M->Xsize, M->Ysize,
			vips_image_get_scale(M), vips_image_get_offset(M), conv->layers, conv->cluster);
	else
		result = vips_hip_conv(in, out, VIPS_MATRIX(M, 0, 0), M->Xsize, M->Ysize,
			vips_image_get_scale(M), vips_image_get_offset(M), conv->precision);
	g_object_unref(M);

	return result;
}

static int
vips_conv_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	return vips_conv_hip_run(op, in, out, FALSE);
}

static int
vips_convsep_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	return vips_conv_hip_run(op, in, out, TRUE);
}

/* The region form of conv_hip (integer and float precision): one plan, vips_hip_conv_gen per
 * strip; a strip reads mask_height / 2 rows above and the rest below (convi.c:778-782). */
typedef struct _ConvStrip {
	VipsHipConv *conv;
	int mask_height;
} ConvStrip;

static void
vips_conv_hip_strip_close(VipsHipOp *op, void *plan)
{
	ConvStrip *p = (ConvStrip *) plan;

	if (p) {
		vips_hip_conv_free(p->conv);
		g_free(p);
	}
}

static int
vips_conv_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsConvHip *conv = (VipsConvHip *) op;
	ConvStrip *p;
	VipsImage *M;

	if (conv->precision == VIPS_PRECISION_APPROXIMATE)
		return 1;
	if (vips_check_matrix("conv_hip", conv->mask, &M))
		return -1;
	p = g_new0(ConvStrip, 1);
	p->mask_height = M->Ysize;
	p->conv = vips_hip_conv_new(VIPS_MATRIX(M, 0, 0), M->Xsize, M->Ysize,
		vips_image_get_scale(M), vips_image_get_offset(M), conv->precision);
	g_object_unref(M);
	if (!p->conv) {
		g_free(p);
		return hip_fail("conv_hip");
	}
	*plan = p;

	return 0;
}

static void
vips_conv_hip_strip_need(VipsHipOp *op, void *plan, int out_top, int out_rows, int *in_top, int *in_rows)
{
	const int mask_height = ((ConvStrip *) plan)->mask_height;

	*in_top = out_top - mask_height / 2;
	*in_rows = out_rows + mask_height - 1;
}

static int
vips_conv_hip_strip_run(VipsHipOp *op, void *plan, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return vips_hip_conv_gen(((ConvStrip *) plan)->conv, in, out);
}

HIP_SUBCLASS_FULL(VipsConvHip, vips_conv_hip, "conv_hip", "convolution operation (MI355X)",
	HIP_STRIPS(vips_conv_hip))
HIP_SUBCLASS(VipsConvsepHip, vips_convsep_hip, "convsep_hip", "separable convolution operation (MI355X)")

#define CONV_ARGS(class) \
	VIPS_ARG_IMAGE(class, "mask", 20, "Mask", "Input matrix image", \
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsConvHip, mask)); \
	VIPS_ARG_ENUM(class, "precision", 103, "Precision", "Convolve with this precision", \
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsConvHip, precision), \
		VIPS_TYPE_PRECISION, VIPS_PRECISION_FLOAT); \
	VIPS_ARG_INT(class, "layers", 104, "Layers", "Use this many layers in approximation", \
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsConvHip, layers), 1, 1000, 5); \
	VIPS_ARG_INT(class, "cluster", 105, "Cluster", "Cluster lines closer than this in approximation", \
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsConvHip, cluster), 1, 100, 1);

static void
vips_conv_hip_args(VipsConvHipClass *class)
{
	CONV_ARGS(class)
}

static void
vips_convsep_hip_args(VipsConvsepHipClass *class)
{
	CONV_ARGS(class)
}

static void
vips_conv_hip_init(VipsConvHip *conv)
{
	conv->precision = VIPS_PRECISION_FLOAT;
	conv->layers = 5;
	conv->cluster = 1;
}

static void
vips_convsep_hip_init(VipsConvsepHip *conv)
{
	conv->precision = VIPS_PRECISION_FLOAT;
	conv->layers = 5;
	conv->cluster = 1;
}

/* gaussblur_hip: convolution/gaussblur.c:118-175 */
typedef struct _VipsGaussblurHip {
	VipsHipOp parent_instance;
	double sigma, min_ampl;
	VipsPrecision precision;
} VipsGaussblurHip;

static int
vips_gaussblur_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsGaussblurHip *g = (VipsGaussblurHip *) op;

	return vips_hip_gaussblur(in, out, g->sigma, g->min_ampl, g->precision);
}

HIP_SUBCLASS(VipsGaussblurHip, vips_gaussblur_hip, "gaussblur_hip", "gaussian blur (MI355X)")

static void
vips_gaussblur_hip_args(VipsGaussblurHipClass *class)
{
	VIPS_ARG_DOUBLE(class, "sigma", 3, "Sigma", "Sigma of Gaussian",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsGaussblurHip, sigma), 0.0, 1000, 1.5);
	VIPS_ARG_DOUBLE(class, "min_ampl", 3, "Minimum amplitude", "Minimum amplitude of Gaussian",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsGaussblurHip, min_ampl), 0.001, 1.0, 0.2);
	VIPS_ARG_ENUM(class, "precision", 4, "Precision", "Convolve with this precision",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsGaussblurHip, precision),
		VIPS_TYPE_PRECISION, VIPS_PRECISION_INTEGER);
}

static void
vips_gaussblur_hip_init(VipsGaussblurHip *g)
{
	g->sigma = 1.5;
	g->min_ampl = 0.2;
	g->precision = VIPS_PRECISION_INTEGER;
}

/* sharpen_hip: convolution/sharpen.c:304-395 */
typedef struct _VipsSharpenHip {
	VipsHipOp parent_instance;
	double sigma, x1, y2, y3, m1, m2;
} VipsSharpenHip;

static int
vips_sharpen_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsSharpenHip *s = (VipsSharpenHip *) op;

	return vips_hip_sharpen(in, out, s->sigma, s->x1, s->y2, s->y3, s->m1, s->m2);
}

HIP_SUBCLASS(VipsSharpenHip, vips_sharpen_hip, "sharpen_hip", "unsharp masking for print (MI355X)")

static void
vips_sharpen_hip_args(VipsSharpenHipClass *class)
{
	VIPS_ARG_DOUBLE(class, "sigma", 3, "Sigma", "Sigma of Gaussian",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsSharpenHip, sigma), 0.000001, 10.0, 0.5);
	VIPS_ARG_DOUBLE(class, "x1", 5, "x1", "Flat/jaggy threshold",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsSharpenHip, x1), 0, 1000000, 2.0);
	VIPS_ARG_DOUBLE(class, "y2", 6, "y2", "Maximum brightening",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsSharpenHip, y2), 0, 1000000, 10.0);
	VIPS_ARG_DOUBLE(class, "y3", 7, "y3", "Maximum darkening",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsSharpenHip, y3), 0, 1000000, 20.0);
	VIPS_ARG_DOUBLE(class, "m1", 8, "m1", "Slope for flat areas",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsSharpenHip, m1), 0, 1000000, 0.0);
	VIPS_ARG_DOUBLE(class, "m2", 9, "m2", "Slope for jaggy areas",
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsSharpenHip, m2), 0, 1000000, 3.0);
}

static void
vips_sharpen_hip_init(VipsSharpenHip *s)
{
	s->sigma = 0.5;
	s->x1 = 2.0;
	s->y2 = 10.0;
	s->y3 = 20.0;
	s->m1 = 0.0;
	s->m2 = 3.0;
}

/* colourspace_hip: colour/colourspace.c:614-650 */
typedef struct _VipsColourspaceHip {
	VipsHipOp parent_instance;
	VipsInterpretation space;
} VipsColourspaceHip;

static int
vips_colourspace_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsColourspaceHip *c = (VipsColourspaceHip *) op;

	return vips_hip_colourspace(in, out, c->space);
}

/* gaussblur_hip -> colourspace_hip as one call: vips_hip_gaussblur_colourspace runs both blur
 * passes and the colour route in one kernel on 3-band float images (BASELINE config 3), and
 * the two operations otherwise. */
static int
vips_colourspace_hip_fuse(VipsHipOp *op, VipsHipOp *up, VipsHipImage *up_in, VipsHipImage **out)
{
	VipsColourspaceHip *c = (VipsColourspaceHip *) op;
	VipsGaussblurHip *g;

	if (!G_TYPE_CHECK_INSTANCE_TYPE(up, vips_gaussblur_hip_get_type()))
		return 1;
	g = (VipsGaussblurHip *) up;

	return vips_hip_gaussblur_colourspace(up_in, out, g->sigma, g->min_ampl, g->precision, c->space);
}

HIP_SUBCLASS_FULL(VipsColourspaceHip, vips_colourspace_hip, "colourspace_hip",
	"convert to a new colorspace (MI355X)", class->fuse = vips_colourspace_hip_fuse;)

static void
vips_colourspace_hip_args(VipsColourspaceHipClass *class)
{
	VIPS_ARG_ENUM(class, "space", 6, "Space", "Destination color space",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsColourspaceHip, space),
		VIPS_TYPE_INTERPRETATION, VIPS_INTERPRETATION_sRGB);
}

static void
vips_colourspace_hip_init(VipsColourspaceHip *c)
{
	c->space = VIPS_INTERPRETATION_sRGB;
}

/* cast_hip: conversion/cast.c:470-520 */
typedef struct _VipsCastHip {
	VipsHipOp parent_instance;
	VipsBandFormat format;
} VipsCastHip;

static int
vips_cast_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	VipsCastHip *c = (VipsCastHip *) op;

	return vips_hip_cast(in, out, c->format);
}

HIP_SUBCLASS(VipsCastHip, vips_cast_hip, "cast_hip", "cast an image (MI355X)")

static void
vips_cast_hip_args(VipsCastHipClass *class)
{
	VIPS_ARG_ENUM(class, "format", 6, "Format", "Format to cast to",
		VIPS_ARGUMENT_REQUIRED_INPUT, G_STRUCT_OFFSET(VipsCastHip, format),
		VIPS_TYPE_BAND_FORMAT, VIPS_FORMAT_UCHAR);
}

static void
vips_cast_hip_init(VipsCastHip *c)
{
	c->format = VIPS_FORMAT_UCHAR;
}

/* premultiply_hip / unpremultiply_hip: conversion/premultiply.c:273-330, unpremultiply.c:340-400 */
typedef struct _VipsPremultiplyHip {
	VipsHipOp parent_instance;
	gboolean uchar;
} VipsPremultiplyHip;

typedef VipsPremultiplyHip VipsUnpremultiplyHip;

static int
vips_premultiply_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	return vips_hip_premultiply(in, out, ((VipsPremultiplyHip *) op)->uchar);
}

static int
vips_unpremultiply_hip_compute(VipsHipOp *op, VipsHipImage *in, VipsHipImage **out)
{
	return vips_hip_unpremultiply(in, out, ((VipsPremultiplyHip *) op)->uchar);
}

HIP_SUBCLASS(VipsPremultiplyHip, vips_premultiply_hip, "premultiply_hip", "premultiply image alpha (MI355X)")
HIP_SUBCLASS(VipsUnpremultiplyHip, vips_unpremultiply_hip, "unpremultiply_hip",
	"unpremultiply image alpha (MI355X)")

#define PREMUL_ARGS(class) \
	VIPS_ARG_BOOL(class, "uchar", 116, "Uchar", "Use the uchar fast path", \
		VIPS_ARGUMENT_OPTIONAL_INPUT, G_STRUCT_OFFSET(VipsPremultiplyHip, uchar), FALSE);

static void
vips_premultiply_hip_args(VipsPremultiplyHipClass *class)
{
	PREMUL_ARGS(class)
}

static void
vips_unpremultiply_hip_args(VipsUnpremultiplyHipClass *class)
{
	PREMUL_ARGS(class)
}

static void
vips_premultiply_hip_init(VipsPremultiplyHip *p)
{
}

static void
vips_unpremultiply_hip_init(VipsUnpremultiplyHip *p)
{
}

/* ------------------------------------------------------------------ registration */

/* Register every class. Called by GModule when libvips (or the test shim) opens the
 * module, the same entry point libvips/module/heif.c:53-78 uses.
 */
G_MODULE_EXPORT const gchar *
g_module_check_init(GModule *module)
{
	vips_reduce_hip_get_type();
	vips_reduceh_hip_get_type();
	vips_reducev_hip_get_type();
	vips_shrink_hip_get_type();
	vips_shrinkh_hip_get_type();
	vips_shrinkv_hip_get_type();
	vips_resize_hip_get_type();
	vips_thumbnail_hip_get_type();
	vips_thumbnail_file_hip_get_type();
	vips_conv_hip_get_type();
	vips_convsep_hip_get_type();
	vips_gaussblur_hip_get_type();
	vips_sharpen_hip_get_type();
	vips_colourspace_hip_get_type();
	vips_cast_hip_get_type();
	vips_premultiply_hip_get_type();
	vips_unpremultiply_hip_get_type();

	/* types registered by a module must never be unloaded */
	g_module_make_resident(module);

	return NULL;
}
