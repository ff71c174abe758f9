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
#include <stdio.h>
#include <string.h>

#include <vips/vips.h>
#include <vips/vector.h>

#include "vips_hip.h"

#define HIP_META "vips-hip-image"

/* ------------------------------------------------------------------ device link */

/* The device-resident twin of a VipsImage, attached to it as a VipsArea. */
typedef struct _HipLink {
	VipsHipImage *dev;
	VipsImage *owner; /* metadata is copied down pipelines: only valid on its owner */
} HipLink;

static int
hip_link_free(void *data, void *unused)
{
	HipLink *link = (HipLink *) data;

	vips_hip_image_unref(link->dev);
	g_free(link);

	return 0;
}

static void
hip_link_attach(VipsImage *image, VipsHipImage *dev)
{
	HipLink *link = g_new(HipLink, 1);

	link->dev = dev;
	link->owner = image;
	vips_image_set_area(image, HIP_META, (VipsCallbackFn) hip_link_free, link);
}

static VipsHipImage *
hip_link_find(VipsImage *image)
{
	const void *data;

	if (vips_image_get_typeof(image, HIP_META) &&
		!vips_image_get_area(image, HIP_META, &data)) {
		const HipLink *link = (const HipLink *) data;

		if (link->owner == image)
			return link->dev;
	}

	return NULL;
}

static int
hip_fail(const char *domain)
{
	vips_error(domain, "%s", vips_hip_error_buffer());
	vips_hip_error_clear();

	return -1;
}

/* ------------------------------------------------------------------ base class */

typedef struct _VipsHipOp {
	VipsOperation parent_instance;

	VipsImage *in;
	VipsImage *out;

	/* The result: on the device, and (lazily) on the host for generate. */
	VipsHipImage *result;
	VipsPel *host;
	GMutex lock;
} VipsHipOp;

typedef struct _VipsHipOpClass {
	VipsOperationClass parent_class;

	/* Run the operation on a device-resident image. */
	int (*compute)(struct _VipsHipOp *op, VipsHipImage *in, VipsHipImage **out);
} VipsHipOpClass;

#define VIPS_TYPE_HIP_OP (vips_hip_op_get_type())
#define VIPS_HIP_OP(obj) (G_TYPE_CHECK_INSTANCE_CAST((obj), VIPS_TYPE_HIP_OP, VipsHipOp))
#define VIPS_HIP_OP_GET_CLASS(obj) (G_TYPE_INSTANCE_GET_CLASS((obj), VIPS_TYPE_HIP_OP, VipsHipOpClass))

G_DEFINE_ABSTRACT_TYPE(VipsHipOp, vips_hip_op, VIPS_TYPE_OPERATION);

/* The per-thread sequence: owns the stream this worker's copies run on. */
static void *
vips_hip_op_start(VipsImage *out, void *a, void *b)
{
	/* NULL selects the library's own per-thread stream; creating it here and dropping it
	 * in stop matches the sequence contract (one per worker, under image->sslock). */
	if (vips_hip_set_stream(NULL)) {
		hip_fail("vips_hip");
		return NULL;
	}

	return (void *) out;
}

static int
vips_hip_op_stop(void *seq, void *a, void *b)
{
	return 0;
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

	/* First demand: bring the device result to the host, once. */
	g_mutex_lock(&op->lock);
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

static int
vips_hip_op_build(VipsObject *object)
{
	VipsObjectClass *class = VIPS_OBJECT_GET_CLASS(object);
	VipsHipOp *op = VIPS_HIP_OP(object);
	VipsHipOpClass *hclass = VIPS_HIP_OP_GET_CLASS(op);
	VipsImage **t = (VipsImage **) vips_object_local_array(object, 2);

	VipsImage *in;
	VipsHipImage *dev;
	VipsHipImage *fresh = NULL;

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

	/* A device-resident input (made by another *_hip op) is used as it is; anything
	 * else is rendered to memory and uploaded once. */
	if (!(dev = hip_link_find(op->in))) {
		if (!(t[1] = vips_image_copy_memory(in)))
			return -1;
		if (!(fresh = vips_hip_image_new_from_memory(VIPS_IMAGE_ADDR(t[1], 0, 0),
				  t[1]->Xsize, t[1]->Ysize, t[1]->Bands, t[1]->BandFmt, t[1]->Type)))
			return hip_fail(class->nickname);
		dev = fresh;
	}

	/* Follow the host library's vector switch (iofuncs/vector.cpp:98-113): in a Highway-built
	 * libvips this selects the 8-bit-mantissa arithmetic its own convi uses on uchar images,
	 * so *_hip results keep matching the built-ins; FALSE in a scalar build. */
	vips_hip_vector_set_enabled(vips_vector_isenabled());

	if (hclass->compute(op, dev, &op->result)) {
		vips_hip_image_unref(fresh);
		return hip_fail(class->nickname);
	}
	/* The kernels were queued on THIS thread's stream; generate runs on libvips worker
	 * threads with their own (non-blocking) streams, and a downstream *_hip op may build on
	 * yet another thread: finish the work before anyone else can see the result, and before
	 * the uploaded input goes back to the pool. */
	if (vips_hip_synchronize()) {
		vips_hip_image_unref(fresh);
		return hip_fail(class->nickname);
	}
	vips_hip_image_unref(fresh);

	g_object_set(object, "out", vips_image_new(), NULL);
	if (vips_image_pipelinev(op->out, VIPS_DEMAND_STYLE_ANY, in, NULL))
		return -1;
	op->out->Xsize = vips_hip_image_get_width(op->result);
	op->out->Ysize = vips_hip_image_get_height(op->result);
	op->out->Bands = vips_hip_image_get_bands(op->result);
	op->out->BandFmt = (VipsBandFormat) vips_hip_image_get_format(op->result);
	op->out->Type = (VipsInterpretation) vips_hip_image_get_interpretation(op->result);

	if (vips_image_generate(op->out,
			vips_hip_op_start, vips_hip_op_gen, vips_hip_op_stop, in, op))
		return -1;

	/* Downstream *_hip ops pick the device copy up from here. The link holds its own
	 * handle on the same pixels. */
	hip_link_attach(op->out,
		vips_hip_image_new_from_device(vips_hip_image_get_data(op->result),
			op->out->Xsize, op->out->Ysize, op->out->Bands, op->out->BandFmt, op->out->Type));

	return 0;
}

static void
vips_hip_op_dispose(GObject *gobject)
{
	VipsHipOp *op = VIPS_HIP_OP(gobject);

	VIPS_FREE(op->host);
	if (op->result) {
		vips_hip_image_unref(op->result);
		op->result = NULL;
	}

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

#define HIP_SUBCLASS(TypeName, type_name, nick, desc) \
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
		type_name##_args(class); \
	}

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

HIP_SUBCLASS(VipsReduceHip, vips_reduce_hip, "reduce_hip", "reduce an image (MI355X)")

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
	hip_link_attach(thumbnail->out,
		vips_hip_image_new_from_device(vips_hip_image_get_data(thumbnail->result),
			thumbnail->out->Xsize, thumbnail->out->Ysize, thumbnail->out->Bands,
			thumbnail->out->BandFmt, thumbnail->out->Type));

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

HIP_SUBCLASS(VipsConvHip, vips_conv_hip, "conv_hip", "convolution operation (MI355X)")
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

HIP_SUBCLASS(VipsColourspaceHip, vips_colourspace_hip, "colourspace_hip",
	"convert to a new colorspace (MI355X)")

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
