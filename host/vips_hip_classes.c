/* The class table of the vips-hip module: one GObject subclass of VipsHipOp per *_hip operation -- the original
 * operation's arguments and defaults (libvips/resample, convolution, colour, conversion: each
 * class cites its original) and the hooks the evaluation engine calls (whole-image form, region form for the strip
 * loop).  Textually part of vips_hip_module.c, which holds the engine (device link, base class, host cache, strip
 * producer / ring) and the registration; split out so that neither file has to be read for the other.
 */
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

#define HIP_HALO(type_name) class->halo = type_name##_halo;

/* ---- the region form of the whole resample family
 *
 * vips_reduce / vips_resize / vips_shrink are, per axis, an optional integer box shrink (the
 * `gap` pre-shrink, reduceh.cpp:430-455 / reducev.cpp:894-917, or vips_shrink's own) and an
 * optional residual reduce, vertical axis first (reduce.c:98-121, resize.c:207-228, shrink.c:77-119):
 *     shrinkv(int_v) -> reducev(rv) -> shrinkh(int_h) -> reduceh(rh)
 * Each stage has a generate replacement in the C ABI that works in whole-image coordinates, so a
 * strip of output rows is made by walking its row range back through the vertical stages
 * (vips_hip_reducev_need, x int_v) and the four gens forward, the rows between them on the device.
 */
typedef struct _ResampleStrip {
	/* upsizing (both scales >= 1): vips_resize's scale-only affine (resize.c:230-300), one
	 * generate replacement that works in whole-image coordinates */
	gboolean upsize;
	double hscale, vscale, idx, idy;
	int interpolate;

	int int_v, int_h;
	VipsHipReduce *rv, *rh;
	int w0, h0; /* input */
	int h1;     /* rows after shrinkv */
	int h2;     /* ... after reducev = output rows */
	int w1;     /* columns after shrinkh */
	int w2;     /* ... after reduceh = output columns */
} ResampleStrip;

/* one axis of vips_reduceh_build / vips_reducev_build: the output size, the integer pre-shrink
 * `gap` buys and the residual factor (reduceh.cpp:396-481, reducev.cpp:859-941) */
static int
resample_axis(const char *nick, int in_size, double shrink, VipsKernel kernel, double gap,
	int *int_shrink, int *shrunk_size, VipsHipReduce **reduce, int *out_size)
{
	int size = (int) ((double) in_size / shrink + 0.5);
	double extra = size * shrink - in_size;
	double residual = shrink;

	*int_shrink = 1;
	*shrunk_size = in_size;
	*reduce = NULL;
	if (size <= 0) {
		vips_error(nick, "%s", "image has shrunk to nothing");
		return -1;
	}
	if (gap > 0.0 && kernel != VIPS_KERNEL_NEAREST) {
		const int k = (int) floor((double) in_size / size / gap);

		if (k > 1) {
			*int_shrink = k;
			residual /= k;
			extra /= k;
			*shrunk_size = vips_hip_shrink_out_size(in_size, k, 1); /* "ceil", TRUE */
		}
	}
	*out_size = residual == 1.0 ? *shrunk_size : size;
	if (residual != 1.0 &&
		!(*reduce = vips_hip_reduce_new(kernel, residual, *shrunk_size, size, extra)))
		return hip_fail(nick);

	return 0;
}

static void
resample_strip_close(VipsHipOp *op, void *plan)
{
	ResampleStrip *p = (ResampleStrip *) plan;

	if (p) {
		vips_hip_reduce_free(p->rv);
		vips_hip_reduce_free(p->rh);
		g_free(p);
	}
}

/* vshrink / hshrink >= 1: the factors of the two axes (1 = untouched); int_only: box shrinks of
 * exactly these (integer) factors, rounding up when ceil is set, no reduce */
static int
resample_strip_open(VipsHipOp *op, VipsImage *in, double vshrink, double hshrink, VipsKernel kernel, double gap,
	gboolean int_only, gboolean ceil, void **plan)
{
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;
	ResampleStrip *p = g_new0(ResampleStrip, 1);

	p->w0 = in->Xsize;
	p->h0 = in->Ysize;
	p->int_v = p->int_h = 1;
	p->h1 = p->h2 = p->h0;
	p->w1 = p->w2 = p->w0;
	if (int_only) {
		p->int_v = (int) vshrink;
		p->int_h = (int) hshrink;
		p->h1 = p->h2 = p->int_v > 1 ? vips_hip_shrink_out_size(p->h0, p->int_v, ceil) : p->h0;
		p->w1 = p->w2 = p->int_h > 1 ? vips_hip_shrink_out_size(p->w0, p->int_h, ceil) : p->w0;
	}
	else if ((vshrink != 1.0 && resample_axis(nick, p->h0, vshrink, kernel, gap, &p->int_v, &p->h1, &p->rv, &p->h2)) ||
		(hshrink != 1.0 && resample_axis(nick, p->w0, hshrink, kernel, gap, &p->int_h, &p->w1, &p->rh, &p->w2))) {
		resample_strip_close(op, p);
		return -1;
	}
	if (p->h2 != op->out->Ysize || p->w2 != op->out->Xsize || p->h1 <= 0 || p->w1 <= 0) {
		/* not the decomposition the original operation's header came from: whole image only */
		resample_strip_close(op, p);
		return 1;
	}
	*plan = p;

	return 0;
}

static void
resample_strip_need(VipsHipOp *op, void *plan, int out_top, int out_rows, int *in_top, int *in_rows)
{
	ResampleStrip *p = (ResampleStrip *) plan;
	int top = out_top, rows = out_rows;

	if (p->upsize) {
		/* output row y reads input rows around y / vscale: the bicubic stencil (4 rows) and the
		 * centre-sampling displacement lie well inside a margin of 4 rows either side */
		*in_top = (int) floor(out_top / p->vscale) - 4;
		*in_rows = (int) ceil((out_top + out_rows) / p->vscale) + 4 - *in_top;
		return;
	}
	if (p->rv)
		vips_hip_reducev_need(p->rv, out_top, out_rows, &top, &rows);
	*in_top = top * p->int_v;
	*in_rows = rows * p->int_v;
}

static int
resample_strip_run(VipsHipOp *op, void *plan, const VipsHipRegion *in, const VipsHipRegion *out)
{
	ResampleStrip *p = (ResampleStrip *) plan;
	VipsHipImage *tmp[3] = { NULL, NULL, NULL };
	VipsHipRegion cur = *in, next;
	int n = 0, result = 0;
	int top1 = out->top, rows1 = out->height;

	if (p->upsize)
		return vips_hip_upsize_gen(in, out, p->hscale, p->vscale, p->idx, p->idy, p->interpolate, 0)
			? hip_fail(VIPS_OBJECT_GET_CLASS(op)->nickname)
			: 0;
	if (p->rv)
		vips_hip_reducev_need(p->rv, out->top, out->height, &top1, &rows1);

	/* a stage writes into a device image of its own unless it is the last one, which writes `out` */
#define STAGE(LAST, WIDTH, TOP, ROWS, IM_W, IM_H, CALL) \
	do { \
		if (LAST) \
			next = *out; \
		else { \
			if (!(tmp[n] = vips_hip_image_new((WIDTH), (ROWS), in->bands, in->format, 0))) { \
				result = -1; \
				break; \
			} \
			vips_hip_image_region(tmp[n], &next); \
			next.top = (TOP); \
			next.im_width = (IM_W); \
			next.im_height = (IM_H); \
			n++; \
		} \
		if (CALL) \
			result = -1; \
		cur = next; \
	} while (0)

	if (!result && p->int_v > 1)
		STAGE(!p->rv && p->int_h == 1 && !p->rh, p->w0, top1, rows1, p->w0, p->h1,
			vips_hip_shrinkv_gen(p->int_v, &cur, &next));
	/* 16: the fat-strip height the reference's sink evaluates reducev in (thread.c:301-325), what
	 * the whole-image path uses (strips are multiples of 16 lines) */
	if (!result && p->rv)
		STAGE(p->int_h == 1 && !p->rh, p->w0, out->top, out->height, p->w0, p->h2,
			vips_hip_reducev_gen_tiled(p->rv, &cur, &next, 16));
	if (!result && p->int_h > 1)
		STAGE(!p->rh, p->w1, out->top, out->height, p->w1, p->h2, vips_hip_shrinkh_gen(p->int_h, &cur, &next));
	if (!result && p->rh)
		STAGE(TRUE, p->w2, out->top, out->height, p->w2, p->h2, vips_hip_reduceh_gen(p->rh, &cur, &next));
#undef STAGE
	/* (the pool orders reuse of these blocks behind the kernels: same thread, same stream) */
	for (int i = 0; i < 3; i++)
		vips_hip_image_unref(tmp[i]);

	return result ? hip_fail(VIPS_OBJECT_GET_CLASS(op)->nickname) : 0;
}

#define HIP_RESAMPLE_STRIPS(type_name) \
	class->strip_open = type_name##_strip_open; \
	class->strip_need = resample_strip_need; \
	class->strip_run = resample_strip_run; \
	class->strip_close = resample_strip_close;

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

/* The region form (images over the HBM budget): RGBA uchar with an even integer factor and no
 * pre-shrink takes the fused kernel per strip, everything else the chain of generate
 * replacements. */
static int
vips_reduce_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsReduceHip *reduce = (VipsReduceHip *) op;

	if (reduce->kernel == VIPS_KERNEL_NEAREST)
		return 1;
	return resample_strip_open(op, in, reduce->vshrink, reduce->hshrink, reduce->kernel, reduce->gap, FALSE, FALSE,
		plan);
}

static int
vips_reduce_hip_strip_run(VipsHipOp *op, void *plan, const VipsHipRegion *in, const VipsHipRegion *out)
{
	ResampleStrip *p = (ResampleStrip *) plan;

	if (p->int_v == 1 && p->int_h == 1 && p->rv && p->rh) {
		const int r = vips_hip_reduce_gen_tiled(p->rv, p->rh, in, out, 16);

		if (r <= 0)
			return r;
	}
	return resample_strip_run(op, plan, in, out);
}

#define VIPS_REDUCE_HIP_STRIPS \
	class->strip_open = vips_reduce_hip_strip_open; \
	class->strip_need = resample_strip_need; \
	class->strip_run = vips_reduce_hip_strip_run; \
	class->strip_close = resample_strip_close;

HIP_SUBCLASS_FULL(VipsReduceHip, vips_reduce_hip, "reduce_hip", "reduce an image (MI355X)",
	VIPS_REDUCE_HIP_STRIPS)

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

static int
vips_reduceh_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsReduce1Hip *r = (VipsReduce1Hip *) op;

	if (r->kernel == VIPS_KERNEL_NEAREST)
		return 1;
	return resample_strip_open(op, in, 1.0, r->shrink, r->kernel, r->gap, FALSE, FALSE, plan);
}

static int
vips_reducev_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsReduce1Hip *r = (VipsReduce1Hip *) op;

	if (r->kernel == VIPS_KERNEL_NEAREST)
		return 1;
	return resample_strip_open(op, in, r->shrink, 1.0, r->kernel, r->gap, FALSE, FALSE, plan);
}

HIP_SUBCLASS_FULL(VipsReducehHip, vips_reduceh_hip, "reduceh_hip", "shrink an image horizontally (MI355X)",
	HIP_RESAMPLE_STRIPS(vips_reduceh_hip))
HIP_SUBCLASS_FULL(VipsReducevHip, vips_reducev_hip, "reducev_hip", "shrink an image vertically (MI355X)",
	HIP_RESAMPLE_STRIPS(vips_reducev_hip))

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

/* shrink.c:77-119: integer factors are the two box shrinks; anything else is vips_reducev /
 * vips_reduceh with "gap", 1.0 */
static int
vips_shrink_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsShrinkHip *shrink = (VipsShrinkHip *) op;

	if ((int) shrink->hshrink == shrink->hshrink && (int) shrink->vshrink == shrink->vshrink)
		return resample_strip_open(op, in, shrink->vshrink, shrink->hshrink, VIPS_KERNEL_LANCZOS3, 0.0, TRUE,
			shrink->ceil, plan);
	return resample_strip_open(op, in, shrink->vshrink, shrink->hshrink, VIPS_KERNEL_LANCZOS3, 1.0, FALSE, FALSE,
		plan);
}

HIP_SUBCLASS_FULL(VipsShrinkHip, vips_shrink_hip, "shrink_hip", "shrink an image (MI355X)",
	HIP_RESAMPLE_STRIPS(vips_shrink_hip))

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

static int
vips_shrinkh_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsShrink1Hip *s = (VipsShrink1Hip *) op;

	return s->shrink == 1 ? 1 : resample_strip_open(op, in, 1.0, s->shrink, VIPS_KERNEL_LANCZOS3, 0.0, TRUE, s->ceil, plan);
}

static int
vips_shrinkv_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsShrink1Hip *s = (VipsShrink1Hip *) op;

	return s->shrink == 1 ? 1 : resample_strip_open(op, in, s->shrink, 1.0, VIPS_KERNEL_LANCZOS3, 0.0, TRUE, s->ceil, plan);
}

HIP_SUBCLASS_FULL(VipsShrinkhHip, vips_shrinkh_hip, "shrinkh_hip", "shrink an image horizontally (MI355X)",
	HIP_RESAMPLE_STRIPS(vips_shrinkh_hip))
HIP_SUBCLASS_FULL(VipsShrinkvHip, vips_shrinkv_hip, "shrinkv_hip", "shrink an image vertically (MI355X)",
	HIP_RESAMPLE_STRIPS(vips_shrinkv_hip))

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

/* Both halves of vips_resize have a region form: downsizing (resize.c:207-228: vips_reducev then
 * vips_reduceh, each with its `gap` pre-shrink) and upsizing (the scale-only vips_affine,
 * resize.c:230-300); the nearest kernel (vips_subsample, vips_zoom: resize.c:165-203, 257-266) and
 * one axis up with the other down go through whole. */
static int
vips_resize_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsResizeHip *resize = (VipsResizeHip *) op;
	double hscale = resize->scale;
	double vscale = vips_object_argument_isset(VIPS_OBJECT(op), "vscale") ? resize->vscale : resize->scale;

	if (resize->kernel == VIPS_KERNEL_NEAREST || hscale <= 0.0 || vscale <= 0.0)
		return 1;
	if (hscale >= 1.0 && vscale >= 1.0) {
		/* pure upsizing: vips_affine with the matrix (hscale, 0, 0, vscale), centre sampling and
		 * the interpolator the kernel maps to (resize.c:118-133, 268-300) */
		ResampleStrip *p;

		if (hscale == 1.0 && vscale == 1.0)
			return 1;
		p = g_new0(ResampleStrip, 1);
		p->upsize = TRUE;
		p->hscale = hscale;
		p->vscale = vscale;
		p->idx = 0.5 * (1.0 - 1.0 / hscale);
		p->idy = 0.5 * (1.0 - 1.0 / vscale);
		p->interpolate = resize->kernel == VIPS_KERNEL_LINEAR ? 1 : 2; /* bilinear : bicubic */
		if (vips_hip_affine_out_size(in->Xsize, hscale) != op->out->Xsize ||
			vips_hip_affine_out_size(in->Ysize, vscale) != op->out->Ysize) {
			g_free(p);
			return 1;
		}
		*plan = p;
		return 0;
	}
	if (hscale > 1.0 || vscale > 1.0)
		return 1; /* one axis up, one down: whole image */
	/* "Don't let either axis drop below 1 px." (resize.c:197-200) */
	hscale = VIPS_MAX(hscale, 1.0 / in->Xsize);
	vscale = VIPS_MAX(vscale, 1.0 / in->Ysize);
	return resample_strip_open(op, in, 1.0 / vscale, 1.0 / hscale, resize->kernel, resize->gap, FALSE, FALSE, plan);
}

HIP_SUBCLASS_FULL(VipsResizeHip, vips_resize_hip, "resize_hip", "resize an image (MI355X)",
	HIP_RESAMPLE_STRIPS(vips_resize_hip))

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

/* The plain case -- a 3-band uchar sRGB image, not linear, no crop: a resize by the factor
 * vips_thumbnail_calculate_shrink picks (thumbnail.c:413-467) -- has the resize's region form;
 * alpha (premultiply), linear light and crops go through whole. */
static int
vips_thumbnail_hip_strip_open(VipsHipOp *op, VipsImage *in, void **plan)
{
	VipsThumbnailHip *thumbnail = (VipsThumbnailHip *) op;
	const int width = thumbnail->width;
	const int height = vips_object_argument_isset(VIPS_OBJECT(op), "height") ? thumbnail->height : width;
	double hshrink, vshrink;

	if (thumbnail->linear || thumbnail->crop != VIPS_INTERESTING_NONE || in->Bands != 3 ||
		in->BandFmt != VIPS_FORMAT_UCHAR || in->Type != VIPS_INTERPRETATION_sRGB)
		return 1;
	hshrink = (double) in->Xsize / width;
	vshrink = (double) in->Ysize / height;
	if (thumbnail->size != VIPS_SIZE_FORCE) {
		if (!(hshrink < vshrink))
			vshrink = hshrink;
		else
			hshrink = vshrink;
	}
	if (thumbnail->size == VIPS_SIZE_UP || hshrink <= 1.0 || vshrink <= 1.0)
		return 1;
	hshrink = VIPS_MIN(hshrink, in->Xsize);
	vshrink = VIPS_MIN(vshrink, in->Ysize);
	/* (through 1 / scale, as vips_hip_thumbnail_image -> vips_hip_resize computes it) */
	hshrink = 1.0 / (1.0 / hshrink);
	vshrink = 1.0 / (1.0 / vshrink);
	return resample_strip_open(op, in, vshrink, hshrink, VIPS_KERNEL_LANCZOS3, 2.0, FALSE, FALSE, plan);
}

HIP_SUBCLASS_FULL(VipsThumbnailHip, vips_thumbnail_hip, "thumbnail_image_hip",
	"generate thumbnail from image (MI355X)", HIP_RESAMPLE_STRIPS(vips_thumbnail_hip))

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

/* a per-pixel operation: a strip reads exactly its own rows */
static int
hip_pointwise_halo(VipsHipOp *op, VipsImage *in, int *above, int *below)
{
	*above = *below = 0;

	return 0;
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
/* convsep.c:61-118: the mask runs along both axes; n taps read n / 2 rows above and the rest
 * below (the same window whatever the precision: the approximate form's box sums included) */
static int
vips_convsep_hip_halo(VipsHipOp *op, VipsImage *in, int *above, int *below)
{
	VipsConvHip *conv = (VipsConvHip *) op;
	VipsImage *M;
	int n;

	if (vips_check_matrix("convsep_hip", conv->mask, &M))
		return -1;
	n = M->Xsize * M->Ysize;
	g_object_unref(M);
	*above = n / 2;
	*below = n - 1 - n / 2;

	return 0;
}

HIP_SUBCLASS_FULL(VipsConvsepHip, vips_convsep_hip, "convsep_hip", "separable convolution operation (MI355X)",
	HIP_HALO(vips_convsep_hip))

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

/* gaussblur.c:71-116: vips_gaussmat(sigma, min_ampl, separable, precision) then vips_convsep:
 * the mask's width decides the rows a strip reads */
static int
vips_gaussblur_hip_halo(VipsHipOp *op, VipsImage *in, int *above, int *below)
{
	VipsGaussblurHip *g = (VipsGaussblurHip *) op;
	int n;

	if (g->sigma < 0.2) { /* gaussblur.c:88-92: a copy */
		*above = *below = 0;
		return 0;
	}
	if ((n = vips_hip_gaussmat(g->sigma, g->min_ampl, 1, g->precision, NULL, 0, NULL)) < 0)
		return hip_fail("gaussblur_hip");
	*above = n / 2;
	*below = n - 1 - n / 2;

	return 0;
}

HIP_SUBCLASS_FULL(VipsGaussblurHip, vips_gaussblur_hip, "gaussblur_hip", "gaussian blur (MI355X)",
	HIP_HALO(vips_gaussblur_hip))

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

/* sharpen.c:176-228: everything is per pixel except the blur of L with
 * vips_gaussmat(sigma, 0.1, separable, integer) */
static int
vips_sharpen_hip_halo(VipsHipOp *op, VipsImage *in, int *above, int *below)
{
	VipsSharpenHip *s = (VipsSharpenHip *) op;
	int n;

	if ((n = vips_hip_gaussmat(s->sigma, 0.1, 1, VIPS_PRECISION_INTEGER, NULL, 0, NULL)) < 0)
		return hip_fail("sharpen_hip");
	*above = n / 2;
	*below = n - 1 - n / 2;

	return 0;
}

HIP_SUBCLASS_FULL(VipsSharpenHip, vips_sharpen_hip, "sharpen_hip", "unsharp masking for print (MI355X)",
	HIP_HALO(vips_sharpen_hip))

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
	"convert to a new colorspace (MI355X)", class->fuse = vips_colourspace_hip_fuse; class->halo = hip_pointwise_halo;)

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

HIP_SUBCLASS_FULL(VipsCastHip, vips_cast_hip, "cast_hip", "cast an image (MI355X)", class->halo = hip_pointwise_halo;)

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

HIP_SUBCLASS_FULL(VipsPremultiplyHip, vips_premultiply_hip, "premultiply_hip", "premultiply image alpha (MI355X)",
	class->halo = hip_pointwise_halo;)
HIP_SUBCLASS_FULL(VipsUnpremultiplyHip, vips_unpremultiply_hip, "unpremultiply_hip",
	"unpremultiply image alpha (MI355X)", class->halo = hip_pointwise_halo;)

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
