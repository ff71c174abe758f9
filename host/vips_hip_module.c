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

static guint64
hip_env_bytes(const char *name, guint64 fallback)
{
	const char *env = g_getenv(name);
	guint64 budget = fallback;

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

/* HBM budget (bytes of input + output an operation may hold on the device at once) above
 * which strip-capable operations work through the image in row strips.
 * $VIPS_HIP_BUDGET, with an optional k / m / g suffix; default 64 GiB.
 */
static guint64
hip_budget(void)
{
	return hip_env_bytes("VIPS_HIP_BUDGET", (guint64) 64 << 30);
}

/* Host budget: the bytes of a result this module keeps on the HOST at once (plus the two
 * staging buffers of the strip loop).  A result that fits is kept whole; a larger one is held as
 * a ring of row strips that the consumer walks through -- what the reference's sinks do with
 * their two buffers (iofuncs/sinkdisc.c:177-220, sink.c:428-441).
 * $VIPS_HIP_HOST_BUDGET, with an optional k / m / g suffix; default 2 GiB.
 */
static guint64
hip_host_budget(void)
{
	return hip_env_bytes("VIPS_HIP_HOST_BUDGET", (guint64) 2 << 30);
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
	GCond cond;             /* generate calls wait here: a strip (band) changed state, the evaluation is over */
	GCond pcond;            /* the strip producer waits here: a new demand, a slot nobody reads any more */
	gboolean evaluated;
	gboolean evaluating;    /* a thread is inside hip_eval() (which may wait, lock released, for the
	                         * strip producer's plan): everybody else waits for it */
	char *eval_error;       /* non-NULL: evaluation failed, with this message */
	VipsHipImage *result;   /* the result on the device (NULL for a strip-mined run) */

	/* The host side of the result, for generate: a cache of row items.  A device result comes
	 * down in bands of rows, each on its first touch (a consumer that asks for one tile of a
	 * 17 GB result pays for one band, not for the image); a strip-mined result is made strip by
	 * strip by the producer thread below, and generate calls are served as their strips land. */
	struct _HipCache *cache;
	gboolean striped;       /* items are strips from the producer thread */
	int first_row;          /* the top row of the generate call that started the evaluation */
	GThread *producer;
	gboolean producer_running;
	gboolean producer_quit;
	int setup;              /* the producer's first run: 0 pending, 1 ready, 2 no region form, -1 failed */
	guint64 budget;         /* the HBM budget the strips were sized for */
	int device;             /* the device the first producer run was dealt */
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

	/* Optional, instead of the four hooks above: operations whose output row y reads input rows
	 * y - above .. y + below only (pointwise: 0, 0).  Their region form is the image-level
	 * operation itself run on a window -- the rows of the strip plus that halo -- of which the
	 * strip's rows are kept: at the image's own top and bottom the window ends where the image
	 * ends, so the edge handling is the whole-image operation's.  Returns 1 when this instance
	 * cannot be strip-mined. */
	int (*halo)(struct _VipsHipOp *op, VipsImage *in, int *above, int *below);

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

/* ---- the host side of a result: a bounded cache of row items
 *
 * The reference evaluates any pipeline with bounded memory: a sink asks for tiles, threads
 * compute them (iofuncs/sink.c:469, thread.c:301-325), sinkdisc.c:177-220 writes one buffer behind
 * the one being filled, and progress is reported per tile (sink.c:428-441).  Here the unit is a
 * row item of megabytes to gigabytes -- a strip of a strip-mined result, a band of a device
 * result -- and the host holds `n_slots` of them: every item when the result fits the host
 * budget, else a ring that follows the consumer.  generate calls are served as soon as the item
 * they touch has landed; an item that has left the ring is made (or downloaded) again when
 * somebody comes back for it, the way the reference recomputes a tile that fell out of its cache.
 * Everything below is under op->lock unless it says otherwise.
 */
enum {
	HIP_SLOT_EMPTY = 0,
	HIP_SLOT_FILLING, /* claimed: being pulled / queued / downloaded */
	HIP_SLOT_ISSUED,  /* its download is queued: `event` says when the pixels are there */
	HIP_SLOT_READY
};

typedef struct _HipCache {
	int item_rows; /* rows per item (the last one may be shorter) */
	int n_items;
	int n_slots;
	size_t ls;     /* bytes per row */
	gboolean pinned; /* slots are vips_hip_malloc_host() memory */
	VipsPel **mem; /* per slot, allocated on first use */
	int *slot_item;
	guint8 *slot_state;
	int *slot_pins;      /* generate calls copying out of the slot right now */
	guint64 *slot_tick;  /* last use */
	void **slot_event;   /* ISSUED: recorded behind the slot's download; made by the producer on the
	                      * slot's first use, kept for the cache's life */
	int *item_slot;      /* the slot an item is in, or -1 */
	int *want;           /* per item: generate calls waiting for it */
	guint64 tick;
	int hi_seen;         /* the highest item a consumer has asked for */
} HipCache;

/* what the module holds on the host right now / at most so far (the tests ask: is it bounded?) */
static GMutex hip_host_lock;
static guint64 hip_host_now = 0, hip_host_peak = 0;
/* generate calls served while their operation's producer was still making strips; producer
 * (re)starts (the tests ask: does the output stream? was an evicted strip made again?) */
static volatile gint hip_served_early = 0, hip_producer_starts = 0;

static void
hip_host_account(gint64 delta)
{
	g_mutex_lock(&hip_host_lock);
	hip_host_now += delta;
	hip_host_peak = VIPS_MAX(hip_host_peak, hip_host_now);
	g_mutex_unlock(&hip_host_lock);
}

/* stats[0] = peak host bytes, [1] = host bytes now, [2] = requests served while strips were
 * still being made, [3] = producer starts; reset != 0 restarts the peak from the current level */
G_MODULE_EXPORT void
vips_hip_module_stream_stats(guint64 stats[4], int reset)
{
	g_mutex_lock(&hip_host_lock);
	stats[0] = hip_host_peak;
	stats[1] = hip_host_now;
	if (reset)
		hip_host_peak = hip_host_now;
	g_mutex_unlock(&hip_host_lock);
	stats[2] = (guint64) g_atomic_int_get(&hip_served_early);
	stats[3] = (guint64) g_atomic_int_get(&hip_producer_starts);
}

static HipCache *
hip_cache_new(int total_rows, int item_rows, size_t ls, guint64 host_bytes, gboolean pinned)
{
	HipCache *c = g_new0(HipCache, 1);
	const guint64 item_bytes = (guint64) item_rows * ls;

	c->item_rows = item_rows;
	c->n_items = (total_rows + item_rows - 1) / item_rows;
	c->ls = ls;
	c->pinned = pinned;
	/* everything when it fits; else a ring of at least three: the item being read, the one a
	 * region may straddle into, the one being made */
	if ((guint64) c->n_items * item_bytes <= host_bytes)
		c->n_slots = c->n_items;
	else
		c->n_slots = (int) VIPS_CLIP(3, host_bytes / item_bytes, (guint64) c->n_items);
	c->mem = g_new0(VipsPel *, c->n_slots);
	c->slot_item = g_new(int, c->n_slots);
	c->slot_state = g_new0(guint8, c->n_slots);
	c->slot_pins = g_new0(int, c->n_slots);
	c->slot_tick = g_new0(guint64, c->n_slots);
	c->slot_event = g_new0(void *, c->n_slots);
	c->item_slot = g_new(int, c->n_items);
	c->want = g_new0(int, c->n_items);
	for (int i = 0; i < c->n_slots; i++)
		c->slot_item[i] = -1;
	for (int i = 0; i < c->n_items; i++)
		c->item_slot[i] = -1;
	c->hi_seen = -1;

	return c;
}

static void
hip_cache_free(HipCache *c)
{
	if (!c)
		return;
	for (int i = 0; i < c->n_slots; i++)
		if (c->mem[i]) {
			if (c->pinned)
				vips_hip_free_host(c->mem[i]);
			else
				g_free(c->mem[i]);
			hip_host_account(-(gint64) ((guint64) c->item_rows * c->ls));
		}
	for (int i = 0; i < c->n_slots; i++)
		vips_hip_event_free(c->slot_event[i]);
	g_free(c->mem);
	g_free(c->slot_item);
	g_free(c->slot_state);
	g_free(c->slot_pins);
	g_free(c->slot_tick);
	g_free(c->slot_event);
	g_free(c->item_slot);
	g_free(c->want);
	g_free(c);
}

/* a slot for a new item: an empty one, else the least recently used READY one nobody is copying
 * out of or waiting for; -1 when there is none right now */
static int
hip_cache_victim(HipCache *c)
{
	int best = -1;

	for (int i = 0; i < c->n_slots; i++) {
		if (c->slot_state[i] == HIP_SLOT_EMPTY)
			return i;
		if (c->slot_state[i] == HIP_SLOT_READY && c->slot_pins[i] == 0 && c->want[c->slot_item[i]] == 0 &&
			(best < 0 || c->slot_tick[i] < c->slot_tick[best]))
			best = i;
	}

	return best;
}

static void
hip_cache_claim(HipCache *c, int slot, int item)
{
	if (c->slot_item[slot] >= 0)
		c->item_slot[c->slot_item[slot]] = -1;
	c->slot_item[slot] = item;
	c->item_slot[item] = slot;
	c->slot_state[slot] = HIP_SLOT_FILLING;
	c->slot_tick[slot] = ++c->tick;
}

static void
hip_cache_drop(HipCache *c, int slot)
{
	if (c->slot_item[slot] >= 0)
		c->item_slot[c->slot_item[slot]] = -1;
	c->slot_item[slot] = -1;
	c->slot_state[slot] = HIP_SLOT_EMPTY;
}

/* the slot's memory, allocated on its first use (no lock needed: the slot is claimed) */
static VipsPel *
hip_cache_mem(HipCache *c, int slot)
{
	const size_t bytes = (size_t) c->item_rows * c->ls;

	if (!c->mem[slot]) {
		if (c->pinned && !(c->mem[slot] = (VipsPel *) vips_hip_malloc_host(bytes)))
			vips_hip_error_clear(); /* too large to pin: ordinary memory, its download simply does not overlap */
		if (!c->mem[slot])
			c->mem[slot] = (VipsPel *) g_try_malloc(bytes);
		if (c->mem[slot])
			hip_host_account((gint64) bytes);
	}

	return c->mem[slot];
}

/* ---- the strip loop: images over the HBM budget
 *
 * The stages are pull (libvips' own threaded evaluation of the upstream pipeline, straight into
 * PINNED memory), upload, kernels, download, serve.  One producer thread per evaluating
 * operation walks the strips in the order the consumer asks for them: strip k + 1 is pulled while
 * strip k's upload, kernels and download run on the device, strips alternate between two streams,
 * two pinned staging buffers and two device windows, and each result lands in a slot of the host
 * cache above, where the generate calls that wait for it pick it up -- libvips' workers run
 * beside the device instead of sleeping until the last strip is down.
 */

/* pull target: rows [top, top + rows) of an image into a buffer */
typedef struct _HipPull {
	VipsPel *buf;
	size_t ls;
	int top;
} HipPull;

static int
hip_pull_gen(VipsRegion *region, void *seq, void *a, void *b, gboolean *stop)
{
	HipPull *pull = (HipPull *) b;
	VipsRect *r = &region->valid;
	const size_t ps = VIPS_IMAGE_SIZEOF_PEL(region->im);

	for (int y = 0; y < r->height; y++)
		memcpy(pull->buf + (size_t) (r->top + y - pull->top) * pull->ls + (size_t) r->left * ps,
			VIPS_REGION_ADDR(region, r->left, r->top + y), (size_t) r->width * ps);

	return 0;
}

/* rows [top, top + rows) of @in into @buf, evaluated by libvips' thread pool */
static int
hip_pull_rows(VipsImage *in, int top, int rows, VipsPel *buf)
{
	VipsImage *crop = NULL;
	HipPull pull = { buf, VIPS_IMAGE_SIZEOF_LINE(in), 0 };
	int result;

	if (vips_crop(in, &crop, 0, top, in->Xsize, rows, NULL))
		return -1;
	result = vips_sink(crop, vips_start_one, hip_pull_gen, vips_stop_one, crop, &pull);
	g_object_unref(crop);

	return result;
}

/* The generic region form of operations with a halo hook: the image-level operation on the
 * window, the strip's rows copied out. */
typedef struct _HaloStrip {
	int above, below;
} HaloStrip;

static int
hip_halo_run(VipsHipOp *op, HaloStrip *plan, const VipsHipRegion *in, const VipsHipRegion *out)
{
	VipsHipOpClass *hclass = VIPS_HIP_OP_GET_CLASS(op);
	VipsHipImage *window, *made = NULL;
	int result = -1;

	if (!(window = vips_hip_image_new_from_device(in->data, in->width, in->height, in->bands, in->format,
			  op->ready->Type)))
		return -1;
	if (!hclass->compute(op, window, &made)) {
		const int skip = out->top - in->top;

		if (vips_hip_image_get_width(made) != out->width || vips_hip_image_get_height(made) != in->height ||
			vips_hip_image_get_bands(made) != out->bands || vips_hip_image_get_format(made) != out->format ||
			skip < 0 || skip + out->height > in->height)
			vips_error(VIPS_OBJECT_GET_CLASS(op)->nickname, "%s", "window result does not match the strip");
		else
			result = vips_hip_memcpy_d2d(out->data,
				(char *) vips_hip_image_get_data(made) + (size_t) skip * vips_hip_image_get_stride(made),
				(size_t) out->height * out->stride);
	}
	/* (the pool orders reuse of these blocks behind the copy: same thread, same stream) */
	vips_hip_image_unref(made);
	vips_hip_image_unref(window);

	return result;
}

static const VipsPel *hip_host_in_place(VipsHipOp *op);

/* how many bands of device results have been downloaded in this process (the tests ask: did a
 * small request pay for the whole image?) */
static volatile gint hip_bands_done = 0;

G_MODULE_EXPORT int
vips_hip_module_bands_done(void)
{
	return g_atomic_int_get(&hip_bands_done);
}

/* how many strips the producers have made in this process (the tests ask: was it strip-mined?) */
static volatile gint hip_strips_done = 0;

G_MODULE_EXPORT int
vips_hip_module_strips_done(void)
{
	return g_atomic_int_get(&hip_strips_done);
}

static void
hip_producer_fail(VipsHipOp *op, const char *domain)
{
	/* (called without the lock; the message is what every waiting generate call reports) */
	if (vips_hip_error_buffer()[0])
		hip_fail(domain);
	g_mutex_lock(&op->lock);
	if (!op->eval_error)
		op->eval_error = g_strdup(vips_error_buffer()[0] ? vips_error_buffer() : "evaluation failed");
	g_cond_broadcast(&op->cond);
	g_mutex_unlock(&op->lock);
}

/* The producer: one run = open the plan, make strips for as long as somebody wants one that is
 * not on the host, hand the device and the staging memory back.  A later request for a strip that
 * has left the ring starts another run. */
static void *
hip_producer(void *data)
{
	VipsHipOp *op = (VipsHipOp *) data;
	VipsHipOpClass *hclass = VIPS_HIP_OP_GET_CLASS(op);
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;
	VipsImage *in = op->ready;
	VipsImage *out = op->out;
	const size_t ls = VIPS_IMAGE_SIZEOF_LINE(out);
	const size_t in_ls = VIPS_IMAGE_SIZEOF_LINE(in);
	const gboolean generic = hclass->halo != NULL && !hclass->strip_open;
	const guint64 budget = op->budget;
	const guint64 host_budget = hip_host_budget();
	const VipsPel *resident = NULL;
	HaloStrip halo = { 0, 0 };
	void *plan = NULL;
	VipsPel *stage[2] = { NULL, NULL };
	void *stream[2] = { NULL, NULL };
	void *computed[2] = { NULL, NULL }, *uploaded[2] = { NULL, NULL };
	gboolean upload_seen[2] = { FALSE, FALSE };
	VipsHipImage *dev_in[2] = { NULL, NULL }, *dev_out[2] = { NULL, NULL };
	HipCache *c = NULL;
	int rows = 0, max_in_rows = 0;
	int setup = -1;
	int cursor = 0, pulled = -1, issues = 0, pending = -1;
	gboolean failed = FALSE;

#define STRIP_NEED(TOP, N, IN_TOP, IN_ROWS) \
	do { \
		if (generic) { \
			*(IN_TOP) = (TOP) - halo.above; \
			*(IN_ROWS) = (N) + halo.above + halo.below; \
		} \
		else \
			hclass->strip_need(op, plan, (TOP), (N), (IN_TOP), (IN_ROWS)); \
		if (*(IN_TOP) < 0) { \
			*(IN_ROWS) += *(IN_TOP); \
			*(IN_TOP) = 0; \
		} \
		*(IN_ROWS) = VIPS_MIN(*(IN_ROWS), in->Ysize - *(IN_TOP)); \
	} while (0)

	g_atomic_int_inc(&hip_producer_starts);

	/* NULL selects the library's own per-thread stream (and deals this thread a device; a later
	 * run goes back to the first run's: the slot events live there) */
	if ((op->cache && vips_hip_init(op->device)) || vips_hip_set_stream(NULL))
		goto setup_done;
	if (generic) {
		const int r = hclass->halo(op, in, &halo.above, &halo.below);

		if (r) {
			setup = r > 0 ? 2 : -1;
			goto setup_done;
		}
	}
	else {
		int r;

		if (!hclass->strip_open || !hclass->strip_need || !hclass->strip_run) {
			setup = 2;
			goto setup_done;
		}
		if ((r = hclass->strip_open(op, in, &plan))) {
			plan = NULL;
			setup = r > 0 ? 2 : -1;
			goto setup_done;
		}
	}

	/* an input that already is host memory is uploaded from where it lies: nothing to pull */
	resident = hip_host_in_place(op);

	if (op->cache) /* a later run: the geometry is the first run's */
		rows = op->cache->item_rows;
	else {
		const gboolean whole_fits = (guint64) ls * out->Ysize <= host_budget;

		/* the tallest strip (a multiple of 16 lines: the reference's fat-strip height, which the
		 * vertical reduce re-seeds its position on) of which TWO -- one being pulled and uploaded,
		 * one being computed and downloaded -- fit the HBM budget with their input rows (the
		 * generic form also holds the window's result and the operation's own temporaries: counted
		 * as two more windows), and of which three, with the two staging buffers, fit the host
		 * budget when the whole result does not */
		for (rows = VIPS_ROUND_UP(out->Ysize, 16); rows > 16; rows = VIPS_ROUND_UP(rows / 2, 16)) {
			int in_top, in_rows;
			guint64 dev, host;

			STRIP_NEED(0, VIPS_MIN(rows, out->Ysize), &in_top, &in_rows);
			dev = 2 * ((guint64) in_rows * in_ls + (guint64) rows * ls);
			if (generic)
				dev += 2 * (guint64) in_rows * VIPS_MAX(ls, in_ls);
			host = 3 * (guint64) rows * ls + (resident ? 0 : 2 * (guint64) in_rows * in_ls);
			if (dev <= budget && (whole_fits || host <= host_budget))
				break;
		}
	}
	for (int top = 0; top < out->Ysize; top += rows) {
		int in_top, in_rows;

		STRIP_NEED(top, VIPS_MIN(rows, out->Ysize - top), &in_top, &in_rows);
		max_in_rows = VIPS_MAX(max_in_rows, in_rows);
	}

	/* two staging buffers, pinned; two input windows and two output strips on the device, kept
	 * for the whole run */
	for (int i = 0; i < 2; i++) {
		if (!resident) {
			if (!(stage[i] = (VipsPel *) vips_hip_malloc_host((size_t) max_in_rows * in_ls)))
				goto setup_done;
			hip_host_account((gint64) ((guint64) max_in_rows * in_ls));
		}
		if (!(stream[i] = vips_hip_stream_new()) ||
			!(computed[i] = vips_hip_event_new()) || !(uploaded[i] = vips_hip_event_new()) ||
			!(dev_in[i] = vips_hip_image_new(in->Xsize, max_in_rows, in->Bands, in->BandFmt, in->Type)) ||
			!(dev_out[i] = vips_hip_image_new(out->Xsize, VIPS_MIN(rows, out->Ysize), out->Bands, out->BandFmt, out->Type)))
			goto setup_done;
	}
	op->device = vips_hip_image_get_device(dev_in[0]);
	setup = 1;

setup_done:
	/* (a LATER run -- a request for an evicted strip -- that cannot set up again must fail: "not this form's
	 * case" (2) would leave every waiting generate call restarting the producer for ever) */
	{
		/* (op->cache is written under the lock by an earlier run of this function: read it there too -- ADVICE r5) */
		gboolean had_cache;

		g_mutex_lock(&op->lock);
		had_cache = op->cache != NULL;
		g_mutex_unlock(&op->lock);
		if (setup < 0 || (setup != 1 && had_cache)) {
			if (setup == 2)
				vips_error(nick, "%s", "the strip producer could not be set up again");
			setup = -1;
			hip_producer_fail(op, nick);
		}
	}
	g_mutex_lock(&op->lock);
	if (setup == 1 && !op->cache) {
		const guint64 stage_bytes = resident ? 0 : 2 * (guint64) max_in_rows * in_ls;

		op->cache = hip_cache_new(out->Ysize, rows, ls, host_budget > stage_bytes ? host_budget - stage_bytes : 0, TRUE);
		/* the walk starts at the strip the first consumer is waiting for (its call cannot say so itself: the
		 * cache it would note its wish in is made here) -- a crop at the bottom does not pay for strip 0 */
		if (op->cache && op->first_row > 0)
			cursor = VIPS_MIN(op->first_row / rows, op->cache->n_items - 1);
	}
	if (!op->setup)
		op->setup = setup;
	g_cond_broadcast(&op->cond);
	c = op->cache;

	/* ---- the loop (lock held at the top of every turn) */
	while (setup == 1) {
		const int ahead = VIPS_MAX(1, c->n_slots - 2);
		gboolean can;
		int w = -1, slot = -1;

		if (op->producer_quit || op->eval_error)
			break;
		/* the lowest strip somebody waits for that is neither here nor on its way: when the walk
		 * will not reach it soon, go there */
		for (int i = 0; i < c->n_items; i++)
			if (c->want[i] > 0 && c->item_slot[i] < 0) {
				w = i;
				break;
			}
		if (w >= 0 && (w < cursor || w >= cursor + c->n_slots))
			cursor = w;
		/* skip what is already here (after a jump back) */
		while (cursor < c->n_items && c->item_slot[cursor] >= 0)
			cursor++;
		/* not further ahead of the consumer than the ring can hold beside what it is reading */
		can = cursor < c->n_items && (c->n_slots >= c->n_items || cursor <= VIPS_MAX(c->hi_seen, 0) + ahead);
		if (can)
			slot = hip_cache_victim(c);
		if (slot < 0) {
			if (pending >= 0) {
				/* nothing to queue: make the last strip visible, then look again */
				void *ev = c->slot_event[pending];
				const int p = pending;

				pending = -1;
				g_mutex_unlock(&op->lock);
				if (vips_hip_event_synchronize(ev)) {
					hip_producer_fail(op, nick);
					failed = TRUE;
				}
				g_mutex_lock(&op->lock);
				if (!failed && c->slot_state[p] == HIP_SLOT_ISSUED)
					c->slot_state[p] = HIP_SLOT_READY;
				g_cond_broadcast(&op->cond);
				if (failed)
					break;
				continue;
			}
			if (cursor >= c->n_items && w < 0)
				break; /* every strip made, nobody waiting: the device goes back */
			g_cond_wait(&op->pcond, &op->lock);
			continue;
		}
		hip_cache_claim(c, slot, cursor);
		g_mutex_unlock(&op->lock);

		/* ---- strip `cursor` into `slot`, without the lock */
		{
			const int k = cursor;
			const int b = issues & 1;
			const int top = k * rows;
			const int n = VIPS_MIN(rows, out->Ysize - top);
			VipsPel *mem = hip_cache_mem(c, slot);
			VipsHipRegion ri, ro;
			int in_top, in_rows;

			STRIP_NEED(top, n, &in_top, &in_rows);
			if (!mem) {
				vips_error(nick, "%s", "out of memory for the result");
				failed = TRUE;
			}
			else if (vips_image_iskilled(out)) {
				vips_error(nick, "%s", "killed");
				failed = TRUE;
			}
			/* its input rows: prefetched during the previous strip, or pulled now (the first
			 * strip, or after a jump) into the staging buffer this turn uploads from -- once the
			 * upload that last read that buffer is through */
			if (!failed && !resident && pulled != k) {
				if ((upload_seen[b] && vips_hip_event_synchronize(uploaded[b])) ||
					hip_pull_rows(in, in_top, in_rows, stage[b]))
					failed = TRUE;
				pulled = k;
			}
			if (!c->slot_event[slot] && !(c->slot_event[slot] = vips_hip_event_new()))
				failed = TRUE;
			/* strip k on stream b: its upload may run beside the previous strip's kernels and
			 * download (the other stream); its KERNELS wait for the previous strip's -- the
			 * operation's temporaries come from a pool that orders reuse within one stream, and
			 * kernels of two strips have nothing to gain from running side by side */
			if (!failed &&
				(vips_hip_set_stream(stream[b]) ||
					vips_hip_memcpy_h2d_async(vips_hip_image_get_data(dev_in[b]),
						resident ? resident + (size_t) in_top * in_ls : stage[b], (size_t) in_rows * in_ls) ||
					vips_hip_event_record(uploaded[b]) ||
					(issues > 0 && vips_hip_stream_wait_event(computed[1 - b]))))
				failed = TRUE;
			upload_seen[b] = TRUE;
			if (!failed) {
				vips_hip_image_region(dev_in[b], &ri);
				ri.top = in_top;
				ri.height = in_rows;
				ri.im_width = in->Xsize;
				ri.im_height = in->Ysize;
				vips_hip_image_region(dev_out[b], &ro);
				ro.top = top;
				ro.height = n;
				ro.im_width = out->Xsize;
				ro.im_height = out->Ysize;
				if ((generic ? hip_halo_run(op, &halo, &ri, &ro) : hclass->strip_run(op, plan, &ri, &ro)) ||
					vips_hip_event_record(computed[b]) ||
					vips_hip_memcpy_d2h_async(mem, vips_hip_image_get_data(dev_out[b]), (size_t) n * ls) ||
					vips_hip_event_record(c->slot_event[slot]))
					failed = TRUE;
			}
			if (failed) {
				hip_producer_fail(op, nick);
				g_mutex_lock(&op->lock);
				hip_cache_drop(c, slot);
				break;
			}
			issues++;
			cursor = k + 1;
			g_atomic_int_inc(&hip_strips_done);

			g_mutex_lock(&op->lock);
			c->slot_state[slot] = HIP_SLOT_ISSUED;
			g_cond_broadcast(&op->cond);
			g_mutex_unlock(&op->lock);

			/* the strip before this one has had a whole turn: make it visible (a generate call
			 * that wanted it sooner has waited on its event itself) */
			if (pending >= 0) {
				if (vips_hip_event_synchronize(c->slot_event[pending])) {
					hip_producer_fail(op, nick);
					g_mutex_lock(&op->lock);
					break;
				}
				g_mutex_lock(&op->lock);
				if (c->slot_state[pending] == HIP_SLOT_ISSUED)
					c->slot_state[pending] = HIP_SLOT_READY;
				g_cond_broadcast(&op->cond);
				g_mutex_unlock(&op->lock);
			}
			pending = slot;

			/* meanwhile, on the host: the next strip's rows into the other staging buffer (the
			 * upload that read it, two strips ago, first) -- unless the consumers are elsewhere */
			if (!resident && cursor < c->n_items) {
				const int nb = issues & 1;
				gboolean go;

				g_mutex_lock(&op->lock);
				go = !op->producer_quit && c->item_slot[cursor] < 0 &&
					(c->n_slots >= c->n_items || cursor <= VIPS_MAX(c->hi_seen, 0) + ahead);
				for (int i = 0; i < c->n_items && go; i++)
					if (c->want[i] > 0 && c->item_slot[i] < 0 && (i < cursor || i >= cursor + c->n_slots))
						go = FALSE;
				g_mutex_unlock(&op->lock);
				if (go) {
					const int ntop = cursor * rows;

					STRIP_NEED(ntop, VIPS_MIN(rows, out->Ysize - ntop), &in_top, &in_rows);
					if ((upload_seen[nb] && vips_hip_event_synchronize(uploaded[nb])) ||
						hip_pull_rows(in, in_top, in_rows, stage[nb])) {
						hip_producer_fail(op, nick);
						g_mutex_lock(&op->lock);
						break;
					}
					pulled = cursor;
				}
			}
		}
		g_mutex_lock(&op->lock);
	}
#undef STRIP_NEED

	/* (lock held) this run is over.  Whatever is still queued (a run that was told to quit, or
	 * failed) lands before its memory goes back; then a request that finds no producer running
	 * starts the next run. */
	for (int i = 0; i < 2; i++)
		if (stream[i]) {
			(void) vips_hip_set_stream(stream[i]);
			(void) vips_hip_synchronize();
		}
	for (int i = 0; c && i < c->n_slots; i++) {
		if (c->slot_state[i] == HIP_SLOT_ISSUED && !op->eval_error)
			c->slot_state[i] = HIP_SLOT_READY;
		else if (c->slot_state[i] == HIP_SLOT_ISSUED || c->slot_state[i] == HIP_SLOT_FILLING)
			hip_cache_drop(c, i);
	}
	op->producer_running = FALSE;
	g_cond_broadcast(&op->cond);
	g_mutex_unlock(&op->lock);

	for (int i = 0; i < 2; i++) {
		vips_hip_image_unref(dev_in[i]);
		vips_hip_image_unref(dev_out[i]);
	}
	(void) vips_hip_set_stream(NULL);
	for (int i = 0; i < 2; i++) {
		vips_hip_stream_free(stream[i]);
		vips_hip_event_free(computed[i]);
		vips_hip_event_free(uploaded[i]);
		if (stage[i]) {
			vips_hip_free_host(stage[i]);
			hip_host_account(-(gint64) ((guint64) max_in_rows * in_ls));
		}
	}
	if (!generic && plan && hclass->strip_close)
		hclass->strip_close(op, plan);
	vips_hip_error_clear();
	vips_thread_shutdown();

	return NULL;
}

/* (lock held) start a producer run; the previous run's thread, if any, has set
 * producer_running = FALSE and touches nothing of the operation any more */
static int
hip_producer_start(VipsHipOp *op)
{
	if (op->producer) {
		GThread *old = op->producer;

		op->producer = NULL;
		g_mutex_unlock(&op->lock);
		g_thread_join(old);
		g_mutex_lock(&op->lock);
		if (op->producer_running) /* (somebody else started the next run meanwhile) */
			return 0;
	}
	op->producer_running = TRUE;
	op->producer_quit = FALSE;
	if (!(op->producer = vips_g_thread_new("vips-hip strips", hip_producer, op))) {
		op->producer_running = FALSE;
		return -1;
	}

	return 0;
}

/* (lock held) An image over the HBM budget: start the producer and wait for its plan.  0: the
 * strips are on their way; 1: this instance has no region form; -1: failed. */
static int
hip_eval_strips(VipsHipOp *op, guint64 budget)
{
	VipsHipOpClass *hclass = VIPS_HIP_OP_GET_CLASS(op);

	if (!hclass->halo && !(hclass->strip_open && hclass->strip_need && hclass->strip_run))
		return 1;
	op->budget = budget;
	op->setup = 0;
	if (hip_producer_start(op)) {
		vips_error(VIPS_OBJECT_GET_CLASS(op)->nickname, "%s", "unable to start the strip producer");
		return -1;
	}
	while (!op->setup)
		g_cond_wait(&op->cond, &op->lock);
	if (op->setup == 1) {
		op->striped = TRUE;
		return 0;
	}

	return op->setup == 2 ? 1 : -1;
}

/* An input that already IS host memory (vips_image_new_from_memory(), a loaded-to-memory file,
 * a mapped .v), usable where it lies: its pixels, else NULL.  Costs nothing. */
static const VipsPel *
hip_host_in_place(VipsHipOp *op)
{
	VipsImage *raw = op->in;

	if (raw->Coding == VIPS_CODING_NONE && raw->Xsize == op->ready->Xsize && raw->Ysize == op->ready->Ysize &&
		raw->Bands == op->ready->Bands && raw->BandFmt == op->ready->BandFmt &&
		(raw->dtype == VIPS_IMAGE_SETBUF || raw->dtype == VIPS_IMAGE_SETBUF_FOREIGN ||
			raw->dtype == VIPS_IMAGE_MMAPIN || raw->dtype == VIPS_IMAGE_MMAPINRW) &&
		!vips_image_wio_input(raw) && raw->data)
		return (const VipsPel *) raw->data;

	return NULL;
}

/* The input's pixels on the host.  An image that already IS memory (vips_image_new_from_memory(),
 * a loaded-to-memory file, a mapped .v) is used where it lies: vips_image_decode() of an uncoded
 * image is a header-only vips_copy(), and pulling that "partial" image through
 * vips_image_copy_memory() would allocate and fill a second copy of the whole image on the host
 * (measured on the 1 GiB image of BASELINE config 2: ~300 ms, fifteen times the upload).  Anything
 * else is evaluated into memory by libvips' threaded sink.  *mem is what the caller unrefs. */
static const VipsPel *
hip_host_pixels(VipsHipOp *op, VipsImage **mem)
{
	const VipsPel *in_place = hip_host_in_place(op);

	*mem = NULL;
	if (in_place)
		return in_place;
	if (!(*mem = vips_image_copy_memory(op->ready)))
		return NULL;

	return VIPS_IMAGE_ADDR(*mem, 0, 0);
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
	const VipsPel *pixels;
	VipsImage *mem;

	*fresh = NULL;
	if (op->upstream)
		dev = op->upstream_device(op->upstream);
	if (dev)
		return dev;
	if (!(pixels = hip_host_pixels(op, &mem)))
		return NULL;
	*fresh = vips_hip_image_new_from_memory(pixels, in->Xsize, in->Ysize, in->Bands, in->BandFmt, in->Type);
	VIPS_UNREF(mem);
	if (!*fresh)
		hip_fail(VIPS_OBJECT_GET_CLASS(op)->nickname);

	return *fresh;
}

static void hip_eval(VipsHipOp *op);

/* (lock held) evaluate, once, whoever comes first */
static void
hip_ensure_eval(VipsHipOp *op)
{
	while (op->evaluating)
		g_cond_wait(&op->cond, &op->lock);
	if (!op->evaluated) {
		op->evaluating = TRUE;
		hip_eval(op);
		op->evaluating = FALSE;
		g_cond_broadcast(&op->cond);
	}
}

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
			const int r = hip_eval_strips(op, budget);

			if (r < 0)
				hip_eval_fail(op, class->nickname);
			if (r <= 0)
				return;
			/* r == 1: no region form -- try the whole image (fails loudly if HBM runs out) */
		}
		{
			const VipsPel *pixels = hip_host_pixels(op, &mem);

			if (!pixels ||
				!(fresh = vips_hip_image_new_from_memory(pixels, in->Xsize, in->Ysize, in->Bands, in->BandFmt, in->Type))) {
				VIPS_UNREF(mem);
				hip_eval_fail(op, class->nickname);
				return;
			}
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
	hip_ensure_eval(op);
	result = op->result;
	g_mutex_unlock(&op->lock);

	return result;
}

/* (lock held) The host copy of a device result: address space for the cache now, pixels band by
 * band as they are first asked for (~32 MB bands: large enough for the link, small enough to
 * skip), at most the host budget of them at once. */
static int
hip_bands_open(VipsHipOp *op)
{
	VipsImage *out = op->out;
	const size_t ls = VIPS_IMAGE_SIZEOF_LINE(out);
	const int band_rows = VIPS_CLIP(16, (int) (((size_t) 32 << 20) / ls), out->Ysize);

	op->cache = hip_cache_new(out->Ysize, band_rows, ls, hip_host_budget(), FALSE);

	return 0;
}

/* Rows of item @item for a generate call: its slot (pinned until hip_cache_put) and memory, or
 * NULL with the error set.  Waits for a strip that is on its way; downloads a band of a device
 * result itself. */
static const VipsPel *
hip_cache_get(VipsHipOp *op, int item, int *slot_out)
{
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;
	HipCache *c = op->cache;
	VipsImage *out = op->out;
	int s;

	g_mutex_lock(&op->lock);
	if (item > c->hi_seen) {
		c->hi_seen = item; /* (the producer may walk further now) */
		g_cond_signal(&op->pcond);
	}
	for (;;) {
		if (op->eval_error) {
			vips_error(nick, "%s", op->eval_error);
			g_mutex_unlock(&op->lock);
			return NULL;
		}
		if (vips_image_iskilled(out)) {
			g_mutex_unlock(&op->lock);
			vips_error(nick, "%s", "killed");
			return NULL;
		}
		s = c->item_slot[item];
		if (s >= 0 && c->slot_state[s] == HIP_SLOT_READY)
			break;
		if (s >= 0 && c->slot_state[s] == HIP_SLOT_ISSUED) {
			/* its download is queued: wait for it here rather than for the producer's next turn */
			void *ev = c->slot_event[s];
			int r;

			c->slot_pins[s]++;
			g_mutex_unlock(&op->lock);
			r = vips_hip_event_synchronize(ev);
			g_mutex_lock(&op->lock);
			c->slot_pins[s]--;
			if (r) {
				g_mutex_unlock(&op->lock);
				hip_fail(nick);
				return NULL;
			}
			if (c->slot_item[s] == item && c->slot_state[s] == HIP_SLOT_ISSUED) {
				c->slot_state[s] = HIP_SLOT_READY;
				g_cond_broadcast(&op->cond);
			}
			continue;
		}
		if (op->striped) {
			/* a strip that is being pulled, or is not on the host (any more): the producer makes it */
			c->want[item]++;
			if (!op->producer_running && hip_producer_start(op)) {
				c->want[item]--;
				g_mutex_unlock(&op->lock);
				vips_error(nick, "%s", "unable to start the strip producer");
				return NULL;
			}
			g_cond_signal(&op->pcond);
			g_cond_wait(&op->cond, &op->lock);
			c->want[item]--;
			continue;
		}
		/* a band of a device result: this thread downloads it (another thread may be at it) */
		if (s >= 0 || (s = hip_cache_victim(c)) < 0) {
			g_cond_wait(&op->cond, &op->lock);
			continue;
		}
		hip_cache_claim(c, s, item);
		g_mutex_unlock(&op->lock);
		{
			const int top = item * c->item_rows;
			const int rows = VIPS_MIN(c->item_rows, out->Ysize - top);
			VipsPel *mem = hip_cache_mem(c, s);
			int r = 0;

			if (!mem) {
				vips_error(nick, "%s", "out of memory for the result");
				r = -1;
			}
			/* (vips_hip_memcpy_d2h binds nothing: run where the result lives) */
			else if (vips_hip_init(vips_hip_image_get_device(op->result)) ||
				vips_hip_memcpy_d2h(mem,
					(const char *) vips_hip_image_get_data(op->result) + (size_t) top * vips_hip_image_get_stride(op->result),
					(size_t) rows * c->ls))
				r = hip_fail(nick);
			g_mutex_lock(&op->lock);
			if (r) {
				hip_cache_drop(c, s);
				g_cond_broadcast(&op->cond);
				g_mutex_unlock(&op->lock);
				return NULL;
			}
			c->slot_state[s] = HIP_SLOT_READY;
			g_atomic_int_inc(&hip_bands_done);
			g_cond_broadcast(&op->cond);
		}
	}
	c->slot_pins[s]++;
	c->slot_tick[s] = ++c->tick;
	if (op->striped && op->producer_running)
		g_atomic_int_inc(&hip_served_early);
	g_mutex_unlock(&op->lock);
	*slot_out = s;

	return c->mem[s];
}

static void
hip_cache_put(VipsHipOp *op, int slot)
{
	HipCache *c = op->cache;

	g_mutex_lock(&op->lock);
	/* the slot is a candidate for eviction again -- news only for whoever waits for a slot, and
	 * only when the ring is smaller than the result (waking every waiting worker on every tile
	 * served was a thundering herd: 256 workers, thousands of tiles) */
	if (--c->slot_pins[slot] == 0 && c->n_slots < c->n_items) {
		if (op->striped)
			g_cond_signal(&op->pcond);
		else
			g_cond_broadcast(&op->cond);
	}
	g_mutex_unlock(&op->lock);
}

static int
vips_hip_op_gen(VipsRegion *out_region, void *seq, void *a, void *b, gboolean *stop)
{
	VipsHipOp *op = (VipsHipOp *) b;
	VipsRect *r = &out_region->valid;
	VipsImage *out = out_region->im;
	const size_t ps = VIPS_IMAGE_SIZEOF_PEL(out);
	HipCache *c;

	if (vips_image_iskilled(out))
		return -1;

	/* First demand: evaluate.  A result that fits HBM is made here and now (then its bands come
	 * down as they are asked for); an image over the budget only gets its producer started. */
	g_mutex_lock(&op->lock);
	/* (only the call that STARTS the evaluation says where the producer's walk begins: a later call that arrives
	 * before the cache exists must not move it -- ADVICE r5) */
	if (!op->cache && !op->striped && !op->producer_running && !op->setup)
		op->first_row = r->top;
	hip_ensure_eval(op);
	if (op->eval_error) {
		vips_error(VIPS_OBJECT_GET_CLASS(op)->nickname, "%s", op->eval_error);
		g_mutex_unlock(&op->lock);
		return -1;
	}
	if (!op->cache && hip_bands_open(op)) {
		g_mutex_unlock(&op->lock);
		return -1;
	}
	c = op->cache;
	g_mutex_unlock(&op->lock);

	for (int item = r->top / c->item_rows; item <= (r->top + r->height - 1) / c->item_rows; item++) {
		const int top = item * c->item_rows;
		const int y0 = VIPS_MAX(r->top, top);
		const int y1 = VIPS_MIN(r->top + r->height, top + c->item_rows);
		int slot;
		const VipsPel *rows = hip_cache_get(op, item, &slot);

		if (!rows)
			return -1;
		for (int y = y0; y < y1; y++)
			memcpy(VIPS_REGION_ADDR(out_region, r->left, y),
				rows + (size_t) (y - top) * c->ls + (size_t) r->left * ps,
				(size_t) r->width * ps);
		hip_cache_put(op, slot);
	}

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

/* The ORIGINAL operation (the nickname less "_hip") with this operation's arguments, built.  A libvips build()
 * moves no pixels: microseconds, no device.  On failure the original's own words are left in the error buffer
 * (it validates the arguments); where it names itself, THIS operation's name.
 */
static VipsOperation *
hip_twin_build(VipsHipOp *op)
{
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;
	const size_t len = strlen(nick);
	char base[64];
	VipsOperation *twin;

	if (len < 5 || len >= sizeof(base) || strcmp(nick + len - 4, "_hip") != 0) {
		vips_error(nick, "%s", "not a *_hip nickname");
		return NULL;
	}
	memcpy(base, nick, len - 4);
	base[len - 4] = '\0';

	if (!(twin = vips_operation_new(base)))
		return NULL;
	vips_argument_map(VIPS_OBJECT(op), hip_copy_argument, twin, NULL);
	if (vips_object_build(VIPS_OBJECT(twin))) {
		char *said = g_strdup(vips_error_buffer());
		char **line = g_strsplit(said, "\n", -1);
		char *own = g_strdup_printf("%s: ", base);
		gboolean names_itself = FALSE;

		for (int i = 0; line[i]; i++)
			names_itself |= g_str_has_prefix(line[i], own);
		g_free(own);
		if (names_itself) /* (else the buffer stays exactly as the original left it) */
			vips_error_clear();
		for (int i = 0; names_itself && line[i]; i++) {
			char *colon = strstr(line[i], ": ");

			if (!line[i][0])
				continue;
			if (colon) {
				*colon = '\0';
				vips_error(strcmp(line[i], base) == 0 ? nick : line[i], "%s", colon + 2);
			}
			else
				vips_error(nick, "%s", line[i]);
		}
		g_strfreev(line);
		g_free(said);
		vips_object_unref_outputs(VIPS_OBJECT(twin));
		g_object_unref(twin);
		return NULL;
	}
	return twin;
}

/* What will `out` look like?  A drop-in has, by definition, the header the original would
 * produce: build the original, copy its output's header, drop it.  Exact by construction.
 */
static int
hip_twin_header(VipsHipOp *op, VipsImage *out)
{
	VipsOperation *twin;
	VipsImage *twin_out = NULL;

	if (!(twin = hip_twin_build(op)))
		return -1;
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

/* The cases the built-in operation handles and the device path does not (round 6; VERDICT r5 "drop-in holes"):
 * a drop-in hands them to the ORIGINAL operation instead of failing.
 *   - complex band formats, every operation (the device kernels have no complex arithmetic);
 *   - premultiply / unpremultiply of double images (conversion/premultiply.c:155-230 makes double output);
 *   - thumbnail_image of images with fewer than 3 bands other than one-band uchar B_W, not linear -- grey +
 *     alpha, GREY16, linear one-band (resample/thumbnail.c:806-820) -- and with the content-driven crops
 *     (entropy, attention: conversion/smartcrop.c).
 */
static gboolean
hip_wants_original(VipsHipOp *op, VipsImage *in)
{
	const char *nick = VIPS_OBJECT_GET_CLASS(op)->nickname;

	if (vips_band_format_iscomplex(in->BandFmt))
		return TRUE;
	if ((strcmp(nick, "premultiply_hip") == 0 || strcmp(nick, "unpremultiply_hip") == 0) &&
		in->BandFmt == VIPS_FORMAT_DOUBLE)
		return TRUE;
	if (strcmp(nick, "thumbnail_image_hip") == 0) {
		gboolean linear = FALSE;
		int crop = 0;

		g_object_get(op, "linear", &linear, "crop", &crop, NULL);
		if (crop == VIPS_INTERESTING_ENTROPY || crop == VIPS_INTERESTING_ATTENTION)
			return TRUE;
		if (in->Bands < 3 &&
			(linear || in->Bands != 1 || in->BandFmt != VIPS_FORMAT_UCHAR ||
				vips_image_guess_interpretation(in) != VIPS_INTERPRETATION_B_W))
			return TRUE;
	}
	return FALSE;
}

/* `out` = the original operation's output, as libvips' own composite operations hand a sub-operation's
 * image on (vips_image_write: a pipeline link, no pixels now; it keeps the original's image alive). */
static int
hip_delegate(VipsHipOp *op)
{
	VipsOperation *twin;
	VipsImage *twin_out = NULL;
	int result;

	if (!(twin = hip_twin_build(op)))
		return -1;
	g_object_get(twin, "out", &twin_out, NULL);
	g_object_set(op, "out", vips_image_new(), NULL);
	result = vips_image_write(twin_out, op->out);
	g_object_unref(twin_out);
	vips_object_unref_outputs(VIPS_OBJECT(twin));
	g_object_unref(twin);

	return result;
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
	(void) class;
	if (hip_wants_original(op, in))
		return hip_delegate(op);
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

	/* the producer holds no reference: it ends here, before anything it reads goes away */
	g_mutex_lock(&op->lock);
	op->producer_quit = TRUE;
	g_cond_broadcast(&op->cond);
	g_cond_signal(&op->pcond);
	g_mutex_unlock(&op->lock);
	if (op->producer) {
		g_thread_join(op->producer);
		op->producer = NULL;
	}
	hip_cache_free(op->cache);
	op->cache = NULL;
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
	g_cond_init(&op->cond);
	g_cond_init(&op->pcond);
}

/* ------------------------------------------------------------------ subclasses */

/* (the 17 operation classes: arguments, defaults, hooks) */
#include "vips_hip_classes.c"

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
