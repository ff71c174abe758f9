// Image-level resample operations: the host side of vips_reduceh/reducev/reduce/
// shrinkh/shrinkv/shrink/resize.  Each function does what the corresponding
// class build() does in the reference (sizes, integer pre-shrink under `gap`,
// table construction) and then fills the whole output with ONE region op --
// on a 288 GB device the natural "tile" is the image.
#include "internal.h"
#include "resample.h"
#include "colour.h"
#include "reduce_u8.h"

#include <cmath>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>

using namespace vh;

static int g_fatstrip_height = 16; // include/vips/private.h:147-153

extern "C" void vips_hip_set_fatstrip_height(int lines)
{
	g_fatstrip_height = lines;
}

namespace {

// The operation cache (iofuncs/cache.c:990, vips_cache_max = 100): building a
// reduce means 65 x n_point sin() evaluations and two table uploads, so built
// operations are kept and shared, exactly as the reference caches built
// VipsOperations.  Entries are handed out as shared_ptr so an eviction cannot
// pull tables out from under a running call.
struct ReduceKey {
	int device; // the tables of a built reduce live on one device
	int kernel, in_size, out_size;
	double shrink, extra;
	bool operator==(const ReduceKey &o) const
	{
		return device == o.device && kernel == o.kernel && in_size == o.in_size && out_size == o.out_size &&
			memcmp(&shrink, &o.shrink, sizeof(double)) == 0 &&
			memcmp(&extra, &o.extra, sizeof(double)) == 0;
	}
};

typedef std::shared_ptr<VipsHipReduce> ReducePtr;

// leaked on purpose (no device frees from static destructors at exit)
std::mutex &g_cache_mutex = *new std::mutex;
std::list<std::pair<ReduceKey, ReducePtr>> &g_cache =
	*new std::list<std::pair<ReduceKey, ReducePtr>>; // most recently used first
const size_t CACHE_MAX = 100;

ReducePtr reduce_cached(int kernel, double shrink, int in_size, int out_size, double extra)
{
	if (std::isnan(extra))
		extra = out_size * shrink - in_size;
	if (ensure_init())
		return ReducePtr();
	ReduceKey key = { current_device(), kernel, in_size, out_size, shrink, extra };
	{
		std::lock_guard<std::mutex> lock(g_cache_mutex);
		for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
			if (it->first == key) {
				g_cache.splice(g_cache.begin(), g_cache, it);
				return g_cache.front().second;
			}
	}
	VipsHipReduce *raw = vips_hip_reduce_new(kernel, shrink, in_size, out_size, extra);
	if (!raw)
		return ReducePtr();
	ReducePtr r(raw, vips_hip_reduce_free);
	std::lock_guard<std::mutex> lock(g_cache_mutex);
	g_cache.emplace_front(key, r);
	while (g_cache.size() > CACHE_MAX)
		g_cache.pop_back();
	return r;
}

struct ImageRef {
	VipsHipImage *im;
	explicit ImageRef(VipsHipImage *i = nullptr)
		: im(i)
	{
	}
	~ImageRef() { vips_hip_image_unref(im); }
	VipsHipImage *release()
	{
		VipsHipImage *t = im;
		im = nullptr;
		return t;
	}
};

VipsHipImage *like(const VipsHipImage *in, int width, int height)
{
	return vips_hip_image_new(width, height, in->bands, in->format, in->interpretation);
}

// A new reference to the same pixels: vips_image_write()'s pointer copy
// (iofuncs/image.c:2610-2646) for the "factor is 1" early returns.
VipsHipImage *copy_image(const VipsHipImage *in)
{
	VipsHipImage *out = like(in, in->width, in->height);
	if (!out)
		return nullptr;
	if (vips_hip_memcpy_d2d(out->data, in->data, in->stride * in->height)) {
		vips_hip_image_unref(out);
		return nullptr;
	}
	return out;
}

int shrink_axis(VipsHipImage *in, VipsHipImage **out, int shrink, int ceil_mode, bool vertical)
{
	const char *domain = vertical ? "shrinkv" : "shrinkh";
	if (!in || !out) {
		error(domain, "null argument");
		return -1;
	}
	if (shrink < 1) {
		error(domain, "shrink factors should be >= 1");
		return -1;
	}
	if (shrink == 1) {
		*out = copy_image(in);
		return *out ? 0 : -1;
	}
	int size = vips_hip_shrink_out_size(vertical ? in->height : in->width, shrink, ceil_mode);
	if (size <= 0) {
		error(domain, "image has shrunk to nothing");
		return -1;
	}
	ImageRef o(vertical ? like(in, in->width, size) : like(in, size, in->height));
	if (!o.im)
		return -1;
	VipsHipRegion ri, ro;
	vips_hip_image_region(in, &ri);
	vips_hip_image_region(o.im, &ro);
	int result = vertical ? vips_hip_shrinkv_gen(shrink, &ri, &ro)
						  : vips_hip_shrinkh_gen(shrink, &ri, &ro);
	if (result)
		return -1;
	*out = o.release();
	return 0;
}

// vips_reduceh_build / vips_reducev_build up to the generate:
// reduceh.cpp:396-481, reducev.cpp:859-941.  The decisions first (sizes, the integer pre-shrink
// of `gap`, the residual factor), then the pre-shrink, then the residual reduce.
struct AxisPlan {
	int size;            // output size
	int int_shrink;      // box shrink ahead of the reduce (1 = none)
	double residual;     // what the reduce does
	double extra_pixels;
};

int plan_axis(const char *domain, int in_size, double shrink, int kernel, double gap, AxisPlan *p)
{
	if (shrink < 1.0) {
		error(domain, "reduce factor should be >= 1.0");
		return -1;
	}
	// "We need to always round to nearest, so round(), not rint()."
	p->size = (int) ((double) in_size / shrink + 0.5);
	p->extra_pixels = p->size * shrink - in_size;
	p->residual = shrink;
	p->int_shrink = 1;
	if (gap > 0.0 && kernel != VIPS_HIP_KERNEL_NEAREST) {
		if (gap < 1.0) {
			error(domain, "reduce gap should be >= 1.0");
			return -1;
		}
		if (p->size <= 0) {
			error(domain, "image has shrunk to nothing");
			return -1;
		}
		int int_shrink = (int) floor((double) in_size / p->size / gap);
		if (int_shrink > 1) {
			p->int_shrink = int_shrink;
			p->residual /= int_shrink;
			p->extra_pixels /= int_shrink;
		}
	}
	return 0;
}

// the residual reduce of an axis whose pre-shrink (if any) has run: `cur` is its result
int residual_axis(const char *domain, VipsHipImage *cur, VipsHipImage **out, const AxisPlan &p, int kernel,
	bool vertical)
{
	if (p.residual == 1.0) {
		*out = copy_image(cur);
		return *out ? 0 : -1;
	}
	if (p.size <= 0) {
		error(domain, "image has shrunk to nothing");
		return -1;
	}
	ReducePtr rp = reduce_cached(kernel, p.residual, vertical ? cur->height : cur->width, p.size,
		p.extra_pixels);
	if (!rp)
		return -1;
	VipsHipReduce *r = rp.get();
	ImageRef o(vertical ? like(cur, cur->width, p.size) : like(cur, p.size, cur->height));
	int result = -1;
	if (o.im) {
		VipsHipRegion ri, ro;
		vips_hip_image_region(cur, &ri);
		vips_hip_image_region(o.im, &ro);
		// reducev is evaluated by the reference in fatstrip-high generate calls
		// (each re-seeding Y, reducev.cpp:548); reduceh's X is seeded at
		// r->left = 0 of full-width strips (thread.c:301-325).
		result = vertical ? vips_hip_reducev_gen_tiled(r, &ri, &ro, g_fatstrip_height)
						  : vips_hip_reduceh_gen(r, &ri, &ro);
	}
	if (result)
		return -1;
	*out = o.release();
	return 0;
}

int reduce_axis(VipsHipImage *in, VipsHipImage **out, double shrink, int kernel, double gap,
	bool vertical)
{
	const char *domain = vertical ? "reducev" : "reduceh";
	if (!in || !out) {
		error(domain, "null argument");
		return -1;
	}
	AxisPlan p;
	if (plan_axis(domain, vertical ? in->height : in->width, shrink, kernel, gap, &p))
		return -1;
	ImageRef pre;
	VipsHipImage *cur = in;
	if (p.int_shrink > 1) {
		if (shrink_axis(in, &pre.im, p.int_shrink, 1, vertical))
			return -1;
		cur = pre.im;
	}
	return residual_axis(domain, cur, out, p, kernel, vertical);
}

// n uchar images of one size through vips_resize's whole downsizing chain in one launch
// (resize_stream.hip): 0 = done (out[] filled), 1 = not that kernel's case (nothing done), -1 = error
// premul: the images are RGBA and vips_premultiply(uchar) comes first (vips_thumbnail_image, thumbnail.c:848-860): only
// the chain of band kernels can do that on its loads -- anything else answers 1 and the caller premultiplies itself
int resize_down_u8_stream(VipsHipImage *const *in, int n, VipsHipImage **out, const AxisPlan &pv, const AxisPlan &ph,
	int shrunk_width, int kernel, bool premul = false)
{
	if (pv.residual == 1.0 || ph.residual == 1.0 || kernel == VIPS_HIP_KERNEL_NEAREST || n < 1)
		return 1;
	const int shrunk_height =
		pv.int_shrink > 1 ? vips_hip_shrink_out_size(in[0]->height, pv.int_shrink, 1) : in[0]->height;
	if (shrunk_height <= 0)
		return 1;
	ReducePtr rv = reduce_cached(kernel, pv.residual, shrunk_height, pv.size, pv.extra_pixels);
	ReducePtr rh = rv ? reduce_cached(kernel, ph.residual, shrunk_width, ph.size, ph.extra_pixels) : ReducePtr();
	if (!rv || !rh)
		return -1;
	std::vector<ImageRef> o(n);
	std::vector<VipsHipRegion> ri(n), ro(n);
	std::vector<const VipsHipRegion *> pi(n), po(n);
	for (int i = 0; i < n; i++) {
		if (in[i]->width != in[0]->width || in[i]->height != in[0]->height || in[i]->bands != in[0]->bands ||
			in[i]->format != VIPS_HIP_FORMAT_UCHAR)
			return 1;
		o[i].im = like(in[i], ph.size, pv.size);
		if (!o[i].im)
			return -1;
		vips_hip_image_region(in[i], &ri[i]);
		vips_hip_image_region(o[i].im, &ro[i]);
		pi[i] = &ri[i];
		po[i] = &ro[i];
	}
	// residual reduces of exactly 2 on both axes: the static-rotation kernel (resize_stream.hip);
	// any other: the scheduled one (resize_streamg.hip)
	int done = 0;
	if (pv.residual == 2.0 && ph.residual == 2.0 && !premul)
		done = resize_stream_u8_try(rv.get(), pv.int_shrink, rh.get(), ph.int_shrink, shrunk_height, shrunk_width,
			pi.data(), po.data(), n, g_fatstrip_height);
	// any other residual (a size that does not divide the image): shrinkv + reducev as one matrix-core kernel, then
	// shrinkh, then reduceh on the matrix cores (reduce_band.hip) -- three launches that beat the one-kernel chain
	// of resize_streamg.hip (8192^2 x 3 to 1000^2: 0.0515 against 0.079 ms; profiles/NOTES.md R5.5, R5.6)
	// (three launches per image, every one of them spread over the whole image: a single image of any size -- the
	// one-kernel chain walks a small image's rows with a handful of blocks, 50 us against 13 at 1024^2 x 3,
	// tools/band_threshold.py -- and a batch of images of 8 MB and more; a batch of small ones stays with the
	// one-kernel chain, 64 images a launch.  VIPS_HIP_RESIZE_BAND_MIN = that size in bytes)
	const long long band_min = getenv("VIPS_HIP_RESIZE_BAND_MIN") ? atoll(getenv("VIPS_HIP_RESIZE_BAND_MIN")) : 8LL << 20;
	if (done == 0 && !getenv("VIPS_HIP_NO_RESIZE_BAND") && !getenv("VIPS_HIP_STREAMG_ALWAYS") &&
		(premul || n < 4 || (long long) in[0]->width * in[0]->height * in[0]->bands >= band_min)) {
		done = 1;
		for (int i = 0; i < n && done == 1; i++) {
			ImageRef t1(like(in[i], in[i]->width, pv.size));
			if (!t1.im)
				return -1;
			VipsHipRegion r1;
			vips_hip_image_region(t1.im, &r1);
			const int d = shrinkv_reducev_band_try(rv.get(), pv.int_shrink, shrunk_height, pi[i], &r1, g_fatstrip_height, premul);
			if (d < 0)
				return -1;
			if (d == 0) {
				// not its case (a refusal can depend on ONE image: the alignment of caller-wrapped device memory):
				// the one-kernel chain below takes the WHOLE batch and rewrites whatever outputs the images before
				// this one already have -- harmless, every output is written whole
				done = 0;
				break;
			}
			ImageRef t2;
			VipsHipRegion r2 = r1;
			if (ph.int_shrink > 1) {
				// (rows padded to whole dwords: the kernels either side read and write them as dwords)
				t2.im = like(in[i], (shrunk_width + 3) & ~3, pv.size);
				if (!t2.im)
					return -1;
				vips_hip_image_region(t2.im, &r2);
				r2.width = r2.im_width = shrunk_width;
				if (vips_hip_shrinkh_gen(ph.int_shrink, &r1, &r2))
					return -1;
			}
			if (vips_hip_reduceh_gen(rh.get(), &r2, po[i]))
				return -1;
		}
	}
	if (done == 0 && premul)
		return 1; // (un-premultiplied pixels may have reached outputs of the batch: they are dropped with o[])
	if (done == 0)
		done = resize_streamg_u8_try(rv.get(), pv.int_shrink, rh.get(), ph.int_shrink, shrunk_height, shrunk_width,
			pi.data(), po.data(), n, g_fatstrip_height);
	if (done < 0)
		return -1;
	if (done == 0)
		return 1;
	for (int i = 0; i < n; i++)
		out[i] = o[i].release();
	return 0;
}

// n uchar images of one size through the vertical box shrink and the fused tail (reducev ->
// shrinkh -> reduceh, resize_tail.hip), each one launch per 64 images: any scale the tail
// kernel takes.  0 = done (out[] filled), 1 = not its case (nothing done), -1 = error
int resize_down_u8_tail_batch(VipsHipImage *const *in, int n, VipsHipImage **out, const AxisPlan &pv,
	const AxisPlan &ph, int shrunk_width, int kernel)
{
	if (pv.residual == 1.0 || ph.residual == 1.0 || n < 1)
		return 1;
	for (int i = 0; i < n; i++)
		if (in[i]->width != in[0]->width || in[i]->height != in[0]->height || in[i]->bands != in[0]->bands ||
			in[i]->format != VIPS_HIP_FORMAT_UCHAR)
			return 1;
	const int shrunk_height =
		pv.int_shrink > 1 ? vips_hip_shrink_out_size(in[0]->height, pv.int_shrink, 1) : in[0]->height;
	if (shrunk_height <= 0)
		return 1;
	ReducePtr rv = reduce_cached(kernel, pv.residual, shrunk_height, pv.size, pv.extra_pixels);
	ReducePtr rh = rv ? reduce_cached(kernel, ph.residual, shrunk_width, ph.size, ph.extra_pixels) : ReducePtr();
	if (!rv || !rh)
		return -1;
	std::vector<ImageRef> pre(n), o(n);
	std::vector<VipsHipRegion> ri(n), rm(n), ro(n);
	std::vector<const VipsHipRegion *> pi(n), pm(n), po(n);
	for (int i = 0; i < n; i++) {
		vips_hip_image_region(in[i], &ri[i]);
		pi[i] = &ri[i];
		if (pv.int_shrink > 1) {
			pre[i].im = like(in[i], in[i]->width, shrunk_height);
			if (!pre[i].im)
				return -1;
			vips_hip_image_region(pre[i].im, &rm[i]);
		}
		else
			rm[i] = ri[i];
		pm[i] = &rm[i];
		o[i].im = like(in[i], ph.size, pv.size);
		if (!o[i].im)
			return -1;
		vips_hip_image_region(o[i].im, &ro[i]);
		po[i] = &ro[i];
	}
	if (pv.int_shrink > 1) {
		const int done = shrinkv_u8_batch_try(pv.int_shrink, pi.data(), pm.data(), n);
		if (done < 0)
			return -1;
		if (done == 0)
			return 1;
	}
	const int done = resize_tail_u8_try(rv.get(), ph.int_shrink, shrunk_width, rh.get(), pm.data(), po.data(), n,
		g_fatstrip_height);
	if (done < 0)
		return -1;
	if (done == 0)
		return 1; // (a shrinkv already queued wrote only to images dropped here)
	for (int i = 0; i < n; i++)
		out[i] = o[i].release();
	return 0;
}

// vips_resize's downsizing of a uchar image on both axes (resize.c:207-228: reducev with its
// box pre-shrink, then reduceh with its own): the vertical box shrink, then everything else in
// one kernel (resize_tail.hip) when the geometry fits it.  1 = not this function's case.
int resize_down_u8(VipsHipImage *in, VipsHipImage **out, double vshrink, double hshrink, int kernel, double gap,
	bool premul = false)
{
	AxisPlan pv, ph;
	if (plan_axis("reducev", in->height, vshrink, kernel, gap, &pv) ||
		plan_axis("reduceh", in->width, hshrink, kernel, gap, &ph))
		return -1;
	if (pv.residual == 1.0 || ph.residual == 1.0 || pv.size <= 0 || ph.size <= 0)
		return 1;
	const int shrunk_width = ph.int_shrink > 1 ? vips_hip_shrink_out_size(in->width, ph.int_shrink, 1) : in->width;
	if (shrunk_width <= 0)
		return 1;
	{
		// all four operations in one kernel (resize_stream.hip, resize_streamg.hip)
		const int done = resize_down_u8_stream(&in, 1, out, pv, ph, shrunk_width, kernel, premul);
		if (done <= 0 || premul)
			return done;
	}
	ImageRef pre;
	VipsHipImage *cur = in;
	if (pv.int_shrink > 1) {
		if (shrink_axis(in, &pre.im, pv.int_shrink, 1, true))
			return -1;
		cur = pre.im;
	}
	ReducePtr rv = reduce_cached(kernel, pv.residual, cur->height, pv.size, pv.extra_pixels);
	ReducePtr rh = rv ? reduce_cached(kernel, ph.residual, shrunk_width, ph.size, ph.extra_pixels) : ReducePtr();
	if (!rv || !rh)
		return -1;
	ImageRef o(like(cur, ph.size, pv.size));
	if (!o.im)
		return -1;
	VipsHipRegion ri, ro;
	vips_hip_image_region(cur, &ri);
	vips_hip_image_region(o.im, &ro);
	const VipsHipRegion *pri = &ri, *pro = &ro;
	const int done = resize_tail_u8_try(rv.get(), ph.int_shrink, shrunk_width, rh.get(), &pri, &pro, 1,
		g_fatstrip_height);
	if (done < 0)
		return -1;
	if (done > 0) {
		*out = o.release();
		return 0;
	}
	// not the kernel's geometry: the separate operations, from the pre-shrunk image on
	ImageRef t;
	if (residual_axis("reducev", cur, &t.im, pv, kernel, true))
		return -1;
	return reduce_axis(t.im, out, hshrink, kernel, gap, false);
}

} // namespace

namespace vh {

// vips_resize(scale) of n uchar images of one size in one launch, for the batch entry point:
// 0 = done, 1 = neither the streaming kernel's nor the fused tail's case (the caller resizes image
// by image), -1 = error
int resize_batch_u8(VipsHipImage *const *in, int n, VipsHipImage **out, double scale, int kernel, double gap)
{
	if (n < 1 || !in[0] || in[0]->format != VIPS_HIP_FORMAT_UCHAR || kernel == VIPS_HIP_KERNEL_NEAREST)
		return 1;
	if (gap < 0.0)
		gap = 2.0; // resize.c:397
	if (!(scale > 0.0) || scale >= 1.0 || scale < 1.0 / in[0]->width || scale < 1.0 / in[0]->height)
		return 1;
	for (int i = 1; i < n; i++)
		if (!in[i])
			return 1;
	AxisPlan pv, ph;
	if (plan_axis("reducev", in[0]->height, 1.0 / scale, kernel, gap, &pv) ||
		plan_axis("reduceh", in[0]->width, 1.0 / scale, kernel, gap, &ph))
		return -1;
	if (pv.size <= 0 || ph.size <= 0)
		return 1;
	const int shrunk_width =
		ph.int_shrink > 1 ? vips_hip_shrink_out_size(in[0]->width, ph.int_shrink, 1) : in[0]->width;
	if (shrunk_width <= 0)
		return 1;
	const int done = resize_down_u8_stream(in, n, out, pv, ph, shrunk_width, kernel);
	if (done <= 0)
		return done;
	return resize_down_u8_tail_batch(in, n, out, pv, ph, shrunk_width, kernel);
}

// vips_resize(scale) then vips_sharpen (blur mask `coef` / `mask_scale` as convi's integers, LUT
// `lut`: 65536 host ints) of n 3-band uchar sRGB images of one size, ONE kernel for both
// (resize_sharpen.hip): 0 = done, 1 = not that kernel's case (nothing done), -1 = error
int resize_sharpen_batch_u8(VipsHipImage *const *in, int n, VipsHipImage **out, double scale, int kernel, double gap,
	const int *coef, int ncoef, int mask_scale, const int *lut)
{
	if (n < 1 || !in[0] || in[0]->format != VIPS_HIP_FORMAT_UCHAR || in[0]->bands != 3 || kernel == VIPS_HIP_KERNEL_NEAREST)
		return 1;
	if (gap < 0.0)
		gap = 2.0; // resize.c:397
	if (!(scale > 0.0) || scale >= 1.0 || scale < 1.0 / in[0]->width || scale < 1.0 / in[0]->height)
		return 1;
	for (int i = 1; i < n; i++)
		if (!in[i] || in[i]->width != in[0]->width || in[i]->height != in[0]->height || in[i]->bands != 3 ||
			in[i]->format != VIPS_HIP_FORMAT_UCHAR)
			return 1;
	AxisPlan pv, ph;
	if (plan_axis("reducev", in[0]->height, 1.0 / scale, kernel, gap, &pv) ||
		plan_axis("reduceh", in[0]->width, 1.0 / scale, kernel, gap, &ph))
		return -1;
	if (pv.size <= 0 || ph.size <= 0 || pv.residual != 2.0 || ph.residual != 2.0)
		return 1;
	const int shrunk_width = ph.int_shrink > 1 ? vips_hip_shrink_out_size(in[0]->width, ph.int_shrink, 1) : in[0]->width;
	const int shrunk_height = pv.int_shrink > 1 ? vips_hip_shrink_out_size(in[0]->height, pv.int_shrink, 1) : in[0]->height;
	if (shrunk_width <= 0 || shrunk_height <= 0)
		return 1;
	ReducePtr rv = reduce_cached(kernel, pv.residual, shrunk_height, pv.size, pv.extra_pixels);
	ReducePtr rh = rv ? reduce_cached(kernel, ph.residual, shrunk_width, ph.size, ph.extra_pixels) : ReducePtr();
	if (!rv || !rh)
		return -1;
	std::vector<ImageRef> o(n);
	std::vector<VipsHipRegion> ri(n), ro(n);
	std::vector<const VipsHipRegion *> pi(n), po(n);
	for (int i = 0; i < n; i++) {
		o[i].im = like(in[i], ph.size, pv.size);
		if (!o[i].im)
			return -1;
		vips_hip_image_region(in[i], &ri[i]);
		vips_hip_image_region(o[i].im, &ro[i]);
		pi[i] = &ri[i];
		po[i] = &ro[i];
	}
	const int done = resize_sharpen_stream_u8_try(rv.get(), pv.int_shrink, rh.get(), ph.int_shrink, shrunk_height,
		shrunk_width, pi.data(), po.data(), n, g_fatstrip_height, coef, ncoef, mask_scale, lut);
	if (done < 0)
		return -1;
	if (done == 0)
		return 1;
	for (int i = 0; i < n; i++)
		out[i] = o[i].release();
	return 0;
}

// vips_premultiply(uchar) + vips_resize(hscale, vscale, lanczos3, gap 2) of an RGBA uchar image (max_alpha 255) with the
// premultiply on the loads of the first kernel (reduce_band.hip: thumbnail.c:848-860 makes a whole premultiplied image
// first).  0 done, 1 not covered (the caller runs the two operations), -1 error.
int resize_premul_u8(VipsHipImage *in, VipsHipImage **out, double hscale, double vscale)
{
	if (!in || !out || in->format != VIPS_HIP_FORMAT_UCHAR || in->bands != 4 || !(hscale > 0.0 && hscale < 1.0) ||
		!(vscale > 0.0 && vscale < 1.0))
		return 1;
	// (vips_resize_build's floors: "Don't let either axis drop below 1 px")
	if (hscale < 1.0 / in->width || vscale < 1.0 / in->height)
		return 1;
	return resize_down_u8(in, out, 1.0 / vscale, 1.0 / hscale, VIPS_HIP_KERNEL_LANCZOS3, 2.0, true);
}

} // namespace vh

extern "C" {

int vips_hip_shrinkh(VipsHipImage *in, VipsHipImage **out, int hshrink, int ceil_mode)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	return shrink_axis(in, out, hshrink, ceil_mode, false);
}

int vips_hip_shrinkv(VipsHipImage *in, VipsHipImage **out, int vshrink, int ceil_mode)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	return shrink_axis(in, out, vshrink, ceil_mode, true);
}

int vips_hip_reduceh(VipsHipImage *in, VipsHipImage **out, double hshrink, int kernel, double gap)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	return reduce_axis(in, out, hshrink, kernel, gap, false);
}

int vips_hip_reducev(VipsHipImage *in, VipsHipImage **out, double vshrink, int kernel, double gap)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	return reduce_axis(in, out, vshrink, kernel, gap, true);
}

// vips_reduce_build, resample/reduce.c:98-121: vertical first.
int vips_hip_reduce(VipsHipImage *in, VipsHipImage **out, double hshrink, double vshrink,
	int kernel, double gap)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("reduce", "null argument");
		return -1;
	}
	// Fused uchar path when neither axis needs an integer pre-shrink.
	if (in->format == VIPS_HIP_FORMAT_UCHAR && gap <= 0.0 && hshrink > 1.0 && vshrink > 1.0 &&
		kernel != VIPS_HIP_KERNEL_NEAREST) {
		const int height = (int) ((double) in->height / vshrink + 0.5);
		const int width = (int) ((double) in->width / hshrink + 0.5);
		if (width > 0 && height > 0) {
			ReducePtr rvp = reduce_cached(kernel, vshrink, in->height, height, NAN);
			ReducePtr rhp = rvp ? reduce_cached(kernel, hshrink, in->width, width, NAN) : ReducePtr();
			VipsHipReduce *rv = rvp.get(), *rh = rhp.get();
			if (rv && rh) {
				ImageRef o(like(in, width, height));
				int result = -1;
				if (o.im) {
					VipsHipRegion ri, ro;
					vips_hip_image_region(in, &ri);
					vips_hip_image_region(o.im, &ro);
					result = vips_hip_reduce_gen_tiled(rv, rh, &ri, &ro, g_fatstrip_height);
				}
				if (result < 0)
					return -1;
				if (result == 0) {
					*out = o.release();
					return 0;
				}
				// result > 0: geometry not covered by the fused kernel
			}
			else
				return -1;
		}
	}

	ImageRef t0;
	if (reduce_axis(in, &t0.im, vshrink, kernel, gap, true))
		return -1;
	return reduce_axis(t0.im, out, hshrink, kernel, gap, false);
}

// vips_shrink_build, resample/shrink.c:77-119
int vips_hip_shrink(VipsHipImage *in, VipsHipImage **out, double hshrink, double vshrink,
	int ceil_mode)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("shrink", "null argument");
		return -1;
	}
	const int hshrink_int = (int) hshrink;
	const int vshrink_int = (int) vshrink;
	ImageRef t0;
	if (hshrink_int != hshrink || vshrink_int != vshrink) {
		if (reduce_axis(in, &t0.im, vshrink, VIPS_HIP_KERNEL_LANCZOS3, 1.0, true))
			return -1;
		return reduce_axis(t0.im, out, hshrink, VIPS_HIP_KERNEL_LANCZOS3, 1.0, false);
	}
	// ushort, both factors above 1: the two box shrinks in one kernel (resample16: the image read once, the
	// 1 / vshrink-size intermediate never made)
	if (in->format == VIPS_HIP_FORMAT_USHORT && hshrink_int > 1 && vshrink_int > 1) {
		const int h1 = vips_hip_shrink_out_size(in->height, vshrink_int, ceil_mode);
		const int w1 = vips_hip_shrink_out_size(in->width, hshrink_int, ceil_mode);
		if (h1 > 0 && w1 > 0) {
			ImageRef o(like(in, w1, h1));
			if (!o.im)
				return -1;
			VipsHipRegion ri, ro;
			vips_hip_image_region(in, &ri);
			vips_hip_image_region(o.im, &ro);
			const int done = vh::shrinkbox16_try(hshrink_int, vshrink_int, &ri, &ro);
			if (done < 0)
				return -1;
			if (done > 0) {
				*out = o.release();
				return 0;
			}
		}
	}
	if (shrink_axis(in, &t0.im, vshrink_int, ceil_mode, true))
		return -1;
	return shrink_axis(t0.im, out, hshrink_int, ceil_mode, false);
}

// vips_resize_build, resample/resize.c:135-329.
int vips_hip_resize(VipsHipImage *in, VipsHipImage **out, double scale, double vscale_arg,
	int kernel, double gap)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("resize", "null argument");
		return -1;
	}
	double hscale = scale;
	double vscale = vscale_arg > 0.0 ? vscale_arg : scale;
	if (gap < 0.0)
		gap = 2.0; // resize.c:397
	if (hscale <= 0.0 || vscale <= 0.0) {
		error("resize", "scale must be > 0");
		return -1;
	}
	ImageRef t1, t2, t3;
	VipsHipImage *cur = in;
	if (kernel == VIPS_HIP_KERNEL_NEAREST) {
		// the int part of the scale by vips_subsample, resize.c:165-203
		int int_hshrink, int_vshrink;
		if (gap < 1.0) {
			int_hshrink = (int) floor(1.0 / hscale);
			int_vshrink = (int) floor(1.0 / vscale);
		}
		else {
			const int target_width = (int) (in->width * hscale + 0.5);   // VIPS_ROUND_UINT
			const int target_height = (int) (in->height * vscale + 0.5);
			if (target_width <= 0 || target_height <= 0) {
				error("resize", "image has shrunk to nothing");
				return -1;
			}
			int_hshrink = (int) floor((double) in->width / target_width / gap);
			int_vshrink = (int) floor((double) in->height / target_height / gap);
		}
		int_hshrink = int_hshrink > 1 ? int_hshrink : 1;
		int_vshrink = int_vshrink > 1 ? int_vshrink : 1;
		if (int_vshrink > 1 || int_hshrink > 1) {
			const int sw = in->width / int_hshrink, sh = in->height / int_vshrink;
			if (sw <= 0 || sh <= 0) {
				error("subsample", "image has shrunk to nothing");
				return -1;
			}
			t1.im = vips_hip_image_new(sw, sh, in->bands, in->format, in->interpretation);
			if (!t1.im)
				return -1;
			VipsHipRegion ri, ro;
			vips_hip_image_region(in, &ri);
			vips_hip_image_region(t1.im, &ro);
			if (vips_hip_subsample_gen(&ri, &ro, int_hshrink, int_vshrink))
				return -1;
			cur = t1.im;
			hscale *= int_hshrink;
			vscale *= int_vshrink;
		}
	}
	// "Don't let either axis drop below 1 px."
	if (hscale < 1.0 / cur->width)
		hscale = 1.0 / cur->width;
	if (vscale < 1.0 / cur->height)
		vscale = 1.0 / cur->height;

	// any residual downsizing (the integer pre-shrink of the other kernels lives in reduce_axis)
	if (vscale < 1.0 && hscale < 1.0 && cur->format == VIPS_HIP_FORMAT_UCHAR && kernel != VIPS_HIP_KERNEL_NEAREST) {
		const int done = resize_down_u8(cur, &t2.im, 1.0 / vscale, 1.0 / hscale, kernel, gap);
		if (done < 0)
			return -1;
		if (done == 0) {
			*out = t2.release();
			return 0;
		}
	}
	if (vscale < 1.0) {
		if (reduce_axis(cur, &t2.im, 1.0 / vscale, kernel, gap, true))
			return -1;
		cur = t2.im;
	}
	if (hscale < 1.0) {
		if (reduce_axis(cur, &t3.im, 1.0 / hscale, kernel, gap, false))
			return -1;
		cur = t3.im;
	}

	// any upsizing, resize.c:230-300
	if (hscale > 1.0 || vscale > 1.0) {
		const int interpolate = kernel == VIPS_HIP_KERNEL_NEAREST ? VIPS_HIP_INTERPOLATE_NEAREST
			: kernel == VIPS_HIP_KERNEL_LINEAR                     ? VIPS_HIP_INTERPOLATE_BILINEAR
																   : VIPS_HIP_INTERPOLATE_BICUBIC;
		// "For centre sampling, shift by 0.5 down and right.  Except if this is nearest"
		const double idx = kernel == VIPS_HIP_KERNEL_NEAREST ? 0.0 : 0.5 * (1.0 - 1.0 / hscale);
		const double idy = kernel == VIPS_HIP_KERNEL_NEAREST ? 0.0 : 0.5 * (1.0 - 1.0 / vscale);
		VipsHipRegion ri, ro;
		vips_hip_image_region(cur, &ri);
		if (kernel == VIPS_HIP_KERNEL_NEAREST && hscale == floor(hscale) && vscale == floor(vscale)) {
			const int xfac = (int) floor(hscale), yfac = (int) floor(vscale);
			ImageRef o(vips_hip_image_new(cur->width * xfac, cur->height * yfac, cur->bands, cur->format,
				cur->interpretation));
			if (!o.im)
				return -1;
			vips_hip_image_region(o.im, &ro);
			if (vips_hip_zoom_gen(&ri, &ro, xfac, yfac))
				return -1;
			*out = o.release();
			return 0;
		}
		// one axis only: the other one's scale is exactly 1 (its displacement is still passed)
		const double a = hscale > 1.0 ? hscale : 1.0;
		const double d = vscale > 1.0 ? vscale : 1.0;
		ImageRef o(vips_hip_image_new(vips_hip_affine_out_size(cur->width, a),
			vips_hip_affine_out_size(cur->height, d), cur->bands, cur->format, cur->interpretation));
		if (!o.im)
			return -1;
		vips_hip_image_region(o.im, &ro);
		if (vips_hip_upsize_gen(&ri, &ro, a, d, idx, idy, interpolate, 0))
			return -1;
		*out = o.release();
		return 0;
	}

	if (cur == in) {
		*out = copy_image(in);
		return *out ? 0 : -1;
	}
	*out = cur == t3.im ? t3.release() : cur == t2.im ? t2.release() : t1.release();
	return 0;
}

} // extern "C"
