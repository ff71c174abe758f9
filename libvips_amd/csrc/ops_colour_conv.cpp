// Image-level convolution and colour operations: the host side of vips_conv /
// vips_convsep / vips_gaussblur / vips_sharpen / vips_colourspace / vips_cast.
// Each mirrors the build() of the corresponding reference class and then runs the
// region ops once over the whole (device-resident) image.
#include "colour.h"
#include "conv.h"
#include "reduce_u8.h"

#include <atomic>
#include <thread>
#include <string>
#include <cmath>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

using namespace vh;

namespace {

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

// Operation cache (the role of iofuncs/cache.c for this path): a mask's device tables are
// built and uploaded once, not per call -- a table upload synchronises the stream, which on a
// 1024^2 thumbnail costs more than the kernels.  LRU, shared_ptr so an eviction cannot pull
// tables out from under a running call; leaked on purpose (no device frees at exit).
typedef std::shared_ptr<VipsHipConv> ConvPtr;

struct ConvKey {
	int device; // a mask's device tables live on one device
	std::vector<double> mask;
	int mw, mh, precision;
	double scale, offset;
	bool operator==(const ConvKey &o) const
	{
		return device == o.device && mw == o.mw && mh == o.mh && precision == o.precision &&
			memcmp(&scale, &o.scale, sizeof(double)) == 0 &&
			memcmp(&offset, &o.offset, sizeof(double)) == 0 && mask.size() == o.mask.size() &&
			memcmp(mask.data(), o.mask.data(), mask.size() * sizeof(double)) == 0;
	}
};

std::mutex &g_conv_mutex = *new std::mutex;
std::list<std::pair<ConvKey, ConvPtr>> &g_conv_cache = *new std::list<std::pair<ConvKey, ConvPtr>>;
const size_t CONV_CACHE_MAX = 64;

ConvPtr conv_cached(const double *mask, int mw, int mh, double scale, double offset, int precision)
{
	if (!mask || mw <= 0 || mh <= 0 || (long long) mw * mh > 65536)
		return ConvPtr(vips_hip_conv_new(mask, mw, mh, scale, offset, precision), vips_hip_conv_free);
	if (ensure_init())
		return ConvPtr();
	ConvKey key = { current_device(), std::vector<double>(mask, mask + (size_t) mw * mh), mw, mh, precision, scale, offset };
	{
		std::lock_guard<std::mutex> lock(g_conv_mutex);
		for (auto it = g_conv_cache.begin(); it != g_conv_cache.end(); ++it)
			if (it->first == key) {
				g_conv_cache.splice(g_conv_cache.begin(), g_conv_cache, it);
				return g_conv_cache.front().second;
			}
	}
	VipsHipConv *raw = vips_hip_conv_new(mask, mw, mh, scale, offset, precision);
	if (!raw)
		return ConvPtr();
	ConvPtr c(raw, vips_hip_conv_free);
	std::lock_guard<std::mutex> lock(g_conv_mutex);
	g_conv_cache.emplace_front(std::move(key), c);
	while (g_conv_cache.size() > CONV_CACHE_MAX)
		g_conv_cache.pop_back();
	return c;
}

// the same for the box / line decompositions of precision=approximate (approx.hip): the
// clustering is host work worth keeping
typedef std::shared_ptr<VipsHipConva> ConvaPtr;

struct ConvaKey {
	ConvKey mask; // precision field: 1 = separable
	int layers, cluster;
	bool operator==(const ConvaKey &o) const { return layers == o.layers && cluster == o.cluster && mask == o.mask; }
};

std::list<std::pair<ConvaKey, ConvaPtr>> &g_conva_cache = *new std::list<std::pair<ConvaKey, ConvaPtr>>;

ConvaPtr conva_cached(const double *mask, int mw, int mh, double scale, double offset, int layers, int cluster,
	bool separable)
{
	auto make = [&]() {
		return separable ? vips_hip_convasep_new(mask, mw * mh, scale, offset, layers)
						 : vips_hip_conva_new(mask, mw, mh, scale, offset, layers, cluster);
	};
	if (!mask || mw <= 0 || mh <= 0 || (long long) mw * mh > 65536)
		return ConvaPtr(make(), vips_hip_conva_free);
	if (ensure_init())
		return ConvaPtr();
	ConvaKey key = { { current_device(), std::vector<double>(mask, mask + (size_t) mw * mh), mw, mh, separable ? 1 : 0, scale, offset },
		layers, cluster };
	{
		std::lock_guard<std::mutex> lock(g_conv_mutex);
		for (auto it = g_conva_cache.begin(); it != g_conva_cache.end(); ++it)
			if (it->first == key) {
				g_conva_cache.splice(g_conva_cache.begin(), g_conva_cache, it);
				return g_conva_cache.front().second;
			}
	}
	VipsHipConva *raw = make();
	if (!raw)
		return ConvaPtr();
	ConvaPtr c(raw, vips_hip_conva_free);
	std::lock_guard<std::mutex> lock(g_conv_mutex);
	g_conva_cache.emplace_front(std::move(key), c);
	while (g_conva_cache.size() > CONV_CACHE_MAX)
		g_conva_cache.pop_back();
	return c;
}

// the sharpen LUT (sharpen.c:230-257), cached by its parameters
struct LutKey {
	int device;
	double p[5];
	bool operator==(const LutKey &o) const { return device == o.device && memcmp(p, o.p, sizeof(p)) == 0; }
};
typedef std::shared_ptr<int> LutPtr;
std::list<std::pair<LutKey, LutPtr>> &g_lut_cache = *new std::list<std::pair<LutKey, LutPtr>>;

// sharpen.c:230-257
void sharpen_lut_host(double x1, double y2, double y3, double m1, double m2, std::vector<int> &lut)
{
	lut.resize(65536);
	for (int i = 0; i < 65536; i++) {
		double v = (i - 32767) / 327.67;
		double y;

		if (v < -x1)
			y = (v + x1) * m2 + -x1 * m1;
		else if (v < x1)
			y = v * m1;
		else
			y = (v - x1) * m2 + x1 * m1;

		if (y < -y3)
			y = -y3;
		if (y > y2)
			y = y2;

		lut[i] = rint(y * 327.67);
	}
}

LutPtr sharpen_lut_cached(double x1, double y2, double y3, double m1, double m2)
{
	if (ensure_init())
		return LutPtr();
	LutKey key = { current_device(), { x1, y2, y3, m1, m2 } };
	{
		std::lock_guard<std::mutex> lock(g_conv_mutex);
		for (auto it = g_lut_cache.begin(); it != g_lut_cache.end(); ++it)
			if (it->first == key) {
				g_lut_cache.splice(g_lut_cache.begin(), g_lut_cache, it);
				return g_lut_cache.front().second;
			}
	}
	std::vector<int> lut;
	sharpen_lut_host(x1, y2, y3, m1, m2, lut);
	int *d_lut = (int *) upload(lut.data(), lut.size() * sizeof(int));
	if (!d_lut)
		return LutPtr();
	LutPtr l(d_lut, [](int *d) { vips_hip_free(d); });
	std::lock_guard<std::mutex> lock(g_conv_mutex);
	g_lut_cache.emplace_front(key, l);
	while (g_lut_cache.size() > 16)
		g_lut_cache.pop_back();
	return l;
}

// the window of sharpen's LUT that is not constant (sharpen.c:230-257 is flat outside the bends), on the device
// as shorts: what colour.hip's all-in-LDS sharpen kernel copies into LDS.  A window of n = 0 with lut_win =
// nullptr: this LUT does not fit (the callers take the kernel that reads the LUT through global memory)
std::shared_ptr<vh::SharpenLutWindow> sharpen_lut_window_cached(double x1, double y2, double y3, double m1, double m2)
{
	// (leaked on purpose, like g_lut_cache: the deleter frees device memory, which must not happen from a static
	// destructor after the runtime's own have run)
	static std::mutex &mutex = *new std::mutex;
	static std::list<std::pair<LutKey, std::shared_ptr<vh::SharpenLutWindow>>> &cache =
		*new std::list<std::pair<LutKey, std::shared_ptr<vh::SharpenLutWindow>>>;
	LutKey key = { current_device(), { x1, y2, y3, m1, m2 } };
	std::lock_guard<std::mutex> lock(mutex);
	for (auto it = cache.begin(); it != cache.end(); ++it)
		if (it->first == key) {
			cache.splice(cache.begin(), cache, it);
			return cache.front().second;
		}
	std::vector<int> lut;
	sharpen_lut_host(x1, y2, y3, m1, m2, lut);
	int lo = 0, hi = 65535;
	while (lo < 65536 && lut[lo] == lut[0])
		lo++;
	while (hi >= 0 && lut[hi] == lut[65535])
		hi--;
	std::shared_ptr<vh::SharpenLutWindow> w(new vh::SharpenLutWindow, [](vh::SharpenLutWindow *p) {
		vips_hip_free(const_cast<short *>(p->lut_win));
		delete p;
	});
	w->below = lut[0];
	w->above = lut[65535];
	// the run of zeros around a difference of 0 (m1 = 0, the default: |difference| < x1): the pixels sharpen leaves alone
	w->zero_lo = 1;
	w->zero_hi = 0;
	if (lut[32768] == 0) {
		int zl = 32768, zh = 32768;
		while (zl > 0 && lut[zl - 1] == 0)
			zl--;
		while (zh < 65535 && lut[zh + 1] == 0)
			zh++;
		w->zero_lo = zl - 32768;
		w->zero_hi = zh - 32768;
	}
	w->lo = lo < 65536 ? lo : 0;
	w->n = lo < 65536 && hi >= lo ? hi - lo + 1 : 0;
	w->lut_win = nullptr;
	bool ok = w->n <= 6144;
	std::vector<short> sv((size_t) w->n + 1, 0);
	for (int k = 0; k < w->n && ok; k++) {
		ok = lut[w->lo + k] >= -32768 && lut[w->lo + k] <= 32767;
		sv[k] = (short) lut[w->lo + k];
	}
	if (ok) {
		w->lut_win = (const short *) upload(sv.data(), sv.size() * sizeof(short));
		if (!w->lut_win)
			vips_hip_error_clear();
	}
	cache.emplace_front(key, w);
	while (cache.size() > 16)
		cache.pop_back();
	return w;
}

int conv_image(VipsHipImage *in, VipsHipImage **out, const double *mask, int mw, int mh,
	double scale, double offset, int precision)
{
	ConvPtr c = conv_cached(mask, mw, mh, scale, offset, precision);
	if (!c)
		return -1;
	const int fmt = vips_hip_conv_out_format(c.get(), in->format);
	ImageRef o(vips_hip_image_new(in->width, in->height, in->bands, fmt, in->interpretation));
	if (!o.im)
		return -1;
	// uchar, integer precision, a small mask: the packed-byte streaming kernel (conv_u8.hip)
	// (not with the Highway variant of convi selected: that one has its own arithmetic)
	if (in->format == VIPS_HIP_FORMAT_UCHAR && precision == VIPS_HIP_PRECISION_INTEGER && !vips_hip_vector_isenabled() &&
		mh > 1 && mw > 1) {
		int r = vh::conv_u8_mfma_2d_try(in, o.im, c.get()); // the matrix cores first (conv_u8_mfma.hip)
		if (r == 1)
			r = vh::conv_u8_2d_try(in, o.im, c.get());
		if (r < 0)
			return -1;
		if (r == 0) {
			*out = o.release();
			return 0;
		}
	}
	// ushort, integer precision, a mask up to 5 x 5: the streaming kernel on 16-bit lanes (conv_u16.hip)
	if (in->format == VIPS_HIP_FORMAT_USHORT && precision == VIPS_HIP_PRECISION_INTEGER && (mh > 1 || mw > 1)) {
		const int r = vh::conv_u16_2d_try(in, o.im, c.get());
		if (r < 0)
			return -1;
		if (r == 0) {
			*out = o.release();
			return 0;
		}
	}
	VipsHipRegion ri, ro;
	vips_hip_image_region(in, &ri);
	vips_hip_image_region(o.im, &ro);
	if (vips_hip_conv_gen(c.get(), &ri, &ro))
		return -1;
	// the plan's tables go back to the pool behind the kernel on this stream
	*out = o.release();
	return 0;
}

// vips_image_guess_interpretation (iofuncs/header.c:666-759) for the formats and tags
// this library carries; vips_image_default_interpretation :590-652.
int guess_interpretation(const VipsHipImage *im)
{
	bool sane = true;
	const int fmt = im->format;
	const bool is8 = fmt == VIPS_HIP_FORMAT_UCHAR || fmt == VIPS_HIP_FORMAT_CHAR;
	const bool isuint = fmt == VIPS_HIP_FORMAT_UCHAR || fmt == VIPS_HIP_FORMAT_USHORT ||
		fmt == VIPS_HIP_FORMAT_UINT;
	const bool isfloat = fmt == VIPS_HIP_FORMAT_FLOAT || fmt == VIPS_HIP_FORMAT_DOUBLE;
	int want_bands = 0;
	switch (im->interpretation) {
	case VIPS_HIP_INTERPRETATION_B_W:
	case VIPS_HIP_INTERPRETATION_GREY16:
		want_bands = 1;
		break;
	case VIPS_HIP_INTERPRETATION_XYZ:
	case VIPS_HIP_INTERPRETATION_LAB:
	case VIPS_HIP_INTERPRETATION_LABS:
	case VIPS_HIP_INTERPRETATION_sRGB:
	case VIPS_HIP_INTERPRETATION_RGB16:
	case VIPS_HIP_INTERPRETATION_scRGB:
		want_bands = 3;
		break;
	default:
		break;
	}
	if (im->bands < want_bands)
		sane = false;
	switch (im->interpretation) {
	case VIPS_HIP_INTERPRETATION_MULTIBAND:
		sane = false;
		break;
	case VIPS_HIP_INTERPRETATION_scRGB:
		if (!isfloat)
			sane = false;
		break;
	case VIPS_HIP_INTERPRETATION_LABS:
		if (isuint || is8)
			sane = false;
		break;
	case VIPS_HIP_INTERPRETATION_RGB16:
	case VIPS_HIP_INTERPRETATION_GREY16:
		if (is8)
			sane = false;
		break;
	default:
		break;
	}
	if (sane)
		return im->interpretation;
	switch (fmt) {
	case VIPS_HIP_FORMAT_UCHAR:
	case VIPS_HIP_FORMAT_SHORT:
	case VIPS_HIP_FORMAT_UINT:
	case VIPS_HIP_FORMAT_INT:
	case VIPS_HIP_FORMAT_FLOAT:
	case VIPS_HIP_FORMAT_DOUBLE:
		if (im->bands <= 2)
			return VIPS_HIP_INTERPRETATION_B_W;
		if (im->bands <= 4)
			return VIPS_HIP_INTERPRETATION_sRGB;
		return VIPS_HIP_INTERPRETATION_MULTIBAND;
	case VIPS_HIP_FORMAT_USHORT:
		if (im->bands <= 2)
			return VIPS_HIP_INTERPRETATION_GREY16;
		if (im->bands <= 4)
			return VIPS_HIP_INTERPRETATION_RGB16;
		return VIPS_HIP_INTERPRETATION_MULTIBAND;
	default:
		return VIPS_HIP_INTERPRETATION_MULTIBAND;
	}
}

// vips_interpretation_max_alpha, iofuncs/header.c:194-206
double max_alpha(int interpretation)
{
	switch (interpretation) {
	case VIPS_HIP_INTERPRETATION_GREY16:
	case VIPS_HIP_INTERPRETATION_RGB16:
		return 65535.0;
	case VIPS_HIP_INTERPRETATION_scRGB:
		return 1.0;
	default:
		return 255.0;
	}
}

const char *interpretation_nick(int v)
{
	switch (v) {
	case VIPS_HIP_INTERPRETATION_MULTIBAND: return "multiband";
	case VIPS_HIP_INTERPRETATION_B_W: return "b-w";
	case VIPS_HIP_INTERPRETATION_XYZ: return "xyz";
	case VIPS_HIP_INTERPRETATION_LAB: return "lab";
	case VIPS_HIP_INTERPRETATION_LABS: return "labs";
	case VIPS_HIP_INTERPRETATION_sRGB: return "srgb";
	case VIPS_HIP_INTERPRETATION_RGB16: return "rgb16";
	case VIPS_HIP_INTERPRETATION_GREY16: return "grey16";
	case VIPS_HIP_INTERPRETATION_scRGB: return "scrgb";
	default: return "unknown";
	}
}

// One entry of vips_colour_routes[] (colourspace.c:223-520), restricted to the spaces
// of this library: the colour steps, or a plain vips_cast_* (route == cast only).
struct Route {
	int from, to;
	int n;
	int steps[5];
	int cast_format; // for the x -> x identity routes
};

enum {
	S_sRGB2scRGB = VIPS_HIP_COLOUR_sRGB2scRGB,
	S_scRGB2XYZ = VIPS_HIP_COLOUR_scRGB2XYZ,
	S_XYZ2Lab = VIPS_HIP_COLOUR_XYZ2Lab,
	S_Lab2XYZ = VIPS_HIP_COLOUR_Lab2XYZ,
	S_XYZ2scRGB = VIPS_HIP_COLOUR_XYZ2scRGB,
	S_scRGB2sRGB = VIPS_HIP_COLOUR_scRGB2sRGB,
	S_Lab2LabS = VIPS_HIP_COLOUR_Lab2LabS,
	S_LabS2Lab = VIPS_HIP_COLOUR_LabS2Lab
};

#define XYZ VIPS_HIP_INTERPRETATION_XYZ
#define LAB VIPS_HIP_INTERPRETATION_LAB
#define LABS VIPS_HIP_INTERPRETATION_LABS
#define scRGB VIPS_HIP_INTERPRETATION_scRGB
#define sRGB VIPS_HIP_INTERPRETATION_sRGB

const Route routes[] = {
	{ XYZ, XYZ, 0, {}, VIPS_HIP_FORMAT_FLOAT },
	{ XYZ, LAB, 1, { S_XYZ2Lab }, -1 },
	{ XYZ, LABS, 2, { S_XYZ2Lab, S_Lab2LabS }, -1 },
	{ XYZ, scRGB, 1, { S_XYZ2scRGB }, -1 },
	{ XYZ, sRGB, 2, { S_XYZ2scRGB, S_scRGB2sRGB }, -1 },

	{ LAB, XYZ, 1, { S_Lab2XYZ }, -1 },
	{ LAB, LAB, 0, {}, VIPS_HIP_FORMAT_FLOAT },
	{ LAB, LABS, 1, { S_Lab2LabS }, -1 },
	{ LAB, scRGB, 2, { S_Lab2XYZ, S_XYZ2scRGB }, -1 },
	{ LAB, sRGB, 3, { S_Lab2XYZ, S_XYZ2scRGB, S_scRGB2sRGB }, -1 },

	{ LABS, XYZ, 2, { S_LabS2Lab, S_Lab2XYZ }, -1 },
	{ LABS, LAB, 1, { S_LabS2Lab }, -1 },
	{ LABS, LABS, 0, {}, VIPS_HIP_FORMAT_SHORT },
	{ LABS, scRGB, 3, { S_LabS2Lab, S_Lab2XYZ, S_XYZ2scRGB }, -1 },
	{ LABS, sRGB, 4, { S_LabS2Lab, S_Lab2XYZ, S_XYZ2scRGB, S_scRGB2sRGB }, -1 },

	{ scRGB, XYZ, 1, { S_scRGB2XYZ }, -1 },
	{ scRGB, LAB, 2, { S_scRGB2XYZ, S_XYZ2Lab }, -1 },
	{ scRGB, LABS, 3, { S_scRGB2XYZ, S_XYZ2Lab, S_Lab2LabS }, -1 },
	{ scRGB, scRGB, 0, {}, VIPS_HIP_FORMAT_FLOAT },
	{ scRGB, sRGB, 1, { S_scRGB2sRGB }, -1 },

	{ sRGB, XYZ, 2, { S_sRGB2scRGB, S_scRGB2XYZ }, -1 },
	{ sRGB, LAB, 3, { S_sRGB2scRGB, S_scRGB2XYZ, S_XYZ2Lab }, -1 },
	{ sRGB, LABS, 4, { S_sRGB2scRGB, S_scRGB2XYZ, S_XYZ2Lab, S_Lab2LabS }, -1 },
	{ sRGB, scRGB, 1, { S_sRGB2scRGB }, -1 },
	{ sRGB, sRGB, 0, {}, VIPS_HIP_FORMAT_UCHAR },
};

#undef XYZ
#undef LAB
#undef LABS
#undef scRGB
#undef sRGB

int step_out_interpretation(int step)
{
	switch (step) {
	case VIPS_HIP_COLOUR_sRGB2scRGB:
	case VIPS_HIP_COLOUR_sRGB2scRGB16:
	case VIPS_HIP_COLOUR_XYZ2scRGB:
		return VIPS_HIP_INTERPRETATION_scRGB;
	case VIPS_HIP_COLOUR_scRGB2XYZ:
	case VIPS_HIP_COLOUR_Lab2XYZ:
		return VIPS_HIP_INTERPRETATION_XYZ;
	case VIPS_HIP_COLOUR_XYZ2Lab:
	case VIPS_HIP_COLOUR_LabS2Lab:
		return VIPS_HIP_INTERPRETATION_LAB;
	case VIPS_HIP_COLOUR_scRGB2sRGB:
		return VIPS_HIP_INTERPRETATION_sRGB;
	case VIPS_HIP_COLOUR_scRGB2sRGB16:
		return VIPS_HIP_INTERPRETATION_RGB16;
	case VIPS_HIP_COLOUR_Lab2LabS:
		return VIPS_HIP_INTERPRETATION_LABS;
	default:
		return VIPS_HIP_INTERPRETATION_MULTIBAND;
	}
}

int step_out_format(int step)
{
	switch (step) {
	case VIPS_HIP_COLOUR_scRGB2sRGB: return VIPS_HIP_FORMAT_UCHAR;
	case VIPS_HIP_COLOUR_scRGB2sRGB16: return VIPS_HIP_FORMAT_USHORT;
	case VIPS_HIP_COLOUR_Lab2LabS: return VIPS_HIP_FORMAT_SHORT;
	default: return VIPS_HIP_FORMAT_FLOAT;
	}
}

int cast_image(VipsHipImage *in, VipsHipImage **out, int format)
{
	// same format: vips_cast / vips_colourspace return the input (a pointer copy in the
	// reference).  Library-owned pixels are shared; caller-owned device memory is copied.
	if (format == in->format) {
		if (VipsHipImage *shared = image_share(in)) {
			*out = shared;
			return 0;
		}
		ImageRef c(vips_hip_image_new(in->width, in->height, in->bands, format, in->interpretation));
		if (!c.im || vips_hip_memcpy_d2d(c.im->data, in->data, in->stride * in->height))
			return -1;
		*out = c.release();
		return 0;
	}
	ImageRef o(vips_hip_image_new(in->width, in->height, in->bands, format, in->interpretation));
	if (!o.im)
		return -1;
	VipsHipRegion ri, ro;
	vips_hip_image_region(in, &ri);
	vips_hip_image_region(o.im, &ro);
	if (vips_hip_cast_gen(&ri, &ro))
		return -1;
	*out = o.release();
	return 0;
}

} // namespace

extern "C" {

// vips_conv_build, convolution/conv.c:62-118
int vips_hip_conv(VipsHipImage *in, VipsHipImage **out, const double *mask, int mask_width,
	int mask_height, double scale, double offset, int precision)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out || !mask) {
		error("conv", "null argument");
		return -1;
	}
	// conv.c:99-107 with the class defaults layers = 5, cluster = 1 (:159-160)
	if (precision == VIPS_HIP_PRECISION_APPROXIMATE)
		return vips_hip_conva(in, out, mask, mask_width, mask_height, scale, offset, 5, 1);
	return conv_image(in, out, mask, mask_width, mask_height, scale, offset, precision);
}

// vips_conva_build, convolution/conva.c:1231-1280
int vips_hip_conva(VipsHipImage *in, VipsHipImage **out, const double *mask, int mask_width,
	int mask_height, double scale, double offset, int layers, int cluster)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("conva", "null argument");
		return -1;
	}
	ConvaPtr c = conva_cached(mask, mask_width, mask_height, scale, offset, layers, cluster, false);
	if (!c)
		return -1;
	ImageRef o(vips_hip_image_new(in->width, in->height, in->bands, in->format, in->interpretation));
	if (!o.im)
		return -1;
	{
		const int r = vh::conva_fast_image(in, o.im, c.get());
		if (r < 0)
			return -1;
		if (r == 0) {
			*out = o.release();
			return 0;
		}
	}
	VipsHipRegion ri, ro;
	vips_hip_image_region(in, &ri);
	vips_hip_image_region(o.im, &ro);
	if (vips_hip_conva_gen(c.get(), &ri, &ro))
		return -1;
	*out = o.release();
	return 0;
}

// vips_convasep_build, convolution/convasep.c:775-828: horizontal pass, then vertical
int vips_hip_convasep(VipsHipImage *in, VipsHipImage **out, const double *mask, int mask_n,
	double scale, double offset, int layers)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("convasep", "null argument");
		return -1;
	}
	if (mask_n <= 0) {
		error("convasep", "separable matrix images must have width or height 1");
		return -1;
	}
	ConvaPtr c = conva_cached(mask, mask_n, 1, scale, offset, layers, 1, true);
	if (!c)
		return -1;
	ImageRef o(vips_hip_image_new(in->width, in->height, in->bands, in->format, in->interpretation));
	if (!o.im)
		return -1;
	// 8/16-bit images whose sums cannot wrap: both passes in the fused separable kernel
	const int r = vh::convasep_fused(in, o.im, c.get());
	if (r < 0)
		return -1;
	if (r == 0) {
		*out = o.release();
		return 0;
	}
	ImageRef t(vips_hip_image_new(in->width, in->height, in->bands, in->format, in->interpretation));
	if (!t.im)
		return -1;
	VipsHipRegion ri, rt, ro;
	vips_hip_image_region(in, &ri);
	vips_hip_image_region(t.im, &rt);
	vips_hip_image_region(o.im, &ro);
	if (vips_hip_convasep_gen(c.get(), &ri, &rt, 0) || vips_hip_convasep_gen(c.get(), &rt, &ro, 1))
		return -1;
	*out = o.release();
	return 0;
}

// vips_convsep_build, convolution/convsep.c:61-118: conv(M) then conv(rot90(M), offset 0)
int vips_hip_convsep(VipsHipImage *in, VipsHipImage **out, const double *mask, int mask_n,
	double scale, double offset, int precision)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out || !mask) {
		error("convsep", "null argument");
		return -1;
	}
	// convsep.c:81-87 with the class default layers = 5 (:155)
	if (precision == VIPS_HIP_PRECISION_APPROXIMATE)
		return vips_hip_convasep(in, out, mask, mask_n, scale, offset, 5);
	if (mask_n <= 0) {
		error("convsep", "separable matrix images must have width or height 1");
		return -1;
	}
	// float images, and uchar / ushort / short with integer precision: both passes in one
	// streaming kernel (convsep_f32.hip)
	// (with the Highway variant of convi selected, uchar images take the two plain passes)
	const bool vector_uchar = in->format == VIPS_HIP_FORMAT_UCHAR && vips_hip_vector_isenabled();
	if (in->format == VIPS_HIP_FORMAT_FLOAT ||
		(precision == VIPS_HIP_PRECISION_INTEGER && !vector_uchar &&
			(in->format == VIPS_HIP_FORMAT_UCHAR || in->format == VIPS_HIP_FORMAT_USHORT ||
				in->format == VIPS_HIP_FORMAT_SHORT))) {
		ConvPtr c = conv_cached(mask, mask_n, 1, scale, offset, precision);
		if (!c)
			return -1;
		ImageRef o(vips_hip_image_new(in->width, in->height, in->bands, in->format, in->interpretation));
		if (!o.im)
			return -1;
		int r = vh::convsep_stream_fused(in, o.im, c.get(), 0.0, nullptr, 0);
		if (r == 1)
			r = vh::conv_u8_mfma_sep_try(in, o.im, c.get(), 0.0);
		if (r == 1)
			r = vh::conv_u16_mfma_sep_try(in, o.im, c.get(), 0.0);
		if (r == 1)
			r = vh::conv_u8_sep_try(in, o.im, c.get(), 0.0);
		if (r == 1)
			r = vh::convsep_f32_fused(in, o.im, c.get(), 0.0);
		if (r < 0)
			return -1;
		if (r == 0) {
			*out = o.release();
			return 0;
		}
	}
	ImageRef t1;
	// first pass: the mask as given (1 row of mask_n)
	if (conv_image(in, &t1.im, mask, mask_n, 1, scale, offset, precision))
		return -1;
	// second pass: vips_rot(D90) turns the row into a column (same element order
	// top to bottom), same scale, offset forced to 0
	return conv_image(t1.im, out, mask, 1, mask_n, scale, 0.0, precision);
}

// vips_gaussblur_build, convolution/gaussblur.c:71-116
int vips_hip_gaussblur(VipsHipImage *in, VipsHipImage **out, double sigma, double min_ampl,
	int precision)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("gaussblur", "null argument");
		return -1;
	}
	if (sigma < 0.2) {
		ImageRef o(vips_hip_image_new(in->width, in->height, in->bands, in->format,
			in->interpretation));
		if (!o.im || vips_hip_memcpy_d2d(o.im->data, in->data, in->stride * in->height))
			return -1;
		*out = o.release();
		return 0;
	}
	int width = vips_hip_gaussmat(sigma, min_ampl, 1, precision, nullptr, 0, nullptr);
	if (width < 0)
		return -1;
	std::vector<double> mask(width);
	double scale = 1.0;
	if (vips_hip_gaussmat(sigma, min_ampl, 1, precision, mask.data(), width, &scale) < 0)
		return -1;
	return vips_hip_convsep(in, out, mask.data(), width, scale, 0.0, precision);
}

int vips_hip_cast(VipsHipImage *in, VipsHipImage **out, int format)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("cast", "null argument");
		return -1;
	}
	return cast_image(in, out, format);
}

static int premultiply_image(VipsHipImage *in, VipsHipImage **out, int uchar, int inverse)
{
	if (!in || !out) {
		error("premultiply", "null argument");
		return -1;
	}
	const char *domain = inverse ? "unpremultiply" : "premultiply";
	if (in->bands == 1) { // "Trivial case: fall back to copy()."
		return cast_image(in, out, in->format);
	}
	if (in->format == VIPS_HIP_FORMAT_DOUBLE) {
		error(domain, "double images are outside the HIP path");
		return -1;
	}
	const bool fast = uchar && in->format == VIPS_HIP_FORMAT_UCHAR;
	ImageRef o(vips_hip_image_new(in->width, in->height, in->bands,
		fast ? VIPS_HIP_FORMAT_UCHAR : VIPS_HIP_FORMAT_FLOAT, in->interpretation));
	if (!o.im)
		return -1;
	VipsHipRegion ri, ro;
	vips_hip_image_region(in, &ri);
	vips_hip_image_region(o.im, &ro);
	if (vips_hip_premultiply_gen(&ri, &ro, max_alpha(in->interpretation), uchar, inverse))
		return -1;
	*out = o.release();
	return 0;
}

int vips_hip_premultiply(VipsHipImage *in, VipsHipImage **out, int uchar)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	return premultiply_image(in, out, uchar, 0);
}

int vips_hip_unpremultiply(VipsHipImage *in, VipsHipImage **out, int uchar)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	return premultiply_image(in, out, uchar, 1);
}

// vips_colourspace_build, colour/colourspace.c:551-612
int vips_hip_colourspace(VipsHipImage *in, VipsHipImage **out, int space)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("colourspace", "null argument");
		return -1;
	}
	int interpretation = guess_interpretation(in);
	const Route *route = nullptr;
	for (const Route &r : routes)
		if (r.from == interpretation && r.to == space) {
			route = &r;
			break;
		}
	if (!route) {
		error("vips_colourspace", "no known route from '%s' to '%s'",
			interpretation_nick(interpretation), interpretation_nick(space));
		return -1;
	}
	if (route->n == 0) {
		// vips_cast_float / vips_cast_uchar / vips_cast_short; interpretation unchanged
		return cast_image(in, out, route->cast_format);
	}
	if (in->bands < 3) {
		error("colourspace", "image must have at least 3 bands");
		return -1;
	}

	// The stored input: the fused route kernel reads uchar / ushort / short / float and
	// applies the cast the first step's build() would insert; other formats get that
	// vips_cast as a pass of its own first.
	ImageRef pre;
	VipsHipImage *cur = in;
	const int first = route->steps[0];
	const bool readable = cur->format == VIPS_HIP_FORMAT_UCHAR || cur->format == VIPS_HIP_FORMAT_USHORT ||
		cur->format == VIPS_HIP_FORMAT_SHORT || cur->format == VIPS_HIP_FORMAT_FLOAT;
	if (!readable) {
		int want = VIPS_HIP_FORMAT_FLOAT;
		if (first == VIPS_HIP_COLOUR_sRGB2scRGB)
			want = VIPS_HIP_FORMAT_UCHAR;
		else if (first == VIPS_HIP_COLOUR_LabS2Lab)
			want = VIPS_HIP_FORMAT_SHORT;
		if (cast_image(cur, &pre.im, want))
			return -1;
		cur = pre.im;
	}

	// Extra bands: every step rescales them in float when the alpha range changes
	// (vips_linear1) and casts them to its output format (colour.c:249-296).  Those
	// roundings do not compose, so images with extra bands run the chain one step per
	// pass like the reference; 3-band images take the fused route.
	const double alpha_scale = 1.0;
	const int last = route->steps[route->n - 1];
	const bool stepwise = cur->bands > 3;

	if (!stepwise) {
		ImageRef o(vips_hip_image_new(cur->width, cur->height, cur->bands, step_out_format(last),
			step_out_interpretation(last)));
		if (!o.im)
			return -1;
		VipsHipRegion ri, ro;
		vips_hip_image_region(cur, &ri);
		vips_hip_image_region(o.im, &ro);
		if (vips_hip_colour_route_gen(route->steps, route->n, alpha_scale, &ri, &ro))
			return -1;
		*out = o.release();
		return 0;
	}

	// images with extra bands: one pass per step, exactly the reference's chain
	ImageRef hold;
	int before = interpretation;
	for (int s = 0; s < route->n; s++) {
		const int st = route->steps[s];
		const int after = step_out_interpretation(st);
		ImageRef o(vips_hip_image_new(cur->width, cur->height, cur->bands, step_out_format(st), after));
		if (!o.im)
			return -1;
		VipsHipRegion ri, ro;
		vips_hip_image_region(cur, &ri);
		vips_hip_image_region(o.im, &ro);
		if (vips_hip_colour_route_gen(&st, 1, max_alpha(after) / max_alpha(before), &ri, &ro))
			return -1;
		vips_hip_image_unref(hold.im);
		hold.im = o.release();
		cur = hold.im;
		before = after;
	}
	*out = hold.release();
	return 0;
}

// vips_gaussblur() followed by vips_colourspace() (gaussblur.c:71-116, colourspace.c:551-612):
// BASELINE config 3.  On a 3-band float image whose route has float colour steps only, both
// convolution passes and the route run in ONE streaming kernel (convsep_stream.hip): the
// blurred image is never written.  Every other case is the two operations, one after the
// other.  The pixels are the same either way.
int vips_hip_gaussblur_colourspace(VipsHipImage *in, VipsHipImage **out, double sigma, double min_ampl,
	int precision, int space)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("gaussblur", "null argument");
		return -1;
	}
	if (in->format == VIPS_HIP_FORMAT_FLOAT && in->bands == 3 && sigma >= 0.2 &&
		precision != VIPS_HIP_PRECISION_APPROXIMATE) {
		const int interpretation = guess_interpretation(in); // gaussblur keeps the interpretation
		const Route *route = nullptr;
		for (const Route &r : routes)
			if (r.from == interpretation && r.to == space) {
				route = &r;
				break;
			}
		if (route && route->n > 0 && step_out_format(route->steps[route->n - 1]) == VIPS_HIP_FORMAT_FLOAT) {
			const int width = vips_hip_gaussmat(sigma, min_ampl, 1, precision, nullptr, 0, nullptr);
			if (width < 0)
				return -1;
			std::vector<double> mask(width);
			double scale = 1.0;
			if (vips_hip_gaussmat(sigma, min_ampl, 1, precision, mask.data(), width, &scale) < 0)
				return -1;
			ConvPtr c = conv_cached(mask.data(), width, 1, scale, 0.0, precision);
			if (!c)
				return -1;
			ImageRef o(vips_hip_image_new(in->width, in->height, 3, VIPS_HIP_FORMAT_FLOAT,
				step_out_interpretation(route->steps[route->n - 1])));
			if (!o.im)
				return -1;
			const int r = vh::convsep_stream_fused(in, o.im, c.get(), 0.0, route->steps, route->n);
			if (r < 0)
				return -1;
			if (r == 0) {
				*out = o.release();
				return 0;
			}
		}
	}
	ImageRef blurred;
	if (vips_hip_gaussblur(in, &blurred.im, sigma, min_ampl, precision))
		return -1;
	return vips_hip_colourspace(blurred.im, out, space);
}

static int sharpen_fused_images(VipsHipImage *const *in, int n_images, VipsHipImage **out, double sigma, double x1,
	double y2, double y3, double m1, double m2);

// vips_sharpen's blur mask as the convi C path's integers (sharpen.c:214-218, convi.c:886-915) when
// it has at most 5 taps -- the case of the fused sharpen kernels; false: a wider mask (or an error,
// which the unfused path then reports)
static bool sharpen_small_mask(double sigma, std::vector<int> &coef, int *scale)
{
	const int n = vips_hip_gaussmat(sigma, 0.1, 1, VIPS_HIP_PRECISION_INTEGER, nullptr, 0, nullptr);
	if (n < 1 || n > 5)
		return false;
	std::vector<double> mask(n);
	double s = 1.0;
	if (vips_hip_gaussmat(sigma, 0.1, 1, VIPS_HIP_PRECISION_INTEGER, mask.data(), n, &s) < 0) {
		vips_hip_error_clear();
		return false;
	}
	coef.resize(n);
	for (int k = 0; k < n; k++)
		coef[k] = (int) rint(mask[k]);
	*scale = (int) rint(s);
	return true;
}

static bool sharpen_images_fusable(VipsHipImage *const *in, int n)
{
	for (int i = 0; i < n; i++)
		if (!in[i] || in[i]->format != VIPS_HIP_FORMAT_UCHAR || in[i]->bands != 3 ||
			in[i]->interpretation != VIPS_HIP_INTERPRETATION_sRGB || in[i]->width != in[0]->width ||
			in[i]->height != in[0]->height)
			return false;
	return n > 0;
}

// The two CU partitions of the batched pipeline (see vips_hip_resize_sharpen_batch): streams
// restricted to 3/4 and 1/4 of the CUs: the low and the high run of consecutive mask bits.  (What
// the runtime honours on this part, measured by tools/c4_masks.py: runs of at least 8 consecutive
// bits restrict a queue, in effect in groups of 32 CUs -- 48 or 56 bits behave like 32; masks
// interleaved finer than that leave it on every CU.)  Made once per device and kept (a masked
// stream is a hardware queue with a fixed mask).
struct BatchStreams {
	hipStream_t resize = nullptr, sharpen = nullptr;
};
static BatchStreams *batch_streams()
{
	static std::mutex mutex;
	static std::map<int, BatchStreams> by_device;
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess)
		return nullptr;
	std::lock_guard<std::mutex> lock(mutex);
	auto it = by_device.find(device);
	if (it != by_device.end())
		return it->second.resize ? &it->second : nullptr;
	BatchStreams &bs = by_device[device];
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
		return nullptr;
	const int cus = prop.multiProcessorCount;
	const int share = getenv("VIPS_HIP_BATCH_SHARPEN_CUS") ? atoi(getenv("VIPS_HIP_BATCH_SHARPEN_CUS")) : cus / 4;
	if (share == 0) {
		// $VIPS_HIP_BATCH_SHARPEN_CUS=0: two plain streams, both kernels on every CU (the sharpen's at the lowest
		// priority the device offers: it fills what the resize leaves)
		int least = 0, greatest = 0;
		(void) hipDeviceGetStreamPriorityRange(&least, &greatest);
		if (hipStreamCreateWithPriority(&bs.resize, hipStreamNonBlocking, greatest) != hipSuccess ||
			hipStreamCreateWithPriority(&bs.sharpen, hipStreamNonBlocking, least) != hipSuccess) {
			bs.resize = bs.sharpen = nullptr;
			(void) hipGetLastError();
			return nullptr;
		}
		return &bs;
	}
	if (cus < 16 || cus > 1024 || share < 8 || share > cus - 8)
		return nullptr;
	std::vector<uint32_t> lo((cus + 31) / 32, 0u), hi((cus + 31) / 32, 0u);
	for (int i = 0; i < cus; i++)
		(i < cus - share ? lo : hi)[i / 32] |= 1u << (i % 32);
	if (hipExtStreamCreateWithCUMask(&bs.resize, (uint32_t) lo.size(), lo.data()) != hipSuccess ||
		hipExtStreamCreateWithCUMask(&bs.sharpen, (uint32_t) hi.size(), hi.data()) != hipSuccess) {
		bs.resize = bs.sharpen = nullptr; // (the plain one-stream order then)
		(void) hipGetLastError();
		return nullptr;
	}
	return &bs;
}
// BASELINE config 4, the batched thumbnail pipeline: vips_resize(scale) [then vips_sharpen()] on
// n independent images.  libvips runs such a batch as n pipelines over its thread pool
// (iofuncs/threadpool.c:625); here n_threads host threads each take the next image, every thread
// on its own stream: the many small kernels of one image's sharpen stage overlap the two big
// streaming kernels of the next images' resize instead of each paying its launch latency alone.
// sigma < 0 skips the sharpen.  Returns the number of images that failed (their outs[] are
// NULL; the first error message is left in the caller's error buffer); all work is complete
// on return.
// ... on the device the calling thread drives (every image of `in` lives there)
// wait = false: a uniform batch returns as soon as it is queued (everything it queued is ordered on
// the calling thread's stream: the join below); any other batch, and every failure, waits as before
static int resize_sharpen_batch_here(VipsHipImage *const *in, int n, VipsHipImage **out, double scale, int kernel,
	double gap, double sigma, double x1, double y2, double y3, double m1, double m2, int n_threads, bool wait = true)
{
	if (n_threads < 1)
		n_threads = 1;
	if (n_threads > n)
		n_threads = n;
	for (int i = 0; i < n; i++)
		out[i] = nullptr;
	// A batch of uchar images of one size whose resize is the streaming kernel's case: one launch
	// per 64 images for the whole resize chain (resize_stream.hip) and one for the sharpen.
	//
	// The resize chain is bound by HBM, the sharpen by the FP64 pipe.  Queued one behind the other
	// on one stream they cost the sum of their times (0.040 + 0.010 ms per 8192 x 8192 image);
	// left to share every CU (two plain streams) each gets in the other's way and the sum stays
	// (round 2: 0.044 + 0.016 side by side).  So the CUs are PARTITIONED: the resize of chunk k + 1
	// runs on a stream masked to 3/4 of the CUs of every XCD (hipExtStreamCreateWithCUMask) next to
	// the sharpen of chunk k on the other quarter -- measured on the MI355X (tools/c4_cumask.py):
	// resize 0.0405 ms per image on 256 CUs, 0.0440 on 192; sharpen x 4 on 64 CUs = about the same
	// time, so a pair of chunks takes what the resize alone takes on 192 CUs.  The first chunk's
	// resize and the last chunk's sharpen have nothing to run beside and take every CU.  $VIPS_HIP_BATCH_OVERLAP=0: one
	// stream, one stage after the other.
	if (n > 0 && !getenv("VIPS_HIP_NO_BATCH_LAUNCH")) {
		const int chunk = 64;
		// resize and sharpen of 3-band sRGB images in ONE kernel (resize_sharpen.hip): nothing to
		// partition, no thumbnail in memory between the two.  Bit-exact, but on the MI355X the
		// sharpen's arithmetic inside the streaming blocks costs more than the partition it removes
		// (0.051 ms per 8192 x 8192 image against 0.043: DESIGN.md 3.2), so it runs on request only:
		// $VIPS_HIP_RESIZE_SHARPEN=1
		std::vector<int> coef;
		int mask_scale = 1;
		const bool small_mask = sigma >= 0.0 && sharpen_small_mask(sigma, coef, &mask_scale);
		const bool fusable = small_mask && sharpen_images_fusable(in, n);
		const char *one_kernel = getenv("VIPS_HIP_RESIZE_SHARPEN");
		if (fusable && one_kernel && atoi(one_kernel) > 0) {
			std::vector<int> lut;
			sharpen_lut_host(x1, y2, y3, m1, m2, lut);
			const int r = vh::resize_sharpen_batch_u8(in, n, out, scale, kernel, gap, coef.data(), (int) coef.size(),
				mask_scale, lut.data());
			if (r < 0)
				return -1;
			if (r == 0) {
				if (wait && vips_hip_synchronize()) {
					for (int i = 0; i < n; i++) {
						vips_hip_image_unref(out[i]);
						out[i] = nullptr;
					}
					return -1;
				}
				return 0;
			}
		}
		std::vector<ImageRef> small(n); // held to the end: several streams read and write them
		std::vector<VipsHipImage *> ps(n, nullptr);
		const int first = n < chunk ? n : chunk;
		const char *ov = getenv("VIPS_HIP_BATCH_OVERLAP");
		// (the partitions only when the ONE-kernel sharpen will run on them: the unfused sharpen takes
		// its temporaries from the pool, which orders reuse within one stream only)
		BatchStreams *bs = sigma >= 0.0 && n > chunk && fusable && !getenv("VIPS_HIP_NO_FUSED_SHARPEN") &&
				!(ov && atoi(ov) == 0)
			? batch_streams()
			: nullptr;
		hipStream_t main_stream = stream();
		std::vector<hipEvent_t> events;
		auto event_on = [&](hipStream_t s) -> hipEvent_t {
			hipEvent_t ev = nullptr;
			if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
				return nullptr;
			events.push_back(ev);
			return hipEventRecord(ev, s) == hipSuccess ? ev : nullptr;
		};
		int bad = 0;
		if (bs) {
			// both partitions start behind whatever the caller queued on its stream
			hipEvent_t ev = event_on(main_stream);
			if (!ev || hipStreamWaitEvent(bs->resize, ev, 0) != hipSuccess || hipStreamWaitEvent(bs->sharpen, ev, 0) != hipSuccess)
				bad = 1;
		}
		// the first chunk's resize has no sharpen to run beside: it takes every CU (the caller's
		// stream), and the resize partition starts behind it
		int r = 0;
		hipEvent_t first_done = nullptr;
		if (!bad) {
			r = vh::resize_batch_u8(in, first, ps.data(), scale, kernel, gap);
			if (bs && r == 0) {
				first_done = event_on(main_stream);
				if (!first_done || hipStreamWaitEvent(bs->resize, first_done, 0) != hipSuccess)
					bad = 1;
			}
		}
		if (r < 0 || bad) {
			for (hipEvent_t ev : events)
				(void) hipEventDestroy(ev);
			return -1;
		}
		if (r == 0) {
			for (int base = 0; base < n && !bad; base += chunk) {
				const int cnt = n - base < chunk ? n - base : chunk;
				const bool last = base + cnt >= n;
				if (base > 0) {
					ScopedStream on(bs ? bs->resize : nullptr);
					const int rr = vh::resize_batch_u8(in + base, cnt, ps.data() + base, scale, kernel, gap);
					if (rr < 0)
						bad = 1;
					else if (rr > 0) // a chunk of another geometry: image by image
						for (int i = base; i < base + cnt && !bad; i++)
							if (!in[i] || vips_hip_resize(in[i], &ps[i], scale, -1.0, kernel, gap))
								bad = 1;
				}
				for (int i = base; i < base + cnt; i++)
					small[i].im = ps[i];
				if (bad)
					break;
				if (sigma < 0.0) {
					for (int i = base; i < base + cnt; i++)
						out[i] = small[i].release();
					continue;
				}
				// the sharpen of this chunk: behind its resize, on the sharpen partition -- the
				// last chunk's on the caller's stream (every CU), which then also waits for the rest
				hipStream_t sharpen_on = nullptr;
				if (bs) {
					sharpen_on = last ? main_stream : bs->sharpen;
					hipEvent_t ev = base == 0 ? first_done : event_on(bs->resize);
					if (!ev || hipStreamWaitEvent(sharpen_on, ev, 0) != hipSuccess)
						bad = 1;
				}
				if (bad)
					break;
				{
					ScopedStream on(sharpen_on);
					int done = sharpen_fused_images(ps.data() + base, cnt, out + base, sigma, x1, y2, y3, m1, m2);
					if (done > 0) {
						// not the one-kernel sharpen's case after all: the separate operations take pool
						// blocks for their temporaries, which are only ordered within ONE stream -- so
						// nothing else of the batch may be in flight around them
						if (bs && (hipStreamSynchronize(bs->resize) != hipSuccess || hipStreamSynchronize(bs->sharpen) != hipSuccess ||
									  hipStreamSynchronize(main_stream) != hipSuccess))
							bad = 1;
						done = 0;
						for (int i = base; i < base + cnt && !done && !bad; i++)
							done = vips_hip_sharpen(ps[i], &out[i], sigma, x1, y2, y3, m1, m2);
						if (bs && hipStreamSynchronize(sharpen_on ? sharpen_on : main_stream) != hipSuccess)
							bad = 1;
					}
					if (done)
						bad = 1;
				}
			}
			if (bs) {
				// the caller's stream ends behind both partitions, so that everything the batch
				// queued is ordered on it
				for (hipStream_t s : { bs->resize, bs->sharpen }) {
					hipEvent_t ev = event_on(s);
					if (!ev || hipStreamWaitEvent(main_stream, ev, 0) != hipSuccess)
						bad = 1;
				}
			}
			// (queued form: the intermediate thumbnails go back to the pool below while the device
			// still reads them -- the pool hands a block to this thread's later work only, and that
			// work is queued behind the join; an event destroyed before it completes is released
			// when it does)
			if (wait || bad) {
				if (vips_hip_synchronize())
					bad = 1;
				if (bs && (hipStreamSynchronize(bs->resize) != hipSuccess || hipStreamSynchronize(bs->sharpen) != hipSuccess))
					bad = 1;
			}
			for (hipEvent_t ev : events)
				(void) hipEventDestroy(ev);
			if (bad) {
				for (int i = 0; i < n; i++) {
					vips_hip_image_unref(out[i]);
					out[i] = nullptr;
				}
				return -1;
			}
			return 0;
		}
		for (hipEvent_t ev : events)
			(void) hipEventDestroy(ev);
	}
	std::atomic<int> next(0), failed(0);
	std::mutex err_mutex;
	std::string first_error;
	auto worker = [&]() {
		for (;;) {
			const int i = next.fetch_add(1);
			if (i >= n)
				break;
			out[i] = nullptr;
			ImageRef small;
			int r = in[i] ? vips_hip_resize(in[i], &small.im, scale, -1.0, kernel, gap) : -1;
			if (!r) {
				if (sigma >= 0.0)
					r = vips_hip_sharpen(small.im, &out[i], sigma, x1, y2, y3, m1, m2);
				else
					out[i] = small.release();
			}
			if (r) {
				out[i] = nullptr;
				failed.fetch_add(1);
				std::lock_guard<std::mutex> lock(err_mutex);
				if (first_error.empty())
					first_error = in[i] ? vips_hip_error_buffer() : "resize_sharpen_batch: null image\n";
				vips_hip_error_clear(); // the error buffer is per thread
			}
		}
	};
	if (n_threads <= 1) {
		worker();
		if (vips_hip_synchronize())
			return -1;
	}
	else {
		std::vector<std::thread> pool;
		// results cross to the caller's thread: every pool thread finishes its stream (and gives
		// it back) before it ends
		for (int t = 0; t < n_threads; t++)
			pool.emplace_back([&]() {
				worker();
				release_thread_stream();
			});
		for (std::thread &t : pool)
			t.join();
	}
	if (!first_error.empty())
		error("resize_sharpen_batch", "%s", first_error.c_str());
	return failed.load();
}

// The batch SCATTER of BASELINE config 4 inside one process: the images of a batch may live on
// several devices (whoever loaded them dealt them out: vips_hip_devices()); each device's share
// runs on a host thread bound to that device -- its own pool, plan caches and streams -- with no
// data moving between devices.  One device: the calling thread does the work itself.
static int resize_sharpen_batch_any(VipsHipImage *const *in, int n, VipsHipImage **out, double scale, int kernel,
	double gap, double sigma, double x1, double y2, double y3, double m1, double m2, int n_threads, bool wait)
{
	if (!in || !out || n < 0) {
		error("resize_sharpen_batch", "null argument");
		return -1;
	}
	if (ensure_init())
		return -1;
	std::map<int, std::vector<int>> by_device;
	for (int i = 0; i < n; i++)
		by_device[in[i] ? in[i]->device : current_device()].push_back(i);
	if (by_device.size() <= 1) {
		if (n > 0 && vh::bind_to(in[by_device.begin()->second[0]]))
			return -1;
		return resize_sharpen_batch_here(in, n, out, scale, kernel, gap, sigma, x1, y2, y3, m1, m2, n_threads, wait);
	}
	for (int i = 0; i < n; i++)
		out[i] = nullptr;
	std::mutex err_mutex;
	std::string first_error;
	std::atomic<int> failed(0);
	std::vector<std::thread> workers;
	for (auto &group : by_device) {
		const int device = group.first;
		const std::vector<int> &idx = group.second;
		workers.emplace_back([&, device]() {
			std::vector<VipsHipImage *> sub_in(idx.size()), sub_out(idx.size(), nullptr);
			for (size_t k = 0; k < idx.size(); k++)
				sub_in[k] = in[idx[k]];
			int r = vips_hip_init(device);
			if (!r)
				r = resize_sharpen_batch_here(sub_in.data(), (int) idx.size(), sub_out.data(), scale, kernel, gap, sigma,
					x1, y2, y3, m1, m2, n_threads);
			for (size_t k = 0; k < idx.size(); k++)
				out[idx[k]] = sub_out[k];
			if (r) {
				failed.fetch_add(r < 0 ? (int) idx.size() : r);
				std::lock_guard<std::mutex> lock(err_mutex);
				if (first_error.empty())
					first_error = vips_hip_error_buffer();
			}
			release_thread_stream(); // results cross to the caller's thread
		});
	}
	for (std::thread &t : workers)
		t.join();
	if (!first_error.empty())
		error("resize_sharpen_batch", "%s", first_error.c_str());
	return failed.load() == n ? -1 : failed.load();
}

int vips_hip_resize_sharpen_batch(VipsHipImage *const *in, int n, VipsHipImage **out, double scale, int kernel,
	double gap, double sigma, double x1, double y2, double y3, double m1, double m2, int n_threads)
{
	return resize_sharpen_batch_any(in, n, out, scale, kernel, gap, sigma, x1, y2, y3, m1, m2, n_threads, true);
}

// ... returning as soon as the batch is QUEUED when it is one the library runs in batch launches on
// the caller's device (the usual case of a thumbnail service: same-sized uchar images): the results
// are then ordered on the calling thread's stream like any other operation's -- use them in stream
// order, or vips_hip_synchronize() -- and the host prepares the next batch while this one runs.
int vips_hip_resize_sharpen_batch_queue(VipsHipImage *const *in, int n, VipsHipImage **out, double scale, int kernel,
	double gap, double sigma, double x1, double y2, double y3, double m1, double m2, int n_threads)
{
	return resize_sharpen_batch_any(in, n, out, scale, kernel, gap, sigma, x1, y2, y3, m1, m2, n_threads, false);
}

// vips_extract_area (conversion/extract.c:137-187): a rectangle of the image, as a new image
int vips_hip_extract_area(VipsHipImage *in, VipsHipImage **out, int left, int top, int width, int height)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("extract_area", "null argument");
		return -1;
	}
	if (width <= 0 || height <= 0 || left < 0 || top < 0 || (long long) left + width > in->width ||
		(long long) top + height > in->height) {
		error("extract_area", "bad extract area");
		return -1;
	}
	ImageRef o(vips_hip_image_new(width, height, in->bands, in->format, in->interpretation));
	if (!o.im)
		return -1;
	const size_t pel = (size_t) in->bands * format_sizeof(in->format);
	const unsigned char *src = (const unsigned char *) in->data + (size_t) top * in->stride + (size_t) left * pel;
	if (hipMemcpy2DAsync(o.im->data, o.im->stride, src, in->stride, (size_t) width * pel, (size_t) height,
			hipMemcpyDeviceToDevice, stream()) != hipSuccess) {
		error("extract_area", "device copy failed");
		return -1;
	}
	*out = o.release();
	return 0;
}

// vips_thumbnail_build, resample/thumbnail.c:678-1067, for in-memory images
// (vips_thumbnail_image: no pre-shrink on load, no pages).
int vips_hip_thumbnail_image(VipsHipImage *in, VipsHipImage **out, int width, int height, int size,
	int linear)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	return vips_hip_thumbnail_image_crop(in, out, width, height, size, linear, 0);
}

// ... with the crop argument: a VipsInteresting (include/vips/conversion.h:97-107).  The
// positional modes are here (none 0, centre 1, low 4, high 5, all 6: smartcrop.c:359-400); the
// content-driven ones (entropy 2, attention 3) are outside the path.
int vips_hip_thumbnail_image_crop(VipsHipImage *in, VipsHipImage **out, int width, int height, int size,
	int linear, int crop)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	const char *domain = "thumbnail";
	if (!in || !out) {
		error(domain, "null argument");
		return -1;
	}
	if (crop == 2 || crop == 3) {
		error(domain, "crop modes 'entropy' and 'attention' are outside the HIP path");
		return -1;
	}
	if (crop < 0 || crop > 6) {
		error(domain, "bad crop mode %d", crop);
		return -1;
	}
	if (width <= 0) {
		error(domain, "parameter width not set");
		return -1;
	}
	if (height <= 0)
		height = width; // thumbnail.c:720-721
	if (size < 0 || size > 3) {
		error(domain, "bad size mode %d", size);
		return -1;
	}

	// processing space (thumbnail.c:763-823)
	ImageRef space;
	VipsHipImage *cur;
	if (in->bands < 3) {
		// B_W is the processing space of images with fewer than 3 bands (thumbnail.c:806-820); a
		// one-band uchar B_W image (a greyscale JPEG) is already in it.  GREY16 (linear) and
		// grey + alpha are outside the path.
		if (linear || in->bands != 1 || in->format != VIPS_HIP_FORMAT_UCHAR ||
			guess_interpretation(in) != VIPS_HIP_INTERPRETATION_B_W) {
			error(domain, "images with fewer than 3 bands are outside the HIP path unless they are "
						  "one-band uchar B_W, not linear");
			return -1;
		}
		cur = in;
	}
	else {
		// already there (the usual sRGB uchar photograph): vips_colourspace is a pointer copy then, and this
		// function only reads the result -- caller-owned device memory, which vips_hip_colourspace would have
		// to copy to own, is read where it is
		const int target = linear ? VIPS_HIP_INTERPRETATION_scRGB : VIPS_HIP_INTERPRETATION_sRGB;
		const int have = guess_interpretation(in);
		const Route *route = nullptr;
		for (const Route &r : routes)
			if (r.from == have && r.to == target) {
				route = &r;
				break;
			}
		if (route && route->n == 0 && route->cast_format == in->format)
			cur = in;
		else {
			if (vips_hip_colourspace(in, &space.im, target))
				return -1;
			cur = space.im;
		}
	}

	// vips_thumbnail_calculate_shrink, thumbnail.c:413-467 (crop NONE, no rotate)
	double hshrink = (double) cur->width / width;
	double vshrink = (double) cur->height / height;
	// fit the box (the bigger shrink wins) or, when cropping, fill it (the smaller one)
	const bool horizontal = crop != 0 ? (hshrink < vshrink) : !(hshrink < vshrink);
	if (size != 3) { // != VIPS_SIZE_FORCE
		if (horizontal)
			vshrink = hshrink;
		else
			hshrink = vshrink;
	}
	if (size == 1) { // VIPS_SIZE_UP
		hshrink = hshrink < 1 ? hshrink : 1;
		vshrink = vshrink < 1 ? vshrink : 1;
	}
	else if (size == 2) { // VIPS_SIZE_DOWN
		hshrink = hshrink > 1 ? hshrink : 1;
		vshrink = vshrink > 1 ? vshrink : 1;
	}
	hshrink = hshrink < cur->width ? hshrink : cur->width;
	vshrink = vshrink < cur->height ? vshrink : cur->height;

	// vips_image_hasalpha: premultiply before shrinking (thumbnail.c:848-860), staying in
	// uchar when the image is uchar
	int unpremultiplied_format = -1;
	ImageRef pre;
	ImageRef resized;
	if (cur->bands > 3 && hshrink != 1.0 && vshrink != 1.0) {
		unpremultiplied_format = cur->format;
		// RGBA uchar with 8-bit alpha: the premultiply on the loads of the resize's first kernel (no premultiplied
		// image in between: 268 MB written and read again for an 8192 x 8192 input) when the chain of band kernels
		// takes the resize; else the operation, then the resize
		if (cur->format == VIPS_HIP_FORMAT_UCHAR && cur->bands == 4 && max_alpha(cur->interpretation) == 255.0) {
			const int r = vh::resize_premul_u8(cur, &resized.im, 1.0 / hshrink, 1.0 / vshrink);
			if (r < 0)
				return -1;
		}
		if (!resized.im) {
			if (vips_hip_premultiply(cur, &pre.im, cur->format == VIPS_HIP_FORMAT_UCHAR))
				return -1;
			cur = pre.im;
		}
	}

	if (!resized.im && vips_hip_resize(cur, &resized.im, 1.0 / hshrink, 1.0 / vshrink, VIPS_HIP_KERNEL_LANCZOS3, 2.0))
		return -1;

	if (unpremultiplied_format >= 0) { // thumbnail.c:886-904
		ImageRef un;
		if (unpremultiplied_format == VIPS_HIP_FORMAT_UCHAR) {
			if (vips_hip_unpremultiply(resized.im, &un.im, 1))
				return -1;
		}
		else {
			ImageRef f;
			if (vips_hip_unpremultiply(resized.im, &f.im, 0) ||
				vips_hip_cast(f.im, &un.im, unpremultiplied_format))
				return -1;
		}
		vips_hip_image_unref(resized.im);
		resized.im = un.release();
	}

	if (linear) { // thumbnail.c:973-987: back to sRGB
		ImageRef back;
		if (vips_hip_colourspace(resized.im, &back.im, VIPS_HIP_INTERPRETATION_sRGB))
			return -1;
		vips_hip_image_unref(resized.im);
		resized.im = back.release();
	}
	if (crop != 0) { // thumbnail.c:1010-1038 -> vips_smartcrop's positional modes
		const VipsHipImage *r = resized.im;
		int crop_width = width < r->width ? width : r->width;
		int crop_height = height < r->height ? height : r->height;
		int left = 0, top = 0;
		if (crop == 1) {
			left = (r->width - crop_width) / 2;
			top = (r->height - crop_height) / 2;
		}
		else if (crop == 5) {
			left = r->width - crop_width;
			top = r->height - crop_height;
		}
		else if (crop == 6) {
			crop_width = r->width;
			crop_height = r->height;
		}
		return vips_hip_extract_area(resized.im, out, left, top, crop_width, crop_height);
	}
	*out = resized.release();
	return 0;
}

// vips_sharpen_build, convolution/sharpen.c:171-302
// vips_sharpen on n 3-band uchar sRGB images of one size (every thumbnail): the LabS round trip,
// the blur of L and the LUT in one kernel (colour.hip sharpen_fused_u8) when the blur mask has at
// most 5 taps.  0 = done (out[] filled), 1 = not that kernel's case, -1 = error
static int sharpen_fused_images(VipsHipImage *const *in, int n_images, VipsHipImage **out, double sigma, double x1,
	double y2, double y3, double m1, double m2)
{
	for (int i = 0; i < n_images; i++)
		if (!in[i] || in[i]->format != VIPS_HIP_FORMAT_UCHAR || in[i]->bands != 3 ||
			in[i]->interpretation != VIPS_HIP_INTERPRETATION_sRGB ||
			guess_interpretation(in[i]) != VIPS_HIP_INTERPRETATION_sRGB || in[i]->width != in[0]->width ||
			in[i]->height != in[0]->height)
			return 1;
	const Route *to = nullptr, *from = nullptr;
	for (const Route &r : routes) {
		if (r.from == VIPS_HIP_INTERPRETATION_sRGB && r.to == VIPS_HIP_INTERPRETATION_LABS)
			to = &r;
		if (r.from == VIPS_HIP_INTERPRETATION_LABS && r.to == VIPS_HIP_INTERPRETATION_sRGB)
			from = &r;
	}
	const int n = vips_hip_gaussmat(sigma, 0.1, 1, VIPS_HIP_PRECISION_INTEGER, nullptr, 0, nullptr);
	if (!(to && from && to->n > 0 && from->n > 0 && n >= 1 && n <= 5))
		return 1;
	std::vector<double> mask(n);
	double scale = 1.0;
	if (vips_hip_gaussmat(sigma, 0.1, 1, VIPS_HIP_PRECISION_INTEGER, mask.data(), n, &scale) < 0)
		return -1;
	// the convi C path's integers: rint of the mask and of its scale (convi.c:886-915); zero taps
	// are squeezed out by the reference: same sums
	std::vector<int> coef(n);
	for (int k = 0; k < n; k++)
		coef[k] = (int) rint(mask[k]);
	LutPtr lut = sharpen_lut_cached(x1, y2, y3, m1, m2);
	if (!lut)
		return -1;
	std::vector<ImageRef> o(n_images);
	std::vector<VipsHipRegion> ri(n_images), ro(n_images);
	std::vector<const VipsHipRegion *> pi(n_images), po(n_images);
	for (int i = 0; i < n_images; i++) {
		o[i].im = vips_hip_image_new(in[i]->width, in[i]->height, 3, VIPS_HIP_FORMAT_UCHAR, VIPS_HIP_INTERPRETATION_sRGB);
		if (!o[i].im)
			return -1;
		vips_hip_image_region(in[i], &ri[i]);
		vips_hip_image_region(o[i].im, &ro[i]);
		pi[i] = &ri[i];
		po[i] = &ro[i];
	}
	std::shared_ptr<vh::SharpenLutWindow> win = sharpen_lut_window_cached(x1, y2, y3, m1, m2);
	const int r = vh::sharpen_fused_u8(pi.data(), po.data(), n_images, to->steps, to->n, from->steps, from->n,
		coef.data(), n, (int) rint(scale), lut.get(), win.get());
	if (r)
		return r;
	for (int i = 0; i < n_images; i++)
		out[i] = o[i].release();
	return 0;
}

int vips_hip_sharpen(VipsHipImage *in, VipsHipImage **out, double sigma, double x1, double y2,
	double y3, double m1, double m2)
{
	if (in && vh::bind_to(in)) // run where the pixels live
		return -1;
	if (!in || !out) {
		error("sharpen", "null argument");
		return -1;
	}
	const int old_interpretation = in->interpretation;
	{
		const int r = sharpen_fused_images(&in, 1, out, sigma, x1, y2, y3, m1, m2);
		if (r <= 0)
			return r;
	}
	ImageRef labs;
	if (vips_hip_colourspace(in, &labs.im, VIPS_HIP_INTERPRETATION_LABS))
		return -1;
	if (labs.im->bands < 3) {
		error("sharpen", "image must have at least 3 bands");
		return -1;
	}

	// "Stop at 10% of max ... We always sharpen a short, so there's no point using a
	// float mask."
	int width = vips_hip_gaussmat(sigma, 0.1, 1, VIPS_HIP_PRECISION_INTEGER, nullptr, 0, nullptr);
	if (width < 0)
		return -1;
	std::vector<double> mask(width);
	double scale = 1.0;
	if (vips_hip_gaussmat(sigma, 0.1, 1, VIPS_HIP_PRECISION_INTEGER, mask.data(), width, &scale) < 0)
		return -1;

	// vips_cast_short: colourspace(LABS) already produced short
	ImageRef shorts;
	VipsHipImage *cur = labs.im;
	if (cur->format != VIPS_HIP_FORMAT_SHORT) {
		if (cast_image(cur, &shorts.im, VIPS_HIP_FORMAT_SHORT))
			return -1;
		cur = shorts.im;
	}

	LutPtr lut = sharpen_lut_cached(x1, y2, y3, m1, m2);
	if (!lut)
		return -1;
	int *d_lut = lut.get();

	// extract L, blur it with the integer separable mask (sharpen.c:274-278)
	ImageRef L(vips_hip_image_new(cur->width, cur->height, 1, VIPS_HIP_FORMAT_SHORT,
		cur->interpretation));
	ImageRef blurred, sharp(vips_hip_image_new(cur->width, cur->height, cur->bands,
							   VIPS_HIP_FORMAT_SHORT, VIPS_HIP_INTERPRETATION_LABS));
	int result = -1;
	if (L.im && sharp.im) {
		VipsHipRegion rc, rl;
		vips_hip_image_region(cur, &rc);
		vips_hip_image_region(L.im, &rl);
		if (!band_cast(&rc, 0, &rl, 0, 1) &&
			!vips_hip_convsep(L.im, &blurred.im, mask.data(), width, scale, 0.0,
				VIPS_HIP_PRECISION_INTEGER)) {
			VipsHipRegion rb, rs;
			vips_hip_image_region(blurred.im, &rb);
			vips_hip_image_region(sharp.im, &rs);
			result = vips_hip_sharpen_gen(d_lut, &rc, &rb, &rs);
		}
	}
	if (result)
		return -1;

	// back to where we came from (sharpen.c:295-297)
	return vips_hip_colourspace(sharp.im, out, old_interpretation);
}

} // extern "C"
