// JPEG shrink-on-load in front of the device path: what
// `vips_thumbnail("x.jpg", ...)` does before any pixel reaches a kernel.
//
//   vips_thumbnail_find_jpegshrink  resample/thumbnail.c:488-519: pick the DCT-domain block
//                                   shrink 1/2/4/8 that leaves at least a factor of two for the
//                                   final resize (1 in linear mode)
//   read_jpeg_header / _generate    foreign/jpeg2vips.c:517-640, 800-905: libjpeg with
//                                   scale_num / scale_denom = 1 / shrink and otherwise its
//                                   defaults, output cropped to image size / shrink rounded
//                                   DOWN (libjpeg rounds up and pads), CMYK inverted
//   vips_thumbnail_build            thumbnail.c:678-1067 on the pre-shrunk image: the
//                                   thumbnail_image pipeline of ops_colour_conv.cpp
//
// The entropy decode is libjpeg's and runs on the host (the same IJG library the reference
// build links, loaded with dlopen so that libvipship.so itself does not depend on it); the
// pre-shrunk image -- 1/4 to 1/64 of the pixels -- is uploaded and everything after it runs on
// the device.  Auto-rotation (EXIF orientation other than 1) and ICC colour management (an
// embedded profile in linear mode) are outside the path: such files are refused rather than
// thumbnailed wrongly.
#include "internal.h"

#include <csetjmp>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include <dlfcn.h>

#ifdef VIPS_HIP_HAVE_JPEGLIB
#include <jpeglib.h>
#endif

using namespace vh;

namespace {

// vips_thumbnail_calculate_shrink (thumbnail.c:413-467; crop none, no rotation) -- the same
// arithmetic vips_hip_thumbnail_image applies to the image it is handed.
void calculate_shrink(int in_width, int in_height, int width, int height, int size, int crop, double *hshrink,
	double *vshrink)
{
	double hs = (double) in_width / width;
	double vs = (double) in_height / height;
	const bool horizontal = crop != 0 ? (hs < vs) : !(hs < vs);
	if (size != 3) { // != VIPS_SIZE_FORCE
		if (horizontal)
			vs = hs;
		else
			hs = vs;
	}
	if (size == 1) { // VIPS_SIZE_UP
		hs = hs < 1 ? hs : 1;
		vs = vs < 1 ? vs : 1;
	}
	else if (size == 2) { // VIPS_SIZE_DOWN
		hs = hs > 1 ? hs : 1;
		vs = vs > 1 ? vs : 1;
	}
	hs = hs < in_width ? hs : in_width;
	vs = vs < in_height ? vs : in_height;
	*hshrink = hs;
	*vshrink = vs;
}

#ifdef VIPS_HIP_HAVE_JPEGLIB

// the entry points of libjpeg this file calls, bound at first use
struct JpegApi {
	void *handle;
	struct jpeg_error_mgr *(*std_error)(struct jpeg_error_mgr *);
	void (*create_decompress)(j_decompress_ptr, int, size_t);
	void (*destroy_decompress)(j_decompress_ptr);
	void (*stdio_src)(j_decompress_ptr, FILE *);
	void (*save_markers)(j_decompress_ptr, int, unsigned int);
	int (*read_header)(j_decompress_ptr, boolean);
	void (*calc_output_dimensions)(j_decompress_ptr);
	boolean (*start_decompress)(j_decompress_ptr);
	JDIMENSION (*read_scanlines)(j_decompress_ptr, JSAMPARRAY, JDIMENSION);
	bool ok;
};

JpegApi *jpeg_api()
{
	static JpegApi api;
	static std::once_flag once;
	std::call_once(once, [] {
		memset(&api, 0, sizeof(api));
		const char *names[] = { "libjpeg.so.9", "/opt/conda/lib/libjpeg.so.9", nullptr };
		for (int i = 0; names[i] && !api.handle; i++)
			api.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
		if (!api.handle)
			return;
		api.std_error = (decltype(api.std_error)) dlsym(api.handle, "jpeg_std_error");
		api.create_decompress = (decltype(api.create_decompress)) dlsym(api.handle, "jpeg_CreateDecompress");
		api.destroy_decompress = (decltype(api.destroy_decompress)) dlsym(api.handle, "jpeg_destroy_decompress");
		api.stdio_src = (decltype(api.stdio_src)) dlsym(api.handle, "jpeg_stdio_src");
		api.save_markers = (decltype(api.save_markers)) dlsym(api.handle, "jpeg_save_markers");
		api.read_header = (decltype(api.read_header)) dlsym(api.handle, "jpeg_read_header");
		api.calc_output_dimensions =
			(decltype(api.calc_output_dimensions)) dlsym(api.handle, "jpeg_calc_output_dimensions");
		api.start_decompress = (decltype(api.start_decompress)) dlsym(api.handle, "jpeg_start_decompress");
		api.read_scanlines = (decltype(api.read_scanlines)) dlsym(api.handle, "jpeg_read_scanlines");
		api.ok = api.std_error && api.create_decompress && api.destroy_decompress && api.stdio_src &&
			api.save_markers && api.read_header && api.calc_output_dimensions && api.start_decompress &&
			api.read_scanlines;
	});
	return api.ok ? &api : nullptr;
}

struct ErrorManager {
	struct jpeg_error_mgr pub;
	jmp_buf jump;
	char message[JMSG_LENGTH_MAX];
};

void on_error_exit(j_common_ptr cinfo)
{
	ErrorManager *em = (ErrorManager *) cinfo->err;
	(*cinfo->err->format_message)(cinfo, em->message);
	longjmp(em->jump, 1);
}

void on_output_message(j_common_ptr) // warnings: fail_on none (jpeg2vips.c:868-877)
{
}

// EXIF orientation (tag 0x0112 of IFD0) from an APP1 segment; 0 when there is none.
int exif_orientation(const unsigned char *d, unsigned int n)
{
	if (n < 14 || memcmp(d, "Exif\0\0", 6) != 0)
		return 0;
	const unsigned char *t = d + 6;
	const unsigned int tn = n - 6;
	bool big;
	if (t[0] == 'M' && t[1] == 'M')
		big = true;
	else if (t[0] == 'I' && t[1] == 'I')
		big = false;
	else
		return 0;
	auto u16 = [&](unsigned int at) -> unsigned int {
		return big ? (t[at] << 8 | t[at + 1]) : (t[at + 1] << 8 | t[at]);
	};
	auto u32 = [&](unsigned int at) -> unsigned int {
		return big ? ((unsigned int) t[at] << 24 | t[at + 1] << 16 | t[at + 2] << 8 | t[at + 3])
				   : ((unsigned int) t[at + 3] << 24 | t[at + 2] << 16 | t[at + 1] << 8 | t[at]);
	};
	const unsigned int ifd = u32(4);
	if (ifd > tn || tn - ifd < 2)
		return 0;
	const unsigned int entries = u16(ifd);
	for (unsigned int i = 0; i < entries; i++) {
		const unsigned int at = ifd + 2 + 12 * i;
		if (at + 12 > tn)
			break;
		if (u16(at) == 0x0112)
			return (int) u16(at + 8);
	}
	return 0;
}

// Decode @path with the block shrink @shrink.  header_only: fill @h and stop.  Otherwise
// h->width * h->height * h->bands bytes go to @pixels.  Plain C inside (setjmp).
int decode(const char *path, int shrink, VipsHipJpegHeader *h, unsigned char *pixels, size_t size,
	bool header_only)
{
	JpegApi *api = jpeg_api();
	if (!api) {
		error("jpegload", "libjpeg.so.9 not found: JPEG loading is not available");
		return -1;
	}
	if (shrink != 1 && shrink != 2 && shrink != 4 && shrink != 8) {
		error("jpegload", "bad shrink factor %d", shrink); // jpeg2vips.c / jpegload.c: 1, 2, 4, 8
		return -1;
	}
	FILE *f = fopen(path, "rb");
	if (!f) {
		error("jpegload", "unable to open \"%s\"", path);
		return -1;
	}
	struct jpeg_decompress_struct cinfo;
	ErrorManager em;
	// written between setjmp and a possible longjmp, read after it
	unsigned char *volatile line = nullptr;
	volatile bool created = false;

	cinfo.err = api->std_error(&em.pub);
	em.pub.error_exit = on_error_exit;
	em.pub.output_message = on_output_message;
	if (setjmp(em.jump)) {
		error("jpegload", "%s", em.message);
		if (created)
			api->destroy_decompress(&cinfo);
		free(line);
		fclose(f);
		return -1;
	}
	api->create_decompress(&cinfo, JPEG_LIB_VERSION, sizeof(struct jpeg_decompress_struct));
	created = true;
	api->stdio_src(&cinfo, f);
	api->save_markers(&cinfo, JPEG_APP0 + 1, 0xffff);
	api->save_markers(&cinfo, JPEG_APP0 + 2, 0xffff);

	api->read_header(&cinfo, TRUE);
	cinfo.scale_denom = shrink;
	cinfo.scale_num = 1;
	api->calc_output_dimensions(&cinfo);

	h->image_width = (int) cinfo.image_width;
	h->image_height = (int) cinfo.image_height;
	// strictly round down: libjpeg rounds up and pads with black (jpeg2vips.c:628-640)
	h->width = (int) cinfo.image_width / shrink;
	h->height = (int) cinfo.image_height / shrink;
	h->bands = cinfo.output_components;
	bool invert = false;
	switch (cinfo.out_color_space) {
	case JCS_GRAYSCALE:
		h->interpretation = VIPS_HIP_INTERPRETATION_B_W;
		break;
	case JCS_CMYK:
		h->interpretation = 15; // VIPS_INTERPRETATION_CMYK
		invert = true;
		break;
	default:
		h->interpretation = VIPS_HIP_INTERPRETATION_sRGB;
		break;
	}
	h->orientation = 0;
	h->has_icc = 0;
	for (jpeg_saved_marker_ptr p = cinfo.marker_list; p; p = p->next) {
		if (p->marker == JPEG_APP0 + 1 && !h->orientation)
			h->orientation = exif_orientation(p->data, p->data_length);
		if (p->marker == JPEG_APP0 + 2 && p->data_length > 14 && memcmp(p->data, "ICC_PROFILE", 11) == 0)
			h->has_icc = 1;
	}
	int result = 0;
	if (h->width < 1 || h->height < 1) {
		error("jpegload", "image has shrunk to nothing");
		result = -1;
	}
	else if (!header_only) {
		const size_t row_out = (size_t) h->width * h->bands;
		if (size < row_out * h->height) {
			error("jpegload", "buffer too small");
			result = -1;
		}
		else {
			const size_t row_lib = (size_t) cinfo.output_width * cinfo.output_components;
			line = (unsigned char *) malloc(row_lib);
			api->start_decompress(&cinfo);
			for (int y = 0; y < h->height; y++) {
				JSAMPROW rows[1] = { line };
				const unsigned char *src = line;
				api->read_scanlines(&cinfo, rows, 1);
				unsigned char *q = pixels + row_out * y;
				if (invert)
					for (size_t x = 0; x < row_out; x++)
						q[x] = 255 - src[x];
				else
					memcpy(q, src, row_out);
			}
		}
	}
	// the rows the crop does not need are never asked for: no jpeg_finish_decompress
	api->destroy_decompress(&cinfo);
	free(line);
	fclose(f);
	return result;
}

#endif // VIPS_HIP_HAVE_JPEGLIB

bool is_jpeg(const char *path)
{
	FILE *f = fopen(path, "rb");
	if (!f)
		return false;
	unsigned char m[2] = { 0, 0 };
	const size_t n = fread(m, 1, 2, f);
	fclose(f);
	return n == 2 && m[0] == 0xff && m[1] == 0xd8;
}

} // namespace

extern "C" {

int vips_hip_thumbnail_find_jpegshrink(int in_width, int in_height, int width, int height, int size,
	int linear, int crop)
{
	if (in_width <= 0 || in_height <= 0 || width <= 0) {
		error("thumbnail", "bad dimensions");
		return -1;
	}
	if (height <= 0)
		height = width;
	double hshrink, vshrink;
	calculate_shrink(in_width, in_height, width, height, size, crop, &hshrink, &vshrink);
	const double shrink = hshrink < vshrink ? hshrink : vshrink;
	// libjpeg shrinks in Y of YCbCr, not in linear light (thumbnail.c:497-501)
	if (linear)
		return 1;
	if (shrink >= 16)
		return 8;
	if (shrink >= 8)
		return 4;
	if (shrink >= 4)
		return 2;
	return 1;
}

int vips_hip_jpeg_read_header(const char *path, int shrink, VipsHipJpegHeader *header)
{
	if (!path || !header) {
		error("jpegload", "null argument");
		return -1;
	}
#ifdef VIPS_HIP_HAVE_JPEGLIB
	return decode(path, shrink, header, nullptr, 0, true);
#else
	error("jpegload", "built without jpeglib.h: JPEG loading is not available");
	return -1;
#endif
}

int vips_hip_jpeg_read_to_memory(const char *path, int shrink, void *host_data, size_t size)
{
	if (!path || !host_data) {
		error("jpegload", "null argument");
		return -1;
	}
#ifdef VIPS_HIP_HAVE_JPEGLIB
	VipsHipJpegHeader h;
	return decode(path, shrink, &h, (unsigned char *) host_data, size, false);
#else
	error("jpegload", "built without jpeglib.h: JPEG loading is not available");
	return -1;
#endif
}

VipsHipImage *vips_hip_image_new_from_jpeg(const char *path, int shrink)
{
	VipsHipJpegHeader h;
	if (vips_hip_jpeg_read_header(path, shrink, &h))
		return nullptr;
	std::vector<unsigned char> pixels((size_t) h.width * h.height * h.bands);
	if (vips_hip_jpeg_read_to_memory(path, shrink, pixels.data(), pixels.size()))
		return nullptr;
	return vips_hip_image_new_from_memory(pixels.data(), h.width, h.height, h.bands, VIPS_HIP_FORMAT_UCHAR,
		h.interpretation);
}

// vips_thumbnail (thumbnail.c:1130-1330 file class + :549-676 open + :678-1067 build) for
// JPEG and .v files.
int vips_hip_thumbnail(const char *path, VipsHipImage **out, int width, int height, int size, int linear,
	int crop)
{
	if (!path || !out) {
		error("thumbnail", "null argument");
		return -1;
	}
	if (width <= 0) {
		error("thumbnail", "parameter width not set");
		return -1;
	}
	VipsHipImage *loaded = nullptr;
	if (is_jpeg(path)) {
		VipsHipJpegHeader h;
		if (vips_hip_jpeg_read_header(path, 1, &h))
			return -1;
		if (h.orientation > 1) {
			error("thumbnail", "\"%s\" needs auto-rotation (EXIF orientation %d): outside the HIP path", path,
				h.orientation);
			return -1;
		}
		// an embedded profile only matters when it would be used: in linear mode the reference
		// imports through it (thumbnail.c:769-790); otherwise, with no export profile asked for,
		// it is carried as metadata and the pixels are not touched
		if (h.has_icc && linear) {
			error("thumbnail", "\"%s\" carries an ICC profile and linear mode would import through it: "
							   "colour management is outside the HIP path", path);
			return -1;
		}
		if (h.bands != 1 && h.bands != 3) {
			error("thumbnail", "\"%s\": CMYK JPEGs need an ICC import: outside the HIP path", path);
			return -1;
		}
		const int factor = vips_hip_thumbnail_find_jpegshrink(h.width, h.height, width, height, size, linear, crop);
		if (factor < 0)
			return -1;
		loaded = vips_hip_image_new_from_jpeg(path, factor);
	}
	else
		loaded = vips_hip_image_new_from_vfile(path);
	if (!loaded)
		return -1;
	const int r = vips_hip_thumbnail_image_crop(loaded, out, width, height, size, linear, crop);
	vips_hip_image_unref(loaded);
	return r;
}

// A batch of files on `n_threads` host threads: every thread decodes, uploads and thumbnails on
// its own stream, so the host cores' entropy decoding (the bottleneck once resize runs at HBM
// speed) and the device work of different files overlap.  The reference gets the same effect
// from a caller running vips_thumbnail() on several threads; BASELINE config C4 is this shape.
// Returns the number of files that failed; outs[i] is NULL and errors[i] (if given, 256 bytes
// each) holds the message for those.
int vips_hip_thumbnail_batch(const char *const *paths, int n, VipsHipImage **outs, char *errors, int width,
	int height, int size, int linear, int crop, int n_threads)
{
	if (!paths || !outs || n < 0) {
		error("thumbnail", "null argument");
		return -1;
	}
	if (n_threads < 1)
		n_threads = 1;
	if (n_threads > n)
		n_threads = n;
	std::atomic<int> next(0), failed(0);
	auto worker = [&]() {
		for (;;) {
			const int i = next.fetch_add(1);
			if (i >= n)
				break;
			outs[i] = nullptr;
			if (vips_hip_thumbnail(paths[i], &outs[i], width, height, size, linear, crop)) {
				outs[i] = nullptr;
				failed.fetch_add(1);
				if (errors) {
					strncpy(errors + (size_t) i * 256, vips_hip_error_buffer(), 255);
					errors[(size_t) i * 256 + 255] = 0;
				}
				vips_hip_error_clear(); // the error buffer is per thread
			}
			else if (errors)
				errors[(size_t) i * 256] = 0;
		}
	};
	if (n_threads <= 1) {
		worker();
		(void) vips_hip_synchronize();
	}
	else {
		std::vector<std::thread> pool;
		// results cross to the caller's thread, and a pool thread's cached blocks go to the
		// global list when it ends: finish (and give back) the thread's stream first
		for (int t = 0; t < n_threads; t++)
			pool.emplace_back([&]() {
				worker();
				release_thread_stream();
			});
		for (std::thread &t : pool)
			t.join();
	}
	return failed.load();
}

} // extern "C"
