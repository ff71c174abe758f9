// The libvips native ".v" format next to the device path: load a .v
// file straight into HBM and save a device image as one, with the disc and the PCIe link kept
// busy at the same time.
//
// Format (doc/file-format.md; iofuncs/vips.c:283-298 field table, :301-394
// vips__read_header_bytes, :396-441 vips__write_header_bytes): a 64-byte header -- the magic
// number written most-significant byte first, then Xsize, Ysize, Bands, Bbits, BandFmt, Coding,
// Type (int32), Xres, Yres (float32), Length (int32), Compression, Level (int16), Xoffset,
// Yoffset (int32) in the byte order the magic names, zero padding -- followed by the pixels as
// band-interleaved scanlines without padding, followed by an optional XML block of metadata.
// A file shorter than header + pixels is refused (iofuncs/image.c:966-979); anything behind the
// pixels is metadata this path does not carry.
//
// What a sink does with two buffers on the CPU (sinkdisc.c:195-220: one buffer is written in
// the background while workers fill the other) is done here with two pinned host buffers: the
// disc read of chunk k+1 overlaps the host-to-device copy of chunk k, and the device-to-host
// copy of chunk k+1 overlaps the write() of chunk k.
#include "internal.h"

#include <cerrno>
#include <cstdio>
#include <cstring>

#include <sys/stat.h>

using namespace vh;

namespace {

const unsigned int MAGIC_INTEL = 0xb6a6f208u; // include/vips/image.h:49-56
const unsigned int MAGIC_SPARC = 0x08f2a6b6u;
const size_t HEADER_BYTES = 64;
const size_t CHUNK_BYTES = 32u << 20;

// VipsInterpretation values with a name (include/vips/image.h:94-118)
bool known_interpretation(int v)
{
	return v == 0 || v == 1 || v == 10 || v == 12 || v == 13 || (v >= 15 && v <= 19) || (v >= 21 && v <= 31);
}

unsigned int get32(const unsigned char *p, bool msb)
{
	return msb ? ((unsigned int) p[0] << 24 | (unsigned int) p[1] << 16 | (unsigned int) p[2] << 8 | p[3])
			   : ((unsigned int) p[3] << 24 | (unsigned int) p[2] << 16 | (unsigned int) p[1] << 8 | p[0]);
}

void put32(unsigned char *p, unsigned int v)
{
	p[0] = (unsigned char) v;
	p[1] = (unsigned char) (v >> 8);
	p[2] = (unsigned char) (v >> 16);
	p[3] = (unsigned char) (v >> 24);
}

float as_float(unsigned int bits)
{
	float f;
	memcpy(&f, &bits, 4);
	return f;
}

unsigned int as_bits(float f)
{
	unsigned int bits;
	memcpy(&bits, &f, 4);
	return bits;
}

int clip(int lo, int v, int hi)
{
	return v < lo ? lo : (v > hi ? hi : v);
}

// vips__read_header_bytes, iofuncs/vips.c:301-394
int parse_header(const char *path, const unsigned char *h, VipsHipVHeader *out)
{
	const unsigned int magic = get32(h, true); // the magic is always stored MSB first
	if (magic != MAGIC_INTEL && magic != MAGIC_SPARC) {
		error("VipsImage", "\"%s\" is not a VIPS image", path);
		return -1;
	}
	const bool msb = magic == MAGIC_SPARC;
	out->msb_first = msb;
	out->width = clip(1, (int) get32(h + 4, msb), 10000000); // VIPS_MAX_COORD
	out->height = clip(1, (int) get32(h + 8, msb), 10000000);
	out->bands = clip(1, (int) get32(h + 12, msb), 10000000);
	out->format = clip(0, (int) get32(h + 20, msb), 9);
	out->coding = (int) get32(h + 24, msb);
	out->interpretation = (int) get32(h + 28, msb);
	const float xres = as_float(get32(h + 32, msb)), yres = as_float(get32(h + 36, msb));
	out->xres = xres > 0 ? xres : 0;
	out->yres = yres > 0 ? yres : 0;
	out->xoffset = (int) get32(h + 48, msb);
	out->yoffset = (int) get32(h + 52, msb);
	if (!known_interpretation(out->interpretation))
		out->interpretation = -1; // VIPS_INTERPRETATION_ERROR
	// VipsCoding: 0 none, 2 LABQ, 6 RAD
	if (out->coding != 0 && out->coding != 2 && out->coding != 6) {
		error("VipsImage", "unknown coding");
		return -1;
	}
	if (out->coding == 2 && (out->bands != 4 || out->format != VIPS_HIP_FORMAT_UCHAR)) {
		error("VipsImage", "malformed LABQ image");
		return -1;
	}
	if (out->coding == 6 && (out->bands != 4 || out->format != VIPS_HIP_FORMAT_UCHAR)) {
		error("VipsImage", "malformed RAD image");
		return -1;
	}
	out->data_offset = (long long) HEADER_BYTES;
	// The header is untrusted: 1e7 x 1e7 x 1e7 x 16 bytes wraps a long long.  Checked products, and
	// nothing the device could hold anyway (1 TiB) gets past here.
	long long size = 0;
	if (__builtin_mul_overflow((long long) out->width, (long long) out->height, &size) ||
		__builtin_mul_overflow(size, (long long) out->bands, &size) ||
		__builtin_mul_overflow(size, (long long) format_sizeof(out->format), &size) ||
		size <= 0 || size > (1LL << 40)) {
		error("VipsImage", "image dimensions %d x %d x %d are too large", out->width, out->height, out->bands);
		return -1;
	}
	out->data_size = size;
	return 0;
}

struct File {
	FILE *f;
	explicit File(FILE *file)
		: f(file)
	{
	}
	~File()
	{
		if (f)
			fclose(f);
	}
};

// two pinned staging buffers with an event each
struct Staging {
	void *buf[2];
	hipEvent_t done[2];
	bool ok;
	Staging()
	{
		buf[0] = buf[1] = nullptr;
		done[0] = done[1] = nullptr;
		ok = hipHostMalloc(&buf[0], CHUNK_BYTES, hipHostMallocDefault) == hipSuccess &&
			hipHostMalloc(&buf[1], CHUNK_BYTES, hipHostMallocDefault) == hipSuccess &&
			hipEventCreateWithFlags(&done[0], hipEventDisableTiming) == hipSuccess &&
			hipEventCreateWithFlags(&done[1], hipEventDisableTiming) == hipSuccess;
	}
	~Staging()
	{
		for (int i = 0; i < 2; i++) {
			if (buf[i])
				(void) hipHostFree(buf[i]);
			if (done[i])
				(void) hipEventDestroy(done[i]);
		}
	}
};

} // namespace

extern "C" {

int vips_hip_vfile_read_header(const char *path, VipsHipVHeader *header)
{
	if (!path || !header) {
		error("VipsImage", "null argument");
		return -1;
	}
	File file(fopen(path, "rb"));
	if (!file.f) {
		error("VipsImage", "unable to open \"%s\": %s", path, strerror(errno));
		return -1;
	}
	unsigned char h[HEADER_BYTES];
	if (fread(h, 1, HEADER_BYTES, file.f) != HEADER_BYTES) {
		error("VipsImage", "unable to read header for \"%s\"", path);
		return -1;
	}
	if (parse_header(path, h, header))
		return -1;
	struct stat st;
	if (fstat(fileno(file.f), &st)) {
		error("VipsImage", "unable to get file stats for \"%s\"", path);
		return -1;
	}
	if ((long long) st.st_size - (long long) HEADER_BYTES < header->data_size) {
		error("VipsImage", "unable to open \"%s\", file too short", path);
		return -1;
	}
	return 0;
}

VipsHipImage *vips_hip_image_new_from_vfile(const char *path)
{
	VipsHipVHeader h;
	if (vips_hip_vfile_read_header(path, &h))
		return nullptr;
	if (h.coding != 0) {
		error("VipsImage", "\"%s\": LABQ / RAD coded files are outside the HIP path", path);
		return nullptr;
	}
	if (h.msb_first) {
		error("VipsImage", "\"%s\": big-endian pixel data is outside the HIP path", path);
		return nullptr;
	}
	if (ensure_init())
		return nullptr;
	VipsHipImage *im = vips_hip_image_new(h.width, h.height, h.bands, h.format,
		h.interpretation < 0 ? 0 : h.interpretation);
	if (!im)
		return nullptr;
	File file(fopen(path, "rb"));
	Staging st;
	if (!file.f || fseek(file.f, (long) h.data_offset, SEEK_SET) || !st.ok) {
		error("VipsImage", "unable to read \"%s\"", path);
		vips_hip_image_unref(im);
		return nullptr;
	}
	const size_t total = (size_t) h.data_size;
	size_t done = 0;
	bool failed = false;
	for (int k = 0; done < total && !failed; k++) {
		const int slot = k & 1;
		const size_t n = total - done < CHUNK_BYTES ? total - done : CHUNK_BYTES;
		// the copy that last used this buffer (chunk k - 2) must have left it
		if (k >= 2 && hipEventSynchronize(st.done[slot]) != hipSuccess)
			failed = true;
		if (!failed && fread(st.buf[slot], 1, n, file.f) != n) {
			error("VipsImage", "unable to read data for \"%s\"", path);
			failed = true;
		}
		if (!failed &&
			(hipMemcpyAsync((char *) im->data + done, st.buf[slot], n, hipMemcpyHostToDevice, stream()) != hipSuccess ||
				hipEventRecord(st.done[slot], stream()) != hipSuccess)) {
			error("VipsImage", "host to device copy failed");
			failed = true;
		}
		done += n;
	}
	if (hipStreamSynchronize(stream()) != hipSuccess && !failed) {
		error("VipsImage", "host to device copy failed");
		failed = true;
	}
	if (failed) {
		vips_hip_image_unref(im);
		return nullptr;
	}
	return im;
}

int vips_hip_image_write_to_vfile(const VipsHipImage *image, const char *path)
{
	if (image && vh::bind_to(image)) // run where the pixels live
		return -1;
	if (!image || !path) {
		error("VipsImage", "null argument");
		return -1;
	}
	if (ensure_init())
		return -1;
	const size_t line = (size_t) image->width * image->bands * format_sizeof(image->format);
	const size_t total = line * image->height;
	if (image->stride != line) {
		error("VipsImage", "padded images are outside the .v writer");
		return -1;
	}
	File file(fopen(path, "wb"));
	if (!file.f) {
		error("VipsImage", "unable to open \"%s\" for writing: %s", path, strerror(errno));
		return -1;
	}
	// vips__write_header_bytes, iofuncs/vips.c:396-441, for a little-endian writer
	unsigned char h[HEADER_BYTES];
	memset(h, 0, sizeof(h));
	h[0] = 0xb6;
	h[1] = 0xa6;
	h[2] = 0xf2;
	h[3] = 0x08;
	put32(h + 4, (unsigned int) image->width);
	put32(h + 8, (unsigned int) image->height);
	put32(h + 12, (unsigned int) image->bands);
	put32(h + 16, (unsigned int) (format_sizeof(image->format) << 3)); // Bbits
	put32(h + 20, (unsigned int) image->format);
	put32(h + 24, 0); // VIPS_CODING_NONE
	put32(h + 28, (unsigned int) image->interpretation);
	put32(h + 32, as_bits(1.0f)); // Xres / Yres: the default of a memory image
	put32(h + 36, as_bits(1.0f));
	if (fwrite(h, 1, HEADER_BYTES, file.f) != HEADER_BYTES) {
		error("VipsImage", "write failed for \"%s\"", path);
		return -1;
	}
	Staging st;
	if (!st.ok) {
		error("VipsImage", "unable to allocate staging buffers");
		return -1;
	}
	// chunk k + 1 comes down from the device while chunk k goes to the disc
	auto fetch = [&](int k, size_t at) -> size_t {
		const size_t n = total - at < CHUNK_BYTES ? total - at : CHUNK_BYTES;
		if (hipMemcpyAsync(st.buf[k & 1], (const char *) image->data + at, n, hipMemcpyDeviceToHost, stream()) != hipSuccess ||
			hipEventRecord(st.done[k & 1], stream()) != hipSuccess)
			return 0;
		return n;
	};
	size_t at = 0;
	size_t n = total ? fetch(0, 0) : 0;
	if (total && !n) {
		error("VipsImage", "device to host copy failed");
		return -1;
	}
	for (int k = 0; at < total; k++) {
		const size_t next_at = at + n;
		size_t next_n = 0;
		if (next_at < total) {
			next_n = fetch(k + 1, next_at);
			if (!next_n) {
				error("VipsImage", "device to host copy failed");
				return -1;
			}
		}
		if (hipEventSynchronize(st.done[k & 1]) != hipSuccess) {
			error("VipsImage", "device to host copy failed");
			return -1;
		}
		if (fwrite(st.buf[k & 1], 1, n, file.f) != n) {
			error("VipsImage", "write failed for \"%s\"", path);
			return -1;
		}
		at = next_at;
		n = next_n;
	}
	if (fflush(file.f)) {
		error("VipsImage", "write failed for \"%s\"", path);
		return -1;
	}
	return 0;
}

} // extern "C"
