// Internal declarations shared by the HIP translation units of libvipship.so.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <memory>

#include "vips_hip.h"

namespace vh {

// iofuncs/error.c shaped error log: "domain: message\n" appended to a
// thread-local buffer; every failing entry point returns -1 after calling it.
void error(const char *domain, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_failed(hipError_t err, const char *what);

#define VH_CHECK(expr) \
	do { \
		hipError_t vh_err_ = (expr); \
		if (vh_err_ != hipSuccess) \
			return vh::hip_failed(vh_err_, #expr); \
	} while (0)

#define VH_CHECK_NULL(expr) \
	do { \
		hipError_t vh_err_ = (expr); \
		if (vh_err_ != hipSuccess) { \
			vh::hip_failed(vh_err_, #expr); \
			return nullptr; \
		} \
	} while (0)

// Make sure a device has been selected for this thread (vips_hip_init(0) on
// first use) -- returns -1 with an error when no GPU is present.
int ensure_init();

// The device the calling thread is bound to (-1: none yet).
int current_device();

hipStream_t stream();
// Synchronise and destroy the calling thread's own stream (threads the library starts itself).
void release_thread_stream();

// The calling thread launches on `s` while this lives (nullptr: no change).  For work the
// library spreads over a second stream itself; whoever does so orders the streams with events
// and keeps pool blocks both streams touch alive until both are synchronised.
struct ScopedStream {
	explicit ScopedStream(hipStream_t s);
	~ScopedStream();
	hipStream_t saved;
	int slot; // the device slot `saved` came from
	bool saved_external, active;
};

// Kernel gates: VIPS_GATE_START/STOP analogue around a launch.
struct Gate {
	explicit Gate(const char *name);
	~Gate();
	const char *name;
	hipEvent_t start;
	bool active;
};

// sizeof one band element
static inline int format_sizeof(int format)
{
	switch (format) {
	case VIPS_HIP_FORMAT_UCHAR:
	case VIPS_HIP_FORMAT_CHAR:
		return 1;
	case VIPS_HIP_FORMAT_USHORT:
	case VIPS_HIP_FORMAT_SHORT:
		return 2;
	case VIPS_HIP_FORMAT_UINT:
	case VIPS_HIP_FORMAT_INT:
	case VIPS_HIP_FORMAT_FLOAT:
		return 4;
	case VIPS_HIP_FORMAT_COMPLEX:
	case VIPS_HIP_FORMAT_DOUBLE:
		return 8;
	case VIPS_HIP_FORMAT_DPCOMPLEX:
		return 16;
	default:
		return 0;
	}
}

static inline bool format_iscomplex(int format)
{
	return format == VIPS_HIP_FORMAT_COMPLEX || format == VIPS_HIP_FORMAT_DPCOMPLEX;
}

static inline bool format_isint(int format)
{
	return format >= VIPS_HIP_FORMAT_UCHAR && format <= VIPS_HIP_FORMAT_INT;
}

// Complex images are processed as twice as many bands of the real type
// (reduceh.cpp:227-228, shrinkh.c:162-163, convi.c:770-771).
static inline int format_real(int format)
{
	if (format == VIPS_HIP_FORMAT_COMPLEX)
		return VIPS_HIP_FORMAT_FLOAT;
	if (format == VIPS_HIP_FORMAT_DPCOMPLEX)
		return VIPS_HIP_FORMAT_DOUBLE;
	return format;
}

static inline int region_elems_per_pel(const VipsHipRegion *r)
{
	return r->bands * (format_iscomplex(r->format) ? 2 : 1);
}

// Common checks on a pair of regions handed to a gen.
int check_region(const char *domain, const VipsHipRegion *r);

// A plan handle's device tables live on ONE device: the device of the thread that first runs
// it (*device < 0: taken now).  Running it from a thread bound to another device fails loudly.
int plan_device(const char *domain, std::atomic<int> *device);

// Small device-resident table upload with caching handled by the callers.
void *upload(const void *host, size_t size);

} // namespace vh

// The image object (layer 3).
struct _VipsHipImage {
	void *data;
	int device = -1; // the device the pixels live on
	int width, height, bands, format, interpretation;
	size_t stride;
	bool owns; // data came from the pool
	// library memory is shared between image objects (vips_copy-style no-op results): the
	// last holder returns it to the pool
	std::shared_ptr<void> hold;
};

namespace vh {
// Run where the data lives: make sure a device is selected and bind the calling thread to the
// device `image` is on (every image-level operation starts with this).
int bind_to(const _VipsHipImage *image);
// A second image object on the same pixels (library-owned images only; nullptr otherwise).
_VipsHipImage *image_share(const _VipsHipImage *in);
} // namespace vh
