// An image held as row strips on several devices of ONE process, and the halo exchange between
// them: BASELINE config 5 (vips_conv 31 x 31 on a 65536 x 65536 ushort image over 8 GPUs) from C,
// the way libvips itself is parallel -- threads inside one process (iofuncs/threadpool.c:625).
//
//   * strip k owns rows [row0_k, row1_k) (a near-equal contiguous split) and lives on devices[k]
//     in a PERSISTENT window: its own rows with room for `halo` rows of its neighbours above and
//     below (clipped at the image edges: kernels clamp there, vips_embed COPY);
//   * vips_hip_strips_exchange() copies every halo row straight from its owner's window into
//     the window that needs it with hipMemcpyPeerAsync (xGMI between the GPUs of a node: peer
//     access is enabled pairwise) -- one copy per (owner, needer) pair, ~2 MB per side for C5, no
//     staging buffers, nothing rebuilt per step;
//   * vips_hip_conv_strips() runs the exchange and then the ordinary region operation
//     (vips_hip_conv_gen) on every window, one host thread per strip bound to the strip's device;
//     the output strips stay on their devices.
//
// The multi-PROCESS form of the same partition (one rank per GPU, RCCL send / recv for the halos)
// is libvips_amd/sharding.py; both give the single-device pixels bit for bit.
#include "conv.h"

#include <string>
#include <thread>
#include <vector>

using namespace vh;

struct _VipsHipStrips {
	int im_width, im_height, bands, format, halo;
	size_t stride;
	struct Strip {
		int device;
		int row0, row1; // own rows
		int top, bottom; // window rows
		void *window;   // device memory on `device`, (bottom - top) * stride bytes
	};
	std::vector<Strip> strips;
};

namespace {

// Bind the calling thread to a device for a scope, then put its previous binding back.
struct ScopedDevice {
	int saved;
	bool ok;
	explicit ScopedDevice(int device)
		: saved(current_device()), ok(vips_hip_init(device) == 0)
	{
	}
	~ScopedDevice()
	{
		if (saved >= 0 && saved != current_device())
			(void) vips_hip_init(saved);
	}
};

void strip_bounds(int total, int n, int k, int *row0, int *row1)
{
	const int base = total / n, extra = total % n;
	*row0 = k * base + (k < extra ? k : extra);
	*row1 = *row0 + base + (k < extra ? 1 : 0);
}

} // namespace

extern "C" {

VipsHipStrips *vips_hip_strips_new(int im_width, int im_height, int bands, int format, int n, const int *devices,
	int halo)
{
	if (ensure_init())
		return nullptr;
	const int es = format_sizeof(format);
	if (im_width <= 0 || im_height <= 0 || bands <= 0 || es == 0 || n <= 0 || n > im_height || halo < 0 || !devices) {
		error("vips_hip_strips_new", "bad parameters");
		return nullptr;
	}
	VipsHipStrips *s = new VipsHipStrips;
	s->im_width = im_width;
	s->im_height = im_height;
	s->bands = bands;
	s->format = format;
	s->halo = halo;
	s->stride = (size_t) im_width * bands * es;
	for (int k = 0; k < n; k++) {
		VipsHipStrips::Strip st;
		st.device = devices[k];
		strip_bounds(im_height, n, k, &st.row0, &st.row1);
		st.top = st.row0 - halo > 0 ? st.row0 - halo : 0;
		st.bottom = st.row1 + halo < im_height ? st.row1 + halo : im_height;
		st.window = nullptr;
		{
			ScopedDevice on(st.device);
			if (on.ok)
				st.window = vips_hip_malloc((size_t) (st.bottom - st.top) * s->stride);
		}
		s->strips.push_back(st);
		if (!st.window) {
			vips_hip_strips_free(s);
			return nullptr;
		}
	}
	// peer access between the devices that exchange rows (already enabled / same device: fine)
	for (int k = 0; k + 1 < n; k++) {
		const int a = s->strips[k].device, b = s->strips[k + 1].device;
		if (a == b)
			continue;
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
			{
				ScopedDevice on(a);
				(void) hipDeviceEnablePeerAccess(b, 0);
			}
			{
				ScopedDevice on(b);
				(void) hipDeviceEnablePeerAccess(a, 0);
			}
			(void) hipGetLastError(); // hipErrorPeerAccessAlreadyEnabled is not an error
		}
	}
	return s;
}

void vips_hip_strips_free(VipsHipStrips *strips)
{
	if (!strips)
		return;
	for (auto &st : strips->strips)
		vips_hip_free(st.window);
	delete strips;
}

int vips_hip_strips_count(const VipsHipStrips *strips)
{
	return strips ? (int) strips->strips.size() : 0;
}

int vips_hip_strips_region(const VipsHipStrips *strips, int k, int *device, VipsHipRegion *own, VipsHipRegion *window)
{
	if (!strips || k < 0 || k >= (int) strips->strips.size()) {
		error("vips_hip_strips_region", "no such strip");
		return -1;
	}
	const auto &st = strips->strips[k];
	if (device)
		*device = st.device;
	for (int which = 0; which < 2; which++) {
		VipsHipRegion *r = which ? window : own;
		if (!r)
			continue;
		const int top = which ? st.top : st.row0, bottom = which ? st.bottom : st.row1;
		r->data = (char *) st.window + (size_t) (top - st.top) * strips->stride;
		r->left = 0;
		r->top = top;
		r->width = strips->im_width;
		r->height = bottom - top;
		r->im_width = strips->im_width;
		r->im_height = strips->im_height;
		r->bands = strips->bands;
		r->format = strips->format;
		r->stride = strips->stride;
	}
	return 0;
}

// Every halo row from its owner's window into the windows that need it.  The strips' own rows
// must be complete (their producers synchronised) on entry; the halos are complete on return.
int vips_hip_strips_exchange(VipsHipStrips *strips)
{
	if (!strips) {
		error("vips_hip_strips_exchange", "null strips");
		return -1;
	}
	const int n = (int) strips->strips.size();
	std::vector<int> touched;
	for (int dst = 0; dst < n; dst++) {
		const auto &d = strips->strips[dst];
		for (int src = 0; src < n; src++) {
			if (src == dst)
				continue;
			const auto &o = strips->strips[src];
			const int lo = d.top > o.row0 ? d.top : o.row0;
			const int hi = d.bottom < o.row1 ? d.bottom : o.row1;
			if (hi <= lo)
				continue;
			// queued on the RECEIVING device's stream of this thread
			ScopedDevice on(d.device);
			if (!on.ok)
				return -1;
			const size_t bytes = (size_t) (hi - lo) * strips->stride;
			void *to = (char *) d.window + (size_t) (lo - d.top) * strips->stride;
			const void *from = (const char *) o.window + (size_t) (lo - o.top) * strips->stride;
			VH_CHECK(hipMemcpyPeerAsync(to, d.device, from, o.device, bytes, stream()));
			touched.push_back(d.device);
		}
	}
	for (int device : touched) {
		ScopedDevice on(device);
		if (!on.ok)
			return -1;
		VH_CHECK(hipStreamSynchronize(stream()));
	}
	return 0;
}

// vips_conv() on an image held as strips: the halo exchange, then every strip's window through
// vips_hip_conv_gen on its own device, all strips at once (one host thread each).  out[k] is
// strip k of the result, an image on devices[k] of the conv's output format.
int vips_hip_conv_strips(VipsHipStrips *strips, VipsHipImage **out, const double *mask, int mask_width, int mask_height,
	double scale, double offset, int precision)
{
	if (!strips || !out || !mask) {
		error("vips_hip_conv_strips", "null argument");
		return -1;
	}
	const int n = (int) strips->strips.size();
	for (int k = 0; k < n; k++)
		out[k] = nullptr;
	if (strips->halo < mask_height / 2 || strips->halo < mask_height - 1 - mask_height / 2) {
		error("vips_hip_conv_strips", "the strips carry %d halo rows, a mask of %d rows needs %d", strips->halo,
			mask_height, mask_height / 2);
		return -1;
	}
	if (vips_hip_strips_exchange(strips))
		return -1;
	std::vector<std::string> errors(n);
	std::vector<std::thread> workers;
	for (int k = 0; k < n; k++)
		workers.emplace_back([&, k]() {
			const auto &st = strips->strips[k];
			int r = vips_hip_init(st.device);
			VipsHipConv *conv = r ? nullptr : vips_hip_conv_new(mask, mask_width, mask_height, scale, offset, precision);
			VipsHipImage *o = nullptr;
			if (!r && conv) {
				VipsHipRegion window, ro;
				(void) vips_hip_strips_region(strips, k, nullptr, nullptr, &window);
				o = vips_hip_image_new(strips->im_width, st.row1 - st.row0, strips->bands,
					vips_hip_conv_out_format(conv, strips->format), 0);
				if (o) {
					vips_hip_image_region(o, &ro);
					ro.top = st.row0;
					ro.im_height = strips->im_height;
					r = vips_hip_conv_gen(conv, &window, &ro);
					if (!r)
						r = vips_hip_synchronize();
				}
				else
					r = -1;
			}
			else
				r = -1;
			vips_hip_conv_free(conv);
			if (r) {
				errors[k] = vips_hip_error_buffer();
				vips_hip_image_unref(o);
				o = nullptr;
			}
			out[k] = o;
			release_thread_stream();
		});
	for (std::thread &t : workers)
		t.join();
	int bad = 0;
	for (int k = 0; k < n; k++)
		if (!out[k]) {
			if (!bad)
				error("vips_hip_conv_strips", "strip %d: %s", k, errors[k].c_str());
			bad++;
		}
	if (bad) {
		for (int k = 0; k < n; k++) {
			vips_hip_image_unref(out[k]);
			out[k] = nullptr;
		}
		return -1;
	}
	return 0;
}

} // extern "C"
