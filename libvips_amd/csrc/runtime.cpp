// Runtime layer of libvipship.so: device selection, per-thread streams, the
// caching HBM pool, pinned staging, the error buffer and kernel gates.
//
// Mirrors the host plumbing a libvips generate leans on, nothing more:
//   error buffer        iofuncs/error.c (vips_error / vips_error_buffer)
//   tracked malloc      iofuncs/memory.c:304,369 (vips_tracked_malloc)
//   per-thread sequence include/vips/image.h:151-154 (start/stop own a stream)
//   gates               include/vips/gate.h:40-56
#include "internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace vh {

static thread_local std::string tls_error;
static thread_local int tls_device = -1;
// the calling thread's own stream per device it has been bound to, and the external stream
// (vips_hip_set_stream) of its current binding
constexpr int MAX_DEVICES = 64;
static thread_local hipStream_t tls_own_stream[MAX_DEVICES];
// (an external stream belongs to the device the thread was bound to when it set it: one slot per
// device, so that a thread that visits another device -- ScopedDevice in strips.cpp, bind_to() --
// finds its stream again when it comes back instead of silently continuing on the library's own)
static thread_local hipStream_t tls_external_dev[MAX_DEVICES];
static thread_local bool tls_stream_external_dev[MAX_DEVICES];
#define tls_external tls_external_dev[tls_device < 0 ? 0 : tls_device]
#define tls_stream_external tls_stream_external_dev[tls_device < 0 ? 0 : tls_device]

static std::mutex g_mutex;
static bool g_checked[MAX_DEVICES]; // gfx950 confirmed
// A THREAD drives one device at a time (hipSetDevice is per thread): vips_hip_init(d) binds the
// calling thread; a thread that never calls it is bound on first use -- to the next entry of
// $VIPS_HIP_DEVICES (a comma list, dealt round-robin over such threads: how one libvips process
// spreads its worker pool over the GPUs of a node, iofuncs/threadpool.c:625), else to
// $VIPS_HIP_DEVICE, else to the device of the process's first vips_hip_init(), else to 0.
// Pools, plan caches and library-made images are per device; an image operation runs on the
// device its input lives on (bind_to).
static std::atomic<int> g_device{ -1 };
static std::atomic<unsigned int> g_next_slot{ 0 };

// $VIPS_HIP_DEVICES, validated; empty when unset or malformed (an error is left for the caller)
static int devices_from_env(std::vector<int> &out)
{
	out.clear();
	const char *env = getenv("VIPS_HIP_DEVICES");
	if (!env || !*env)
		return 0;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return 0;
	const char *p = env;
	while (*p) {
		char *end = nullptr;
		const long d = strtol(p, &end, 10);
		if (end == p || d < 0 || d >= n || d >= MAX_DEVICES || (*end && *end != ',')) {
			error("vips_hip_init", "VIPS_HIP_DEVICES=\"%s\": not a comma list of devices 0..%d", env, n - 1);
			out.clear();
			return -1;
		}
		out.push_back((int) d);
		p = *end ? end + 1 : end;
	}
	return 0;
}

void error(const char *domain, const char *fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	tls_error += domain;
	tls_error += ": ";
	tls_error += buf;
	tls_error += "\n";
}

int hip_failed(hipError_t err, const char *what)
{
	error("vips_hip", "%s failed: %s", what, hipGetErrorString(err));
	return -1;
}

int ensure_init()
{
	if (tls_device >= 0)
		return 0;
	std::vector<int> list;
	if (devices_from_env(list))
		return -1;
	int device;
	if (!list.empty())
		device = list[g_next_slot.fetch_add(1) % list.size()];
	else {
		device = g_device.load();
		if (device < 0) {
			const char *env = getenv("VIPS_HIP_DEVICE");
			device = 0;
			if (env && *env) {
				char *end = nullptr;
				const long d = strtol(env, &end, 10);
				if (end == env || *end || d < 0 || d >= MAX_DEVICES) {
					error("vips_hip_init", "VIPS_HIP_DEVICE=\"%s\" is not a device number", env);
					return -1;
				}
				device = (int) d;
			}
		}
	}
	return vips_hip_init(device);
}

int current_device()
{
	return tls_device;
}

// Run where the data lives: bind the calling thread to the device `image` is on.
int bind_to(const _VipsHipImage *image)
{
	if (ensure_init())
		return -1;
	if (!image || image->device < 0 || image->device == tls_device)
		return 0;
	return vips_hip_init(image->device);
}

int plan_device(const char *domain, std::atomic<int> *device)
{
	if (ensure_init())
		return -1;
	int none = -1;
	if (device->compare_exchange_strong(none, tls_device) || none == tls_device)
		return 0;
	error(domain, "this plan's tables are on device %d, the calling thread drives device %d "
		"(a plan handle belongs to one device: make one per device)", none, tls_device);
	return -1;
}

hipStream_t stream()
{
	if (tls_stream_external)
		return tls_external;
	if (tls_device < 0)
		return nullptr;
	hipStream_t &own = tls_own_stream[tls_device];
	if (!own) {
		if (hipStreamCreateWithFlags(&own, hipStreamNonBlocking) != hipSuccess)
			own = nullptr;
	}
	return own;
}

// For threads the library itself starts (vips_hip_thumbnail_batch): finish and destroy the
// calling thread's own stream before the thread ends -- thread-local streams are otherwise kept
// for the life of the thread and never destroyed (no HIP calls from thread-exit destructors).
void release_thread_stream()
{
	const int bound = tls_device;
	for (int d = 0; d < MAX_DEVICES; d++)
		if (tls_own_stream[d]) {
			if (d != bound)
				(void) hipSetDevice(d);
			(void) hipStreamSynchronize(tls_own_stream[d]);
			(void) hipStreamDestroy(tls_own_stream[d]);
			tls_own_stream[d] = nullptr;
		}
	if (bound >= 0)
		(void) hipSetDevice(bound);
	for (int d = 0; d < MAX_DEVICES; d++) {
		tls_external_dev[d] = nullptr;
		tls_stream_external_dev[d] = false;
	}
}

ScopedStream::ScopedStream(hipStream_t s)
	: saved(tls_external), slot(tls_device < 0 ? 0 : tls_device), saved_external(tls_stream_external),
	  active(s != nullptr)
{
	if (active) {
		tls_external = s;
		tls_stream_external = true;
	}
}

// (restores into the slot it saved from: the thread may have been rebound to another device inside the scope)
ScopedStream::~ScopedStream()
{
	if (active) {
		tls_external_dev[slot] = saved;
		tls_stream_external_dev[slot] = saved_external;
	}
}

_VipsHipImage *image_share(const _VipsHipImage *in)
{
	if (!in || !in->owns || !in->hold)
		return nullptr;
	return new _VipsHipImage(*in);
}

int check_region(const char *domain, const VipsHipRegion *r)
{
	if (!r || !r->data) {
		error(domain, "null region");
		return -1;
	}
	if (r->width <= 0 || r->height <= 0 || r->bands <= 0) {
		error(domain, "empty region");
		return -1;
	}
	if (format_sizeof(r->format) == 0) {
		error(domain, "bad band format %d", r->format);
		return -1;
	}
	return 0;
}

// ---------------------------------------------------------------- the pool
//
// Size-bucketed free lists (power-of-two classes from 256 B, eight classes per octave above 64 MiB,
// 2 MB granules above 1 GiB).
// A freed block goes to the FREEING THREAD's own list and is handed out again only to that
// thread: every thread queues its work on one stream, so a block whose last kernel is still
// in flight can only be reused behind that kernel on the same stream (stream-ordered reuse
// without events).  Blocks cross threads only through the global list, which receives the
// lists of threads that exit and everything at vips_hip_pool_trim(); results cross threads
// only after a synchronize (image download, the module's build()).
struct Pool {
	int device = 0;
	std::mutex mutex;
	std::map<size_t, std::vector<void *>> free_lists; // global: orphaned blocks
	std::unordered_map<void *, size_t> live;
	size_t cached_bytes = 0;
	size_t live_bytes = 0;

	struct Local {
		std::map<size_t, std::vector<void *>> lists;
		Pool *pool = nullptr;
		~Local()
		{
			// thread exit: pointer moves only (no HIP calls: this also runs at process exit)
			if (!pool)
				return;
			std::lock_guard<std::mutex> lock(pool->mutex);
			for (auto &kv : lists)
				for (void *p : kv.second)
					pool->free_lists[kv.first].push_back(p);
		}
	};
	Local &local()
	{
		static thread_local Local l[MAX_DEVICES]; // the calling thread's lists, one per device
		return l[device];
	}

	static size_t bucket(size_t size)
	{
		if (size <= 256)
			return 256;
		if (size > ((size_t) 1 << 30))
			return (size + ((size_t) 1 << 21) - 1) & ~(((size_t) 1 << 21) - 1);
		size_t b = 256;
		while (b < size)
			b <<= 1;
		// above 64 MiB eight classes per octave: a 600 MB strip window costs 640 MB, not 1 GiB
		if (b > ((size_t) 64 << 20)) {
			const size_t step = b >> 4; // an eighth of the octave below b
			return (size + step - 1) / step * step;
		}
		return b;
	}

	void *alloc(size_t size)
	{
		size_t b = bucket(size);
		void *p = nullptr;
		{
			Local &l = local();
			auto it = l.lists.find(b);
			if (it != l.lists.end() && !it->second.empty()) {
				p = it->second.back();
				it->second.pop_back();
			}
		}
		if (!p) {
			std::lock_guard<std::mutex> lock(mutex);
			auto it = free_lists.find(b);
			if (it != free_lists.end() && !it->second.empty()) {
				p = it->second.back();
				it->second.pop_back();
			}
		}
		if (p) {
			std::lock_guard<std::mutex> lock(mutex);
			cached_bytes -= b;
			live[p] = b;
			live_bytes += b;
			return p;
		}
		hipError_t err = hipMalloc(&p, b);
		if (err != hipSuccess) {
			// Out of HBM: drop the cache and retry once.
			trim();
			err = hipMalloc(&p, b);
		}
		if (err != hipSuccess) {
			hip_failed(err, "hipMalloc");
			return nullptr;
		}
		std::lock_guard<std::mutex> lock(mutex);
		live[p] = b;
		live_bytes += b;
		return p;
	}

	void release(void *p)
	{
		if (!p)
			return;
		size_t b;
		{
			std::lock_guard<std::mutex> lock(mutex);
			auto it = live.find(p);
			if (it == live.end())
				return; // not ours
			b = it->second;
			live.erase(it);
			live_bytes -= b;
			cached_bytes += b;
		}
		// (also when the thread is bound to another device right now, IF it has a stream of its own on this
		// device: it may have work queued there from before it re-bound -- that stream orders the reuse, as for
		// any block it frees.  A thread that never ran anything on this device -- a libvips worker disposing of an
		// operation whose result lives elsewhere -- would strand the block in a list nobody allocates from: it
		// goes to the global list, which this device's own threads and trim() reach; nothing of the freeing thread
		// can be pending on it)
		if (device != tls_device && !tls_own_stream[device] && !tls_stream_external_dev[device]) {
			std::lock_guard<std::mutex> lock(mutex);
			free_lists[b].push_back(p);
			return;
		}
		Local &l = local();
		l.pool = this;
		l.lists[b].push_back(p);
	}

	bool owns(void *p)
	{
		std::lock_guard<std::mutex> lock(mutex);
		return live.find(p) != live.end();
	}

	// frees the calling thread's cache and the global list (other threads keep theirs)
	void trim()
	{
		Local &l = local();
		std::lock_guard<std::mutex> lock(mutex);
		for (auto &kv : l.lists)
			for (void *p : kv.second) {
				(void) hipFree(p);
				cached_bytes -= kv.first;
			}
		l.lists.clear();
		for (auto &kv : free_lists)
			for (void *p : kv.second) {
				(void) hipFree(p);
				cached_bytes -= kv.first;
			}
		free_lists.clear();
	}
};

// One pool per device.  Leaked on purpose: other translation units release blocks from their own
// static destructors, and static destruction order across units is unspecified.
static Pool *g_pools[MAX_DEVICES];
static std::mutex &g_pools_mutex = *new std::mutex;

static Pool &pool_of(int device)
{
	if (device < 0 || device >= MAX_DEVICES)
		device = 0;
	std::lock_guard<std::mutex> lock(g_pools_mutex);
	if (!g_pools[device]) {
		g_pools[device] = new Pool;
		g_pools[device]->device = device;
	}
	return *g_pools[device];
}

// the pool of the calling thread's device
static Pool &pool()
{
	return pool_of(current_device());
}

// a block goes back to the pool it came from, whichever device the freeing thread drives
static void pool_release(void *p)
{
	if (!p)
		return;
	Pool &mine = pool();
	if (mine.owns(p)) {
		mine.release(p);
		return;
	}
	for (int d = 0; d < MAX_DEVICES; d++) {
		Pool *other;
		{
			std::lock_guard<std::mutex> lock(g_pools_mutex);
			other = g_pools[d];
		}
		if (other && other != &mine && other->owns(p)) {
			other->release(p);
			return;
		}
	}
}

void *upload(const void *host, size_t size)
{
	void *d = pool().alloc(size);
	if (!d)
		return nullptr;
	// Stream-ordered: the block may have been released to the pool by a plan
	// whose last kernel is still in flight on this stream, so the copy must
	// queue behind it.  The wait keeps the (pageable) host source alive; uploads
	// only happen when an operation is first built, never in steady state.
	hipStream_t s = stream();
	if (hipMemcpyAsync(d, host, size, hipMemcpyHostToDevice, s) != hipSuccess ||
		hipStreamSynchronize(s) != hipSuccess) {
		error("vips_hip", "table upload failed");
		pool_release(d);
		return nullptr;
	}
	return d;
}

// ------------------------------------------------------------------- gates
struct GateRecord {
	std::string name;
	hipEvent_t start, stop;
};
static bool g_gate_enabled = false;
static std::mutex &g_gate_mutex = *new std::mutex;
static std::vector<GateRecord> &g_gate_records = *new std::vector<GateRecord>;

Gate::Gate(const char *name_)
	: name(name_), start(nullptr), active(false)
{
	if (!g_gate_enabled)
		return;
	if (hipEventCreate(&start) != hipSuccess)
		return;
	(void) hipEventRecord(start, stream());
	active = true;
}

Gate::~Gate()
{
	if (!active)
		return;
	hipEvent_t stop;
	if (hipEventCreate(&stop) != hipSuccess)
		return;
	(void) hipEventRecord(stop, stream());
	std::lock_guard<std::mutex> lock(g_gate_mutex);
	g_gate_records.push_back({ name, start, stop });
}

} // namespace vh

using namespace vh;

extern "C" {

int vips_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

int vips_hip_init(int device)
{
	int n = vips_hip_device_count();
	if (n <= 0) {
		error("vips_hip_init", "no HIP device visible (this library has no CPU path)");
		return -1;
	}
	if (device < 0 || device >= n || device >= MAX_DEVICES) {
		error("vips_hip_init", "device %d out of range (have %d)", device, n);
		return -1;
	}
	VH_CHECK(hipSetDevice(device));
	{
		std::lock_guard<std::mutex> lock(g_mutex);
		if (!g_checked[device]) {
			hipDeviceProp_t prop;
			VH_CHECK(hipGetDeviceProperties(&prop, device));
			if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
				error("vips_hip_init", "device %d is %s, this library is built for gfx950 only", device,
					prop.gcnArchName);
				return -1;
			}
			g_checked[device] = true;
		}
		int none = -1;
		g_device.compare_exchange_strong(none, device); // the default of threads that never ask
	}
	tls_device = device; // (the thread's external stream of THIS device, if it set one, is current again)
	return 0;
}

int vips_hip_current_device(void)
{
	return tls_device;
}

int vips_hip_devices(int *devices, int max)
{
	std::vector<int> list;
	if (devices_from_env(list))
		return -1;
	if (list.empty()) {
		if (ensure_init())
			return -1;
		list.push_back(tls_device);
	}
	for (int i = 0; i < (int) list.size() && i < max; i++)
		if (devices)
			devices[i] = list[i];
	return (int) list.size();
}

void vips_hip_shutdown(void)
{
	release_thread_stream();
	vips_hip_pool_trim();
	tls_device = -1; // the thread may bind again, to any device
}

const char *vips_hip_error_buffer(void)
{
	return tls_error.c_str();
}

void vips_hip_error_clear(void)
{
	tls_error.clear();
}

int vips_hip_set_stream(void *s)
{
	if (ensure_init())
		return -1;
	hipStream_t &own = tls_own_stream[tls_device];
	if (own) {
		(void) hipStreamSynchronize(own);
		(void) hipStreamDestroy(own);
		own = nullptr;
	}
	tls_external = (hipStream_t) s;
	tls_stream_external = s != nullptr;
	return 0;
}

void *vips_hip_get_stream(void)
{
	if (ensure_init())
		return nullptr;
	return (void *) stream();
}

// vips_vector_set_enabled / vips_vector_isenabled (iofuncs/vector.cpp:98-113).  The reference's
// vector (Highway) paths are bit-identical to its C paths except convi on uchar, so this switch
// only selects that arithmetic (conv.hip MODE 4).  Off by default: the oracle build has no Highway.
static std::atomic<int> g_vector_enabled(0);

void vips_hip_vector_set_enabled(int enabled)
{
	g_vector_enabled.store(enabled ? 1 : 0);
}

int vips_hip_vector_isenabled(void)
{
	return g_vector_enabled.load();
}

// Float arithmetic mode.  -1 = not decided yet (first use reads $VIPS_HIP_EXACT_FLOAT).
static std::atomic<int> g_exact_float(-1);

void vips_hip_set_exact_float(int enabled)
{
	g_exact_float.store(enabled ? 1 : 0);
}

int vips_hip_get_exact_float(void)
{
	int v = g_exact_float.load();
	if (v < 0) {
		const char *env = getenv("VIPS_HIP_EXACT_FLOAT");
		v = env && atoi(env) != 0 ? 1 : 0;
		g_exact_float.store(v);
	}
	return v;
}

int vips_hip_synchronize(void)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipStreamSynchronize(stream()));
	return 0;
}

void *vips_hip_malloc(size_t size)
{
	if (ensure_init())
		return nullptr;
	return pool().alloc(size ? size : 1);
}

void vips_hip_free(void *ptr)
{
	pool_release(ptr);
}

// Pinned host memory is expensive to make (the pages are locked and mapped: ~0.1 s per GiB), and
// the strip loop of the libvips module wants the same few staging buffers for every evaluation:
// freed blocks are kept -- a handful, by exact size -- and handed out again.  The cache is what a
// long-running libvips process keeps page-locked between evaluations, so it is small: the two
// staging buffers and the ring slots of one strip loop under the module's default budgets
// ($VIPS_HIP_POOL_PINNED bytes overrides it; vips_hip_pool_trim() empties it).
static std::mutex &g_pinned_mutex = *new std::mutex;
static std::unordered_map<void *, size_t> &g_pinned_live = *new std::unordered_map<void *, size_t>;
static std::vector<std::pair<size_t, void *>> &g_pinned_free = *new std::vector<std::pair<size_t, void *>>;
constexpr size_t PINNED_CACHE_BLOCKS = 8;
static size_t pinned_cache_bytes()
{
	static const size_t v = [] {
		const char *e = getenv("VIPS_HIP_POOL_PINNED");
		const long long n = e ? atoll(e) : 0;
		return n > 0 ? (size_t) n : (size_t) 4 << 30;
	}();
	return v;
}

void *vips_hip_malloc_host(size_t size)
{
	if (ensure_init())
		return nullptr;
	size = size ? size : 1;
	{
		std::lock_guard<std::mutex> lock(g_pinned_mutex);
		for (size_t i = 0; i < g_pinned_free.size(); i++)
			if (g_pinned_free[i].first == size) {
				void *p = g_pinned_free[i].second;
				g_pinned_free.erase(g_pinned_free.begin() + i);
				g_pinned_live[p] = size;
				return p;
			}
	}
	void *p = nullptr;
	VH_CHECK_NULL(hipHostMalloc(&p, size, hipHostMallocDefault));
	std::lock_guard<std::mutex> lock(g_pinned_mutex);
	g_pinned_live[p] = size;
	return p;
}

void vips_hip_free_host(void *ptr)
{
	if (!ptr)
		return;
	std::vector<void *> drop;
	{
		std::lock_guard<std::mutex> lock(g_pinned_mutex);
		auto it = g_pinned_live.find(ptr);
		if (it == g_pinned_live.end())
			drop.push_back(ptr); // not ours to cache
		else {
			g_pinned_free.emplace_back(it->second, ptr);
			g_pinned_live.erase(it);
			size_t total = 0;
			for (auto &b : g_pinned_free)
				total += b.first;
			while (g_pinned_free.size() > PINNED_CACHE_BLOCKS || total > pinned_cache_bytes()) {
				total -= g_pinned_free.front().first;
				drop.push_back(g_pinned_free.front().second);
				g_pinned_free.erase(g_pinned_free.begin());
			}
		}
	}
	for (void *p : drop)
		(void) hipHostFree(p);
}

static void pinned_trim()
{
	std::vector<std::pair<size_t, void *>> drop;
	{
		std::lock_guard<std::mutex> lock(g_pinned_mutex);
		drop.swap(g_pinned_free);
	}
	for (auto &b : drop)
		(void) hipHostFree(b.second);
}

int vips_hip_memcpy_h2d(void *dst, const void *src, size_t size)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, stream()));
	VH_CHECK(hipStreamSynchronize(stream()));
	return 0;
}

int vips_hip_memcpy_d2h(void *dst, const void *src, size_t size)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, stream()));
	VH_CHECK(hipStreamSynchronize(stream()));
	return 0;
}

int vips_hip_memcpy_h2d_async(void *dst, const void *src, size_t size)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, stream()));
	return 0;
}

int vips_hip_memcpy_d2h_async(void *dst, const void *src, size_t size)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, stream()));
	return 0;
}

void *vips_hip_stream_new(void)
{
	if (ensure_init())
		return nullptr;
	hipStream_t s = nullptr;
	VH_CHECK_NULL(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	return (void *) s;
}

void vips_hip_stream_free(void *s)
{
	if (!s)
		return;
	// the stream may be remembered in the slot of ANY device this thread has been bound to (bind_to() moves a
	// thread to an image's device and does not move it back): forget it everywhere before it is destroyed
	for (int d = 0; d < MAX_DEVICES; d++)
		if (tls_external_dev[d] == (hipStream_t) s) {
			tls_external_dev[d] = nullptr;
			tls_stream_external_dev[d] = false;
		}
	(void) hipStreamSynchronize((hipStream_t) s);
	(void) hipStreamDestroy((hipStream_t) s);
}

int vips_hip_memcpy_d2d(void *dst, const void *src, size_t size)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToDevice, stream()));
	return 0;
}

int vips_hip_memcpy2d_h2d(void *dst, size_t dpitch, const void *src, size_t spitch,
	size_t width_bytes, size_t height)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height,
		hipMemcpyHostToDevice, stream()));
	VH_CHECK(hipStreamSynchronize(stream()));
	return 0;
}

int vips_hip_memcpy2d_d2h(void *dst, size_t dpitch, const void *src, size_t spitch,
	size_t width_bytes, size_t height)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height,
		hipMemcpyDeviceToHost, stream()));
	VH_CHECK(hipStreamSynchronize(stream()));
	return 0;
}

size_t vips_hip_pool_bytes(void)
{
	size_t total = 0;
	for (int d = 0; d < MAX_DEVICES; d++) {
		Pool *p;
		{
			std::lock_guard<std::mutex> lock(g_pools_mutex);
			p = g_pools[d];
		}
		if (p) {
			std::lock_guard<std::mutex> lock(p->mutex);
			total += p->cached_bytes + p->live_bytes;
		}
	}
	return total;
}

// the calling thread's caches and the orphaned blocks of every device, and the pinned cache
void vips_hip_pool_trim(void)
{
	pinned_trim();
	for (int d = 0; d < MAX_DEVICES; d++) {
		Pool *p;
		{
			std::lock_guard<std::mutex> lock(g_pools_mutex);
			p = g_pools[d];
		}
		if (p)
			p->trim();
	}
}

void *vips_hip_event_new(void)
{
	if (ensure_init())
		return nullptr;
	hipEvent_t e;
	VH_CHECK_NULL(hipEventCreate(&e));
	return (void *) e;
}

void vips_hip_event_free(void *event)
{
	if (event)
		(void) hipEventDestroy((hipEvent_t) event);
}

int vips_hip_event_record(void *event)
{
	if (ensure_init())
		return -1;
	VH_CHECK(hipEventRecord((hipEvent_t) event, stream()));
	return 0;
}

int vips_hip_event_synchronize(void *event)
{
	if (!event) {
		error("vips_hip_event_synchronize", "null event");
		return -1;
	}
	VH_CHECK(hipEventSynchronize((hipEvent_t) event));
	return 0;
}

int vips_hip_stream_wait_event(void *event)
{
	if (ensure_init())
		return -1;
	if (!event) {
		error("vips_hip_stream_wait_event", "null event");
		return -1;
	}
	VH_CHECK(hipStreamWaitEvent(stream(), (hipEvent_t) event, 0));
	return 0;
}

double vips_hip_event_elapsed_ms(void *start, void *stop)
{
	float ms = 0.f;
	if (hipEventSynchronize((hipEvent_t) stop) != hipSuccess)
		return -1.0;
	if (hipEventElapsedTime(&ms, (hipEvent_t) start, (hipEvent_t) stop) != hipSuccess)
		return -1.0;
	return (double) ms;
}

void vips_hip_gate_enable(int enable)
{
	g_gate_enabled = enable != 0;
}

void vips_hip_gate_reset(void)
{
	std::lock_guard<std::mutex> lock(g_gate_mutex);
	for (auto &r : g_gate_records) {
		(void) hipEventDestroy(r.start);
		(void) hipEventDestroy(r.stop);
	}
	g_gate_records.clear();
}

int vips_hip_gate_query(const char *name, double *total_ms)
{
	if (!name)
		name = "";
	(void) hipDeviceSynchronize();
	std::lock_guard<std::mutex> lock(g_gate_mutex);
	int n = 0;
	double total = 0.0;
	size_t len = strlen(name);
	for (auto &r : g_gate_records) {
		if (strncmp(r.name.c_str(), name, len) != 0)
			continue;
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
			total += ms;
			n += 1;
		}
	}
	if (total_ms)
		*total_ms = total;
	return n;
}

int vips_hip_gate_report(char *buf, int size)
{
	(void) hipDeviceSynchronize();
	std::lock_guard<std::mutex> lock(g_gate_mutex);
	std::map<std::string, std::pair<int, double>> totals;
	for (auto &r : g_gate_records) {
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess)
			continue;
		auto &t = totals[r.name];
		t.first += 1;
		t.second += ms;
	}
	std::string text;
	for (auto &kv : totals) {
		char line[256];
		snprintf(line, sizeof(line), "%s %d %.6f\n", kv.first.c_str(), kv.second.first,
			kv.second.second);
		text += line;
	}
	if (buf && size > 0) {
		strncpy(buf, text.c_str(), size - 1);
		buf[size - 1] = '\0';
	}
	return (int) totals.size();
}

// ------------------------------------------------------------------ images

// width * height * bands * element size must fit comfortably (1 TiB cap): every kernel does
// its address arithmetic in size_t / long long from these four numbers
static bool image_bytes_overflow(int width, int height, int bands, int es)
{
	long long size = 0;
	return __builtin_mul_overflow((long long) width, (long long) height, &size) ||
		__builtin_mul_overflow(size, (long long) bands, &size) ||
		__builtin_mul_overflow(size, (long long) es, &size) || size > (1LL << 40);
}

VipsHipImage *vips_hip_image_new(int width, int height, int bands, int format,
	int interpretation)
{
	if (ensure_init())
		return nullptr;
	int es = format_sizeof(format);
	if (width <= 0 || height <= 0 || bands <= 0 || es == 0) {
		error("vips_hip_image_new", "bad image parameters %dx%dx%d format %d",
			width, height, bands, format);
		return nullptr;
	}
	if (image_bytes_overflow(width, height, bands, es)) {
		error("vips_hip_image_new", "image %dx%dx%d is too large", width, height, bands);
		return nullptr;
	}
	VipsHipImage *im = new VipsHipImage;
	im->width = width;
	im->height = height;
	im->bands = bands;
	im->format = format;
	im->interpretation = interpretation;
	im->stride = (size_t) width * bands * es;
	im->owns = true;
	im->device = current_device();
	im->data = pool().alloc(im->stride * height);
	if (!im->data) {
		delete im;
		return nullptr;
	}
	im->hold = std::shared_ptr<void>(im->data, [](void *p) { pool_release(p); });
	return im;
}

VipsHipImage *vips_hip_image_new_from_memory(const void *host_data, int width, int height,
	int bands, int format, int interpretation)
{
	VipsHipImage *im = vips_hip_image_new(width, height, bands, format, interpretation);
	if (!im)
		return nullptr;
	if (vips_hip_memcpy_h2d(im->data, host_data, im->stride * height)) {
		vips_hip_image_unref(im);
		return nullptr;
	}
	return im;
}

VipsHipImage *vips_hip_image_new_from_device(void *device_data, int width, int height,
	int bands, int format, int interpretation)
{
	if (ensure_init())
		return nullptr;
	int es = format_sizeof(format);
	if (!device_data || width <= 0 || height <= 0 || bands <= 0 || es == 0) {
		error("vips_hip_image_new_from_device", "bad image parameters");
		return nullptr;
	}
	if (image_bytes_overflow(width, height, bands, es)) {
		error("vips_hip_image_new_from_device", "image %dx%dx%d is too large", width, height, bands);
		return nullptr;
	}
	VipsHipImage *im = new VipsHipImage;
	im->width = width;
	im->height = height;
	im->bands = bands;
	im->format = format;
	im->interpretation = interpretation;
	im->stride = (size_t) width * bands * es;
	im->owns = false;
	im->data = device_data;
	// the caller's memory: taken to be on the device the calling thread drives
	im->device = current_device();
	return im;
}

void vips_hip_image_unref(VipsHipImage *image)
{
	if (!image)
		return;
	delete image; // library memory goes back to the pool with its last holder
}

void vips_hip_image_unref_many(VipsHipImage **images, int n)
{
	if (!images)
		return;
	for (int i = 0; i < n; i++) {
		vips_hip_image_unref(images[i]);
		images[i] = nullptr;
	}
}

int vips_hip_image_write_to_memory(const VipsHipImage *image, void *host_data)
{
	if (!image || !host_data) {
		error("vips_hip_image_write_to_memory", "null argument");
		return -1;
	}
	if (bind_to(image))
		return -1;
	return vips_hip_memcpy_d2h(host_data, image->data, image->stride * image->height);
}

// the getters of a NULL image answer 0 / NULL instead of crashing
void *vips_hip_image_get_data(const VipsHipImage *image) { return image ? image->data : nullptr; }
int vips_hip_image_get_device(const VipsHipImage *image) { return image ? image->device : -1; }
int vips_hip_image_get_width(const VipsHipImage *image) { return image ? image->width : 0; }
int vips_hip_image_get_height(const VipsHipImage *image) { return image ? image->height : 0; }
int vips_hip_image_get_bands(const VipsHipImage *image) { return image ? image->bands : 0; }
int vips_hip_image_get_format(const VipsHipImage *image) { return image ? image->format : -1; }
int vips_hip_image_get_interpretation(const VipsHipImage *image) { return image ? image->interpretation : -1; }
size_t vips_hip_image_get_stride(const VipsHipImage *image) { return image ? image->stride : 0; }

void vips_hip_image_region(const VipsHipImage *image, VipsHipRegion *region)
{
	if (!region)
		return;
	if (!image) {
		memset(region, 0, sizeof(*region));
		return;
	}
	region->data = image->data;
	region->left = 0;
	region->top = 0;
	region->width = image->width;
	region->height = image->height;
	region->im_width = image->width;
	region->im_height = image->height;
	region->bands = image->bands;
	region->format = image->format;
	region->stride = image->stride;
}

} // extern "C"
