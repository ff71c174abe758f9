// TEST INFRASTRUCTURE ONLY -- a stand-in for the HIP runtime in which memory is host memory,
// copies are memcpy, and KERNELS DO NOTHING.  LD_PRELOADed in front of libamdhip64.so by
// tests/test_host_glue_mock.py so that the HOST side of libvipship.so (plans, dispatch, sizes,
// file loaders and savers, thread pools, the libvips module's build / generate plumbing) can be
// driven end to end, and checked for shapes, copies and crashes, on a box without a GPU.  It says
// nothing about pixels a kernel would have made, it is never linked, and nothing outside tests/
// knows it exists.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime_api.h>

extern "C" {

// $MOCK_HIP_DEVICES fake devices (default 1); the current device is per thread, as in HIP
static int mock_devices()
{
	const char *e = getenv("MOCK_HIP_DEVICES");
	const int n = e ? atoi(e) : 1;
	return n < 1 ? 1 : n;
}
static thread_local int g_current_device = 0;
static long g_set_device_calls[16];

hipError_t hipGetDeviceCount(int *count)
{
	*count = mock_devices();
	return hipSuccess;
}

hipError_t hipSetDevice(int d)
{
	if (d < 0 || d >= mock_devices())
		return hipErrorInvalidDevice;
	g_current_device = d;
	if (d < 16)
		__sync_fetch_and_add(&g_set_device_calls[d], 1);
	return hipSuccess;
}

hipError_t hipGetDevice(int *d)
{
	*d = g_current_device;
	return hipSuccess;
}

// how often a thread bound itself to device d: lets the tests see work spread over devices
long mock_hip_set_device_calls(int d) { return d >= 0 && d < 16 ? g_set_device_calls[d] : 0; }

hipError_t hipDeviceCanAccessPeer(int *can, int, int)
{
	*can = 1;
	return hipSuccess;
}

hipError_t hipDeviceEnablePeerAccess(int, unsigned int) { return hipSuccess; }

static long g_peer_copies = 0, g_peer_bytes = 0;

hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, size_t size, hipStream_t)
{
	if (size)
		memmove(dst, src, size);
	__sync_fetch_and_add(&g_peer_copies, 1);
	__sync_fetch_and_add(&g_peer_bytes, (long) size);
	return hipSuccess;
}

long mock_hip_peer_copies(void) { return g_peer_copies; }
long mock_hip_peer_bytes(void) { return g_peer_bytes; }

hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *prop, int)
{
	memset(prop, 0, sizeof(*prop));
	strcpy(prop->name, "mock gfx950");
	strcpy(prop->gcnArchName, "gfx950:sramecc+:xnack-");
	prop->multiProcessorCount = 256;
	return hipSuccess;
}

const char *hipGetErrorString(hipError_t) { return "mock hip error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
static long g_syncs = 0; // host waits: device / stream / event synchronize
long mock_hip_syncs(void) { return g_syncs; }
hipError_t hipDeviceSynchronize(void)
{
	__sync_fetch_and_add(&g_syncs, 1);
	return hipSuccess;
}

static long g_mallocs = 0;
long mock_hip_mallocs(void) { return g_mallocs; }

hipError_t hipMalloc(void **p, size_t size)
{
	__sync_fetch_and_add(&g_mallocs, 1);
	*p = malloc(size ? size : 1);
	return *p ? hipSuccess : hipErrorOutOfMemory;
}

hipError_t hipFree(void *p)
{
	free(p);
	return hipSuccess;
}

hipError_t hipHostMalloc(void **p, size_t size, unsigned int)
{
	*p = malloc(size ? size : 1);
	return *p ? hipSuccess : hipErrorOutOfMemory;
}

hipError_t hipHostFree(void *p)
{
	free(p);
	return hipSuccess;
}

// $MOCK_HIP_FAIL_D2H_AFTER=N: the N + 1-th device-to-host copy from now on fails (the tests ask:
// does a failure inside a worker thread reach the caller?).  Read per call; counted per process.
static long g_d2h = 0;

hipError_t hipMemcpyAsync(void *dst, const void *src, size_t size, hipMemcpyKind kind, hipStream_t)
{
	if (kind == hipMemcpyDeviceToHost) {
		const char *e = getenv("MOCK_HIP_FAIL_D2H_AFTER");
		if (!e)
			g_d2h = 0;
		else if (__sync_fetch_and_add(&g_d2h, 1) >= atol(e))
			return hipErrorUnknown;
	}
	if (size)
		memmove(dst, src, size);
	return hipSuccess;
}

hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width,
	size_t height, hipMemcpyKind, hipStream_t)
{
	for (size_t y = 0; y < height; y++)
		memmove((char *) dst + y * dpitch, (const char *) src + y * spitch, width);
	return hipSuccess;
}

hipError_t hipMemsetAsync(void *dst, int value, size_t size, hipStream_t)
{
	memset(dst, value, size);
	return hipSuccess;
}

static int g_streams = 0;

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int)
{
	*s = (hipStream_t) malloc(8);
	__sync_fetch_and_add(&g_streams, 1);
	return hipSuccess;
}

hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *)
{
	*s = (hipStream_t) malloc(8);
	__sync_fetch_and_add(&g_streams, 1);
	return hipSuccess;
}

hipError_t hipStreamDestroy(hipStream_t s)
{
	free(s);
	__sync_fetch_and_sub(&g_streams, 1);
	return hipSuccess;
}

// how many streams are alive: lets the tests see a leak
int mock_hip_live_streams(void) { return g_streams; }

hipError_t hipStreamSynchronize(hipStream_t)
{
	__sync_fetch_and_add(&g_syncs, 1);
	return hipSuccess;
}

hipError_t hipEventCreate(hipEvent_t *e)
{
	*e = (hipEvent_t) malloc(8);
	return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int)
{
	*e = (hipEvent_t) malloc(8);
	return hipSuccess;
}

hipError_t hipEventDestroy(hipEvent_t e)
{
	free(e);
	return hipSuccess;
}

hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t)
{
	__sync_fetch_and_add(&g_syncs, 1);
	return hipSuccess;
}

hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t)
{
	*ms = 0.f;
	return hipSuccess;
}

hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// kernel<<<...>>> is "push the configuration; if that worked, call the stub, which pops it and
// calls hipLaunchKernel": keep the pair here so launches reach the counter below
struct MockConfig {
	dim3 grid, block;
	size_t shmem;
	hipStream_t stream;
};
static thread_local MockConfig g_config;

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
	g_config = MockConfig{ grid, block, shmem, stream };
	return hipSuccess;
}

hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
	*grid = g_config.grid;
	*block = g_config.block;
	*shmem = g_config.shmem;
	*stream = g_config.stream;
	return hipSuccess;
}

static long g_launches = 0;

// the kernels do nothing
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t)
{
	__sync_fetch_and_add(&g_launches, 1);
	return hipSuccess;
}

long mock_hip_launches(void) { return g_launches; }

} // extern "C"
