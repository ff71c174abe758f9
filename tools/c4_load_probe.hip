// Load skeleton of resize_stream.hip on BASELINE config 4's geometry: 64 images of 8192 x 8192 x 3
// uchar, blocks walking down 2 KB (or 4 KB) column strips of a 24576-byte-pitch image, `DEPTH`
// row loads in flight per lane, nothing computed (the rows are XORed into one register).  What
// does the ACCESS PATTERN alone reach, against the kernel's 4.97 TB/s?
// build: hipcc --offload-arch=gfx950 -O3 tools/c4_load_probe.hip -o /tmp/c4_load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Geo {
	long long image_bytes;
	int pitch;       // bytes per image row
	int rows;        // image rows
	int nstrips, strip_pitch, span;
	int nsegs, seg_rows, overlap; // input rows per segment, rows re-read above a segment
	int n_images, grouped;
};

template <int NT, int VB, int DEPTH, bool NTL>
__global__ void __launch_bounds__(NT) walk(const unsigned char *base, Geo g, unsigned *sink)
{
	typedef unsigned int vec __attribute__((ext_vector_type(VB / 4)));
	typedef const vec __attribute__((address_space(1))) *GIn;
	const int wg = blockIdx.x;
	int strip, unit;
	if (g.grouped) {
		const int grp = (wg >> 3) / g.nstrips;
		strip = (wg >> 3) - grp * g.nstrips;
		unit = grp * 8 + (wg & 7);
	}
	else {
		unit = wg / g.nstrips;
		strip = wg - unit * g.nstrips;
	}
	if (unit >= g.nsegs * g.n_images)
		return;
	const int img = unit / g.nsegs, seg = unit - img * g.nsegs;
	int start = strip * g.strip_pitch;
	if (start > g.pitch - g.span)
		start = g.pitch - g.span;
	const unsigned char *p = base + (long long) img * g.image_bytes + start;
	const unsigned lane = threadIdx.x * VB;
	int r0 = seg * g.seg_rows - g.overlap;
	int r1 = (seg + 1) * g.seg_rows;
	if (r0 < 0)
		r0 = 0;
	if (r1 > g.rows)
		r1 = g.rows;
	vec ring[DEPTH];
	auto load = [&](int r) -> vec {
		const int rc = r < g.rows - 1 ? r : g.rows - 1;
		const unsigned off = (unsigned) rc * (unsigned) g.pitch + lane;
		GIn q = (GIn) (p + off);
		return NTL ? __builtin_nontemporal_load(q) : *q;
	};
#pragma unroll
	for (int i = 0; i < DEPTH; i++)
		ring[i] = load(r0 + i);
	unsigned acc = 0;
	for (int r = r0; r < r1; r += DEPTH) {
#pragma unroll
		for (int i = 0; i < DEPTH; i++) {
#pragma unroll
			for (int k = 0; k < VB / 4; k++)
				acc ^= ring[i][k];
			ring[i] = load(r + DEPTH + i);
		}
	}
	if (acc == 0x12345678)
		*sink = acc;
}

template <int NT, int VB, int DEPTH, bool NTL>
static void run(const char *name, const unsigned char *buf, Geo g, unsigned *sink, int blocks_per_cu_lds)
{
	g.span = NT * VB;
	// strips as the kernel cuts them: (2 tw + 12) shrunk columns of 4 pixels of 3 bytes in the span
	int tw = ((g.span - 3) / 12 - 12) / 2;
	g.strip_pitch = tw * 2 * 4 * 3;
	g.nstrips = (1024 + tw - 1) / tw;
	const long long units = (long long) g.nsegs * g.n_images;
	g.grouped = 1;
	const long long blocks = (units + 7) / 8 * 8 * g.nstrips;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const size_t lds = blocks_per_cu_lds ? (size_t) (160 * 1024 / blocks_per_cu_lds - 1024) : 0;
	if (lds > 64 * 1024)
		CHECK(hipFuncSetAttribute((const void *) walk<NT, VB, DEPTH, NTL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
	float best = 1e9f;
	for (int rep = 0; rep < 4; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((walk<NT, VB, DEPTH, NTL>), dim3((unsigned) blocks), dim3(NT), lds, 0, buf, g, sink);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best)
			best = ms;
	}
	const double alg = (double) g.image_bytes * g.n_images;
	printf("%-34s strips %2d x %4d B, %d segs of %d rows, %5lld blocks, lds %3zu KB: %.3f ms = %.4f ms/image, %4.0f GB/s of the images\n",
		name, g.nstrips, g.span, g.nsegs, g.seg_rows, blocks, lds / 1024, best, best / g.n_images, alg / best / 1e6);
	fflush(stdout);
}

int main()
{
	Geo g;
	g.pitch = 8192 * 3;
	g.rows = 8192;
	g.image_bytes = (long long) g.pitch * g.rows;
	g.n_images = 64;
	g.overlap = 48;
	unsigned char *buf;
	unsigned *sink;
	CHECK(hipMalloc(&buf, (size_t) g.image_bytes * g.n_images));
	CHECK(hipMalloc(&sink, 4));
	CHECK(hipMemset(buf, 1, (size_t) g.image_bytes * g.n_images));
	for (int pass = 0; pass < 2; pass++) {
		printf("---- pass %d\n", pass);
		for (int nsegs : { 3, 4, 8 }) {
			g.nsegs = nsegs;
			g.seg_rows = (g.rows + nsegs - 1) / nsegs;
			run<512, 4, 8, false>("512x4 d8 (shipped shape)", buf, g, sink, 2);
			if (nsegs != 3)
				continue;
			run<512, 4, 8, true>("512x4 d8 nt", buf, g, sink, 2);
			run<512, 4, 4, false>("512x4 d4", buf, g, sink, 2);
			run<512, 4, 16, false>("512x4 d16", buf, g, sink, 2);
			run<512, 4, 8, false>("512x4 d8 3 blocks/CU", buf, g, sink, 3);
			run<512, 4, 8, false>("512x4 d8 4 blocks/CU", buf, g, sink, 4);
			run<256, 8, 8, false>("256x8 d8", buf, g, sink, 4);
			run<256, 8, 4, false>("256x8 d4", buf, g, sink, 4);
			run<256, 8, 4, true>("256x8 d4 nt", buf, g, sink, 4);
			run<256, 8, 4, false>("256x8 d4 8 blocks/CU", buf, g, sink, 8);
			run<256, 16, 4, false>("256x16 d4 (4 KB spans)", buf, g, sink, 4);
			run<256, 16, 4, true>("256x16 d4 nt", buf, g, sink, 4);
			run<256, 16, 8, false>("256x16 d8", buf, g, sink, 4);
			run<512, 8, 4, false>("512x8 d4 (4 KB spans)", buf, g, sink, 2);
			run<512, 8, 4, true>("512x8 d4 nt", buf, g, sink, 2);
			run<512, 8, 8, false>("512x8 d8", buf, g, sink, 2);
			run<1024, 4, 4, false>("1024x4 d4 (4 KB spans)", buf, g, sink, 1);
			run<1024, 4, 8, false>("1024x4 d8", buf, g, sink, 1);
		}
	}
	return 0;
}
