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

// STORE: every 56 rows the block writes 7 output rows of its strip (what the kernel's horizontal pass
// does): 1 = 237-byte fragments at their unaligned places in 3072-byte rows, a byte per lane;
// 2 = the same bytes as dwords; 3 = 256-byte fragments on 256-byte boundaries (rows of 13 x 256),
// a dword per lane; 4 = those as dwordx4; 5 = the 7 fragments of a slab next to each other (1792
// contiguous bytes)
template <int NT, int VB, int DEPTH, bool NTL, int WORK = 0, int STORE = 0, int JIT = 0>
__global__ void __launch_bounds__(NT) walk(const unsigned char *base, Geo g, unsigned *sink, unsigned char *outbuf = nullptr,
	unsigned char *const *outs = nullptr)
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
	int since = 0, slab = 0, flushed_slab = 0;
	unsigned last_slot = (unsigned) (__builtin_amdgcn_s_memrealtime() >> (STORE + 6));
	// (outs: one allocation per image, as the library's pool hands them out, instead of one for all)
	unsigned char *oimg = outs ? outs[img] : outbuf + (size_t) img * (1024 * 3328);
	const int orow0 = seg * ((g.seg_rows + 7) / 8 + 8);
	for (int r = r0; r < r1; r += DEPTH) {
#pragma unroll
		for (int i = 0; i < DEPTH; i++) {
#pragma unroll
			for (int k = 0; k < VB / 4; k++) {
				unsigned v = ring[i][k];
				// WORK dependent VALU instructions per loaded dword (the kernel's vertical pass:
				// ~9 VALU + ~4 SALU per dword and row)
#pragma unroll
				for (int w = 0; w < WORK; w++)
					v = v * 1664525u + acc;
				acc ^= v;
			}
			ring[i] = load(r + DEPTH + i);
		}
		if (STORE >= 7) {
			// the rows of finished slabs wait; every block writes what it holds when the chip-wide
			// 100 MHz clock crosses a multiple of 2^(STORE + 6) ticks (7: 82 us, 8: 164 us, 9: 328 us):
			// the writes of the whole chip fall into short windows instead of trickling
			since += DEPTH;
			if (since >= 56) {
				since = 0;
				slab++;
				const unsigned long long now = __builtin_amdgcn_s_memrealtime();
				const unsigned slot = (unsigned) (now >> (STORE + 6));
				if (slot != last_slot || slab - flushed_slab >= 14 || r + DEPTH >= r1) {
					last_slot = slot;
					const int t = threadIdx.x;
					for (int y = orow0 + flushed_slab * 7 + (t >> 6); y < orow0 + slab * 7 && y < 1024; y += NT / 64) {
						const int b0 = ((strip * 237) & ~3) + 4 * (t & 63);
						if ((t & 63) < 60 && b0 + 4 <= 3072)
							*(unsigned *) (oimg + (size_t) y * 3072 + b0) = acc;
					}
					flushed_slab = slab;
				}
			}
		}
		else if (STORE) {
			since += DEPTH;
			if (since >= 56) {
				since = 0;
				const int t = threadIdx.x;
				for (int k = 0; k < 7; k++) {
					// JIT: strip s writes s % 8 slabs late, so that the fragments of one output row reach
					// the L2 ~10 us apart (blocks of the real kernel drift; the probe's run in step)
					const int y = orow0 + (slab - (JIT ? strip % 8 : 0)) * 7 + k;
					if (y >= 1024 || y < orow0)
						break;
					if (STORE == 1) {
						if (t < 237 && strip * 237 + t < 3072)
							oimg[(size_t) y * 3072 + strip * 237 + t] = (unsigned char) acc;
					}
					else if (STORE == 2) {
						const int b0 = ((strip * 237) & ~3) + 4 * t;
						if (t < 60 && b0 + 4 <= 3072)
							*(unsigned *) (oimg + (size_t) y * 3072 + b0) = acc;
					}
					else if (STORE == 3) {
						if (t < 64)
							*(unsigned *) (oimg + (size_t) y * 3328 + strip * 256 + 4 * t) = acc;
					}
					else if (STORE == 4) {
						if (t < 16)
							*(uint4 *) (oimg + (size_t) y * 3328 + strip * 256 + 16 * t) = make_uint4(acc, acc, acc, acc);
					}
					else if (STORE == 5) {
						if (t < 64)
							*(unsigned *) (oimg + ((size_t) (orow0 / 7 + slab) * 13 + strip) * 1792 + k * 256 + 4 * t) = acc;
					}
				}
				slab++;
			}
		}
	}
	if (acc == 0x12345678)
		*sink = acc;
}

template <int NT, int VB, int DEPTH, bool NTL, int WORK = 0, int STORE = 0, int JIT = 0>
static void run(const char *name, const unsigned char *buf, Geo g, unsigned *sink, int blocks_per_cu_lds, unsigned char *outbuf = nullptr,
	unsigned char *const *outs = nullptr)
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
		CHECK(hipFuncSetAttribute((const void *) walk<NT, VB, DEPTH, NTL, WORK, STORE, JIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
	float best = 1e9f;
	for (int rep = 0; rep < 4; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((walk<NT, VB, DEPTH, NTL, WORK, STORE, JIT>), dim3((unsigned) blocks), dim3(NT), lds, 0, buf, g, sink, outbuf, outs);
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
	unsigned char *outbuf;
	CHECK(hipMalloc(&outbuf, (size_t) 64 * 1024 * 3328 + (16 << 20)));
	// 64 separate 4 MB blocks, with other allocations in between (what a pool that calls hipMalloc per block gets)
	unsigned char *sep[64], *sep3[64];
	void *spacer;
	for (int i = 0; i < 64; i++) {
		CHECK(hipMalloc(&sep[i], 4 << 20));
		CHECK(hipMalloc(&spacer, 12345));
		CHECK(hipMalloc(&sep3[i], 3 * 1024 * 1024 + 4096));
	}
	unsigned char **outs4, **outs3;
	CHECK(hipMalloc(&outs4, sizeof(sep)));
	CHECK(hipMalloc(&outs3, sizeof(sep3)));
	CHECK(hipMemcpy(outs4, sep, sizeof(sep), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(outs3, sep3, sizeof(sep3), hipMemcpyHostToDevice));
	for (int pass = 0; pass < 2; pass++) {
		printf("---- pass %d\n", pass);
		for (int nsegs : { 3, 4, 8 }) {
			g.nsegs = nsegs;
			g.seg_rows = (g.rows + nsegs - 1) / nsegs;
			run<512, 4, 8, false>("512x4 d8 (shipped shape)", buf, g, sink, 2);
			if (nsegs != 3)
				continue;
			run<512, 4, 8, false, 6, 0>("512x4 d8 + 6 VALU, no stores", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 1>("  + 237 B fragments, bytes", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 2>("  + 237 B fragments, dwords", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 3>("  + 256 B aligned, dwords", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 4>("  + 256 B aligned, dwordx4", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 5>("  + 1792 B contiguous per slab", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 7, 0>("  + 237 B dwords, chip-wide windows 82 us", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 8, 0>("  + 237 B dwords, windows 164 us", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 9, 0>("  + 237 B dwords, windows 328 us (14-slab cap)", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 1, 0>("  + 237 B bytes, 64 x hipMalloc(4 MB)", buf, g, sink, 2, outbuf, outs4);
			run<512, 4, 8, false, 6, 2, 0>("  + 237 B dwords, 64 x hipMalloc(4 MB)", buf, g, sink, 2, outbuf, outs4);
			run<512, 4, 8, false, 6, 2, 0>("  + 237 B dwords, 64 x hipMalloc(3 MB+)", buf, g, sink, 2, outbuf, outs3);
			run<512, 4, 8, false, 6, 1, 1>("  + 237 B bytes, strips apart in time", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 2, 1>("  + 237 B dwords, apart in time", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 3, 1>("  + 256 B aligned, apart in time", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6, 5, 1>("  + 1792 B contiguous, apart in time", buf, g, sink, 2, outbuf);
			run<512, 4, 8, false, 6>("512x4 d8 + 6 VALU per dword", buf, g, sink, 2);
			run<512, 4, 8, false, 12>("512x4 d8 + 12 VALU per dword", buf, g, sink, 2);
			run<512, 4, 12, false, 12>("512x4 d12 + 12 VALU per dword", buf, g, sink, 2);
			run<512, 4, 16, false, 12>("512x4 d16 + 12 VALU per dword", buf, g, sink, 2);
			run<512, 4, 8, false, 12>("512x4 d8 + 12 VALU, 3 blocks/CU", buf, g, sink, 3);
			run<512, 4, 8, false, 12>("512x4 d8 + 12 VALU, 4 blocks/CU", buf, g, sink, 4);
			run<256, 8, 8, false, 12>("256x8 d8 + 12 VALU, 4 blocks/CU", buf, g, sink, 4);
			run<256, 8, 4, false, 12>("256x8 d4 + 12 VALU, 4 blocks/CU", buf, g, sink, 4);
			run<256, 8, 4, false, 12>("256x8 d4 + 12 VALU, 8 blocks/CU", buf, g, sink, 8);
			run<512, 4, 8, false, 20>("512x4 d8 + 20 VALU per dword", buf, g, sink, 2);
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
