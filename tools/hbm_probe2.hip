// Strip-read probe, round 2: which load mechanism streams the fused reduce's access pattern
// (blocks walking down column strips of a 64 KB-pitch image) fastest on this part?
//   plain  : global_load_dwordx2 / dwordx4 into registers (what round 1 shipped), +- nt
//   ldsdma : global_load_lds_dwordx4 (1 KiB per wave-instruction) into an LDS ring, the waves
//            then ds_read their columns back; loads stay in flight across the barrier
//            (hand-counted s_waitcnt vmcnt), +- nt
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_probe2.hip -o tools/hbm_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int PITCH = 65536; // bytes per image row (16384 RGBA pixels)

// ---- plain register loads: VB bytes per lane per row, 256 threads -> strip of 256*VB bytes
template <int VB, bool NT>
__global__ void __launch_bounds__(256) strip_plain(const unsigned char *__restrict__ in, int rows_per_block,
	int strips, int strip_pitch, unsigned *sink)
{
	typedef unsigned int vec __attribute__((ext_vector_type(VB / 4)));
	const int strip = blockIdx.x % strips;
	const int seg = blockIdx.x / strips;
	const unsigned char *p = in + (size_t) seg * rows_per_block * PITCH + (size_t) strip * strip_pitch +
		threadIdx.x * VB;
	unsigned acc = 0;
	for (int r = 0; r < rows_per_block; r += 8) {
		vec v[8];
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const vec *q = reinterpret_cast<const vec *>(p + (size_t) (r + i) * PITCH);
			v[i] = NT ? __builtin_nontemporal_load(q) : *q;
		}
#pragma unroll
		for (int i = 0; i < 8; i++)
#pragma unroll
			for (int k = 0; k < VB / 4; k++)
				acc ^= v[i][k];
	}
	if (acc == 0x12345678)
		*sink = acc;
}

// ---- LDS-DMA ring.  Block = WAVES waves; a row of the strip is SW bytes = SW/1024 pieces of
// 1 KiB (one wave-instruction each); a slot holds R rows; the ring has G slots.  Wave w issues
// the pieces p with p % WAVES == w of a slot (every wave issues PPW = R*SW/1024/WAVES pieces).
template <bool NT>
static __device__ __forceinline__ void glds16(const unsigned char *gsrc, unsigned lds_dst)
{
	unsigned keep;
	if (NT)
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
					 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
	else
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
					 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int N>
static __device__ __forceinline__ void wait_vm()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int WAVES, int SW, int R, int G, bool NT>
__global__ void __launch_bounds__(WAVES * 64) strip_ldsdma(const unsigned char *__restrict__ in,
	int rows_per_block, int strips, int strip_pitch, unsigned *sink)
{
	constexpr int PIECES = R * SW / 1024; // per slot
	constexpr int PPW = PIECES / WAVES;
	static_assert(PIECES % WAVES == 0, "pieces per wave");
	constexpr int SLOT = R * SW;
	extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
	const int t = threadIdx.x;
	const int lane = t & 63;
	const int w = __builtin_amdgcn_readfirstlane(t >> 6);
	const int strip = blockIdx.x % strips;
	const int seg = blockIdx.x / strips;
	const unsigned char *base = in + (size_t) seg * rows_per_block * PITCH + (size_t) strip * strip_pitch;
	const unsigned ring0 = (unsigned) (size_t) ring; // LDS byte address of the ring
	const int nslots = rows_per_block / R;

	auto issue = [&](int s) {
		const unsigned dst = ring0 + (unsigned) ((s % G) * SLOT);
#pragma unroll
		for (int k = 0; k < PPW; k++) {
			const int piece = k * WAVES + w; // 1 KiB pieces of the slot, row-major
			const int row = piece / (SW / 1024), part = piece % (SW / 1024);
			glds16<NT>(base + (size_t) (s * R + row) * PITCH + part * 1024 + lane * 16,
				dst + (unsigned) (piece * 1024));
		}
	};

	// prologue: G - 1 slots in flight
#pragma unroll
	for (int s = 0; s < G - 1; s++)
		if (s < nslots)
			issue(s);
	unsigned acc = 0;
	for (int s = 0; s < nslots; s++) {
		// slot s landed for this wave when at most G - 2 younger slots are outstanding
		if (s + G - 2 < nslots)
			wait_vm<(G - 2) * PPW>();
		else
			wait_vm<0>();
		__builtin_amdgcn_s_barrier(); // everyone's pieces of slot s landed; everyone done reading slot s - 1
		if (s + G - 1 < nslots)
			issue(s + G - 1); // into the slot read at iteration s - 1
		// consume: every thread reads its SW/256 bytes of each row
		const unsigned char *slot = ring + (s % G) * SLOT;
#pragma unroll
		for (int i = 0; i < R; i++) {
			if (SW / (WAVES * 64) == 8) {
				const uint2 v = *reinterpret_cast<const uint2 *>(slot + i * SW + t * 8);
				acc ^= v.x ^ v.y;
			}
			else if (SW / (WAVES * 64) == 16) {
				const uint4 v = *reinterpret_cast<const uint4 *>(slot + i * SW + t * 16);
				acc ^= v.x ^ v.y ^ v.z ^ v.w;
			}
			else {
				const unsigned v = *reinterpret_cast<const unsigned *>(slot + i * SW + t * 4);
				acc ^= v;
			}
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	if (acc == 0x12345678)
		*sink = acc;
}


// ---- the fused reduce's exact tile geometry: tiles_x x tiles_y tiles, tile (bx, by) reads `rows`
// rows from row by * row_pitch and SW bytes from byte column bx * col_pitch; XCD-contiguous tile
// ranges and the serpentine walk as in reduce_u8.hip.  `lds_pad` bytes of dynamic LDS bound
// the blocks per CU like the real kernel's planes / stage do.
struct TileGeo {
	int tiles_x, tiles_y, rows, row_pitch, col_pitch;
};

static __device__ __forceinline__ bool tile_of_block(const TileGeo &g, int *bx, int *by)
{
	const int per_xcd = gridDim.x / 8;
	const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (tile >= g.tiles_x * g.tiles_y)
		return false;
	*by = tile / g.tiles_x;
	*bx = tile - *by * g.tiles_x;
	return true;
}

template <int VB, bool NT, int DEPTH, int THREADS = 256>
__global__ void __launch_bounds__(THREADS) tile_plain(const unsigned char *__restrict__ in, TileGeo g, unsigned *sink)
{
	typedef unsigned int vec __attribute__((ext_vector_type(VB / 4)));
	extern __shared__ unsigned char pad[];
	int bx, by;
	if (!tile_of_block(g, &bx, &by))
		return;
	const int dir = (by & 1) ? -1 : 1;
	const int r0 = (by & 1) ? by * g.row_pitch + g.rows - 1 : by * g.row_pitch;
	const unsigned char *p = in + (size_t) bx * g.col_pitch + threadIdx.x * VB;
	unsigned acc = 0;
	for (int r = 0; r < g.rows; r += DEPTH) {
		vec v[DEPTH];
#pragma unroll
		for (int i = 0; i < DEPTH; i++) {
			const vec *q = reinterpret_cast<const vec *>(p + (size_t) (r0 + dir * (r + i)) * PITCH);
			v[i] = NT ? __builtin_nontemporal_load(q) : *q;
		}
#pragma unroll
		for (int i = 0; i < DEPTH; i++)
#pragma unroll
			for (int k = 0; k < VB / 4; k++)
				acc ^= v[i][k];
	}
	if (acc == 0x12345678) {
		*sink = acc;
		pad[threadIdx.x] = 1;
	}
}

// two buffers of DEPTH rows, refilled as soon as consumed (the fused kernel's pattern: DEPTH..2*DEPTH
// loads in flight per wave); WAITALL = wait for everything outstanding before each refill
template <int VB, bool NT, int DEPTH, bool WAITALL>
__global__ void __launch_bounds__(256) tile_pipe(const unsigned char *__restrict__ in, TileGeo g, unsigned *sink)
{
	typedef unsigned int vec __attribute__((ext_vector_type(VB / 4)));
	extern __shared__ unsigned char pad[];
	int bx, by;
	if (!tile_of_block(g, &bx, &by))
		return;
	const int dir = (by & 1) ? -1 : 1;
	const int r0 = (by & 1) ? by * g.row_pitch + g.rows - 1 : by * g.row_pitch;
	const unsigned char *p = in + (size_t) bx * g.col_pitch + threadIdx.x * VB;
	unsigned acc = 0;
	vec v[2][DEPTH];
	auto load = [&](int b, int r) __attribute__((always_inline)) {
#pragma unroll
		for (int i = 0; i < DEPTH; i++) {
			const vec *q = reinterpret_cast<const vec *>(p + (size_t) (r0 + dir * (r + i)) * PITCH);
			v[b][i] = NT ? __builtin_nontemporal_load(q) : *q;
		}
	};
	auto eat = [&](int b) __attribute__((always_inline)) {
#pragma unroll
		for (int i = 0; i < DEPTH; i++)
#pragma unroll
			for (int k = 0; k < VB / 4; k++)
				acc ^= v[b][i][k];
	};
	load(0, 0);
	load(1, DEPTH);
	for (int r = 0; r < g.rows; r += 2 * DEPTH) {
		eat(0);
		if (WAITALL)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (r + 2 * DEPTH < g.rows)
			load(0, r + 2 * DEPTH);
		eat(1);
		if (WAITALL)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (r + 3 * DEPTH < g.rows)
			load(1, r + 3 * DEPTH);
	}
	if (acc == 0x12345678) {
		*sink = acc;
		pad[threadIdx.x] = 1;
	}
}

template <int WAVES, int SW, int R, int G, bool NT>
__global__ void __launch_bounds__(WAVES * 64) tile_ldsdma(const unsigned char *__restrict__ in, TileGeo g,
	unsigned *sink)
{
	constexpr int PIECES = R * SW / 1024; // per slot
	constexpr int PPW = PIECES / WAVES;
	static_assert(PIECES % WAVES == 0, "pieces per wave");
	constexpr int SLOT = R * SW;
	extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
	const int t = threadIdx.x;
	const int lane = t & 63;
	const int w = __builtin_amdgcn_readfirstlane(t >> 6);
	int bx, by;
	if (!tile_of_block(g, &bx, &by))
		return;
	const int dir = (by & 1) ? -1 : 1;
	const int r0 = (by & 1) ? by * g.row_pitch + g.rows - 1 : by * g.row_pitch;
	const unsigned char *base = in + (size_t) bx * g.col_pitch;
	const unsigned ring0 = (unsigned) (size_t) ring;
	const int nslots = g.rows / R;

	auto issue = [&](int s) {
		const unsigned dst = ring0 + (unsigned) ((s % G) * SLOT);
#pragma unroll
		for (int k = 0; k < PPW; k++) {
			const int piece = k * WAVES + w;
			const int row = piece / (SW / 1024), part = piece % (SW / 1024);
			glds16<NT>(base + (size_t) (r0 + dir * (s * R + row)) * PITCH + part * 1024 + lane * 16,
				dst + (unsigned) (piece * 1024));
		}
	};
#pragma unroll
	for (int s = 0; s < G - 1; s++)
		if (s < nslots)
			issue(s);
	unsigned acc = 0;
	for (int s = 0; s < nslots; s++) {
		if (s + G - 2 < nslots)
			wait_vm<(G - 2) * PPW>();
		else
			wait_vm<0>();
		__builtin_amdgcn_s_barrier();
		if (s + G - 1 < nslots)
			issue(s + G - 1);
		const unsigned char *slot = ring + (s % G) * SLOT;
#pragma unroll
		for (int i = 0; i < R; i++) {
			if (SW / (WAVES * 64) == 8) {
				const uint2 v = *reinterpret_cast<const uint2 *>(slot + i * SW + t * 8);
				acc ^= v.x ^ v.y;
			}
			else {
				const uint4 v = *reinterpret_cast<const uint4 *>(slot + i * SW + t * 16);
				acc ^= v.x ^ v.y ^ v.z ^ v.w;
			}
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	if (acc == 0x12345678)
		*sink = acc;
}

template <typename F>
static double time_ms(F f, int reps)
{
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	f();
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	for (int i = 0; i < reps; i++)
		f();
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms;
	CHECK(hipEventElapsedTime(&ms, a, b));
	CHECK(hipGetLastError());
	return ms / reps;
}

static const size_t BYTES = (size_t) 1 << 30;
static void *g_in;
static unsigned *g_sink;

template <int VB, bool NT>
static void run_plain(const char *name, int rows, int pitch)
{
	const int sw = 256 * VB;
	const int strips = pitch == sw ? PITCH / sw : (PITCH - sw) / pitch + 1;
	const int segs = 16384 / rows;
	double ms = time_ms([&] { strip_plain<VB, NT><<<strips * segs, 256>>>((const unsigned char *) g_in, rows, strips, pitch, g_sink); }, 20);
	const double bytes = (double) strips * segs * rows * sw;
	printf("%-34s strip %4d B pitch %4d rows/blk %4d blocks %5d: %.4f ms  %.0f GB/s (requested)\n", name, sw,
		pitch, rows, strips * segs, ms, bytes / ms / 1e6);
}

template <int WAVES, int SW, int R, int G, bool NT>
static void run_dma(const char *name, int rows, int pitch)
{
	const int strips = pitch == SW ? PITCH / SW : (PITCH - SW) / pitch + 1;
	const int segs = 16384 / rows;
	const size_t lds = (size_t) G * R * SW;
	CHECK(hipFuncSetAttribute((const void *) strip_ldsdma<WAVES, SW, R, G, NT>,
		hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
	double ms = time_ms([&] {
		strip_ldsdma<WAVES, SW, R, G, NT><<<strips * segs, WAVES * 64, lds>>>((const unsigned char *) g_in, rows, strips, pitch, g_sink);
	}, 20);
	const double bytes = (double) strips * segs * rows * SW;
	printf("%-34s strip %4d B pitch %4d rows/blk %4d blocks %5d lds %3zu KB: %.4f ms  %.0f GB/s (requested)\n",
		name, SW, pitch, rows, strips * segs, lds / 1024, ms, bytes / ms / 1e6);
}


template <int VB, bool NT, int DEPTH, int THREADS = 256>
static void run_tile_plain(const char *name, TileGeo g, int lds_pad)
{
	const int tiles = g.tiles_x * g.tiles_y, grid = (tiles + 7) / 8 * 8;
	CHECK(hipFuncSetAttribute((const void *) tile_plain<VB, NT, DEPTH, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	double ms = time_ms([&] { tile_plain<VB, NT, DEPTH, THREADS><<<grid, THREADS, lds_pad>>>((const unsigned char *) g_in, g, g_sink); }, 20);
	const double bytes = (double) tiles * g.rows * THREADS * VB;
	printf("%-30s %2dx%2d tiles rows %4d pitch %4d B lds %3d KB: %.4f ms  %.0f GB/s requested (%.3fx of 1 GiB)\n", name,
		g.tiles_x, g.tiles_y, g.rows, g.col_pitch, lds_pad / 1024, ms, bytes / ms / 1e6, bytes / BYTES);
}

template <int VB, bool NT, int DEPTH, bool WAITALL>
static void run_tile_pipe(const char *name, TileGeo g, int lds_pad)
{
	const int tiles = g.tiles_x * g.tiles_y, grid = (tiles + 7) / 8 * 8;
	CHECK(hipFuncSetAttribute((const void *) tile_pipe<VB, NT, DEPTH, WAITALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	double ms = time_ms([&] { tile_pipe<VB, NT, DEPTH, WAITALL><<<grid, 256, lds_pad>>>((const unsigned char *) g_in, g, g_sink); }, 20);
	const double bytes = (double) tiles * g.rows * 256 * VB;
	printf("%-30s %2dx%2d tiles rows %4d pitch %4d B lds %3d KB: %.4f ms  %.0f GB/s requested (%.3fx of 1 GiB)\n", name,
		g.tiles_x, g.tiles_y, g.rows, g.col_pitch, lds_pad / 1024, ms, bytes / ms / 1e6, bytes / BYTES);
}

template <int WAVES, int SW, int R, int G, bool NT>
static void run_tile_dma(const char *name, TileGeo g, int lds_total)
{
	const int tiles = g.tiles_x * g.tiles_y, grid = (tiles + 7) / 8 * 8;
	const int lds = lds_total > G * R * SW ? lds_total : G * R * SW;
	CHECK(hipFuncSetAttribute((const void *) tile_ldsdma<WAVES, SW, R, G, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	double ms = time_ms([&] { tile_ldsdma<WAVES, SW, R, G, NT><<<grid, WAVES * 64, lds>>>((const unsigned char *) g_in, g, g_sink); }, 20);
	const double bytes = (double) tiles * g.rows * SW;
	printf("%-30s %2dx%2d tiles rows %4d pitch %4d B lds %3d KB: %.4f ms  %.0f GB/s requested (%.3fx of 1 GiB)\n", name,
		g.tiles_x, g.tiles_y, g.rows, g.col_pitch, lds / 1024, ms, bytes / ms / 1e6, bytes / BYTES);
}

static void tile_section()
{
	// the shipped kernel's two geometries (unaligned 59-pixel tiles, aligned 56-pixel tiles)
	const TileGeo u59 = { 35, 29, 608, 568, 1888 };  // oht 71: 29 * 568 = 16472 (last tile overruns: pad)
	const TileGeo a56 = { 37, 27, 648, 608, 1792 };  // oht 76
	for (int lds : { 0, 40 * 1024 }) {
		run_tile_plain<8, false, 8>("tile plain8 d8", u59, lds);
		run_tile_plain<8, true, 8>("tile plain8 nt d8", u59, lds);
		run_tile_plain<8, true, 8>("tile plain8 nt d8 aligned", a56, lds);
		run_tile_plain<8, true, 4>("tile plain8 nt d4 aligned", a56, lds);
		run_tile_plain<8, true, 16>("tile plain8 nt d16 aligned", a56, lds);
	}
	// LDS-DMA, 256 threads, 2 KB spans, 4 blocks / CU (40 KB each)
	run_tile_dma<4, 2048, 4, 2, true>("tile dma w4 2K r4 g2 nt", a56, 40 * 1024);
	run_tile_dma<4, 2048, 4, 3, true>("tile dma w4 2K r4 g3 nt", a56, 40 * 1024);
	run_tile_dma<4, 2048, 4, 4, true>("tile dma w4 2K r4 g4 nt", a56, 40 * 1024);
	run_tile_dma<4, 2048, 8, 2, true>("tile dma w4 2K r8 g2 nt", a56, 40 * 1024);
	run_tile_dma<4, 2048, 4, 4, false>("tile dma w4 2K r4 g4", a56, 40 * 1024);
	run_tile_dma<4, 2048, 4, 4, true>("tile dma w4 2K r4 g4 nt u59", u59, 40 * 1024);
	// 512 threads, 4 KB spans (1024 pixels), 2 blocks / CU: 123-pixel tiles (unaligned) / 120 (aligned)
	const TileGeo w123 = { 17, 30, 592, 552, 3936 };
	const TileGeo w120 = { 18, 28, 632, 592, 3840 };
	run_tile_dma<8, 4096, 4, 3, true>("tile dma w8 4K r4 g3 nt w123", w123, 80 * 1024);
	run_tile_dma<8, 4096, 4, 4, true>("tile dma w8 4K r4 g4 nt w123", w123, 80 * 1024);
	run_tile_dma<8, 4096, 4, 3, true>("tile dma w8 4K r4 g3 nt w120", w120, 80 * 1024);
	run_tile_dma<8, 4096, 4, 3, false>("tile dma w8 4K r4 g3 w123", w123, 80 * 1024);
	// one block per CU: 15 tiles down, 17 across
	const TileGeo big = { 17, 15, 1136, 1096, 3936 };
	run_tile_dma<8, 4096, 4, 4, true>("tile dma w8 4K r4 g4 nt big", big, 0);
	run_tile_dma<8, 4096, 4, 8, true>("tile dma w8 4K r4 g8 nt big", big, 0);
	run_tile_dma<16, 4096, 4, 8, true>("tile dma w16 4K r4 g8 nt big", big, 0);
}

// round 3: 1024-pixel tiles (512 threads x 8 B or 256 threads x 16 B), two blocks per CU
static void wide_section()
{
	const TileGeo u59 = { 35, 29, 608, 568, 1888 };
	const TileGeo w123 = { 17, 30, 592, 552, 3936 };  // oht 69
	const TileGeo w120 = { 18, 28, 632, 592, 3840 };  // line-aligned pitch, oht 74
	const TileGeo w123t = { 17, 15, 1136, 1096, 3936 }; // one block per CU
	const TileGeo u59t = { 35, 15, 1136, 1096, 1888 };  // 256 threads, two blocks per CU, tall
	const int K80 = 80 * 1024, K40 = 40 * 1024;
	run_tile_plain<8, false, 8>("u59 256x8 d8", u59, K40);
	run_tile_plain<8, false, 4>("u59 256x8 d4", u59, K40);
	run_tile_plain<8, true, 4>("u59 256x8 nt d4", u59, K40);
	run_tile_plain<8, false, 8, 512>("w123 512x8 d8", w123, K80);
	run_tile_plain<8, true, 8, 512>("w123 512x8 nt d8", w123, K80);
	run_tile_plain<8, false, 4, 512>("w123 512x8 d4", w123, K80);
	run_tile_plain<8, true, 4, 512>("w123 512x8 nt d4", w123, K80);
	run_tile_plain<8, true, 2, 512>("w123 512x8 nt d2", w123, K80);
	run_tile_plain<8, false, 4, 512>("w120 512x8 d4 aligned", w120, K80);
	run_tile_plain<8, true, 4, 512>("w120 512x8 nt d4 aligned", w120, K80);
	run_tile_plain<8, true, 8, 512>("w120 512x8 nt d8 aligned", w120, K80);
	run_tile_plain<16, false, 8>("w123 256x16 d8", w123, K80);
	run_tile_plain<16, true, 8>("w123 256x16 nt d8", w123, K80);
	run_tile_plain<16, false, 4>("w123 256x16 d4", w123, K80);
	run_tile_plain<16, true, 4>("w123 256x16 nt d4", w123, K80);
	run_tile_plain<16, true, 4>("w120 256x16 nt d4 aligned", w120, K80);
	run_tile_plain<16, true, 8>("w120 256x16 nt d8 aligned", w120, K80);
	run_tile_plain<16, true, 4>("w123 256x16 nt d4 4blk/CU", w123, K40);
	run_tile_plain<8, true, 4, 512>("w123t 512x8 nt d4 1blk/CU", w123t, 0);
	run_tile_plain<8, true, 8, 512>("w123t 512x8 nt d8 1blk/CU", w123t, 0);
	run_tile_plain<8, true, 4>("u59t 256x8 nt d4 2blk/CU", u59t, K80);
	run_tile_plain<8, true, 8>("u59t 256x8 nt d8 2blk/CU", u59t, K80);
	run_tile_plain<8, false, 8>("u59t 256x8 d8 2blk/CU", u59t, K80);
}

// round 3: how many rows in flight per wave does the shipped geometry want?
static void depth_section()
{
	const TileGeo u59 = { 35, 29, 608, 568, 1888 };
	const int K40 = 40 * 1024;
	run_tile_plain<8, false, 2>("u59 d2", u59, K40);
	run_tile_plain<8, false, 3>("u59 d3", u59, K40);
	run_tile_plain<8, false, 4>("u59 d4", u59, K40);
	run_tile_plain<8, false, 5>("u59 d5", u59, K40);
	run_tile_plain<8, false, 6>("u59 d6", u59, K40);
	run_tile_plain<8, false, 8>("u59 d8", u59, K40);
	run_tile_plain<8, true, 4>("u59 nt d4", u59, K40);
	run_tile_pipe<8, false, 4, false>("u59 pipe 2x4", u59, K40);
	run_tile_pipe<8, false, 4, true>("u59 pipe 2x4 waitall", u59, K40);
	run_tile_pipe<8, true, 4, true>("u59 pipe 2x4 waitall nt", u59, K40);
	run_tile_pipe<8, false, 2, false>("u59 pipe 2x2", u59, K40);
	run_tile_pipe<8, false, 2, true>("u59 pipe 2x2 waitall", u59, K40);
	run_tile_pipe<8, false, 3, false>("u59 pipe 2x3", u59, K40);
	run_tile_pipe<8, false, 8, false>("u59 pipe 2x8", u59, K40);
}

int main()
{
	CHECK(hipMalloc(&g_in, BYTES + (64 << 20)));
	CHECK(hipMalloc(&g_sink, 4));
	CHECK(hipMemset(g_in, 1, BYTES));
	if (getenv("PROBE_DEPTH")) {
		for (int rep = 0; rep < 2; rep++) {
			printf("---- depth, pass %d\n", rep);
			depth_section();
		}
		return 0;
	}
	if (getenv("PROBE_WIDE")) {
		for (int rep = 0; rep < 2; rep++) {
			printf("---- wide tiles, pass %d\n", rep);
			wide_section();
		}
		return 0;
	}
	if (getenv("PROBE_TILES")) {
		for (int rep = 0; rep < 2; rep++) {
			printf("---- tiles, pass %d\n", rep);
			tile_section();
		}
		return 0;
	}
	for (int rep = 0; rep < 2; rep++) {
		printf("---- pass %d\n", rep);
		for (int rows : { 512, 1024 }) {
			run_plain<8, false>("plain 8B/lane", rows, 2048);
			run_plain<8, true>("plain 8B/lane nt", rows, 2048);
			run_plain<16, false>("plain 16B/lane", rows, 4096);
			run_plain<16, true>("plain 16B/lane nt", rows, 4096);
		}
		// overlapping strips (the real kernel's 1888-byte tile pitch: 8 % halo columns)
		run_plain<8, false>("plain 8B/lane halo", 512, 1888);
		run_plain<8, true>("plain 8B/lane nt halo", 512, 1888);
		run_plain<16, false>("plain 16B/lane halo", 1024, 3936);
		run_plain<16, true>("plain 16B/lane nt halo", 1024, 3936);
		// LDS-DMA: 256 threads, 2 KB strips
		run_dma<4, 2048, 4, 3, false>("dma w4 2K r4 g3", 512, 2048);
		run_dma<4, 2048, 4, 3, true>("dma w4 2K r4 g3 nt", 512, 2048);
		run_dma<4, 2048, 4, 4, true>("dma w4 2K r4 g4 nt", 512, 2048);
		run_dma<4, 2048, 8, 3, true>("dma w4 2K r8 g3 nt", 512, 2048);
		run_dma<4, 2048, 8, 2, true>("dma w4 2K r8 g2 nt", 512, 2048);
		run_dma<4, 2048, 4, 4, true>("dma w4 2K r4 g4 nt halo", 512, 1888);
		// 512 threads, 4 KB strips
		run_dma<8, 4096, 4, 3, false>("dma w8 4K r4 g3", 1024, 4096);
		run_dma<8, 4096, 4, 3, true>("dma w8 4K r4 g3 nt", 1024, 4096);
		run_dma<8, 4096, 4, 4, true>("dma w8 4K r4 g4 nt", 1024, 4096);
		run_dma<8, 4096, 8, 2, true>("dma w8 4K r8 g2 nt", 1024, 4096);
		run_dma<8, 4096, 4, 3, true>("dma w8 4K r4 g3 nt halo", 1024, 3936);
		// 256 threads, 4 KB strips (16 B per lane per row from LDS)
		run_dma<4, 4096, 4, 3, true>("dma w4 4K r4 g3 nt", 1024, 4096);
		run_dma<4, 4096, 2, 4, true>("dma w4 4K r2 g4 nt", 1024, 4096);
		// 1024 threads, 8 KB strips
		run_dma<16, 8192, 4, 3, true>("dma w16 8K r4 g3 nt", 2048, 8192);
		run_dma<16, 8192, 4, 4, true>("dma w16 8K r4 g4 nt", 2048, 8192);
	}
	return 0;
}
