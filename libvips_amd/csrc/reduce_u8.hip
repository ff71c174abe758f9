// uchar fast paths for gfx950 -- see reduce_u8.h.
//
// reduce_fused_u8x4<S, D>: vips_reduce() on uchar RGBA with an even integer
// shrink S on both axes and a constant coefficient phase (input size a
// multiple of S gives phase 0: SURVEY.md appendix "C2 phase arithmetic").
// One launch does reducev (reducev.cpp:418-459) AND reduceh
// (reduceh.cpp:269-328); the vertically reduced scanlines live only in LDS.
//
//   workgroup = 256 threads = one output tile (OWT x OHT pixels)
//   thread t  = input columns col0 + 2t, col0 + 2t + 1 (8 contiguous bytes per
//               row: a wave reads 512 contiguous bytes per scanline)
//   vertical  : the thread walks down the tile's input rows in groups of S.
//               Row S*g + i is tap k = S*d + i of output row g - d, d < D, so D
//               accumulator sets are live and one output row completes per
//               group -- every input byte is loaded from HBM once per tile and
//               used D times from registers.  Rows are paired so one
//               v_dot2_i32_i16 does two taps: v_perm_b32 builds (row r, row r+1)
//               i16 pairs of one channel, the coefficient pair is a scalar.
//   LDS       : a finished row is stored as u16 pairs, planar per channel
//               (plane[row][channel][column]); after R rows a barrier, then
//   horizontal: every thread takes output pixels of the R x OWT strip; its 4
//               channels are D*S/2 dot2 over consecutive LDS dwords
//               (ds_read_b128, 16-byte aligned for S = 8, conflict-free across
//               a wave), rounds, packs RGBA and stores one dword.
//
// Integer arithmetic is exact, so the i32 sums equal the reference's whatever
// the summation order; rounding/clipping is templates.h:152-157.
#include "reduce_u8.h"
#include "kernel_stmt.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <cstring>
#include <vector>

namespace vh {

typedef short short2v __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ int dot2(unsigned int pix, unsigned int coef, int acc)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, pix),
		__builtin_bit_cast(short2v, coef), acc, false);
}

// (sum + 2048) >> 12, clipped to 0..255 (templates.h:152-157).
//
// The empty asm keeps the shift and the clamp apart on purpose: when hipcc
// (ROCm 7.2) sees shift+clamp of two values being packed into bytes it selects
// gfx950's v_ashr_pk_u8_i32 and then treats bits 31:16 of the result as zero,
// but the hardware leaves the old register contents there -- OR-ing a third
// channel in at bit 16 picked up garbage (found as +1..+9 errors in the blue
// channel only, see DESIGN.md "toolchain findings").
static __device__ __forceinline__ int fin_u8(int s)
{
	s = (s + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT;
	VH_VECTOR1(s);
	return min(max(s, 0), 255);
}

constexpr int FUSED_THREADS = 256;
constexpr int FUSED_SPAN = 2 * FUSED_THREADS; // input columns per tile

struct FusedArgs {
	const unsigned char *in;
	long long in_stride;
	int in_left, in_top;   // origin of the input window
	int in_right;          // in_left + window width
	int pairs;             // MFMA kernel: every tile can fetch whole pixel pairs (see load_rows)
	int stagger;           // MFMA kernel: groups of phase shift between the 4 blocks of a CU
	int burst_rows;        // MFMA kernel: staged output rows per burst (a multiple of 8)
	int im_width, im_height;
	unsigned char *out;
	long long out_stride;
	int out_width, out_height; // region being generated
	int fx0, fy0;              // first tap (un-embedded input coords) of output (0, 0) of the region
	int owt, oht;              // tile size in output pixels
	int tiles_x, tiles;
	int aligned8;           // input base and stride are multiples of 8 bytes
	int small_window;       // the input window spans < 2 GB: 32-bit byte offsets are safe
	int debug;              // VIPS_HIP_FUSED_DEBUG ablation bits (profiling only; 0 in production)
	int xshift;             // MFMA kernel: a tile's lanes start this many columns left of its first tap,
	                        // so that every wave's 512-byte row segment starts on a 128-byte line
};

// Coefficients travel BY VALUE in the kernel-argument segment: they are read with
// scalar loads (s_load from kernarg memory, dynamic scalar offset), need no device
// allocation and cannot alias the pixel stores.
//   cv / ch = vertical / horizontal i16 coefficient pairs (lo half = even tap).
template <int S, int D>
struct FusedCoefs {
	unsigned int cv[D * (S / 2)];
	unsigned int cv_flip[D * (S / 2)]; // taps reversed, for tiles walked bottom-up
	unsigned int ch[D * (S / 2)];
};

// The accumulator set `slot` holds output row j with j mod D == slot; at input
// group g (rot = g mod D) that row is d = (rot - slot) mod D groups old, i.e. the
// group's rows are its taps S*d .. S*d + S-1.  Accumulators therefore never move:
// the scalar coefficient block rotates instead (one s_load per group).
template <int S, int D>
struct FusedStep {
	static constexpr int PLANE = FUSED_SPAN / 2;
	static constexpr int NP = S * D / 2;
	typedef const unsigned int __attribute__((address_space(4))) *KernargWords;

	// Rows first_row + dir * i, i < S (dir = -1 when the tile is walked bottom-up).
	static __device__ __forceinline__ void load(const FusedArgs &a, uint2 (&px)[S], int first_row,
		int dir, int ca, int cb, bool interior)
	{
		if (interior) {
#pragma unroll
			for (int i = 0; i < S; i++) {
				const int row = min(max(first_row + dir * i, 0), a.im_height - 1) - a.in_top;
				px[i] = *reinterpret_cast<const uint2 *>(a.in + row * a.in_stride + 4 * ca);
			}
		}
		else {
#pragma unroll
			for (int i = 0; i < S; i++) {
				const int row = min(max(first_row + dir * i, 0), a.im_height - 1) - a.in_top;
				const unsigned char *line = a.in + row * a.in_stride;
				px[i].x = *reinterpret_cast<const unsigned int *>(line + 4 * ca);
				px[i].y = *reinterpret_cast<const unsigned int *>(line + 4 * cb);
			}
		}
	}

	// Group ROT (mod D): accumulator set `slot` is d = (ROT - slot) mod D groups old, so
	// this group's rows are its taps S*d .. S*d + S-1.  ROT is a template argument, so the
	// accumulators never move and every coefficient is a kernarg scalar at a fixed offset.
	template <int ROT>
	static __device__ __forceinline__ void accumulate(const uint2 (&px)[S], int (&acc)[D][8],
		KernargWords kcv)
	{
#pragma unroll
		for (int i = 0; i < S; i += 2) {
#pragma unroll
			for (int p = 0; p < 2; p++) {
				const unsigned int ra = p ? px[i].y : px[i].x;
				const unsigned int rb = p ? px[i + 1].y : px[i + 1].x;
#pragma unroll
				for (int c = 0; c < 4; c++) {
					// bytes: [ra.c, 0, rb.c, 0]
					const unsigned int pair =
						__builtin_amdgcn_perm(rb, ra, 0x0c000c00u | (unsigned) c | ((4u + c) << 16));
#pragma unroll
					for (int s = 0; s < D; s++) {
						constexpr int dummy = 0;
						(void) dummy;
						const int d = (ROT - s + D) % D;
						acc[s][p * 4 + c] = dot2(pair, kcv[d * (S / 2) + i / 2], acc[s][p * 4 + c]);
					}
				}
			}
		}
	}

	// Round accumulator set SLOT into LDS row `lds_row` (when it is a real row) and clear it.
	template <int SLOT>
	static __device__ __forceinline__ void retire(int (&acc)[D][8], unsigned int *lds, int lds_row,
		int t, bool store)
	{
		if (store) {
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const unsigned int v = (unsigned) fin_u8(acc[SLOT][c]) |
					((unsigned) fin_u8(acc[SLOT][4 + c]) << 16);
				lds[(lds_row * 4 + c) * PLANE + t] = v;
			}
		}
#pragma unroll
		for (int c = 0; c < 8; c++)
			acc[SLOT][c] = 0;
	}

	// One batch = D consecutive groups (ROT = 0 .. D-1), statically unrolled, with the
	// next group's rows always in flight; group g completes output row g - (D - 1), which
	// lands in LDS row ROT.
	template <int ROT>
	static __device__ __forceinline__ void batch(const FusedArgs &a, uint2 (&cur)[S], uint2 (&nxt)[S],
		int g0, int ngroups, int (&acc)[D][8], unsigned int *lds, KernargWords kcv, int t, int row0,
		int dir, int ca, int cb, bool interior, int oh)
	{
		if constexpr (ROT < D) {
			const int g = g0 + ROT;
			if (g < ngroups) {
				if (g + 1 < ngroups)
					load(a, nxt, row0 + dir * S * (g + 1), dir, ca, cb, interior);
				accumulate<ROT>(cur, acc, kcv);
				const int j = g - (D - 1);
				retire<(ROT + 1) % D>(acc, lds, ROT, t, j >= 0 && j < oh);
			}
			batch<ROT + 1>(a, nxt, cur, g0, ngroups, acc, lds, kcv, t, row0, dir, ca, cb, interior, oh);
		}
	}
};

template <int S, int D>
__global__ void __launch_bounds__(FUSED_THREADS, 4)
reduce_fused_u8x4(FusedArgs a, FusedCoefs<S, D> k_by_value)
{
	// Index the coefficient block where it lies in the kernarg segment (constant
	// address space, scalar loads at immediate offsets).
	typedef FusedStep<S, D> Step;
	typedef typename Step::KernargWords KernargWords;
	static_assert(sizeof(FusedArgs) % alignof(FusedCoefs<S, D>) == 0, "kernarg layout");
	const KernargWords kcv = (KernargWords) ((const char __attribute__((address_space(4))) *)
										   __builtin_amdgcn_kernarg_segment_ptr() +
		sizeof(FusedArgs));
	const KernargWords kch = kcv + 2 * (S * D / 2);
	(void) k_by_value;
	constexpr int NP = S * D / 2; // coefficient pairs
	constexpr int PLANE = FUSED_SPAN / 2; // dwords per (row, channel) plane
	__shared__ __attribute__((aligned(16))) unsigned int lds[D * 4 * PLANE];

	// XCD-aware tile order: block b runs on XCD b % 8, so give every XCD a
	// contiguous run of tiles (row-major): horizontally adjacent tiles share
	// their (D-1)*S-column halo through one L2.
	const int per_xcd = gridDim.x / 8;
	const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (tile >= a.tiles)
		return;

	const int t = threadIdx.x;
	const int bx = tile % a.tiles_x;
	const int by = tile / a.tiles_x;
	const int x0 = bx * a.owt;
	const int y0 = by * a.oht;
	const int ow = min(a.owt, a.out_width - x0);
	const int oh = min(a.oht, a.out_height - y0);

	// this thread's two input columns, clamped to the image (vips_embed COPY)
	const int tile_col0 = a.fx0 + S * x0;
	const int col0 = tile_col0 + 2 * t;
	const int ca = min(max(col0, 0), a.im_width - 1) - a.in_left;
	const int cb = min(max(col0 + 1, 0), a.im_width - 1) - a.in_left;
	// block-uniform: no column of this tile touches the left/right edge, so every
	// thread reads 8 aligned contiguous bytes per row
	const bool interior = a.aligned8 && tile_col0 >= 0 && tile_col0 + FUSED_SPAN <= a.im_width &&
		(((tile_col0 - a.in_left) & 1) == 0);
	// Serpentine: odd tile rows are walked bottom-up, so a tile reads the (D-1)*S halo
	// rows it shares with its vertical neighbour at the same moment the neighbour does
	// (all tiles are resident and advance in step) and one of the two reads hits L2 /
	// Infinity Cache instead of HBM.  Bottom-up is the same code on the flipped
	// problem: rows counted from the last one, taps reversed (k.cv_flip).
	const bool flip = (by & 1) != 0;
	const int dir = flip ? -1 : 1;
	const int row0 = flip ? a.fy0 + S * (y0 + oh - 1) + S * D - 1 : a.fy0 + S * y0;
	const KernargWords kcv_dir = flip ? kcv + S * D / 2 : kcv;

	int acc[D][8];
#pragma unroll
	for (int d = 0; d < D; d++)
#pragma unroll
		for (int c = 0; c < 8; c++)
			acc[d][c] = 0;

	const int ngroups = oh + D - 1;
	uint2 buf0[S], buf1[S];
	Step::load(a, buf0, row0, dir, ca, cb, interior);

	for (int g0 = 0; g0 < ngroups; g0 += D) {
		Step::template batch<0>(a, buf0, buf1, g0, ngroups, acc, lds, kcv_dir, t, row0, dir, ca, cb, interior, oh);
		if (D & 1) {
			// an odd number of steps leaves the prefetched rows in the other buffer
#pragma unroll
			for (int i = 0; i < S; i++)
				buf0[i] = buf1[i];
		}

		// ---- horizontal pass over the rows this batch completed:
		// j = g0 + r - (D - 1) for r = 0 .. D-1, kept in LDS row r
		const int jlo = max(g0 - (D - 1), 0);
		const int jhi = min(g0, oh - 1); // inclusive
		if (jhi < jlo)
			continue;
		__syncthreads();
		const int nrows = jhi - jlo + 1;
		const int r_lo = jlo - (g0 - (D - 1));
		const int items = nrows * ow;
		for (int it = t; it < items; it += FUSED_THREADS) {
			const int rr = it / ow;
			const int xo = it - rr * ow;
			unsigned int rgba = 0;
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const unsigned int *src = &lds[((r_lo + rr) * 4 + c) * PLANE + xo * (S / 2)];
				int sum = 0;
				if (S % 8 == 0) {
#pragma unroll
					for (int q = 0; q < NP; q += 4) {
						const uint4 v = *reinterpret_cast<const uint4 *>(src + q);
						sum = dot2(v.x, kch[q], sum);
						sum = dot2(v.y, kch[q + 1], sum);
						sum = dot2(v.z, kch[q + 2], sum);
						sum = dot2(v.w, kch[q + 3], sum);
					}
				}
				else if (S % 4 == 0) {
#pragma unroll
					for (int q = 0; q < NP; q += 2) {
						const uint2 v = *reinterpret_cast<const uint2 *>(src + q);
						sum = dot2(v.x, kch[q], sum);
						sum = dot2(v.y, kch[q + 1], sum);
					}
				}
				else {
#pragma unroll
					for (int q = 0; q < NP; q++)
						sum = dot2(src[q], kch[q], sum);
				}
				rgba |= (unsigned) fin_u8(sum) << (8 * c);
			}
			const int jj = jlo + rr; // row of the (possibly flipped) tile
			unsigned int *dst = reinterpret_cast<unsigned int *>(
				a.out + (long long) (y0 + (flip ? oh - 1 - jj : jj)) * a.out_stride);
			dst[x0 + xo] = rgba;
		}
		__syncthreads();
	}
}

// ------------------------------------------------ both passes on the matrix cores
//
// reduce_fused_u8x4_mfma<D, NB> (S = 8): the same tiles and thread <-> column mapping as
// reduce_fused_u8x4, but every tap runs on the MFMA pipe with v_mfma_f32_4x4x4_16b_f16
// (16 independent 4x4x4 blocks, block = 4 lanes):
//
//   B[k][j]  lane j of the block supplies 4 halves = ITS OWN column's bytes of input rows
//            k = 0..3 (a quad of the 8-row group) -- lanes keep their columns, no shuffles
//   A[i][k]  lane i of the block supplies the coefficients of accumulator row i for
//            those 4 rows (the same in all 16 blocks)
//   D[i][j]  lane j, register i: 4 of the 8 live output rows of lane j's column
//
// so one instruction does 16 multiply-adds per lane (v_dot2: 2) on a pipe that the rest of
// the kernel leaves idle.  Exactness: a pixel byte p is used as the f16 DENORMAL with bit
// pattern 0x00pp = p * 2^-24 (one v_perm, no conversion; the MFMA honours f16 denormals,
// tools/mfma_probe.hip), coefficients (|c| < 2048) are exact halves, products are exact in
// f32 and every partial sum is (an integer below 2^23) * 2^-24, so the f32 accumulator holds
// exactly n * 2^-24 with n = sum c * p.  Retire: y = fma(acc, 2^12, 2^-13) = n / 4096 + 2^-13
// exactly, and v_cvt_pk_u8_f32 (round to nearest, saturate 0..255; the 2^-13 turns every
// tie into "up") gives clip((n + 2048) >> 12) -- reduceh.cpp:120-141's rounding -- and
// packs the byte, two instructions per sample.  The host checks the bounds and otherwise
// keeps the VALU kernel.
//
// 8 accumulator rows ("slots") per column rotate through the D <= 8 tap groups: at group g
// (ROT = g mod 8) slot s is d = (ROT - s) mod 8 groups old (d >= D: idle, zero coefficients).
// ROT is a template argument; the A operands come from a 1 KB LDS table indexed by it.
// The horizontal pass is the same computation along x on the u8 T planes in LDS.
//
// Output rows are staged in LDS for the whole tile and written in one burst at its end: on
// this part a 1.5 % stream of writes trickling into a streaming read costs 13 % of the
// read rate (tools/write_probe.hip: 0.175 -> 0.198 ms per GiB), a burst at the end 4 %.
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int MFMA_SLOTS = 8;

struct MfmaTables {
	// [flip][ROT 8][quad 2][half 2][row 4] x 4 halves
	unsigned short a[2][MFMA_SLOTS * 2 * 2 * 4 * 4];
	unsigned short ah[MFMA_SLOTS * 2 * 2 * 4 * 4]; // the same for the horizontal taps
};

constexpr int MFMA_TABLE_ENTRIES = MFMA_SLOTS * 2 * 2 * 4; // half4v entries per table
constexpr int HSEG_OUT = 8;                                // outputs per horizontal segment
// bytes per (row, channel) T plane: 512 samples + 4, so that the 32 planes a half-wave of the
// horizontal pass reads (8 rows x 4 channels, one dword each) fall in 32 distinct banks
// Geometry of a block of NTH threads (256: four blocks per CU, 59-pixel tiles; 512: two blocks
// per CU, 123-pixel tiles -- half the halo columns per input byte).  A lane owns two pixels.
template <int NTH>
struct MfmaGeo {
	static constexpr int SPAN = 2 * NTH;            // input columns per tile
	static constexpr int PLANE = SPAN + 4;          // odd number of dwords: see above
	static constexpr int PLANES_BYTES = MFMA_SLOTS * 4 * PLANE;
	static constexpr int STAGE_PITCH = SPAN / 8 - 4; // dwords per staged output row (owt <= SPAN/8 - 5)
	static constexpr int BLOCKS_PER_CU = 1024 / NTH;
	// (160 KB / blocks per CU) - planes - tables, in staged rows
	static constexpr int MAX_OHT = NTH == 256 ? 88 : 92;
	// stage_rows = rows the stage must hold: a burst leaves as soon as burst_rows are complete,
	// and a horizontal pass completes at most 8 more
	static constexpr size_t lds_bytes(int stage_rows)
	{
		return (size_t) PLANES_BYTES + 2 * MFMA_TABLE_ENTRIES * 8 + (size_t) stage_rows * STAGE_PITCH * 4;
	}
};

template <int D, bool NT = false, int PROF = 0, bool PAIRS = false, int NTH = FUSED_THREADS, int LATE = 0, int XPLANE = 0>
struct MfmaStep {
	static constexpr int S = 8;
	static constexpr int MFMA_PLANE = XPLANE ? XPLANE : MfmaGeo<NTH>::PLANE; // (XPLANE: the exchange kernel's wider planes)

	// Rows first_row + dir * i, I0 <= i < I0 + N.  The launcher only picks this kernel for
	// windows < 2 GB, so every address is the uniform base (an SGPR pair) plus one 32-bit lane
	// offset: the saddr form of global_load, no 64-bit VALU address arithmetic.  Interior
	// tiles fetch their two pixels as one dwordx2; edge tiles clamp each column.
	//
	// `interior` is the tile's (block-uniform) load mode:
	//   1  interior: one dwordx2 per row at column ca
	//   2  edge tile whose pixel pairs never straddle the clamp (even tile origin, even clamp
	//      bounds): one dwordx2 per row from the clamped PAIR at column ca, then lanes that lie
	//      wholly outside duplicate the edge pixel (cb = 1: y = x, left; cb = 2: x = y, right).
	//      The kernel runs ONE residency round, so it ends when its slowest tile ends: with
	//      two dword loads per lane and row the edge tiles were that tile.
	//   0  anything else: clamp each column, two dword loads
	template <int I0, int N>
	static __device__ __forceinline__ void load_rows(const FusedArgs &a, uint2 (&px)[S], int first_row,
		int dir, int ca, int cb, int interior)
	{
		const unsigned int stride32 = (unsigned int) a.in_stride;
		if (PAIRS || interior) { // a PAIRS kernel is only launched when every tile is mode 1 or 2
#pragma unroll
			for (int i = I0; i < I0 + N; i++) {
				const int row = min(max(first_row + dir * i, 0), a.im_height - 1) - a.in_top;
				const unsigned int off = (unsigned int) row * stride32 + (unsigned int) (4 * ca);
				typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
				const u32x2 *src = reinterpret_cast<const u32x2 *>(a.in + (size_t) off);
				const u32x2 v = NT ? __builtin_nontemporal_load(src) : *src;
				px[i] = make_uint2(v.x, v.y);
			}
			// LATE (the default): the fix-up is applied where the rows are CONSUMED (quad()), not here.
			// Here it made every wave wait for the loads it had just issued -- s_waitcnt vmcnt(3..0)
			// right behind the four global_loads -- so a wave's own matrix work never ran under
			// its own loads (round 3, read off the ISA: 0.1998 -> 0.1928 ms on one box).
			if (PAIRS && LATE == 0) { // branch-free: interior lanes carry cb = 0 (the compares live in SGPR masks)
#pragma unroll
				for (int i = I0; i < I0 + N; i++) {
					const unsigned int x = px[i].x, y = px[i].y;
					px[i].y = cb == 1 ? x : y;
					px[i].x = cb == 2 ? y : x;
				}
			}
		}
		else {
#pragma unroll
			for (int i = I0; i < I0 + N; i++) {
				const int row = min(max(first_row + dir * i, 0), a.im_height - 1) - a.in_top;
				const unsigned int base = (unsigned int) row * stride32;
				px[i].x = *reinterpret_cast<const unsigned int *>(a.in + (size_t) (base + (unsigned int) (4 * ca)));
				px[i].y = *reinterpret_cast<const unsigned int *>(a.in + (size_t) (base + (unsigned int) (4 * cb)));
			}
		}
	}

	// channel C of rows r0..r3 as four f16 denormals
	template <int C>
	static __device__ __forceinline__ half4v make_b(unsigned int r0, unsigned int r1, unsigned int r2,
		unsigned int r3)
	{
		constexpr unsigned int sel = 0x0c000c00u | (unsigned) C | ((4u + C) << 16);
		uint2 v;
		v.x = __builtin_amdgcn_perm(r1, r0, sel);
		v.y = __builtin_amdgcn_perm(r3, r2, sel);
		return __builtin_bit_cast(half4v, v);
	}

	// four consecutive T bytes as four f16 denormals
	static __device__ __forceinline__ half4v bytes_b(unsigned int w)
	{
		uint2 v;
		v.x = __builtin_amdgcn_perm(0u, w, 0x0c010c00u);
		v.y = __builtin_amdgcn_perm(0u, w, 0x0c030c02u);
		return __builtin_bit_cast(half4v, v);
	}

	// acc = n * 2^-24 -> clip((n + 2048) >> 12) into byte `byte` of `old`
	static __device__ __forceinline__ unsigned int fin_pack(float acc, unsigned int byte, unsigned int old)
	{
		return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(acc, 4096.0f, 0x1p-13f), byte, old);
	}

	// One quad (rows 4*Q .. 4*Q+3 of the group) of both pixels; once its B operands exist the
	// quad's buffer registers are refilled with the rows of group g + NB.
	template <int ROT, int Q>
	static __device__ __forceinline__ void quad(const FusedArgs &a, uint2 (&px)[S], float4v (&acc)[8][2],
		const half4v *lane_a /* &table[lane & 3] */, bool more, int next_row, int dir, int ca, int cb,
		int interior)
	{
		if constexpr (PROF == 16) { // profiling: loads only -- consume the rows, refill, no arithmetic
#pragma unroll
			for (int i = 4 * Q; i < 4 * Q + 4; i++)
				VH_USE2(px[i].x, px[i].y);
			if (more)
				load_rows<4 * Q, 4>(a, px, next_row, dir, ca, cb, interior);
			return;
		}
		const half4v a0 = lane_a[((ROT * 2 + Q) * 2 + 0) * 4];
		const half4v a1 = lane_a[((ROT * 2 + Q) * 2 + 1) * 4];
		uint2 row[4];
#pragma unroll
		for (int i = 0; i < 4; i++) {
			row[i] = px[4 * Q + i];
			if (PAIRS && LATE != 0) { // the edge fix-up of load_rows, at the point of use
				const unsigned int x = row[i].x, y = row[i].y;
				row[i].y = cb == 1 ? x : y;
				row[i].x = cb == 2 ? y : x;
			}
		}
#pragma unroll
		for (int p = 0; p < 2; p++) {
			const unsigned int r0 = p ? row[0].y : row[0].x;
			const unsigned int r1 = p ? row[1].y : row[1].x;
			const unsigned int r2 = p ? row[2].y : row[2].x;
			const unsigned int r3 = p ? row[3].y : row[3].x;
			half4v b[4];
			b[0] = make_b<0>(r0, r1, r2, r3);
			b[1] = make_b<1>(r0, r1, r2, r3);
			b[2] = make_b<2>(r0, r1, r2, r3);
			b[3] = make_b<3>(r0, r1, r2, r3);
			if constexpr (PROF == 8) { // profiling: arithmetic only (rows stay, but opaque to the compiler)
				if (p == 1) {
#pragma unroll
					for (int i = 4 * Q; i < 4 * Q + 4; i++)
						VH_VECTOR2(px[i].x, px[i].y);
				}
			}
			else if (p == 1 && more)
				load_rows<4 * Q, 4>(a, px, next_row, dir, ca, cb, interior);
#pragma unroll
			for (int c = 0; c < 4; c++) {
				acc[p * 4 + c][0] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, b[c], acc[p * 4 + c][0], 0, 0, 0);
				acc[p * 4 + c][1] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, b[c], acc[p * 4 + c][1], 0, 0, 0);
			}
		}
	}

	// Slot (ROT - (D - 1)) mod 8 has seen all its taps: round it into T row `lds_row`.
	template <int ROT>
	static __device__ __forceinline__ void retire(float4v (&acc)[8][2], unsigned char *planes, int lds_row,
		int t, bool store)
	{
		constexpr int SLOT = (ROT - (D - 1) + 2 * MFMA_SLOTS) % MFMA_SLOTS;
		constexpr int H = SLOT >> 2, I = SLOT & 3;
		if (store) {
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const unsigned int v = fin_pack(acc[4 + c][H][I], 1, fin_pack(acc[c][H][I], 0, 0));
				*reinterpret_cast<unsigned short *>(planes + (lds_row * 4 + c) * MFMA_PLANE + 2 * t) =
					(unsigned short) v;
			}
		}
#pragma unroll
		for (int o = 0; o < 8; o++)
			acc[o][H][I] = 0.0f;
	}

	// The horizontal pass is the same computation along x: a lane owns one (row, channel)
	// line segment of the T planes and walks it in groups of 8 samples; sample group G
	// is d = (G - s) mod 8 groups into output xo = first + s, one output retires per group.
	// The four channel lanes of a quad OR their bytes together (two quad_perm DPP moves)
	// and lane O / 2 keeps the RGBA pixel of output O.
	template <int G>
	static __device__ __forceinline__ void hwalk(float4v (&hacc)[2], const unsigned char *line,
		const half4v *lane_ah, int hc, unsigned int (&pix)[2])
	{
		constexpr int NG = HSEG_OUT + D - 1;
		if constexpr (G < NG) {
			constexpr int ROT = G % MFMA_SLOTS;
			const half4v b0 = bytes_b(*reinterpret_cast<const unsigned int *>(line + 8 * G));
			const half4v b1 = bytes_b(*reinterpret_cast<const unsigned int *>(line + 8 * G + 4));
			hacc[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 0) * 2 + 0) * 4], b0, hacc[0], 0, 0, 0);
			hacc[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 0) * 2 + 1) * 4], b0, hacc[1], 0, 0, 0);
			hacc[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 1) * 2 + 0) * 4], b1, hacc[0], 0, 0, 0);
			hacc[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 1) * 2 + 1) * 4], b1, hacc[1], 0, 0, 0);
			constexpr int SLOT = (ROT - (D - 1) + 2 * MFMA_SLOTS) % MFMA_SLOTS;
			constexpr int H = SLOT >> 2, I = SLOT & 3;
			if constexpr (G >= D - 1) {
				constexpr int O = G - (D - 1);
				int v = (int) fin_pack(hacc[H][I], (unsigned int) hc, 0);
				v |= __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
				v |= __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); // quad_perm [2,3,0,1]
				if (hc == O / 2)
					pix[O & 1] = (unsigned int) v;
			}
			hacc[H][I] = 0.0f;
			if constexpr ((G & 1) == 1)
				__builtin_amdgcn_sched_barrier(0); // keep the unrolled walk's LDS reads from piling up
			hwalk<G + 1>(hacc, line, lane_ah, hc, pix);
		}
	}

	// hwalk that also hands out the UNROUNDED sums of the segment's outputs 0, 1, 2 (raw[0..2]) and 5, 6, 7
	// (raw[3..5]): the exchange kernel's partial sums of the outputs that straddle a tile boundary
	template <int G>
	static __device__ __forceinline__ void hwalk_x(float4v (&hacc)[2], const unsigned char *line, const half4v *lane_ah,
		int hc, unsigned int (&pix)[2], float (&raw)[6])
	{
		constexpr int NG = HSEG_OUT + D - 1;
		if constexpr (G < NG) {
			constexpr int ROT = G % MFMA_SLOTS;
			const half4v b0 = bytes_b(*reinterpret_cast<const unsigned int *>(line + 8 * G));
			const half4v b1 = bytes_b(*reinterpret_cast<const unsigned int *>(line + 8 * G + 4));
			hacc[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 0) * 2 + 0) * 4], b0, hacc[0], 0, 0, 0);
			hacc[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 0) * 2 + 1) * 4], b0, hacc[1], 0, 0, 0);
			hacc[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 1) * 2 + 0) * 4], b1, hacc[0], 0, 0, 0);
			hacc[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 1) * 2 + 1) * 4], b1, hacc[1], 0, 0, 0);
			constexpr int SLOT = (ROT - (D - 1) + 2 * MFMA_SLOTS) % MFMA_SLOTS;
			constexpr int H = SLOT >> 2, I = SLOT & 3;
			if constexpr (G >= D - 1) {
				constexpr int O = G - (D - 1);
				if constexpr (O < 3)
					raw[O] = hacc[H][I];
				if constexpr (O >= 5)
					raw[O - 2] = hacc[H][I];
				int v = (int) fin_pack(hacc[H][I], (unsigned int) hc, 0);
				v |= __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
				v |= __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); // quad_perm [2,3,0,1]
				if (hc == O / 2)
					pix[O & 1] = (unsigned int) v;
			}
			hacc[H][I] = 0.0f;
			if constexpr ((G & 1) == 1)
				__builtin_amdgcn_sched_barrier(0);
			hwalk_x<G + 1>(hacc, line, lane_ah, hc, pix, raw);
		}
	}

	// NB = prefetch depth: group g lives in ring buffer g mod NB (NB divides 8, so the index is
	// static) and each of its quads is refilled with group g + NB as soon as it has been consumed.
	template <int ROT, int NB>
	static __device__ __forceinline__ void batch(const FusedArgs &a, uint2 (&px)[NB][S], int g0, int ngroups,
		float4v (&acc)[8][2], unsigned char *planes, const half4v *lane_a, int t, int row0, int dir, int ca,
		int cb, int interior, int oh)
	{
		if constexpr (ROT < MFMA_SLOTS) {
			const int g = g0 + ROT;
			if (g >= 0 && g < ngroups) {
				// (the branches around a group and around its refill stay: without them -- every
				// group refilling, the last one with its own rows again -- the compiler's waits
				// become exact, eight rows stay in flight per wave, and the kernel is SLOWER: 0.1977
				// against 0.1932 ms; branch-free over whole batches with padded tiles: 0.234.  Fewer
				// requests in flight is what this part's memory system wants, §3.1 of DESIGN.md)
				const bool more = g + NB < ngroups;
				const int next_row = row0 + dir * S * (g + NB);
				quad<ROT, 0>(a, px[ROT % NB], acc, lane_a, more, next_row, dir, ca, cb, interior);
				quad<ROT, 1>(a, px[ROT % NB], acc, lane_a, more, next_row, dir, ca, cb, interior);
				const int j = g - (D - 1);
				retire<ROT>(acc, planes, ROT, t, j >= 0 && j < oh);
			}
			batch<ROT + 1, NB>(a, px, g0, ngroups, acc, planes, lane_a, t, row0, dir, ca, cb, interior, oh);
		}
	}
};

// Tiles are numbered row-major and XCD k (blocks b = k mod 8) takes a contiguous range of
// them, so horizontal neighbours (which read their shared halo columns in lock-step) and
// most vertical neighbours (which the serpentine walk makes meet at their shared halo rows)
// share an L2.  Measured on C2: row-major 0.218 ms, column-major 0.221, no serpentine 0.225.
template <int D, int NB, int OCC, bool NT, int PROF = 0, bool PAIRS = false, int NTH = FUSED_THREADS, int LATE = 0>
__global__ void __launch_bounds__(NTH, OCC)
reduce_fused_u8x4_mfma(FusedArgs a, const MfmaTables *__restrict__ tables)
{
	constexpr int S = 8;
	typedef MfmaStep<D, NT, PROF, PAIRS, NTH, LATE> Step;
	typedef MfmaGeo<NTH> Geo;
	constexpr int MFMA_PLANE = Geo::PLANE, MFMA_PLANES_BYTES = Geo::PLANES_BYTES;
	constexpr int MFMA_STAGE_PITCH = Geo::STAGE_PITCH, FUSED_SPAN = Geo::SPAN;
	VH_DYNAMIC_LDS(unsigned char, lds_raw);
	// T planes (the horizontal walker over-reads the end of a plane by up to 8 * (D - 1)
	// samples: into the next plane / the tables -- any byte is a finite f16 denormal), the two
	// A-operand tables, the staged output rows of the tile
	unsigned char *planes = lds_raw;
	half4v *lds_a = reinterpret_cast<half4v *>(lds_raw + MFMA_PLANES_BYTES);
	half4v *lds_ah = lds_a + MFMA_TABLE_ENTRIES;
	unsigned int *stage = reinterpret_cast<unsigned int *>(lds_ah + MFMA_TABLE_ENTRIES);

	const int per_xcd = gridDim.x / 8;
	const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (tile >= a.tiles)
		return;

	const int t = threadIdx.x;
	const int by = tile / a.tiles_x;
	const int bx = tile - by * a.tiles_x;
	const int x0 = bx * a.owt;
	const int y0 = by * a.oht;
	const int ow = min(a.owt, a.out_width - x0);
	const int oh = min(a.oht, a.out_height - y0);

	const int tile_col0 = a.fx0 + S * x0 - a.xshift;
	const int col0 = tile_col0 + 2 * t;
	// Columns clamp to the image (vips_embed COPY) -- and to the window: the window holds every
	// column an output needs (checked by the host), so this only matters to lanes past the
	// tile's last tap, whose reads must stay inside the window too.
	const int lo = max(0, a.in_left), hi = min(a.im_width, a.in_right) - 1;
	const bool even = a.aligned8 && (((tile_col0 - a.in_left) & 1) == 0);
	int interior = even && tile_col0 >= lo && tile_col0 + FUSED_SPAN <= hi + 1 ? 1 : 0;
	int ca = min(max(col0, lo), hi) - a.in_left;
	int cb = min(max(col0 + 1, lo), hi) - a.in_left;
	if (PAIRS) {
		// see load_rows: the pair is fetched from columns clamped to [lo, hi - 1]; a lane whose
		// first column is left of lo needs pixel lo twice (y = x), one whose second column is
		// right of hi needs pixel hi twice (x = y); dwordx2 loads only need dword alignment
		ca = min(max(col0, lo), hi - 1) - a.in_left;
		cb = col0 < lo ? 1 : (col0 + 1 > hi ? 2 : 0);
	}
	const bool flip = (by & 1) != 0;
	const int dir = flip ? -1 : 1;
	const int row0 = flip ? a.fy0 + S * (y0 + oh - 1) + S * D - 1 : a.fy0 + S * y0;

	// the A-operand tables (vertical: this tile's walking direction), 128 entries of 4 halves
	if (t < MFMA_TABLE_ENTRIES) {
		reinterpret_cast<uint2 *>(lds_a)[t] = reinterpret_cast<const uint2 *>(tables->a[flip ? 1 : 0])[t];
		reinterpret_cast<uint2 *>(lds_ah)[t] = reinterpret_cast<const uint2 *>(tables->ah)[t];
	}
	const half4v *lane_a = lds_a + (t & 3);

	float4v acc[8][2];
#pragma unroll
	for (int o = 0; o < 8; o++)
#pragma unroll
		for (int h = 0; h < 2; h++)
			acc[o][h] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };

	const int ngroups = oh + D - 1;
	uint2 px[NB][S];
#pragma unroll
	for (int b = 0; b < NB; b++)
		if (b < ngroups)
			Step::template load_rows<0, S>(a, px[b], row0 + dir * S * b, dir, ca, cb, interior);
	__syncthreads();

	// The tile's groups are numbered from `off` instead of 0 (ROT = (g + off) mod 8): the blocks
	// sharing a CU get different offsets, so their horizontal passes -- which issue no loads --
	// and their output bursts fall at different times instead of all at once (every tile of the
	// single residency round starts at the same moment and advances at the same rate).
	const int off = a.stagger ? (((int) blockIdx.x / 256) * a.stagger) & 7 : 0;
	int flushed = 0; // rows of the tile already written out
	for (int v0 = 0; v0 < ngroups + off; v0 += MFMA_SLOTS) {
		const int g0 = v0 - off;
		Step::template batch<0, NB>(a, px, g0, ngroups, acc, planes, lane_a, t, row0, dir, ca, cb, interior,
			oh);

		// ---- horizontal pass over the rows this batch completed (T row r <-> group g0 + r)
		const int jlo = max(g0 - (D - 1), 0);
		const int jhi = min(g0 + MFMA_SLOTS - 1 - (D - 1), oh - 1); // inclusive
		if (jhi < jlo)
			continue;
		__syncthreads();
		const int nrows = jhi - jlo + 1;
		const int r_lo = jlo - (g0 - (D - 1));
		if (!(a.debug & 1)) {
			// thread -> (T row, segment of HSEG_OUT outputs, channel)
			const int hc = t & 3, hr = (t >> 2) & 7, hseg = t >> 5;
			const half4v *lane_ah = lds_ah + hc;
			const bool row_ok = hr < nrows;
			const int lrow = r_lo + (row_ok ? hr : 0);
			const unsigned char *line = planes + (lrow * 4 + hc) * MFMA_PLANE + 8 * HSEG_OUT * hseg + a.xshift;
			float4v hacc[2];
			hacc[0] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };
			hacc[1] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };
			unsigned int pix[2] = { 0, 0 };
			Step::template hwalk<0>(hacc, line, lane_ah, hc, pix);
			// lane hc of the quad holds output pixels 2*hc, 2*hc + 1 of the segment
			const int xo = HSEG_OUT * hseg + 2 * hc;
			if (row_ok && xo < MFMA_STAGE_PITCH) {
				const int jj = jlo + hr;
				unsigned int *srow = stage + (jj - flushed) * MFMA_STAGE_PITCH + xo;
				*reinterpret_cast<uint2 *>(srow) = make_uint2(pix[0], pix[1]);
			}
		}
		__syncthreads();

		// ---- output: staged rows leave in bursts of burst_rows (and at the tile's end), a wave
		// per row, a lane per pixel; the next write into the stage is behind the next barrier
		const int done = jhi + 1;
		if ((done - flushed >= a.burst_rows || done == oh) && !(a.debug & 2)) {
			// 16 lanes per row, 4 pixels (one dwordx4 store, dword aligned) per lane: the burst is
			// the kernel's tail, so it wants few, wide store instructions
			constexpr int LPR = NTH / 16; // lanes per row: 4 pixels each
			const int part = t & (LPR - 1);
			for (int r = t / LPR; r < done - flushed; r += 16) {
				const int jj = flushed + r;
				unsigned int *dst = reinterpret_cast<unsigned int *>(
					a.out + (long long) (y0 + (flip ? oh - 1 - jj : jj)) * a.out_stride + (long long) x0 * 4);
				const unsigned int *src = stage + r * MFMA_STAGE_PITCH;
				if (4 * part + 4 <= ow) {
					typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
					typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
					*reinterpret_cast<u32x4_a4 *>(dst + 4 * part) = *reinterpret_cast<const u32x4 *>(src + 4 * part);
				}
				else {
					for (int x = 4 * part; x < ow; x++)
						dst[x] = src[x];
				}
			}
			flushed = done;
		}
		else if (done - flushed >= a.burst_rows || done == oh)
			flushed = done;
	}
}


// ------------------------------------------------ ... without the tiles' horizontal halo (round 6)
//
// reduce_fused_u8x4_mfma reads 512 input columns to make 56 outputs' worth (448): one eighth of its requests are
// columns the tile on either side reads too, and on this part the requests a CU issues bound the stream
// (profiles/NOTES.md 3.1: 2 KB strips without halos stream at 7.1 TB/s, the 59-pixel tiles at 5.8).  Here a tile
// IS 512 aligned columns -- a wave's row segment is four whole 128-byte lines, no column is requested twice -- and
// makes all 64 of its outputs; the six outputs whose 48 taps straddle a tile boundary (three either side) are made
// as PARTIAL SUMS by both tiles, each over its own columns (the T planes carry 48 bytes of zeros either side; at
// the image's edges the edge column, replicated: vips_embed COPY), and a second, tiny kernel adds the two halves
// and rounds them (exact: both are integers below 2^23 in units of 2^-24).  Partial sums and output rows wait in
// LDS and leave in one burst at the tile's end (a trickle of writes would cost the read stream a tenth of its
// rate: tools/write_probe).  Two blocks a CU (78 KB of LDS each), NB row groups in flight per lane, tiles of 128
// output rows: 32 x 16 tiles for BASELINE config 2 = one residency round; requests 1.04 x the image (the
// vertical halo) against 1.22 x.
constexpr int XH = 48;                 // halo bytes either side of a T plane row
constexpr int XPLANE = XH + 512 + XH + 4; // 612 bytes = 153 dwords (odd: 32 planes in 32 banks)
constexpr int XGUARD = 64;             // in front of the planes: segment -1 of plane 0 reads 40 bytes before its row
constexpr int XPLANES_BYTES = XGUARD + MFMA_SLOTS * 4 * XPLANE;
constexpr int XPART = 48;              // floats per row: [side 2][straddling output 6][channel 4]
constexpr int XMAX_OHT = 128;
static constexpr size_t xlds_bytes(int oht)
{
	return (size_t) XPLANES_BYTES + 2 * MFMA_TABLE_ENTRIES * 8 + (size_t) oht * 64 * 4 + (size_t) oht * XPART * 4;
}

template <int D, int NB, int OCC>
__global__ void __launch_bounds__(FUSED_THREADS, OCC)
reduce_fused_u8x4_mfma_x(FusedArgs a, const MfmaTables *__restrict__ tables, float *parts, int *arrivals, int plain,
	int *misplaced)
{
	constexpr int S = 8;
	typedef MfmaStep<D, true, 0, true, FUSED_THREADS, 1, XPLANE> Step;
	VH_DYNAMIC_LDS(unsigned char, lds_raw);
	unsigned char *planes = lds_raw + XGUARD + XH; // byte p of a plane row = the tile's own column p
	half4v *lds_a = reinterpret_cast<half4v *>(lds_raw + XPLANES_BYTES);
	half4v *lds_ah = lds_a + MFMA_TABLE_ENTRIES;
	unsigned int *stage = reinterpret_cast<unsigned int *>(lds_ah + MFMA_TABLE_ENTRIES); // oht rows of 64 pixels
	float *part = reinterpret_cast<float *>(stage + a.oht * 64);                           // oht rows of XPART floats

	// XCD k (blocks b = k mod 8) takes WHOLE rows of tiles, a contiguous run of them: horizontal neighbours -- which
	// hand partial sums to each other -- and most vertical ones share an L2
	const int per_xcd = gridDim.x / 8; // = rows of tiles per XCD x tiles_x (host)
	const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (tile >= a.tiles)
		return;
	const int t = threadIdx.x;
	const int by = tile / a.tiles_x;
	const int bx = tile - by * a.tiles_x;
	const int y0 = by * a.oht;
	const int oh = min(a.oht, a.out_height - y0);
	const int ca = 512 * bx + 2 * t - a.in_left;
	const bool flip = (by & 1) != 0;
	const int dir = flip ? -1 : 1;
	const int row0 = flip ? a.fy0 + S * (y0 + oh - 1) + S * D - 1 : a.fy0 + S * y0;

	// zeros in the planes (the halos stay zero: a group only ever writes its own 512 columns), the tables
	for (int i = t; i < XPLANES_BYTES / 4; i += FUSED_THREADS)
		reinterpret_cast<unsigned int *>(lds_raw)[i] = 0u;
	if (t < MFMA_TABLE_ENTRIES) {
		reinterpret_cast<uint2 *>(lds_a)[t] = reinterpret_cast<const uint2 *>(tables->a[flip ? 1 : 0])[t];
		reinterpret_cast<uint2 *>(lds_ah)[t] = reinterpret_cast<const uint2 *>(tables->ah)[t];
	}
	const half4v *lane_a = lds_a + (t & 3);

	float4v acc[8][2];
#pragma unroll
	for (int o = 0; o < 8; o++)
#pragma unroll
		for (int h = 0; h < 2; h++)
			acc[o][h] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };

	const int ngroups = oh + D - 1;
	uint2 px[NB][S];
#pragma unroll
	for (int b = 0; b < NB; b++)
		if (b < ngroups)
			Step::template load_rows<0, S>(a, px[b], row0 + dir * S * b, dir, ca, 0, 1);
	__syncthreads();

	const bool left_edge = bx == 0, right_edge = bx == a.tiles_x - 1;
	for (int g0 = 0; g0 < ngroups; g0 += MFMA_SLOTS) {
		Step::template batch<0, NB>(a, px, g0, ngroups, acc, planes, lane_a, t, row0, dir, ca, 0, 1, oh);

		// ---- horizontal pass over the rows this batch completed (T row r <-> group g0 + r)
		const int jlo = max(g0 - (D - 1), 0);
		const int jhi = min(g0 + MFMA_SLOTS - 1 - (D - 1), oh - 1); // inclusive
		if (jhi < jlo)
			continue;
		__syncthreads();
		if (left_edge || right_edge) {
			// vips_embed(COPY): the columns beyond the image are its edge column, in every plane row
			for (int i = t; i < MFMA_SLOTS * 4 * (XH / 4); i += FUSED_THREADS) {
				const int rowc = i / (XH / 4), d = i - rowc * (XH / 4);
				unsigned char *prow = planes + rowc * XPLANE;
				if (left_edge)
					*reinterpret_cast<unsigned int *>(prow - XH + 4 * d) = (unsigned int) prow[0] * 0x01010101u;
				if (right_edge)
					*reinterpret_cast<unsigned int *>(prow + 512 + 4 * d) = (unsigned int) prow[511] * 0x01010101u;
			}
			__syncthreads();
		}
		const int nrows = jhi - jlo + 1;
		const int r_lo = jlo - (g0 - (D - 1));
		// thread -> (T row, segment of HSEG_OUT outputs, channel); then the first wave again for the two
		// segments outside the tile (outputs -8 .. -1 and 64 .. 71: the neighbours' straddling outputs)
#pragma unroll 1
		for (int round = 0; round < 2; round++) {
			if (round == 1 && t >= 64)
				break;
			const int hc = t & 3, hr = (t >> 2) & 7;
			const int hseg = round == 0 ? t >> 5 : ((t >> 5) & 1 ? 8 : -1);
			const half4v *lane_ah = lds_ah + hc;
			const bool row_ok = hr < nrows;
			const int lrow = r_lo + (row_ok ? hr : 0);
			const unsigned char *line = planes + (lrow * 4 + hc) * XPLANE + a.fx0 + 8 * HSEG_OUT * hseg;
			float4v hacc[2];
			hacc[0] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };
			hacc[1] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };
			unsigned int pix[2] = { 0, 0 };
			float raw[6];
			Step::template hwalk_x<0>(hacc, line, lane_ah, hc, pix, raw);
			if (!row_ok)
				continue;
			const int jj = jlo + hr;
			if (round == 0)
				*reinterpret_cast<uint2 *>(stage + jj * 64 + HSEG_OUT * hseg + 2 * hc) = make_uint2(pix[0], pix[1]);
			// straddling outputs, side 0: local outputs -3 .. 2, side 1: 61 .. 66
			float *prow = part + jj * XPART + hc;
			if (hseg == 0) {
#pragma unroll
				for (int o = 0; o < 3; o++)
					prow[(3 + o) * 4] = raw[o];
			}
			else if (hseg == 7) {
#pragma unroll
				for (int o = 0; o < 3; o++)
					prow[24 + o * 4] = raw[3 + o];
			}
			else if (hseg == -1) {
#pragma unroll
				for (int o = 0; o < 3; o++)
					prow[o * 4] = raw[3 + o];
			}
			else if (hseg == 8) {
#pragma unroll
				for (int o = 0; o < 3; o++)
					prow[24 + (3 + o) * 4] = raw[o];
			}
		}
		__syncthreads();
	}

	// ---- the tile's end.  Its partial sums leave first (write-through); then it ARRIVES at its two boundaries (an
	// atomic counter each: two arrivals a launch, so the parity of what the atomic returns says who is second, launch
	// after launch without a reset).  The LATER tile of a boundary reads the earlier one's halves -- published before
	// that tile arrived -- adds its own, rounds (MfmaStep::fin_pack: exact integers below 2^23 in units of 2^-24) and
	// writes all six straddling pixels: its own three through the stage, the neighbour's three straight to the
	// image.  The earlier tile leaves those three alone.  Nobody waits for anybody.  At the image's edges the tile
	// holds the whole sums (the replicated edge column) and is "second" by itself.
	if (a.debug & 32) {
		// ($VIPS_HIP_FUSED_DEBUG=32: the two-kernel form -- partial sums by output row, reduce_fused_edges adds them)
		const int part16 = t & 15;
		for (int r = t >> 4; r < oh; r += 16) {
			unsigned int *dst = reinterpret_cast<unsigned int *>(
				a.out + (long long) (y0 + (flip ? oh - 1 - r : r)) * a.out_stride + (long long) (64 * bx) * 4);
			typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
			*reinterpret_cast<u32x4 *>(dst + 4 * part16) = *reinterpret_cast<const u32x4 *>(stage + r * 64 + 4 * part16);
		}
		float *dstp = parts + (size_t) tile * a.oht * XPART;
		for (int i = t; i < oh * (XPART / 4); i += FUSED_THREADS) {
			const int r = i / (XPART / 4), q = i - r * (XPART / 4);
			const int yrel = flip ? oh - 1 - r : r;
			reinterpret_cast<float4 *>(dstp + (size_t) yrel * XPART)[q] = reinterpret_cast<const float4 *>(part + r * XPART)[q];
		}
		return;
	}
	{
		typedef float f32x4 __attribute__((ext_vector_type(4)));
		const bool placed = VH_XCC_ID() == (int) (blockIdx.x & 7u);
		const bool use_plain = plain && placed;
		if (plain && !placed && t == 0)
			VH_STORE_SYS(misplaced, 1);
		float *dstp = parts + (size_t) tile * a.oht * XPART; // (rows in WALK order: the tiles of a row of tiles share it)
		for (int i = t; i < oh * (XPART / 4); i += FUSED_THREADS) {
			f32x4 *dst = reinterpret_cast<f32x4 *>(dstp) + i;
			const f32x4 v = reinterpret_cast<const f32x4 *>(part)[i];
			// The two tiles of a boundary are neighbours in one row of tiles, and a row of tiles belongs to ONE XCD
			// (the tile numbering above, with block b on XCD b % 8: what the part does, checked by a census launch
			// before `plain` is ever set -- xcc_census_kernel -- and by every block for itself here): their hand-off
			// can stay in that XCD's L2 -- plain stores, read by the other tile with loads that skip ITS L1 --
			// instead of going through to memory (12.6 MB of write-through at the kernel's tail: 0.1914 against
			// 0.1867 ms a launch).  A block that finds itself elsewhere writes through and says so (*misplaced, host
			// memory: the host stops using the plain form and reports it -- never seen).
			if (use_plain)
				*dst = v;
			else
				VH_STORE4_SYS(dst, v);
		}
		VH_WAIT_VMCNT(0);
		__syncthreads();
		int *second = reinterpret_cast<int *>(lds_raw); // (the planes are done with)
		if (t < 2) { // (two lanes: the two atomics travel together)
			int *at = arrivals + by * (a.tiles_x + 1) + bx + t;
			const bool edge = t == 0 ? left_edge : right_edge;
			second[t] = edge ? 1 : (atomicAdd(at, 1) & 1);
		}
		__syncthreads();
		const int sec[2] = { second[0], second[1] };
#pragma unroll 1
		for (int side = 0; side < 2; side++) {
			if (!sec[side])
				continue;
			const bool at_edge = side == 0 ? left_edge : right_edge;
			const float *theirs = parts + (size_t) (tile + (side ? 1 : -1)) * a.oht * XPART + (1 - side) * 24;
			for (int i = t; i < oh * 6; i += FUSED_THREADS) {
				const int r = i / 6, o = i - r * 6;
				const float *mine = part + r * XPART + side * 24 + o * 4;
				float v[4] = { mine[0], mine[1], mine[2], mine[3] };
				if (!at_edge) {
					const float *p = theirs + (size_t) r * XPART + o * 4;
#pragma unroll
					for (int c = 0; c < 4; c++)
						v[c] += VH_LOAD_SYS(p + c);
				}
				unsigned int pxl = Step::fin_pack(v[0], 0, 0);
				pxl = Step::fin_pack(v[1], 1, pxl);
				pxl = Step::fin_pack(v[2], 2, pxl);
				pxl = Step::fin_pack(v[3], 3, pxl);
				const int xl = (side ? 61 : -3) + o; // local output
				if (xl >= 0 && xl < 64)
					stage[r * 64 + xl] = pxl;
				else if (!at_edge)
					*reinterpret_cast<unsigned int *>(a.out + (long long) (y0 + (flip ? oh - 1 - r : r)) * a.out_stride +
						(long long) (64 * bx + xl) * 4) = pxl;
			}
		}
		__syncthreads();
		// the burst: 16 lanes a row, 4 pixels each; the first and the last lane's straddling pixels only if this tile
		// made them
		const int part16 = t & 15;
		for (int r = t >> 4; r < oh; r += 16) {
			unsigned int *dst = reinterpret_cast<unsigned int *>(
				a.out + (long long) (y0 + (flip ? oh - 1 - r : r)) * a.out_stride + (long long) (64 * bx) * 4);
			const unsigned int *src = stage + r * 64 + 4 * part16;
			if ((part16 == 0 && !sec[0]) || (part16 == 15 && !sec[1])) {
				if (part16 == 0)
					dst[3] = src[3];
				else
					dst[60] = src[0];
			}
			else {
				typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
				*reinterpret_cast<u32x4 *>(dst + 4 * part16) = *reinterpret_cast<const u32x4 *>(src);
			}
		}
	}
}

// The straddling outputs: boundary k (0 .. tiles_x: 0 and tiles_x are the image's edges, where one tile holds the
// whole sum) x output row x the six outputs, one RGBA pixel a thread: the two tiles' halves added (exact) and
// rounded as every other output is (fin_pack).
__global__ void __launch_bounds__(256)
reduce_fused_edges(FusedArgs a, const float *__restrict__ parts)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const int o = i % 6, k = (i / 6) % (a.tiles_x + 1), y = i / (6 * (a.tiles_x + 1));
	if (y >= a.out_height)
		return;
	const int by = y / a.oht, yrel = y - by * a.oht;
	const int xo = 64 * k - 3 + o;
	if (xo < 0 || xo >= a.out_width)
		return;
	float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	if (k > 0) { // the tile on the left: its side 1
		const float *p = parts + ((size_t) (by * a.tiles_x + k - 1) * a.oht + yrel) * XPART + 24 + o * 4;
		const float4 v = *reinterpret_cast<const float4 *>(p);
		sum = v;
	}
	if (k < a.tiles_x) { // the tile on the right: its side 0
		const float *p = parts + ((size_t) (by * a.tiles_x + k) * a.oht + yrel) * XPART + o * 4;
		const float4 v = *reinterpret_cast<const float4 *>(p);
		sum.x += v.x;
		sum.y += v.y;
		sum.z += v.z;
		sum.w += v.w;
	}
	typedef MfmaStep<6> Step;
	unsigned int px = Step::fin_pack(sum.x, 0, 0);
	px = Step::fin_pack(sum.y, 1, px);
	px = Step::fin_pack(sum.z, 2, px);
	px = Step::fin_pack(sum.w, 3, px);
	*reinterpret_cast<unsigned int *>(a.out + (long long) y * a.out_stride + (long long) xo * 4) = px;
}

// ------------------------------------------------ vertical-only pass on the matrix cores
//
// reducev_u8_mfma<D>: vips_reducev by an integer 8 with one coefficient phase on a uchar image
// of ANY band count.  A scanline is a byte array to a vertical filter, so this is the fused
// kernel's vertical pass alone: a lane owns 8 consecutive bytes of the row, walks down the
// rows in groups of 8 with the same rotating MFMA accumulators, and each group retires one
// output row straight to memory (8 bytes per lane, a wave writes 512 contiguous bytes).  Every
// input byte is read once (the row-pair dot2 kernel re-reads each row n / 8 times through L2).
struct VStreamArgs {
	const unsigned char *in;  // first byte of the columns of the rect, row in_top of the image
	unsigned char *out;
	long long in_stride, out_stride;
	int in_top, im_height;
	int out_height;          // rows of the rect
	int fy0;                 // first tap (input row) of output row 0 of the rect
	int row_u2;              // 8-byte columns per row
	int oht, tiles_x, tiles; // tile = 256 columns x oht output rows
	int alternate;           // every other row of tiles is walked bottom-up
};

template <int D>
struct VStreamStep {
	typedef MfmaStep<D> Base;
	static constexpr int S = 8;

	template <int I0, int N>
	static __device__ __forceinline__ void load_rows(const VStreamArgs &a, uint2 (&px)[S], int first_row, int dir,
		unsigned int coff)
	{
		const unsigned int stride32 = (unsigned int) a.in_stride;
#pragma unroll
		for (int i = I0; i < I0 + N; i++) {
			const int row = min(max(first_row + dir * i, 0), a.im_height - 1) - a.in_top;
			px[i] = *reinterpret_cast<const uint2 *>(a.in + (size_t) ((unsigned int) row * stride32 + coff));
		}
	}

	template <int ROT, int Q>
	static __device__ __forceinline__ void quad(const VStreamArgs &a, uint2 (&px)[S], float4v (&acc)[8][2],
		const half4v *lane_a, bool more, int next_row, int dir, unsigned int coff)
	{
		const half4v a0 = lane_a[((ROT * 2 + Q) * 2 + 0) * 4];
		const half4v a1 = lane_a[((ROT * 2 + Q) * 2 + 1) * 4];
#pragma unroll
		for (int p = 0; p < 2; p++) {
			const unsigned int r0 = p ? px[4 * Q + 0].y : px[4 * Q + 0].x;
			const unsigned int r1 = p ? px[4 * Q + 1].y : px[4 * Q + 1].x;
			const unsigned int r2 = p ? px[4 * Q + 2].y : px[4 * Q + 2].x;
			const unsigned int r3 = p ? px[4 * Q + 3].y : px[4 * Q + 3].x;
			half4v b[4];
			b[0] = Base::template make_b<0>(r0, r1, r2, r3);
			b[1] = Base::template make_b<1>(r0, r1, r2, r3);
			b[2] = Base::template make_b<2>(r0, r1, r2, r3);
			b[3] = Base::template make_b<3>(r0, r1, r2, r3);
			if (p == 1 && more)
				load_rows<4 * Q, 4>(a, px, next_row, dir, coff);
#pragma unroll
			for (int c = 0; c < 4; c++) {
				acc[p * 4 + c][0] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, b[c], acc[p * 4 + c][0], 0, 0, 0);
				acc[p * 4 + c][1] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, b[c], acc[p * 4 + c][1], 0, 0, 0);
			}
		}
	}

	template <int ROT>
	static __device__ __forceinline__ void retire(float4v (&acc)[8][2], unsigned char *dst, bool store)
	{
		constexpr int SLOT = (ROT - (D - 1) + 2 * MFMA_SLOTS) % MFMA_SLOTS;
		constexpr int H = SLOT >> 2, I = SLOT & 3;
		if (store) {
			uint2 v;
			v.x = Base::fin_pack(acc[3][H][I], 3,
				Base::fin_pack(acc[2][H][I], 2, Base::fin_pack(acc[1][H][I], 1, Base::fin_pack(acc[0][H][I], 0, 0))));
			v.y = Base::fin_pack(acc[7][H][I], 3,
				Base::fin_pack(acc[6][H][I], 2, Base::fin_pack(acc[5][H][I], 1, Base::fin_pack(acc[4][H][I], 0, 0))));
			*reinterpret_cast<uint2 *>(dst) = v;
		}
#pragma unroll
		for (int o = 0; o < 8; o++)
			acc[o][H][I] = 0.0f;
	}

	// NB = prefetch depth: group g lives in ring buffer g mod NB (NB divides 8) and each of its quads is refilled
	// with group g + NB as soon as it has been consumed
	template <int ROT, int NB>
	static __device__ __forceinline__ void batch(const VStreamArgs &a, uint2 (&px)[NB][S], int g0, int ngroups,
		float4v (&acc)[8][2], const half4v *lane_a, int row0, int dir, unsigned int coff, unsigned char *out_col,
		int oh, bool active)
	{
		if constexpr (ROT < MFMA_SLOTS) {
			const int g = g0 + ROT;
			if (g < ngroups) {
				const bool more = g + NB < ngroups;
				const int next_row = row0 + dir * S * (g + NB);
				quad<ROT, 0>(a, px[ROT % NB], acc, lane_a, more, next_row, dir, coff);
				quad<ROT, 1>(a, px[ROT % NB], acc, lane_a, more, next_row, dir, coff);
				const int j = g - (D - 1); // row of the (possibly flipped) tile
				retire<ROT>(acc, out_col + (long long) (dir < 0 ? oh - 1 - j : j) * a.out_stride,
					active && j >= 0 && j < oh);
			}
			batch<ROT + 1, NB>(a, px, g0, ngroups, acc, lane_a, row0, dir, coff, out_col, oh, active);
		}
	}
};

template <int D, int NB, int OCC>
__global__ void __launch_bounds__(FUSED_THREADS, OCC)
reducev_u8_mfma(VStreamArgs a, const MfmaTables *__restrict__ tables)
{
	constexpr int S = 8;
	typedef VStreamStep<D> Step;
	__shared__ __attribute__((aligned(16))) half4v lds_a[MFMA_TABLE_ENTRIES];

	// each XCD takes a contiguous range of tiles (row-major: a tile row shares input rows)
	const int per_xcd = gridDim.x / 8;
	const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (tile >= a.tiles)
		return;
	const int t = threadIdx.x;
	const int by = tile / a.tiles_x;
	const int bx = tile - by * a.tiles_x;
	const int y0 = by * a.oht;
	const int oh = min(a.oht, a.out_height - y0);
	const int col = bx * FUSED_THREADS + t;
	const bool active = col < a.row_u2;
	const unsigned int coff = 8u * (unsigned int) min(col, a.row_u2 - 1);
	// Every other row of tiles is walked bottom-up (the flipped problem: rows counted from the last one, taps
	// reversed -- tables->a[1], as in the fused kernel above): a tile and the one below it share 8 (D - 1) input
	// rows, which both now read at about the same time -- the second read is an L2 hit.  With every tile walking
	// down they were read a whole kernel apart: 263 MB fetched for an image of 201 (profiles/r05l_ops_traffic.txt).
	const bool flip = a.alternate && (by & 1);
	const int dir = flip ? -1 : 1;
	const int row0 = flip ? a.fy0 + S * (y0 + oh - 1) + S * D - 1 : a.fy0 + S * y0;

	if (t < MFMA_TABLE_ENTRIES)
		reinterpret_cast<uint2 *>(lds_a)[t] = reinterpret_cast<const uint2 *>(tables->a[flip ? 1 : 0])[t];
	const half4v *lane_a = lds_a + (t & 3);

	float4v acc[8][2];
#pragma unroll
	for (int o = 0; o < 8; o++)
#pragma unroll
		for (int h = 0; h < 2; h++)
			acc[o][h] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };

	const int ngroups = oh + D - 1;
	uint2 px[NB][S];
#pragma unroll
	for (int b = 0; b < NB; b++)
		if (b < ngroups)
			Step::template load_rows<0, S>(a, px[b], row0 + dir * S * b, dir, coff);
	__syncthreads();

	unsigned char *out_col = a.out + (long long) y0 * a.out_stride + coff;
	for (int g0 = 0; g0 < ngroups; g0 += MFMA_SLOTS) {
		Step::template batch<0, NB>(a, px, g0, ngroups, acc, lane_a, row0, dir, coff, out_col, oh, active);
	}
}

// ------------------------------------------------ vips_reduce by 8 on THREE interleaved bands, one kernel (round 6)
//
// reduce_fused_u8x3_mfma<D>: the vertical pass is reducev_u8_mfma's (a scanline is a byte array to a vertical
// filter: a lane owns 8 consecutive BYTES of the row), but the finished T row goes to LDS, interleaved as it lies in
// memory, and every eight T rows the block makes their output pixels: lane (row, segment of 8 outputs, band) walks
// the segment's 13 groups of 8 pixels = 24 bytes with the same rotating accumulators, its band's bytes picked out of
// the three 8-byte LDS reads by two v_perm selectors.  A tile is 2 048 bytes of the row = 682 pixels and makes 80
// outputs (640 + 40 pixels of taps): tiles step by 1 920 bytes, 15 whole lines.  Output rows wait in LDS and leave
// in one burst at the tile's end.  The 25 MB image between reducev and reduceh (8192 x 8192 x 3) is never made.
// Columns outside the image (vips_embed COPY, reduceh.cpp:488-497): their T bytes are the edge pixel's, copied in
// LDS before the horizontal walk; the loads behind them are clamped to any valid dword.  Everything the host
// checks is in launch_fused_u8x3.
struct FusedIArgs {
	const unsigned char *in; // byte 0 of a window row (column in_left), row in_top of the image
	unsigned char *out;
	long long in_stride, out_stride;
	int in_top, im_height;
	int blo, bhi;            // the bytes of a window row that hold image columns: [blo, bhi), multiples of 4
	int tile_b0;             // byte of the first tap of output column 0 (3 (fx0 - in_left): may be negative), a multiple of 4
	int fy0;
	int out_width, out_height;
	int oht, tiles_x, tiles;
	int alternate;
	int aligned16;           // out and out_stride are multiples of 16
};

constexpr int F3_BANDS = 3;
constexpr int F3_OWT = 80;                      // outputs per tile
constexpr int F3_ROW = 8 * FUSED_THREADS;       // bytes of a T row
constexpr int F3_TPITCH = F3_ROW + 8;           // 514 dwords: rows 2 banks apart
constexpr int F3_SPITCH = F3_OWT * F3_BANDS;    // bytes per staged output row (15 x 16)
constexpr int F3_MAX_OHT = 128;
static constexpr size_t f3_lds_bytes(int oht)
{
	return (size_t) 2 * MFMA_SLOTS * F3_TPITCH + 2 * MFMA_TABLE_ENTRIES * 8 + (size_t) oht * F3_SPITCH;
}

template <int D>
struct FusedIStep {
	typedef MfmaStep<D> Base;
	static constexpr int S = 8;

	template <int I0, int N>
	static __device__ __forceinline__ void load_rows(const FusedIArgs &a, uint2 (&px)[S], int first_row, int dir,
		unsigned int o0, unsigned int o1, bool interior)
	{
		const unsigned int stride32 = (unsigned int) a.in_stride;
		typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
		typedef u32x2 __attribute__((aligned(4))) u32x2_a4;
		if (interior) {
#pragma unroll
			for (int i = I0; i < I0 + N; i++) {
				const int row = min(max(first_row + dir * i, 0), a.im_height - 1) - a.in_top;
				const u32x2 v = *reinterpret_cast<const u32x2_a4 *>(a.in + (size_t) ((unsigned int) row * stride32 + o0));
				px[i] = make_uint2(v.x, v.y);
			}
		}
		else {
#pragma unroll
			for (int i = I0; i < I0 + N; i++) {
				const int row = min(max(first_row + dir * i, 0), a.im_height - 1) - a.in_top;
				const unsigned int base = (unsigned int) row * stride32;
				px[i].x = *reinterpret_cast<const unsigned int *>(a.in + (size_t) (base + o0));
				px[i].y = *reinterpret_cast<const unsigned int *>(a.in + (size_t) (base + o1));
			}
		}
	}

	template <int ROT, int Q>
	static __device__ __forceinline__ void quad(const FusedIArgs &a, uint2 (&px)[S], float4v (&acc)[8][2],
		const half4v *lane_a, bool more, int next_row, int dir, unsigned int o0, unsigned int o1, bool interior)
	{
		const half4v a0 = lane_a[((ROT * 2 + Q) * 2 + 0) * 4];
		const half4v a1 = lane_a[((ROT * 2 + Q) * 2 + 1) * 4];
#pragma unroll
		for (int p = 0; p < 2; p++) {
			const unsigned int r0 = p ? px[4 * Q + 0].y : px[4 * Q + 0].x;
			const unsigned int r1 = p ? px[4 * Q + 1].y : px[4 * Q + 1].x;
			const unsigned int r2 = p ? px[4 * Q + 2].y : px[4 * Q + 2].x;
			const unsigned int r3 = p ? px[4 * Q + 3].y : px[4 * Q + 3].x;
			half4v b[4];
			b[0] = Base::template make_b<0>(r0, r1, r2, r3);
			b[1] = Base::template make_b<1>(r0, r1, r2, r3);
			b[2] = Base::template make_b<2>(r0, r1, r2, r3);
			b[3] = Base::template make_b<3>(r0, r1, r2, r3);
			if (p == 1 && more)
				load_rows<4 * Q, 4>(a, px, next_row, dir, o0, o1, interior);
#pragma unroll
			for (int c = 0; c < 4; c++) {
				acc[p * 4 + c][0] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, b[c], acc[p * 4 + c][0], 0, 0, 0);
				acc[p * 4 + c][1] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, b[c], acc[p * 4 + c][1], 0, 0, 0);
			}
		}
	}

	// slot (ROT - (D - 1)) mod 8 has seen all its taps: its 8 bytes into T row ROT
	template <int ROT>
	static __device__ __forceinline__ void retire(float4v (&acc)[8][2], unsigned char *trow_lane, bool store)
	{
		constexpr int SLOT = (ROT - (D - 1) + 2 * MFMA_SLOTS) % MFMA_SLOTS;
		constexpr int H = SLOT >> 2, I = SLOT & 3;
		if (store) {
			uint2 v;
			v.x = Base::fin_pack(acc[3][H][I], 3,
				Base::fin_pack(acc[2][H][I], 2, Base::fin_pack(acc[1][H][I], 1, Base::fin_pack(acc[0][H][I], 0, 0))));
			v.y = Base::fin_pack(acc[7][H][I], 3,
				Base::fin_pack(acc[6][H][I], 2, Base::fin_pack(acc[5][H][I], 1, Base::fin_pack(acc[4][H][I], 0, 0))));
			*reinterpret_cast<uint2 *>(trow_lane + ROT * F3_TPITCH) = v;
		}
#pragma unroll
		for (int o = 0; o < 8; o++)
			acc[o][H][I] = 0.0f;
	}

	// NB = prefetch depth: group g lives in ring buffer g mod NB (NB divides 8) and each of its quads is refilled
	// with group g + NB as soon as it has been consumed
	template <int ROT, int NB>
	static __device__ __forceinline__ void batch(const FusedIArgs &a, uint2 (&px)[NB][S], int g0, int ngroups,
		float4v (&acc)[8][2], const half4v *lane_a, int row0, int dir, unsigned int o0, unsigned int o1, bool interior,
		unsigned char *trow_lane, int oh)
	{
		if constexpr (ROT < MFMA_SLOTS) {
			const int g = g0 + ROT;
			if (g < ngroups) {
				const bool more = g + NB < ngroups;
				const int next_row = row0 + dir * S * (g + NB);
				quad<ROT, 0>(a, px[ROT % NB], acc, lane_a, more, next_row, dir, o0, o1, interior);
				quad<ROT, 1>(a, px[ROT % NB], acc, lane_a, more, next_row, dir, o0, o1, interior);
				const int j = g - (D - 1);
				retire<ROT>(acc, trow_lane, j >= 0 && j < oh);
			}
			batch<ROT + 1, NB>(a, px, g0, ngroups, acc, lane_a, row0, dir, o0, o1, interior, trow_lane, oh);
		}
	}

	// the horizontal walk of one (T row, segment, band): group G = pixels 8 G .. 8 G + 7 of the segment = 24 bytes;
	// the band's bytes are c, c + 3, ... : pixels (0, 1) out of dwords (0, 1), (2, 3) out of (1, 2), (4, 5) out
	// of (3, 4), (6, 7) out of (4, 5) -- selectors sel_a (bytes c, c + 3) and sel_b (bytes c + 2, c + 5)
	template <int G, int FENCE>
	static __device__ __forceinline__ void hwalk(float4v (&hacc)[2], const unsigned char *line, const half4v *lane_ah,
		unsigned int sel_a, unsigned int sel_b, unsigned int (&outb)[HSEG_OUT])
	{
		constexpr int NG = HSEG_OUT + D - 1;
		if constexpr (G < NG) {
			constexpr int ROT = G % MFMA_SLOTS;
			const uint2 w01 = *reinterpret_cast<const uint2 *>(line + 24 * G);
			const uint2 w23 = *reinterpret_cast<const uint2 *>(line + 24 * G + 8);
			const uint2 w45 = *reinterpret_cast<const uint2 *>(line + 24 * G + 16);
			uint2 v0, v1;
			v0.x = __builtin_amdgcn_perm(w01.y, w01.x, sel_a);
			v0.y = __builtin_amdgcn_perm(w23.x, w01.y, sel_b);
			v1.x = __builtin_amdgcn_perm(w45.x, w23.y, sel_a);
			v1.y = __builtin_amdgcn_perm(w45.y, w45.x, sel_b);
			const half4v b0 = __builtin_bit_cast(half4v, v0);
			const half4v b1 = __builtin_bit_cast(half4v, v1);
			hacc[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 0) * 2 + 0) * 4], b0, hacc[0], 0, 0, 0);
			hacc[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 0) * 2 + 1) * 4], b0, hacc[1], 0, 0, 0);
			hacc[0] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 1) * 2 + 0) * 4], b1, hacc[0], 0, 0, 0);
			hacc[1] = __builtin_amdgcn_mfma_f32_4x4x4f16(lane_ah[((ROT * 2 + 1) * 2 + 1) * 4], b1, hacc[1], 0, 0, 0);
			constexpr int SLOT = (ROT - (D - 1) + 2 * MFMA_SLOTS) % MFMA_SLOTS;
			constexpr int H = SLOT >> 2, I = SLOT & 3;
			if constexpr (G >= D - 1)
				outb[G - (D - 1)] = Base::fin_pack(hacc[H][I], 0, 0);
			hacc[H][I] = 0.0f;
			if constexpr (FENCE > 0 && (G % (FENCE > 0 ? FENCE : 1)) == FENCE - 1)
				__builtin_amdgcn_sched_barrier(0); // keep the unrolled walk's LDS reads from piling up
			hwalk<G + 1, FENCE>(hacc, line, lane_ah, sel_a, sel_b, outb);
		}
	}
};

template <int D, int NB, int OCC, int FENCE>
__global__ void __launch_bounds__(FUSED_THREADS, OCC)
reduce_fused_u8x3_mfma(FusedIArgs a, const MfmaTables *__restrict__ tables)
{
	constexpr int S = 8;
	typedef FusedIStep<D> Step;
	VH_DYNAMIC_LDS(unsigned char, lds_raw);
	unsigned char *trows = lds_raw; // two buffers of 8 T rows: batch b's go to buffer b & 1
	half4v *lds_a = reinterpret_cast<half4v *>(lds_raw + 2 * MFMA_SLOTS * F3_TPITCH);
	half4v *lds_ah = lds_a + MFMA_TABLE_ENTRIES;
	unsigned char *stage = reinterpret_cast<unsigned char *>(lds_ah + MFMA_TABLE_ENTRIES);

	// each XCD takes a contiguous range of tiles (row-major: a tile row shares input rows)
	const int per_xcd = gridDim.x / 8;
	const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (tile >= a.tiles)
		return;
	const int t = threadIdx.x;
	const int by = tile / a.tiles_x;
	const int bx = tile - by * a.tiles_x;
	const int x0 = bx * F3_OWT;
	const int y0 = by * a.oht;
	const int ow = min(F3_OWT, a.out_width - x0);
	const int oh = min(a.oht, a.out_height - y0);

	// byte of a window row the tile's T rows start at, and this lane's two dwords of it (clamped into the image's
	// bytes: a clamped dword is overwritten in LDS)
	const int tb = a.tile_b0 + 8 * F3_BANDS * x0;
	const bool interior = tb >= a.blo && tb + F3_ROW <= a.bhi;
	const unsigned int o0 = (unsigned int) min(max(tb + 8 * t, a.blo), a.bhi - 4);
	const unsigned int o1 = (unsigned int) min(max(tb + 8 * t + 4, a.blo), a.bhi - 4);

	const bool flip = a.alternate && (by & 1);
	const int dir = flip ? -1 : 1;
	const int row0 = flip ? a.fy0 + S * (y0 + oh - 1) + S * D - 1 : a.fy0 + S * y0;

	if (t < MFMA_TABLE_ENTRIES) {
		reinterpret_cast<uint2 *>(lds_a)[t] = reinterpret_cast<const uint2 *>(tables->a[flip ? 1 : 0])[t];
		reinterpret_cast<uint2 *>(lds_ah)[t] = reinterpret_cast<const uint2 *>(tables->ah)[t];
	}
	const half4v *lane_a = lds_a + (t & 3);
	const half4v *lane_ah = lds_ah + (t & 3);

	float4v acc[8][2];
#pragma unroll
	for (int o = 0; o < 8; o++)
#pragma unroll
		for (int h = 0; h < 2; h++)
			acc[o][h] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };

	const int ngroups = oh + D - 1;
	uint2 px[NB][S];
#pragma unroll
	for (int b = 0; b < NB; b++)
		if (b < ngroups)
			Step::template load_rows<0, S>(a, px[b], row0 + dir * S * b, dir, o0, o1, interior);
	__syncthreads();

	// thread -> (T row, segment of 8 outputs, band); the last 16 threads walk with thread 239's addresses (their
	// matrix instructions carry no one else's operands, but every lane must hold a valid address) and store nothing
	const int hu = min(t / F3_BANDS, 8 * (F3_OWT / HSEG_OUT) - 1);
	const int hc = t - F3_BANDS * (t / F3_BANDS), hr = hu & 7, hseg = hu >> 3;
	const bool hlane = t < F3_BANDS * 8 * (F3_OWT / HSEG_OUT);
	const unsigned int sel_a = 0x0c000c00u | (unsigned int) hc | ((unsigned int) (hc + 3) << 16);
	const unsigned int sel_b = 0x0c000c00u | (unsigned int) (hc + 2) | ((unsigned int) (hc + 5) << 16);

	// One barrier a batch: a wave that is through its horizontal pass goes on walking into the OTHER buffer while
	// the block's slower waves still read this one; it cannot reach this buffer again before the next barrier.
	for (int g0 = 0; g0 < ngroups; g0 += MFMA_SLOTS) {
		unsigned char *tbuf = trows + ((g0 / MFMA_SLOTS) & 1) * (MFMA_SLOTS * F3_TPITCH);
		Step::template batch<0, NB>(a, px, g0, ngroups, acc, lane_a, row0, dir, o0, o1, interior, tbuf + 8 * t, oh);
		__syncthreads();

		// ---- horizontal pass over the rows this batch completed (T row r <-> group g0 + r); the loads of the next
		// NB groups are in flight across it
		const int jlo = max(g0 - (D - 1), 0);
		const int jhi = min(g0 + MFMA_SLOTS - 1 - (D - 1), oh - 1); // inclusive
		if (jhi < jlo)
			continue;
		const int nrows = jhi - jlo + 1;
		const int r_lo = jlo - (g0 - (D - 1));
		if (!interior) {
			// columns left of the image's first / right of its last: the edge pixel's three bytes
			if (tb < a.blo) {
				const int n = a.blo - tb; // a multiple of 3 (and of 4)
				for (int i = t; i < nrows * n; i += FUSED_THREADS) {
					const int r = i / n, k = i - r * n;
					unsigned char *row = tbuf + (r_lo + r) * F3_TPITCH;
					row[k] = row[n + k % F3_BANDS];
				}
			}
			if (tb + F3_ROW > a.bhi) {
				const int k0 = max(a.bhi - tb, F3_BANDS), n = F3_ROW - k0;
				for (int i = t; i < nrows * n; i += FUSED_THREADS) {
					const int r = i / n, k = i - r * n;
					unsigned char *row = tbuf + (r_lo + r) * F3_TPITCH;
					row[k0 + k] = row[k0 - F3_BANDS + k % F3_BANDS];
				}
			}
			__syncthreads();
		}
		{
			const bool row_ok = hr < nrows;
			const int lrow = r_lo + (row_ok ? hr : 0);
			const unsigned char *line = tbuf + lrow * F3_TPITCH + 8 * F3_BANDS * HSEG_OUT * hseg;
			float4v hacc[2];
			hacc[0] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };
			hacc[1] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };
			unsigned int outb[HSEG_OUT];
			Step::template hwalk<0, FENCE>(hacc, line, lane_ah, sel_a, sel_b, outb);
			if (row_ok && hlane) {
				unsigned char *srow = stage + (jlo + hr) * F3_SPITCH + F3_BANDS * HSEG_OUT * hseg + hc;
#pragma unroll
				for (int k = 0; k < HSEG_OUT; k++)
					srow[F3_BANDS * k] = (unsigned char) outb[k];
			}
		}
	}
	__syncthreads();

	// ---- the tile's rows leave in one burst: 16 lanes a row, 16 bytes a lane
	{
		const int part = t & 15, nbytes = F3_BANDS * ow;
		for (int r = t >> 4; r < oh; r += FUSED_THREADS / 16) {
			if (16 * part >= nbytes)
				continue;
			unsigned char *dst =
				a.out + (long long) (y0 + (flip ? oh - 1 - r : r)) * a.out_stride + (long long) x0 * F3_BANDS + 16 * part;
			const unsigned char *src = stage + r * F3_SPITCH + 16 * part;
			if (a.aligned16 && 16 * part + 16 <= nbytes)
				*reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
			else {
				const int n = min(16, nbytes - 16 * part);
				for (int k = 0; k < n; k++)
					dst[k] = src[k];
			}
		}
	}
}

// ------------------------------------------------ vips_reduceh by 8 on three interleaved bands (round 6)
//
// reduceh_u8x3_mfma<D>: the horizontal half of reduce_fused_u8x3_mfma on its own -- the rows of the image take the
// place of the T rows.  A lane copies its 8 bytes of eight rows into LDS (whole-line loads: the packed vector-ALU
// kernel this replaces and the banded matrix kernel both read a row a lane or stage through bytes), then lane (row,
// segment of 8 outputs, band) walks its 24-byte groups; two batches of rows in flight a lane, two LDS buffers, one
// barrier a batch, the tile's output rows in one burst.  No vertical halo, so a tile is as tall as fills the chip.
struct RhIArgs {
	const unsigned char *in; // byte 0 of a window row (column in_left), the rect's first row
	unsigned char *out;
	long long in_stride, out_stride;
	int blo, bhi, tile_b0;   // as FusedIArgs
	int out_width, rows;
	int oht, tiles_x, tiles;
	int aligned16;
};

constexpr int RH3_NB = 2; // batches of 8 rows in flight a lane
static constexpr size_t rh3_lds_bytes(int oht)
{
	return (size_t) 2 * MFMA_SLOTS * F3_TPITCH + MFMA_TABLE_ENTRIES * 8 + (size_t) oht * F3_SPITCH;
}

template <int D>
__global__ void __launch_bounds__(FUSED_THREADS, 3)
reduceh_u8x3_mfma(RhIArgs a, const MfmaTables *__restrict__ tables)
{
	typedef FusedIStep<D> Step;
	VH_DYNAMIC_LDS(unsigned char, lds_raw);
	unsigned char *trows = lds_raw; // two buffers of 8 rows
	half4v *lds_ah = reinterpret_cast<half4v *>(lds_raw + 2 * MFMA_SLOTS * F3_TPITCH);
	unsigned char *stage = reinterpret_cast<unsigned char *>(lds_ah + MFMA_TABLE_ENTRIES);

	const int per_xcd = gridDim.x / 8;
	const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
	if (tile >= a.tiles)
		return;
	const int t = threadIdx.x;
	const int by = tile / a.tiles_x;
	const int bx = tile - by * a.tiles_x;
	const int x0 = bx * F3_OWT;
	const int y0 = by * a.oht;
	const int ow = min(F3_OWT, a.out_width - x0);
	const int oh = min(a.oht, a.rows - y0);

	const int tb = a.tile_b0 + 8 * F3_BANDS * x0;
	const bool interior = tb >= a.blo && tb + F3_ROW <= a.bhi;
	const unsigned int o0 = (unsigned int) min(max(tb + 8 * t, a.blo), a.bhi - 4);
	const unsigned int o1 = (unsigned int) min(max(tb + 8 * t + 4, a.blo), a.bhi - 4);

	if (t < MFMA_TABLE_ENTRIES)
		reinterpret_cast<uint2 *>(lds_ah)[t] = reinterpret_cast<const uint2 *>(tables->ah)[t];
	const half4v *lane_ah = lds_ah + (t & 3);

	// rows y0 + 8 k + i of the rect (past the tile's last: that one again, never walked)
	const unsigned int stride32 = (unsigned int) a.in_stride;
	auto load = [&](uint2 (&px)[8], int k) {
		typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
		typedef u32x2 __attribute__((aligned(4))) u32x2_a4;
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const unsigned int base = (unsigned int) (y0 + min(8 * k + i, oh - 1)) * stride32;
			if (interior) {
				const u32x2 v = *reinterpret_cast<const u32x2_a4 *>(a.in + (size_t) (base + o0));
				px[i] = make_uint2(v.x, v.y);
			}
			else {
				px[i].x = *reinterpret_cast<const unsigned int *>(a.in + (size_t) (base + o0));
				px[i].y = *reinterpret_cast<const unsigned int *>(a.in + (size_t) (base + o1));
			}
		}
	};
	const int nb = (oh + 7) / 8;
	uint2 px[RH3_NB][8];
#pragma unroll
	for (int p = 0; p < RH3_NB; p++)
		if (p < nb)
			load(px[p], p);
	__syncthreads();

	const int hu = min(t / F3_BANDS, 8 * (F3_OWT / HSEG_OUT) - 1);
	const int hc = t - F3_BANDS * (t / F3_BANDS), hr = hu & 7, hseg = hu >> 3;
	const bool hlane = t < F3_BANDS * 8 * (F3_OWT / HSEG_OUT);
	const unsigned int sel_a = 0x0c000c00u | (unsigned int) hc | ((unsigned int) (hc + 3) << 16);
	const unsigned int sel_b = 0x0c000c00u | (unsigned int) (hc + 2) | ((unsigned int) (hc + 5) << 16);

	for (int k0 = 0; k0 < nb; k0 += RH3_NB) {
#pragma unroll
		for (int p = 0; p < RH3_NB; p++) {
			const int k = k0 + p;
			if (k >= nb)
				break;
			unsigned char *tbuf = trows + (k & 1) * (MFMA_SLOTS * F3_TPITCH);
#pragma unroll
			for (int i = 0; i < 8; i++)
				*reinterpret_cast<uint2 *>(tbuf + i * F3_TPITCH + 8 * t) = px[p][i];
			if (k + RH3_NB < nb)
				load(px[p], k + RH3_NB);
			__syncthreads();
			const int nrows = min(8, oh - 8 * k);
			if (!interior) {
				if (tb < a.blo) {
					const int n = a.blo - tb;
					for (int i = t; i < nrows * n; i += FUSED_THREADS) {
						const int r = i / n, kk = i - r * n;
						unsigned char *row = tbuf + r * F3_TPITCH;
						row[kk] = row[n + kk % F3_BANDS];
					}
				}
				if (tb + F3_ROW > a.bhi) {
					const int c0 = max(a.bhi - tb, F3_BANDS), n = F3_ROW - c0;
					for (int i = t; i < nrows * n; i += FUSED_THREADS) {
						const int r = i / n, kk = i - r * n;
						unsigned char *row = tbuf + r * F3_TPITCH;
						row[c0 + kk] = row[c0 - F3_BANDS + kk % F3_BANDS];
					}
				}
				__syncthreads();
			}
			const bool row_ok = hr < nrows;
			const unsigned char *line = tbuf + (row_ok ? hr : 0) * F3_TPITCH + 8 * F3_BANDS * HSEG_OUT * hseg;
			float4v hacc[2];
			hacc[0] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };
			hacc[1] = (float4v){ 0.0f, 0.0f, 0.0f, 0.0f };
			unsigned int outb[HSEG_OUT];
			Step::template hwalk<0, 0>(hacc, line, lane_ah, sel_a, sel_b, outb);
			if (row_ok && hlane) {
				unsigned char *srow = stage + (8 * k + hr) * F3_SPITCH + F3_BANDS * HSEG_OUT * hseg + hc;
#pragma unroll
				for (int q = 0; q < HSEG_OUT; q++)
					srow[F3_BANDS * q] = (unsigned char) outb[q];
			}
		}
	}
	__syncthreads();
	{
		const int part = t & 15, nbytes = F3_BANDS * ow;
		for (int r = t >> 4; r < oh; r += FUSED_THREADS / 16) {
			if (16 * part >= nbytes)
				continue;
			unsigned char *dst = a.out + (long long) (y0 + r) * a.out_stride + (long long) x0 * F3_BANDS + 16 * part;
			const unsigned char *src = stage + r * F3_SPITCH + 16 * part;
			if (a.aligned16 && 16 * part + 16 <= nbytes)
				*reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
			else {
				const int n = min(16, nbytes - 16 * part);
				for (int q = 0; q < n; q++)
					dst[q] = src[q];
			}
		}
	}
}

// Is pos[] an arithmetic progression first0 + S*k with one phase?  (What an
// integer shrink of a size-divisible image produces.)
static bool positions_regular(const std::vector<ReducePos> &pos, int *first0, int *step, int *phase)
{
	if (pos.empty())
		return false;
	*first0 = pos[0].first;
	*phase = pos[0].phase;
	*step = pos.size() > 1 ? pos[1].first - pos[0].first : 0;
	for (size_t k = 0; k < pos.size(); k++)
		if (pos[k].first != *first0 + (int) k * *step || pos[k].phase != *phase)
			return false;
	return true;
}

// Pack taps [0, n) of matrixs row `phase`, zero-padded to `total`, as i16 pairs.
static void pack_pairs(const _VipsHipReduce *r, int phase, int total, std::vector<unsigned int> &out)
{
	const short *c = &r->matrixs[(size_t) phase * r->n_point];
	out.resize(total / 2);
	for (int q = 0; q < total / 2; q++) {
		const int k0 = 2 * q, k1 = 2 * q + 1;
		const unsigned short lo = k0 < r->n_point ? (unsigned short) c[k0] : 0;
		const unsigned short hi = k1 < r->n_point ? (unsigned short) c[k1] : 0;
		out[q] = (unsigned int) lo | ((unsigned int) hi << 16);
	}
}

// number of leading taps that matter: trailing zero coefficients are dropped
static int effective_taps(const _VipsHipReduce *r, int phase)
{
	const short *c = &r->matrixs[(size_t) phase * r->n_point];
	int n = r->n_point;
	while (n > 1 && c[n - 1] == 0)
		n--;
	return n;
}

struct FusedPlan {
	unsigned int *d_cv;
	unsigned int *d_ch;
	int fx0, fy0, step, d;
};

// f16 bit pattern of an integer |v| < 2048 (exact)
static unsigned short half_bits(int v)
{
	const _Float16 h = (_Float16) (float) v;
	unsigned short bits;
	memcpy(&bits, &h, sizeof(bits));
	return bits;
}

// The MFMA kernel's A-operand tables (both walking directions) for taps c[0 .. 8*D).
static void mfma_build_tables(const std::vector<int> &taps, const std::vector<int> &taps_h, int D,
	MfmaTables *tab)
{
	const int nt = 8 * D;
	for (int rot = 0; rot < MFMA_SLOTS; rot++)
		for (int q = 0; q < 2; q++)
			for (int h = 0; h < 2; h++)
				for (int i = 0; i < 4; i++) {
					const int d = (rot - (4 * h + i) + 2 * MFMA_SLOTS) % MFMA_SLOTS;
					for (int k = 0; k < 4; k++)
						tab->ah[((((rot * 2 + q) * 2 + h) * 4 + i) * 4) + k] =
							half_bits(d < D ? taps_h[8 * d + 4 * q + k] : 0);
				}
	for (int flip = 0; flip < 2; flip++)
		for (int rot = 0; rot < MFMA_SLOTS; rot++)
			for (int q = 0; q < 2; q++)
				for (int h = 0; h < 2; h++)
					for (int i = 0; i < 4; i++) {
						const int slot = 4 * h + i;
						const int d = (rot - slot + 2 * MFMA_SLOTS) % MFMA_SLOTS;
						for (int k = 0; k < 4; k++) {
							int c = 0;
							if (d < D) {
								const int tap = 8 * d + 4 * q + k;
								c = flip ? taps[nt - 1 - tap] : taps[tap];
							}
							tab->a[flip][((((rot * 2 + q) * 2 + h) * 4 + i) * 4) + k] = half_bits(c);
						}
					}
}

// The kernel with halos, in its shipped form only: whole-pair loads, the edge fix-up at the point of use, streaming
// (nt) loads, 256 threads.  (Rounds 2-5 kept the A/B forms -- the fix-up at the loads, plain loads, 512 threads,
// arithmetic-only and loads-only profiling builds -- behind VIPS_HIP_FUSED_LATE / _NT / _NTH / _DEBUG=8|16; their
// measurements are in profiles/NOTES.md 3.1, the forms are gone.)
template <int D>
static int launch_fused_mfma(const FusedArgs &args, int tiles, const MfmaTables *d_tables)
{
	Gate gate("reduce_fused_u8_mfma");
	typedef MfmaGeo<FUSED_THREADS> Geo;
	const int grid = (tiles + 7) / 8 * 8; // XCD remap wants a multiple of 8
	const int stage_rows = args.burst_rows + 7 < args.oht ? args.burst_rows + 7 : args.oht;
	const size_t lds = Geo::lds_bytes(stage_rows);
	hipLaunchKernelGGL((reduce_fused_u8x4_mfma<D, 1, 4, true, 0, true, FUSED_THREADS, 1>), dim3(grid), dim3(FUSED_THREADS),
		lds, stream(), args, d_tables);
	VH_CHECK(hipGetLastError());
	return 0;
}

// Does block b of a launch run on XCD b % 8 on this device?  (HIP promises nothing; the part's eight command
// processors each take every eighth workgroup.)  One launch of 2 048 blocks, once per device.
__global__ void xcc_census_kernel(int *wrong)
{
	if (threadIdx.x == 0 && VH_XCC_ID() != (int) (blockIdx.x & 7u))
		atomicAdd(wrong, 1);
}

static bool xcd_placement_holds()
{
	constexpr int MAXDEV = 64;
	static std::mutex mutex;
	static signed char state[MAXDEV];
	const int dev = current_device();
	if (dev < 0 || dev >= MAXDEV)
		return false;
	std::lock_guard<std::mutex> lock(mutex);
	if (state[dev] == 0) {
		state[dev] = -1;
		int zero = 0, wrong = 1;
		int *d = (int *) upload(&zero, sizeof(zero));
		if (d) {
			hipLaunchKernelGGL(xcc_census_kernel, dim3(2048), dim3(64), 0, stream(), d);
			if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&wrong, d, sizeof(wrong), hipMemcpyDeviceToHost, stream()) == hipSuccess &&
				hipStreamSynchronize(stream()) == hipSuccess && wrong == 0)
				state[dev] = 1;
			vips_hip_free(d);
		}
		else
			vips_hip_error_clear();
	}
	return state[dev] == 1;
}

// 0 launched, 1 not this kernel's case, -1 error
static int launch_fused_mfma_x(const FusedArgs &all, const VipsHipRegion *in, const VipsHipRegion *out,
	const MfmaTables *d_tables)
{
	const char *e = getenv("VIPS_HIP_FUSED_EXCH");
	if (e && atoi(e) == 0)
		return 1;
	if (in->left != 0 || in->width != in->im_width || out->left != 0 || out->width != out->im_width ||
		(in->im_width & 511) || out->width * 8 != in->im_width)
		return 1;
	if (all.fx0 < -(XH - 24) - 0 || all.fx0 > -16 || (all.fx0 & 3) || all.fx0 < -24)
		return 1;
	// (whole 128-byte lines per wave and row are the point of it; forced by the environment -- the parity tests on
	// host fibers, whose "device" memory is malloc's -- 16 bytes do)
	const uintptr_t in_mask = e ? 15 : 127;
	if (((uintptr_t) in->data & in_mask) || (in->stride & in_mask) || ((uintptr_t) out->data & 15) || (out->stride & 15))
		return 1;
	FusedArgs a = all;
	a.tiles_x = in->im_width / 512;
	// one residency round of 2 blocks a CU: as many rows of tiles as 512 slots allow, tiles of 32 .. 128 rows
	int tiles_y = 512 / a.tiles_x;
	const int least = (out->height + XMAX_OHT - 1) / XMAX_OHT;
	tiles_y = tiles_y < least ? least : tiles_y;
	int oht = (out->height + tiles_y - 1) / tiles_y;
	if (oht < 32)
		oht = 32;
	if (oht > XMAX_OHT)
		return 1;
	tiles_y = (out->height + oht - 1) / oht;
	a.oht = oht;
	a.owt = 64;
	a.tiles = a.tiles_x * tiles_y;
	const int want = e ? 1 : 384; // (by default only launches that fill most of the part; $VIPS_HIP_FUSED_EXCH=1: any)
	if (a.tiles < want || a.tiles > 512)
		return 1;
	const size_t bytes = (size_t) a.tiles * a.oht * XPART * sizeof(float);
	float *parts = (float *) vips_hip_malloc(bytes);
	if (!parts)
		return -1;
	// the arrival counters: one int per tile boundary, zero once and for the life of the calling thread (a launch
	// adds exactly two to each: see the kernel's end); per thread and device, as the stream the launches are
	// ordered on is
	constexpr int MAX_ARRIVALS = 2048;
	if ((a.tiles_x + 1) * tiles_y > MAX_ARRIVALS) {
		vips_hip_free(parts);
		return 1;
	}
	static thread_local std::map<int, int *> arrivals_by_device;
	int *&arrivals = arrivals_by_device[current_device()];
	if (!arrivals) {
		std::vector<int> zeros(MAX_ARRIVALS, 0);
		arrivals = (int *) upload(zeros.data(), zeros.size() * sizeof(int)); // (kept: a thread's 8 KB)
		if (!arrivals) {
			vips_hip_free(parts);
			return -1;
		}
	}
	// the hand-off through the XCD's L2 (see the kernel): rows of tiles dealt whole to the XCDs, the placement checked
	// once, a word of pinned host memory for a block that finds itself elsewhere.  $VIPS_HIP_FUSED_PLAIN=0: through memory
	static thread_local std::map<int, int *> misplaced_by_device;
	int *&misplaced = misplaced_by_device[current_device()];
	if (!misplaced) {
		misplaced = (int *) vips_hip_malloc_host(64);
		if (misplaced)
			*misplaced = 0;
		else
			vips_hip_error_clear();
	}
	static std::atomic<bool> plain_broken(false);
	if (misplaced && *misplaced) {
		*misplaced = 0;
		if (!plain_broken.exchange(true))
			fprintf(stderr, "vips-hip: reduce: a block ran on another XCD than its index says; the hand-off goes through memory from now on\n");
	}
	const char *pe = getenv("VIPS_HIP_FUSED_PLAIN");
	const int plain = misplaced && !plain_broken.load() && !(pe && atoi(pe) == 0) && xcd_placement_holds() ? 1 : 0;
	const size_t lds = xlds_bytes(a.oht);
	const int rows_per_xcd = (tiles_y + 7) / 8;
	const int grid = 8 * rows_per_xcd * a.tiles_x; // (the kernel's numbering: XCD k takes rows k rows_per_xcd ...)
	int rc = 0;
	{
		Gate gate("reduce_fused_u8_mfma_x");
		hipError_t err;
		{
			err = hipFuncSetAttribute((const void *) reduce_fused_u8x4_mfma_x<6, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
				80 * 1024);
			if (err == hipSuccess)
				hipLaunchKernelGGL((reduce_fused_u8x4_mfma_x<6, 4, 2>), dim3(grid), dim3(FUSED_THREADS), lds, stream(), a, d_tables,
					parts, arrivals, plain, misplaced);
		}
		if (err != hipSuccess || hipGetLastError() != hipSuccess)
			rc = -1;
	}
	if (rc == 0 && (a.debug & 32)) {
		Gate gate("reduce_fused_edges");
		const long long threads = (long long) out->height * (a.tiles_x + 1) * 6;
		hipLaunchKernelGGL(reduce_fused_edges, dim3((unsigned int) ((threads + 255) / 256)), dim3(256), 0, stream(), a, parts);
		if (hipGetLastError() != hipSuccess)
			rc = -1;
	}
	vips_hip_free(parts); // (the pool hands the block to this thread's LATER work only: ordered on its stream)
	if (rc)
		error("reduce", "kernel launch failed");
	return rc;
}

// vips_reduce by 8 on a 3-band uchar region in one kernel (reduce_fused_u8x3_mfma).  0: launched; 1: not this
// kernel's case (the caller goes on to reducev, then reduceh); -1: error.  What the kernel assumes and this checks:
// every dword of a window row lies wholly inside or wholly outside the image's columns, and wholly inside or outside
// a tile -- base, stride, the window's first and last image byte and the first tap's byte are multiples of 4
// (a whole image whose width is a multiple of 4 at a 4-byte base always is: the first tap of vips_reduce(8) is
// column -20 or -24); byte offsets fit 32 bits.
static int launch_fused_u8x3(int D, int taps_h, const VipsHipRegion *in, const VipsHipRegion *out, int fx0, int fy0,
	const MfmaTables *d_tables)
{
	if (getenv("VIPS_HIP_NO_FUSED3") || in->bands != F3_BANDS || (D != 6 && D != 7))
		return 1;
	// the last output of a tile must find its taps inside the tile's T row (the walk reads 8 D columns: past the
	// last real tap the coefficients are zero and what it reads there -- any bytes, finite as halves -- counts for nothing)
	if (F3_BANDS * (8 * (F3_OWT - 1) + taps_h) > F3_ROW)
		return 1;
	if (!(in->stride > 0 && (long long) in->stride * in->height < (1LL << 31)))
		return 1;
	if (((uintptr_t) in->data & 3) || (in->stride & 3))
		return 1;
	const int lo = in->left > 0 ? in->left : 0;
	const int hi1 = in->im_width < in->left + in->width ? in->im_width : in->left + in->width;
	const long long blo = 3LL * (lo - in->left), bhi = 3LL * (hi1 - in->left), tb0 = 3LL * ((long long) fx0 - in->left);
	if ((blo & 3) || (bhi & 3) || (tb0 & 3) || bhi - blo < 8 || tb0 < -(1LL << 30) || tb0 > (1LL << 30))
		return 1;
	FusedIArgs a;
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_top = in->top;
	a.im_height = in->im_height;
	a.blo = (int) blo;
	a.bhi = (int) bhi;
	a.tile_b0 = (int) tb0;
	a.fy0 = fy0;
	a.out_width = out->width;
	a.out_height = out->height;
	a.tiles_x = (out->width + F3_OWT - 1) / F3_OWT;
	// a tile re-reads 8 (D - 1) rows of the one above (an L2 hit when the two walk towards each other): the tallest
	// tile that still leaves 1.5 tiles a CU, as reducev_u8_mfma
	// Tile height.  A tile re-reads 8 (D - 1) rows of the one above (an L2 hit when the two walk towards each
	// other), and the launch takes as long as its busiest CU: the tiles it gets -- never fewer than two at a time,
	// one block cannot keep a CU's memory pipe busy -- times the row groups a tile walks.  The height with the
	// smallest product (profiles/r06e_reduce_rgb_sizes*.txt: 1024 ... 20480 squared, the model against the clock).
	int oht = 32;
	auto tiles_at = [&](int h) { return (long long) a.tiles_x * ((out->height + h - 1) / h); };
	{
		long long best = -1;
		for (int h : { 8, 12, 16, 20, 24, 28, 32, 40, 48 }) {
			const long long per_cu = (tiles_at(h) + 255) / 256;
			const long long cost = (2 * (per_cu < 2 ? 2 : per_cu) + 1) * (h + D - 1); // (+ half a tile: the ragged end)
			if (best < 0 || cost <= best) {
				best = cost;
				oht = h;
			}
		}
	}
	if (const char *e = getenv("VIPS_HIP_FUSED3_OHT"))
		oht = atoi(e) > 0 && atoi(e) <= F3_MAX_OHT ? atoi(e) : oht;
	a.oht = oht;
	a.alternate = !getenv("VIPS_HIP_BAND_NO_ALTERNATE");
	a.aligned16 = !(((uintptr_t) out->data & 15) || (out->stride & 15));
	a.tiles = a.tiles_x * ((out->height + oht - 1) / oht);
	const int grid = (a.tiles + 7) / 8 * 8;
	const size_t lds = f3_lds_bytes(oht);
	Gate gate("reduce_fused_u8x3_mfma");
	// two tiles a CU or fewer: two row groups in flight a lane and the horizontal walk's LDS reads all up front
	// (0.0484 -> 0.0456 ms on 8192 x 8192 x 3); more: three blocks a CU with one group in flight (16384 x 16384 x 3:
	// 0.187 ms against 0.206 the other way round) -- profiles/r06e_reduce_rgb.txt
	const bool deep = getenv("VIPS_HIP_FUSED3_DEEP") ? atoi(getenv("VIPS_HIP_FUSED3_DEEP")) != 0 : a.tiles <= 512;
#define F3_GO(DD, NBB, OCC, FF) \
	hipLaunchKernelGGL((reduce_fused_u8x3_mfma<DD, NBB, OCC, FF>), dim3(grid), dim3(FUSED_THREADS), lds, stream(), a, d_tables)
	if (D == 6 && deep)
		F3_GO(6, 2, 2, 0);
	else if (D == 6)
		F3_GO(6, 1, 3, 2);
	else if (deep)
		F3_GO(7, 2, 2, 0);
	else
		F3_GO(7, 1, 3, 2);
#undef F3_GO
	VH_CHECK(hipGetLastError());
	return 0;
}

template <int S, int D>
static int launch_fused(const FusedArgs &args, int tiles, const std::vector<unsigned int> &pairs_v,
	const std::vector<unsigned int> &pairs_h)
{
	FusedCoefs<S, D> k;
	const int np = D * (S / 2);
	for (int q = 0; q < np; q++) {
		k.cv[q] = pairs_v[q];
		// tap k' of the flipped problem is tap S*D-1-k': reverse the pair order and swap halves
		const unsigned int p = pairs_v[np - 1 - q];
		k.cv_flip[q] = (p >> 16) | (p << 16);
		k.ch[q] = pairs_h[q];
	}
	Gate gate("reduce_fused_u8");
	const int grid = (tiles + 7) / 8 * 8; // XCD remap wants a multiple of 8
	hipLaunchKernelGGL((reduce_fused_u8x4<S, D>), dim3(grid), dim3(FUSED_THREADS), 0, stream(), args, k);
	VH_CHECK(hipGetLastError());
	return 0;
}

// ------------------------------------------------- vertical uchar fast kernels
//
// reducev and shrinkv never look across a scanline, so a uchar image of any band count
// is a byte array per row: each thread owns DW consecutive dwords of the row (16 bytes
// when the geometry allows -> 1 KiB contiguous per wave per row) and walks the taps.

struct VerticalArgs {
	const unsigned char *in; // already offset to the first column of the rect
	unsigned char *out;
	long long in_stride, out_stride;
	int in_top, im_height;
	int out_top, out_height;
	int ndw; // dwords per row
};

// reducev.cpp:418-459 / reducev_hwy.cpp:94-268: sum_i k[i] * in[x + i * lskip], +2048, >>12,
// saturate.  Rows are taken in pairs so one v_dot2 does two taps of one byte lane.
template <int DW>
__global__ void __launch_bounds__(256)
reducev_u8_kernel(VerticalArgs a, int n_point, const ReducePos *__restrict__ pos,
	const short *__restrict__ table, int gx, int band)
{
	// Neighbouring output rows share most of their input rows, and an L2 is per XCD: block b
	// runs on XCD b % 8, so give each XCD one contiguous band of output rows (a row-major
	// grid would make all 8 L2s fetch every input row).
	const int local = blockIdx.x / 8;
	const int yb = local / gx;
	const int t = (local - yb * gx) * blockDim.x + threadIdx.x;
	const int y = (blockIdx.x % 8) * band + yb;
	if (t * DW >= a.ndw || y >= a.out_height)
		return;
	constexpr int BATCH = 8; // rows fetched before any is used: 8 loads in flight per lane
	{
		const ReducePos p = pos[y];
		const short *c = table + (size_t) p.phase * n_point;
		int acc[DW][4];
#pragma unroll
		for (int w = 0; w < DW; w++)
#pragma unroll
			for (int k = 0; k < 4; k++)
				acc[w][k] = 0;
		for (int i0 = 0; i0 < n_point; i0 += BATCH) {
			unsigned int v[BATCH][DW];
#pragma unroll
			for (int j = 0; j < BATCH; j++) {
				// past the last tap: re-fetch the last row (the window need not hold more), coefficient 0
				const int r = min(max(p.first + min(i0 + j, n_point - 1), 0), a.im_height - 1) - a.in_top;
				const unsigned int *pr = (const unsigned int *) (a.in + r * a.in_stride) + t * DW;
				if (DW == 4) {
					const uint4 x = *reinterpret_cast<const uint4 *>(pr);
					v[j][0] = x.x, v[j][1 % DW] = x.y, v[j][2 % DW] = x.z, v[j][3 % DW] = x.w;
				}
				else {
#pragma unroll
					for (int w = 0; w < DW; w++)
						v[j][w] = pr[w];
				}
			}
#pragma unroll
			for (int j = 0; j < BATCH; j += 2) {
				const int i = i0 + j;
				const unsigned int lo = i < n_point ? (unsigned short) c[i] : 0u;
				const unsigned int hi = i + 1 < n_point ? (unsigned short) c[i + 1] : 0u;
				const unsigned int coef = lo | (hi << 16);
#pragma unroll
				for (int w = 0; w < DW; w++)
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const unsigned int pair = __builtin_amdgcn_perm(v[j + 1][w], v[j][w],
							0x0c000c00u | (unsigned) k | ((4u + k) << 16));
						acc[w][k] = dot2(pair, coef, acc[w][k]);
					}
			}
		}
		unsigned int *dst = (unsigned int *) (a.out + (long long) y * a.out_stride) + t * DW;
		unsigned int o[DW];
#pragma unroll
		for (int w = 0; w < DW; w++)
			o[w] = (unsigned) fin_u8(acc[w][0]) | ((unsigned) fin_u8(acc[w][1]) << 8) |
				((unsigned) fin_u8(acc[w][2]) << 16) | ((unsigned) fin_u8(acc[w][3]) << 24);
		if (DW == 4)
			*reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1 % DW], o[2 % DW], o[3 % DW]);
		else {
#pragma unroll
			for (int w = 0; w < DW; w++)
				dst[w] = o[w];
		}
	}
}

// shrinkv.c:158-165,218-228 / shrinkv_hwy.cpp:90-203: column sums of vshrink rows, then
// ((sum + vshrink/2) * (2^32 / (256 * vshrink))) >> 24 in unsigned 32-bit arithmetic.
constexpr int SHRINKV_MAXB = 64; // images per launch (blockIdx.z)

// the images of a launch, read where they lie in the kernarg segment
struct ShrinkvPtrs {
	const unsigned char *in[SHRINKV_MAXB]; // already offset to the first column of the rect
	unsigned char *out[SHRINKV_MAXB];
};

template <int DW>
__global__ void __launch_bounds__(256)
shrinkv_u8_kernel(ShrinkvPtrs ptrs_by_value, VerticalArgs a, int vshrink, unsigned int multiplier)
{
	(void) ptrs_by_value;
	typedef const unsigned long long __attribute__((address_space(4))) *KernargPtrs;
	const KernargPtrs kp = (KernargPtrs) __builtin_amdgcn_kernarg_segment_ptr();
	// (pointers made from integers are generic to the compiler: say they are global)
	typedef const unsigned char __attribute__((address_space(1))) *GlobalIn;
	typedef unsigned char __attribute__((address_space(1))) *GlobalOut;
	const GlobalIn in = (GlobalIn) kp[blockIdx.z];
	const GlobalOut out = (GlobalOut) kp[SHRINKV_MAXB + blockIdx.z];
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t * DW >= a.ndw)
		return;
	const unsigned int amend = vshrink / 2;
	for (int y = blockIdx.y; y < a.out_height; y += gridDim.y) {
		const int y0 = (a.out_top + y) * vshrink;
		// even / odd bytes as two u16 lanes each: packed adds, no carries for vshrink <= 256
		unsigned int even[DW], odd[DW];
#pragma unroll
		for (int w = 0; w < DW; w++)
			even[w] = odd[w] = 0;
		for (int i = 0; i < vshrink; i++) {
			const int row = min(y0 + i, a.im_height - 1) - a.in_top;
			typedef unsigned int sv_uint4 __attribute__((ext_vector_type(4)));
			const unsigned int __attribute__((address_space(1))) *p =
				(const unsigned int __attribute__((address_space(1))) *) (in + row * a.in_stride) + t * DW;
			unsigned int v[DW];
			if (DW == 4) {
				const sv_uint4 x = *(const sv_uint4 __attribute__((address_space(1))) *) p;
				v[0] = x.x, v[1 % DW] = x.y, v[2 % DW] = x.z, v[3 % DW] = x.w;
			}
			else {
#pragma unroll
				for (int w = 0; w < DW; w++)
					v[w] = p[w];
			}
#pragma unroll
			for (int w = 0; w < DW; w++) {
				even[w] += v[w] & 0x00ff00ffu;
				odd[w] += (v[w] >> 8) & 0x00ff00ffu;
			}
		}
		unsigned int __attribute__((address_space(1))) *dst =
			(unsigned int __attribute__((address_space(1))) *) (out + (long long) y * a.out_stride) + t * DW;
		unsigned int o[DW];
#pragma unroll
		for (int w = 0; w < DW; w++) {
			const unsigned int b0 = (((even[w] & 0xffffu) + amend) * multiplier) >> 24;
			const unsigned int b2 = (((even[w] >> 16) + amend) * multiplier) >> 24;
			const unsigned int b1 = (((odd[w] & 0xffffu) + amend) * multiplier) >> 24;
			const unsigned int b3 = (((odd[w] >> 16) + amend) * multiplier) >> 24;
			o[w] = (b0 & 0xffu) | ((b1 & 0xffu) << 8) | ((b2 & 0xffu) << 16) | (b3 << 24);
		}
		if (DW == 4) {
			typedef unsigned int sv_uint4 __attribute__((ext_vector_type(4)));
			const sv_uint4 ov = { o[0], o[1 % DW], o[2 % DW], o[3 % DW] };
			*(sv_uint4 __attribute__((address_space(1))) *) dst = ov;
		}
		else {
#pragma unroll
			for (int w = 0; w < DW; w++)
				dst[w] = o[w];
		}
	}
}

// Common geometry of the vertical fast paths: the rect's rows as dword arrays.
static bool vertical_args(const VipsHipRegion *in, const VipsHipRegion *out, VerticalArgs *a, int *dw)
{
	const int bands = in->bands;
	const long long nbytes = (long long) out->width * bands;
	const unsigned char *src = (const unsigned char *) in->data + (size_t) (out->left - in->left) * bands;
	if (nbytes & 3)
		return false;
	if (((uintptr_t) src & 3) || (in->stride & 3) || ((uintptr_t) out->data & 3) || (out->stride & 3))
		return false;
	a->in = src;
	a->out = (unsigned char *) out->data;
	a->in_stride = (long long) in->stride;
	a->out_stride = (long long) out->stride;
	a->in_top = in->top;
	a->im_height = in->im_height;
	a->out_top = out->top;
	a->out_height = out->height;
	a->ndw = (int) (nbytes >> 2);
	const bool wide = !(nbytes & 15) && !((uintptr_t) src & 15) && !(in->stride & 15) &&
		!((uintptr_t) out->data & 15) && !(out->stride & 15);
	*dw = wide ? 4 : 1;
	return true;
}

// The matrix-core streaming kernel: integer-8 shrink, one phase, rows of whole 8-byte columns.
static int reducev_stream_try(const _VipsHipReduce *rc, const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	if (getenv("VIPS_HIP_NO_MFMA"))
		return 0;
	_VipsHipReduce *r = const_cast<_VipsHipReduce *>(rc);
	const long long nbytes = (long long) out->width * in->bands;
	const unsigned char *src = (const unsigned char *) in->data + (size_t) (out->left - in->left) * in->bands;
	if ((nbytes & 7) || ((uintptr_t) src & 7) || (in->stride & 7) || ((uintptr_t) out->data & 7) ||
		(out->stride & 7))
		return 0;
	if (!(in->stride > 0 && (long long) in->stride * in->height < (1LL << 31)))
		return 0;
	std::vector<ReducePos> pv;
	reduce_positions(r, out->top, out->height, tile, pv);
	int fy0, sy, phase;
	if (!positions_regular(pv, &fy0, &sy, &phase) || (out->height > 1 && sy != 8))
		return 0;
	const int n = effective_taps(r, phase);
	const int D = (n + 7) / 8;
	if (D != 6 && D != 7)
		return 0;
	const short *c = &r->matrixs[(size_t) phase * r->n_point];
	std::vector<int> taps(8 * D, 0);
	long long abs_sum = 0;
	int abs_max = 0;
	for (int k = 0; k < 8 * D; k++) {
		if (k < r->n_point)
			taps[k] = c[k];
		const int av = taps[k] < 0 ? -taps[k] : taps[k];
		abs_sum += av;
		abs_max = av > abs_max ? av : abs_max;
	}
	if (!(abs_max < 2048 && abs_sum * 255 < (1 << 23)))
		return 0;
	const MfmaTables *d_tables;
	{
		std::lock_guard<std::mutex> lock(r->mutex);
		auto key = std::make_tuple(-4, phase, 8 * D);
		auto it = r->pos_cache.find(key);
		if (it == r->pos_cache.end()) {
			MfmaTables tab;
			mfma_build_tables(taps, taps, D, &tab);
			void *d = upload(&tab, sizeof(tab));
			if (!d)
				return -1;
			r->pos_cache[key] = (ReducePos *) d;
			d_tables = (const MfmaTables *) d;
		}
		else
			d_tables = (const MfmaTables *) it->second;
	}
	VStreamArgs a;
	a.in = src;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_top = in->top;
	a.im_height = in->im_height;
	a.out_height = out->height;
	a.fy0 = fy0;
	a.row_u2 = (int) (nbytes >> 3);
	a.tiles_x = (a.row_u2 + FUSED_THREADS - 1) / FUSED_THREADS;
	// two residency rounds of tiles (no LDS staging here, so several rounds cost nothing)
	int rows_of_tiles = 2048 / a.tiles_x;
	if (rows_of_tiles < 1)
		rows_of_tiles = 1;
	int oht = (out->height + rows_of_tiles - 1) / rows_of_tiles;
	// (a tile re-reads 8 (D - 1) rows of the one above: 32 output rows a tile where that still leaves 1.5 tiles a
	// CU -- 8192 x 8192 x 3 by 8: 0.0385 -> 0.0361 ms, profiles/r05o_reducev8_oht.txt -- else 16)
	if (oht < 32)
		oht = a.tiles_x * ((out->height + 31) / 32) >= 384 ? 32 : oht < 16 ? 16 : oht;
	if (const char *e = getenv("VIPS_HIP_REDUCEV8_OHT"))
		oht = atoi(e) > 0 ? atoi(e) : oht;
	a.oht = oht;
	a.alternate = !getenv("VIPS_HIP_BAND_NO_ALTERNATE");
	const int tiles_y = (out->height + oht - 1) / oht;
	a.tiles = a.tiles_x * tiles_y;
	const int grid = (a.tiles + 7) / 8 * 8;
	Gate gate("reducev_u8_mfma");
	// at most a tile a CU (images of a few tens of MB): four row groups in flight a lane, two blocks a CU's worth of
	// registers -- 4096 x 4096 x 3: 0.0166 -> 0.0129 ms; from 1.5 tiles a CU on it changes nothing
	// (profiles/r06f_reducev_nb.txt)
	const bool deep = getenv("VIPS_HIP_REDUCEV8_NB") ? atoi(getenv("VIPS_HIP_REDUCEV8_NB")) == 4 : a.tiles <= 256;
	if (D == 6 && deep)
		hipLaunchKernelGGL((reducev_u8_mfma<6, 4, 2>), dim3(grid), dim3(FUSED_THREADS), 0, stream(), a, d_tables);
	else if (D == 6)
		hipLaunchKernelGGL((reducev_u8_mfma<6, 1, 4>), dim3(grid), dim3(FUSED_THREADS), 0, stream(), a, d_tables);
	else if (deep)
		hipLaunchKernelGGL((reducev_u8_mfma<7, 4, 2>), dim3(grid), dim3(FUSED_THREADS), 0, stream(), a, d_tables);
	else
		hipLaunchKernelGGL((reducev_u8_mfma<7, 1, 4>), dim3(grid), dim3(FUSED_THREADS), 0, stream(), a, d_tables);
	if (hipGetLastError() != hipSuccess) {
		error("reducev", "kernel launch failed");
		return -1;
	}
	return 1;
}

// vips_reduceh by 8 with one phase on a 3-band uchar region on the matrix cores (reduceh_u8x3_mfma).  1: launched;
// 0: not this kernel's case; -1: error.  The conditions are launch_fused_u8x3's.
int reduceh_u8x3_try(const _VipsHipReduce *rc, const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	if (getenv("VIPS_HIP_NO_MFMA") || getenv("VIPS_HIP_NO_REDUCEH3") || in->bands != F3_BANDS || out->bands != F3_BANDS)
		return 0;
	{
		// from ~100 MB of input on (8192 x 8192 x 3: 0.061 -> 0.051 ms, 16384 x 16384 x 3: 0.291 -> 0.203); below, the
		// packed vector-ALU kernel's few long blocks are the faster (4096 x 4096 x 3: 0.0216 against 0.0233 ms)
		const long long min_bytes = getenv("VIPS_HIP_REDUCEH3_MIN") ? atoll(getenv("VIPS_HIP_REDUCEH3_MIN")) : 96LL << 20;
		if ((long long) out->height * in->width * F3_BANDS < min_bytes)
			return 0;
	}
	_VipsHipReduce *r = const_cast<_VipsHipReduce *>(rc);
	if (!(in->stride > 0 && (long long) in->stride * in->height < (1LL << 31)))
		return 0;
	if (((uintptr_t) in->data & 3) || (in->stride & 3) || out->top < in->top || out->top + out->height > in->top + in->height)
		return 0;
	std::vector<ReducePos> ph;
	reduce_positions(r, out->left, out->width, tile, ph);
	int fx0, sx, phase;
	if (!positions_regular(ph, &fx0, &sx, &phase) || (out->width > 1 && sx != 8))
		return 0;
	const int nh = effective_taps(r, phase);
	const int D = (nh + 7) / 8;
	if ((D != 6 && D != 7) || F3_BANDS * (8 * (F3_OWT - 1) + nh) > F3_ROW)
		return 0;
	const short *c = &r->matrixs[(size_t) phase * r->n_point];
	std::vector<int> taps(8 * D, 0);
	long long abs_sum = 0;
	int abs_max = 0;
	for (int k = 0; k < 8 * D; k++) {
		if (k < r->n_point)
			taps[k] = c[k];
		const int av = taps[k] < 0 ? -taps[k] : taps[k];
		abs_sum += av;
		abs_max = av > abs_max ? av : abs_max;
	}
	if (!(abs_max < 2048 && abs_sum * 255 < (1 << 23)))
		return 0;
	const int lo = in->left > 0 ? in->left : 0;
	const int hi1 = in->im_width < in->left + in->width ? in->im_width : in->left + in->width;
	const long long blo = 3LL * (lo - in->left), bhi = 3LL * (hi1 - in->left), tb0 = 3LL * ((long long) fx0 - in->left);
	if ((blo & 3) || (bhi & 3) || (tb0 & 3) || bhi - blo < 8 || tb0 < -(1LL << 30) || tb0 > (1LL << 30))
		return 0;
	const MfmaTables *d_tables;
	{
		std::lock_guard<std::mutex> lock(r->mutex);
		auto key = std::make_tuple(-5, phase, 8 * D);
		auto it = r->pos_cache.find(key);
		if (it == r->pos_cache.end()) {
			MfmaTables tab;
			mfma_build_tables(taps, taps, D, &tab);
			void *d = upload(&tab, sizeof(tab));
			if (!d)
				return -1;
			r->pos_cache[key] = (ReducePos *) d;
			d_tables = (const MfmaTables *) d;
		}
		else
			d_tables = (const MfmaTables *) it->second;
	}
	RhIArgs a;
	a.in = (const unsigned char *) in->data + (size_t) (out->top - in->top) * in->stride;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.blo = (int) blo;
	a.bhi = (int) bhi;
	a.tile_b0 = (int) tb0;
	a.out_width = out->width;
	a.rows = out->height;
	a.tiles_x = (out->width + F3_OWT - 1) / F3_OWT;
	// no row is read twice whatever the height, and short tiles -- many blocks -- are what hides the walk: 16 rows
	// (8192 x 8192 x 3: 8 rows 0.0539 ms, 16 0.0510, 24 0.0527, 64 0.0664; profiles/r06l_reduceh3*.txt)
	int oht = 16;
	if (const char *e = getenv("VIPS_HIP_REDUCEH3_OHT"))
		oht = atoi(e) >= 8 && atoi(e) <= 128 ? atoi(e) / 8 * 8 : oht;
	a.oht = oht;
	a.tiles = a.tiles_x * ((out->height + oht - 1) / oht);
	a.aligned16 = !(((uintptr_t) out->data & 15) || (out->stride & 15));
	const int grid = (a.tiles + 7) / 8 * 8;
	const size_t lds = rh3_lds_bytes(oht);
	Gate gate("reduceh_u8x3_mfma");
	if (D == 6)
		hipLaunchKernelGGL((reduceh_u8x3_mfma<6>), dim3(grid), dim3(FUSED_THREADS), lds, stream(), a, d_tables);
	else
		hipLaunchKernelGGL((reduceh_u8x3_mfma<7>), dim3(grid), dim3(FUSED_THREADS), lds, stream(), a, d_tables);
	if (hipGetLastError() != hipSuccess) {
		error("reduceh", "kernel launch failed");
		return -1;
	}
	return 1;
}

int reducev_u8_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const ReducePos *pos, const short *table, int tile)
{
	{
		const int done = reducev_stream_try(r, in, out, tile);
		if (done != 0)
			return done;
	}
	{
		// a coefficient row per output row: the banded matrix on the matrix cores (reduce_band.hip)
		const int done = reducev_band_try(const_cast<_VipsHipReduce *>(r), in, out, tile);
		if (done != 0)
			return done;
	}
	{
		// ... or stream down the rows once on the vector ALU (resample16.hip)
		const int done = reducev8_stream_try(const_cast<_VipsHipReduce *>(r), in, out, tile);
		if (done != 0)
			return done;
	}
	VerticalArgs a;
	int dw;
	if (!vertical_args(in, out, &a, &dw))
		return 0;
	const int threads = (a.ndw + dw - 1) / dw;
	dim3 block(256, 1, 1);
	const int gx = (threads + 255) / 256;
	const int band = (out->height + 7) / 8;
	if ((long long) gx * band * 8 > 0x7fffffffLL)
		return 0;
	dim3 grid(gx * band * 8, 1, 1);
	Gate gate("reducev_u8");
	if (dw == 4)
		hipLaunchKernelGGL(reducev_u8_kernel<4>, grid, block, 0, stream(), a, r->n_point, pos, table, gx,
			band);
	else
		hipLaunchKernelGGL(reducev_u8_kernel<1>, grid, block, 0, stream(), a, r->n_point, pos, table, gx,
			band);
	if (hipGetLastError() != hipSuccess) {
		error("reducev", "kernel launch failed");
		return -1;
	}
	return 1;
}

// vips_shrinkv on n uchar rects of one geometry: one launch per SHRINKV_MAXB of them
int shrinkv_u8_batch_try(int vshrink, const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n)
{
	if (n < 1 || vshrink > 256)
		return 0;
	VerticalArgs a;
	int dw;
	if (!vertical_args(in[0], out[0], &a, &dw))
		return 0;
	std::vector<VerticalArgs> each(n);
	for (int i = 0; i < n; i++) {
		int dwi;
		if (!vertical_args(in[i], out[i], &each[i], &dwi))
			return 0;
		const VerticalArgs &b = each[i];
		if (b.in_stride != a.in_stride || b.out_stride != a.out_stride || b.in_top != a.in_top ||
			b.im_height != a.im_height || b.out_top != a.out_top || b.out_height != a.out_height || b.ndw != a.ndw)
			return 0;
		dw = dwi < dw ? dwi : dw;
	}
	const unsigned int multiplier = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) vshrink));
	const int threads = (a.ndw + dw - 1) / dw;
	dim3 block(256, 1, 1);
	Gate gate("shrinkv_u8");
	for (int base = 0; base < n; base += SHRINKV_MAXB) {
		const int count = n - base < SHRINKV_MAXB ? n - base : SHRINKV_MAXB;
		ShrinkvPtrs p;
		memset(&p, 0, sizeof(p));
		for (int i = 0; i < count; i++) {
			p.in[i] = each[base + i].in;
			p.out[i] = each[base + i].out;
		}
		dim3 grid((threads + 255) / 256, a.out_height < 32768 ? a.out_height : 32768, count);
		if (dw == 4)
			hipLaunchKernelGGL(shrinkv_u8_kernel<4>, grid, block, 0, stream(), p, a, vshrink, multiplier);
		else
			hipLaunchKernelGGL(shrinkv_u8_kernel<1>, grid, block, 0, stream(), p, a, vshrink, multiplier);
		if (hipGetLastError() != hipSuccess) {
			error("shrinkv", "kernel launch failed");
			return -1;
		}
	}
	return 1;
}

int shrinkv_u8_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return shrinkv_u8_batch_try(vshrink, &in, &out, 1);
}


} // namespace vh

using namespace vh;

extern "C" {

int vips_hip_reduce_gen_tiled(const VipsHipReduce *reducev, const VipsHipReduce *reduceh,
	const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	const char *domain = "reduce";
	if (ensure_init())
		return -1;
	if (!reducev || !reduceh) {
		error(domain, "null reduce");
		return -1;
	}
	if (plan_device(domain, &reducev->device) || plan_device(domain, &reduceh->device))
		return -1;
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR ||
		(in->bands != 4 && in->bands != F3_BANDS) || out->bands != in->bands)
		return 1;
	const bool three = in->bands == F3_BANDS; // the matrix-core kernel for interleaved bands, or nothing
	if (in->im_height != reducev->in_size || out->im_height != reducev->out_size ||
		in->im_width != reduceh->in_size || out->im_width != reduceh->out_size) {
		error(domain, "region does not belong to an image of the size these reduces were built for");
		return -1;
	}
	if (((uintptr_t) in->data & 3) || (in->stride & 3) ||
		(!three && (((uintptr_t) out->data & 3) || (out->stride & 3))))
		return 1;

	// Geometry: both axes must step by the same even integer with one phase.
	std::vector<ReducePos> pv, ph;
	reduce_positions(reducev, out->top, out->height, tile, pv);
	reduce_positions(reduceh, out->left, out->width, 0, ph);
	int fy0, sy, phase_y, fx0, sx, phase_x;
	if (!positions_regular(pv, &fy0, &sy, &phase_y) || !positions_regular(ph, &fx0, &sx, &phase_x))
		return 1;
	if (out->height == 1)
		sy = sx;
	if (out->width == 1)
		sx = sy;
	if (sx != sy || sx < 2 || (sx & 1))
		return 1;
	const int S = sx;
	const int nv = effective_taps(reducev, phase_y);
	const int nh = effective_taps(reduceh, phase_x);
	const int nmax = nv > nh ? nv : nh;
	const int D = (nmax + S - 1) / S;
	if (!((S == 8 && (D == 6 || D == 7)) || (S == 4 && (D == 6 || D == 7)) ||
			(S == 2 && (D == 6 || D == 7))))
		return 1;
	if (three && (S != 8 || getenv("VIPS_HIP_NO_MFMA") || getenv("VIPS_HIP_NO_FUSED3")))
		return 1;

	// the input window must cover what the two gens need
	int need0, needn;
	vips_hip_reducev_need(reducev, out->top, out->height, &need0, &needn);
	if (need0 < in->top || need0 + needn > in->top + in->height) {
		error(domain, "input region too small: need rows %d..%d", need0, need0 + needn);
		return -1;
	}
	vips_hip_reduceh_need(reduceh, out->left, out->width, &need0, &needn);
	if (need0 < in->left || need0 + needn > in->left + in->width) {
		error(domain, "input region too small: need columns %d..%d", need0, need0 + needn);
		return -1;
	}

	std::vector<unsigned int> pairs_v, pairs_h;
	pack_pairs(reducev, phase_y, S * D, pairs_v);
	pack_pairs(reduceh, phase_x, S * D, pairs_h);

	FusedArgs args;
	args.in = (const unsigned char *) in->data;
	args.in_stride = (long long) in->stride;
	args.in_left = in->left;
	args.in_right = in->left + in->width;
	args.in_top = in->top;
	args.im_width = in->im_width;
	args.im_height = in->im_height;
	args.out = (unsigned char *) out->data;
	args.out_stride = (long long) out->stride;
	args.out_width = out->width;
	args.out_height = out->height;
	args.aligned8 = !(((uintptr_t) in->data & 7) || (in->stride & 7));
	args.small_window = in->stride > 0 && (long long) in->stride * in->height < (1LL << 31);
	{
		const int debug_bits = getenv("VIPS_HIP_FUSED_DEBUG") ? atoi(getenv("VIPS_HIP_FUSED_DEBUG")) : 0;
		args.debug = debug_bits;
	}
	args.fx0 = fx0;
	args.fy0 = fy0;
	args.xshift = 0;
	{
		// the MFMA kernel's whole-pair loads (load_rows) need two columns to clamp a pair to
		const int lo = in->left > 0 ? in->left : 0;
		const int hi1 = in->im_width < in->left + in->width ? in->im_width : in->left + in->width;
		args.pairs = hi1 - lo >= 2;
	}
	args.stagger = 0;
	args.burst_rows = 1 << 20;
	args.owt = FUSED_SPAN / S - D + 1;
	args.tiles_x = (out->width + args.owt - 1) / args.owt;
	const int mfma_span = 2 * FUSED_THREADS;
	const int mfma_max_oht = MfmaGeo<FUSED_THREADS>::MAX_OHT;

	_VipsHipReduce *rv = const_cast<_VipsHipReduce *>(reducev);
	// S = 8: both passes on the matrix cores when the exactness bounds hold
	// (|c| < 2048 is an exact half, sum |c| * 255 < 2^23 keeps 2n + 1 in 24 bits)
	if (S == 8 && args.small_window && (args.pairs || three) && !getenv("VIPS_HIP_NO_MFMA")) {
		const short *c = &rv->matrixs[(size_t) phase_y * rv->n_point];
		const short *ch = &reduceh->matrixs[(size_t) phase_x * reduceh->n_point];
		std::vector<int> taps(8 * D, 0), taps_h(8 * D, 0);
		long long abs_sum = 0, abs_sum_h = 0;
		int abs_max = 0;
		for (int k = 0; k < 8 * D; k++) {
			if (k < rv->n_point)
				taps[k] = c[k];
			if (k < reduceh->n_point)
				taps_h[k] = ch[k];
			const int av = taps[k] < 0 ? -taps[k] : taps[k];
			const int ah = taps_h[k] < 0 ? -taps_h[k] : taps_h[k];
			abs_sum += av;
			abs_sum_h += ah;
			abs_max = av > abs_max ? av : abs_max;
			abs_max = ah > abs_max ? ah : abs_max;
		}
		if (three && !(abs_max < 2048 && abs_sum * 255 < (1 << 23) && abs_sum_h * 255 < (1 << 23)))
			return 1;
		if (three) {
			const MfmaTables *d_tables;
			{
				std::lock_guard<std::mutex> lock(rv->mutex);
				auto key = std::make_tuple(-3, phase_y * 128 + phase_x, 8 * D);
				auto it = rv->pos_cache.find(key);
				if (it == rv->pos_cache.end()) {
					MfmaTables tab;
					mfma_build_tables(taps, taps_h, D, &tab);
					void *d = upload(&tab, sizeof(tab));
					if (!d)
						return -1;
					rv->pos_cache[key] = (ReducePos *) d;
					d_tables = (const MfmaTables *) d;
				}
				else
					d_tables = (const MfmaTables *) it->second;
			}
			return launch_fused_u8x3(D, nh, in, out, fx0, fy0, d_tables);
		}
		if (abs_max < 2048 && abs_sum * 255 < (1 << 23) && abs_sum_h * 255 < (1 << 23)) {
			// Tile height: ONE residency round (256 CUs x 4 blocks) when the staged rows fit in
			// LDS -- every tile then ends, and bursts its output, at the same time, and
			// neighbouring tiles read their shared halos in lock-step (L2 hits); else the
			// smallest whole number of rounds.
			{
				// Line-aligned tiles: a wave's row segment (64 lanes x 8 bytes) that starts on a
				// 128-byte line costs the memory pipe 4 line requests instead of 5, and on this
				// part the requests a CU can issue, not HBM, bound the stream (tools/hbm_probe2:
				// 2 KB strips at a 1888-byte pitch 5.8 TB/s requested, at a 2048-byte pitch 6.5,
				// 7.1 with nt loads).  So: tile pitch a whole number of lines (owt a multiple
				// of 4 -> 32 * owt bytes), lanes start at the line that holds the first tap.
				// Measured on C2 with whole-pair edge loads: 0.1936 ms aligned (999 tiles of 56
				// columns: more halo), 0.1902 ms with tiles that start at their first tap (1015
				// tiles of 59) -- so alignment is opt-in (VIPS_HIP_FUSED_ALIGN=1).
				// (round 3: with the fix-up at the point of use and nt loads the aligned layout is the
				// faster one on every box measured -- 0.1934 against 0.1979, 0.1920 against 0.1928 --
				// so it is the default where base and stride allow; VIPS_HIP_FUSED_ALIGN=0 for the other)
				const bool align = !(getenv("VIPS_HIP_FUSED_ALIGN") && atoi(getenv("VIPS_HIP_FUSED_ALIGN")) == 0);
				if (align && !(in->stride & 127)) {
					const long long addr = (long long) (uintptr_t) in->data + 4LL * ((long long) fx0 - in->left);
					const int off = (int) (((addr % 128) + 128) % 128); // bytes past a line start
					const int owt = ((mfma_span - off / 4) / S - D + 1) & ~3;
					if (!(off & 15) && owt >= 32) {
						args.xshift = off / 4;
						args.owt = owt;
					}
				}
				// profiling knob: narrower tiles
				const int owt_env = getenv("VIPS_HIP_FUSED_OWT") ? atoi(getenv("VIPS_HIP_FUSED_OWT")) : 0;
				if (owt_env > 0 && owt_env < args.owt)
					args.owt = owt_env;
				args.tiles_x = (out->width + args.owt - 1) / args.owt;
				const int slots = getenv("VIPS_HIP_FUSED_CAP") ? atoi(getenv("VIPS_HIP_FUSED_CAP")) : 256 * 4;
				const int base = slots / args.tiles_x > 0 ? slots / args.tiles_x : 1;
				int oht = out->height;
				for (int k = 1; k <= 4096; k++) {
					oht = (out->height + base * k - 1) / (base * k);
					if (oht <= mfma_max_oht)
						break;
				}
				args.oht = oht < 1 ? 1 : oht;
			}
			{
				const char *e = getenv("VIPS_HIP_FUSED_STAGGER");
				args.stagger = e ? atoi(e) & 7 : 0;
				e = getenv("VIPS_HIP_FUSED_BURST");
				const int burst = e ? atoi(e) : 0;
				args.burst_rows = burst > 0 ? (burst + 7) & ~7 : mfma_max_oht + 8;
			}
			const int tiles_y = (out->height + args.oht - 1) / args.oht;
			const int tiles = args.tiles_x * tiles_y;
			args.tiles = tiles;
			const MfmaTables *d_tables;
			{
				std::lock_guard<std::mutex> lock(rv->mutex);
				auto key = std::make_tuple(-3, phase_y * 128 + phase_x, 8 * D);
				auto it = rv->pos_cache.find(key);
				if (it == rv->pos_cache.end()) {
					MfmaTables tab;
					mfma_build_tables(taps, taps_h, D, &tab);
					void *d = upload(&tab, sizeof(tab));
					if (!d)
						return -1;
					rv->pos_cache[key] = (ReducePos *) d;
					d_tables = (const MfmaTables *) d;
				}
				else
					d_tables = (const MfmaTables *) it->second;
			}
			// round 6: whole images whose width is a multiple of 512 -- tiles without a horizontal halo
			// (reduce_fused_u8x4_mfma_x), the straddling outputs by a second small kernel
			if (D == 6) {
				const int r = launch_fused_mfma_x(args, in, out, d_tables);
				if (r <= 0)
					return r;
			}
			if (D == 6)
				return launch_fused_mfma<6>(args, tiles, d_tables);
			if (D == 7)
				return launch_fused_mfma<7>(args, tiles, d_tables);
		}
	}

	if (three)
		return 1;
	// The VALU kernel.  Tile height: tall tiles amortise the (D-1)*S-row vertical halo; short
	// tiles balance the 256 CUs better.  Measured on C2: two residency rounds (256 CUs x
	// 4 resident blocks x 2) is the sweet spot -- 0.249 ms vs 0.259 ms at one round.
	{
		const int capacity = 256 * 8;
		int rows_of_tiles = capacity / args.tiles_x;
		if (rows_of_tiles < 1)
			rows_of_tiles = 1;
		int oht = (out->height + rows_of_tiles - 1) / rows_of_tiles;
		if (oht < 32)
			oht = 32;
		args.oht = oht;
	}
	const int tiles_y = (out->height + args.oht - 1) / args.oht;
	const int tiles = args.tiles_x * tiles_y;
	args.tiles = tiles;

#define FUSED_CASE(SS, DD) \
	if (S == SS && D == DD) \
		return launch_fused<SS, DD>(args, tiles, pairs_v, pairs_h);
	FUSED_CASE(8, 6)
	FUSED_CASE(8, 7)
	FUSED_CASE(4, 6)
	FUSED_CASE(4, 7)
	FUSED_CASE(2, 6)
	FUSED_CASE(2, 7)
#undef FUSED_CASE
	return 1;
}

int vips_hip_reduce_gen(const VipsHipReduce *reducev, const VipsHipReduce *reduceh,
	const VipsHipRegion *in, const VipsHipRegion *out)
{
	return vips_hip_reduce_gen_tiled(reducev, reduceh, in, out, 0);
}

} // extern "C"
