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
	asm volatile("" : "+v"(s));
	return min(max(s, 0), 255);
}

constexpr int FUSED_THREADS = 256;
constexpr int FUSED_SPAN = 2 * FUSED_THREADS; // input columns per tile

struct FusedArgs {
	const unsigned char *in;
	long long in_stride;
	int in_left, in_top;   // origin of the input window
	int im_width, im_height;
	unsigned char *out;
	long long out_stride;
	int out_width, out_height; // region being generated
	int fx0, fy0;              // first tap (un-embedded input coords) of output (0, 0) of the region
	int owt, oht;              // tile size in output pixels
	int tiles_x, tiles;
	int aligned8;           // input base and stride are multiples of 8 bytes
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
	const short *__restrict__ table)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t * DW >= a.ndw)
		return;
	for (int y = blockIdx.y; y < a.out_height; y += gridDim.y) {
		const ReducePos p = pos[y];
		const short *c = table + (size_t) p.phase * n_point;
		int acc[DW][4];
#pragma unroll
		for (int w = 0; w < DW; w++)
#pragma unroll
			for (int k = 0; k < 4; k++)
				acc[w][k] = 0;
		for (int i = 0; i < n_point; i += 2) {
			const int ra = min(max(p.first + i, 0), a.im_height - 1) - a.in_top;
			const int rb = min(max(p.first + i + 1, 0), a.im_height - 1) - a.in_top;
			const unsigned int lo = (unsigned short) c[i];
			const unsigned int hi = i + 1 < n_point ? (unsigned short) c[i + 1] : 0u;
			const unsigned int coef = lo | (hi << 16);
			const unsigned int *pa = (const unsigned int *) (a.in + ra * a.in_stride) + t * DW;
			const unsigned int *pb = (const unsigned int *) (a.in + rb * a.in_stride) + t * DW;
			unsigned int va[DW], vb[DW];
			if (DW == 4) {
				const uint4 xa = *reinterpret_cast<const uint4 *>(pa);
				const uint4 xb = *reinterpret_cast<const uint4 *>(pb);
				va[0] = xa.x, va[1 % DW] = xa.y, va[2 % DW] = xa.z, va[3 % DW] = xa.w;
				vb[0] = xb.x, vb[1 % DW] = xb.y, vb[2 % DW] = xb.z, vb[3 % DW] = xb.w;
			}
			else {
#pragma unroll
				for (int w = 0; w < DW; w++) {
					va[w] = pa[w];
					vb[w] = pb[w];
				}
			}
#pragma unroll
			for (int w = 0; w < DW; w++)
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const unsigned int pair = __builtin_amdgcn_perm(vb[w], va[w],
						0x0c000c00u | (unsigned) k | ((4u + k) << 16));
					acc[w][k] = dot2(pair, coef, acc[w][k]);
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
template <int DW>
__global__ void __launch_bounds__(256)
shrinkv_u8_kernel(VerticalArgs a, int vshrink, unsigned int multiplier)
{
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
			const unsigned int *p = (const unsigned int *) (a.in + row * a.in_stride) + t * DW;
			unsigned int v[DW];
			if (DW == 4) {
				const uint4 x = *reinterpret_cast<const uint4 *>(p);
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
		unsigned int *dst = (unsigned int *) (a.out + (long long) y * a.out_stride) + t * DW;
		unsigned int o[DW];
#pragma unroll
		for (int w = 0; w < DW; w++) {
			const unsigned int b0 = (((even[w] & 0xffffu) + amend) * multiplier) >> 24;
			const unsigned int b2 = (((even[w] >> 16) + amend) * multiplier) >> 24;
			const unsigned int b1 = (((odd[w] & 0xffffu) + amend) * multiplier) >> 24;
			const unsigned int b3 = (((odd[w] >> 16) + amend) * multiplier) >> 24;
			o[w] = (b0 & 0xffu) | ((b1 & 0xffu) << 8) | ((b2 & 0xffu) << 16) | (b3 << 24);
		}
		if (DW == 4)
			*reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1 % DW], o[2 % DW], o[3 % DW]);
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

int reducev_u8_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const ReducePos *pos, const short *table)
{
	VerticalArgs a;
	int dw;
	if (!vertical_args(in, out, &a, &dw))
		return 0;
	const int threads = (a.ndw + dw - 1) / dw;
	dim3 block(256, 1, 1);
	dim3 grid((threads + 255) / 256, out->height < 32768 ? out->height : 32768, 1);
	Gate gate("reducev_u8");
	if (dw == 4)
		hipLaunchKernelGGL(reducev_u8_kernel<4>, grid, block, 0, stream(), a, r->n_point, pos, table);
	else
		hipLaunchKernelGGL(reducev_u8_kernel<1>, grid, block, 0, stream(), a, r->n_point, pos, table);
	if (hipGetLastError() != hipSuccess) {
		error("reducev", "kernel launch failed");
		return -1;
	}
	return 1;
}

int reduceh_u8_try(const _VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out,
	const ReducePos *pos, const short *table)
{
	return 0;
}

int shrinkv_u8_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	VerticalArgs a;
	int dw;
	if (vshrink > 256 || !vertical_args(in, out, &a, &dw))
		return 0;
	const unsigned int multiplier = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) vshrink));
	const int threads = (a.ndw + dw - 1) / dw;
	dim3 block(256, 1, 1);
	dim3 grid((threads + 255) / 256, out->height < 32768 ? out->height : 32768, 1);
	Gate gate("shrinkv_u8");
	if (dw == 4)
		hipLaunchKernelGGL(shrinkv_u8_kernel<4>, grid, block, 0, stream(), a, vshrink, multiplier);
	else
		hipLaunchKernelGGL(shrinkv_u8_kernel<1>, grid, block, 0, stream(), a, vshrink, multiplier);
	if (hipGetLastError() != hipSuccess) {
		error("shrinkv", "kernel launch failed");
		return -1;
	}
	return 1;
}

int shrinkh_u8_try(int hshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	return 0;
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
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR ||
		in->bands != 4 || out->bands != 4)
		return 1;
	if (in->im_height != reducev->in_size || out->im_height != reducev->out_size ||
		in->im_width != reduceh->in_size || out->im_width != reduceh->out_size) {
		error(domain, "region does not belong to an image of the size these reduces were built for");
		return -1;
	}
	if (((uintptr_t) in->data & 3) || (in->stride & 3) || ((uintptr_t) out->data & 3) ||
		(out->stride & 3))
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
	args.in_top = in->top;
	args.im_width = in->im_width;
	args.im_height = in->im_height;
	args.out = (unsigned char *) out->data;
	args.out_stride = (long long) out->stride;
	args.out_width = out->width;
	args.out_height = out->height;
	args.aligned8 = !(((uintptr_t) in->data & 7) || (in->stride & 7));
	args.fx0 = fx0;
	args.fy0 = fy0;
	args.owt = FUSED_SPAN / S - D + 1;
	args.tiles_x = (out->width + args.owt - 1) / args.owt;
	// Tile height: tall tiles amortise the (D-1)*S-row vertical halo (which the
	// serpentine walk turns into L2 / Infinity-Cache hits anyway); short tiles
	// balance the 256 CUs better.  Measured on C2: two residency waves (256 CUs x
	// 4 resident blocks x 2) is the sweet spot -- 0.249 ms vs 0.259 ms at one wave.
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
