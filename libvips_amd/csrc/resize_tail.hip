// vips_resize()'s tail on uchar images in one kernel: the residual vips_reducev, the integer
// vips_shrinkh and the residual vips_reduceh that follow the vertical box shrink
// (resample/resize.c:207-228 chains reducev.cpp:418-459, shrinkh.c:78-92 and
// reduceh.cpp:216-255 through three images; BASELINE configs 1 and 4, any band count 1..4).
//
// A block makes a TW x TH tile of the final image.  It stages the input rows and the byte span
// its taps reach in LDS (global_load_lds_dword: HBM -> LDS without registers, every load of the
// tile in flight at once; rows that are not dword aligned go through registers), then runs the
// three operations on the tile, each rounding to uchar exactly as the separate operations do:
//   1. reducev: thread = one dword column, walking down the tile's output rows four at a time;
//      two taps per v_dot2_i32_i16 (v_perm_b32 pairs the bytes of two rows); the result row y
//      overwrites staged row y of its own column, which no later output row reads (the first
//      tap row increases with y: checked on the host).  The tap count is a template parameter
//      (every odd count the kernels of reduce produce up to 25, a run-time loop beyond): all
//      LDS reads of four output rows are issued before the first dot product;
//   2. shrinkh: box sums of hshrink pixels, ((sum + h/2) * (2^32 / (256 h))) >> 24, to a second
//      LDS array (columns clamped to the shrunk image: the vips_embed COPY of reduceh);
//   3. reduceh: thread = (row group, pixel, band), taps from that array, (sum + 2048) >> 12, clip.
// The 25 MB + 6 MB of intermediate images per BASELINE config 4 thumbnail are never written.
// Tiles go to XCDs in bands of tile rows, so the rows two vertical neighbours share are read
// from HBM once per XCD.
#include "resample.h"
#include "reduce_u8.h"
#include "kernel_stmt.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace vh {

constexpr int TAIL_MAXB = 64; // images per launch (blockIdx.y)

// the images of a launch, read where they lie in the kernarg segment
struct TailPtrs {
	const unsigned char *in[TAIL_MAXB];
	unsigned char *out[TAIL_MAXB];
};

struct TailArgs {
	long long in_stride, out_stride;
	int width, height, bands;   // the input image (after shrinkv)
	int hshrink, shrunk_width;  // shrinkh
	unsigned int mult8;
	int n_v, n_h;
	int out_width, out_height;
	int tw, th;                 // tile of the output
	int nbx, nby, band;         // tiles across / down; tile rows per XCD
	int pitch;                  // dwords per staged row
	int s_pitch;                // bytes per shrunk row
	int max_rows;               // staged rows
	int np4;                    // dwords per row of the vertical coefficient table
	int off_cv, off_ch, off_first; // byte offsets in LDS of the tables (below)
	int aligned;                // rows start on dword boundaries: LDS-DMA staging
	int debug;                  // $VIPS_HIP_TAIL_DEBUG: skip 1 the staging, 2 reducev, 4 shrinkh, 8 reduceh (timing only)
};

constexpr int TAIL_NT = 256;

// One dword per lane from global memory straight into LDS (see convsep_stream.hip).
static __device__ __forceinline__ void tail_dma_dword(const unsigned char *src, unsigned int voff,
	unsigned int lds_dst)
{
	VH_LDS_DMA_DWORD(src, voff, lds_dst);
}

// (sum + 2048) >> 12, clip (templates.h:152-157).  The shifted value is made opaque before the
// clip for the reason given at fin_u8 in reduce_u8.hip (v_ashr_pk_u8_i32 keeps the upper half of
// its destination on gfx950, the compiler assumes it is zeroed; tests/test_abi.py checks the
// library for the instruction).
static __device__ __forceinline__ unsigned int tail_fin(int s)
{
	s = (s + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT;
	VH_VECTOR1(s);
	return (unsigned int) min(max(s, 0), 255);
}

typedef short tail_short2 __attribute__((ext_vector_type(2)));

// acc + lo16(pix) * lo16(coef) + hi16(pix) * hi16(coef)
static __device__ __forceinline__ int tail_dot2(unsigned int pix, unsigned int coef, int acc)
{
	return __builtin_amdgcn_sdot2(__builtin_bit_cast(tail_short2, pix), __builtin_bit_cast(tail_short2, coef), acc,
		false);
}

// the four bytes of rows w0 (tap 2p) and w1 (tap 2p + 1) against a coefficient pair
static __device__ __forceinline__ void tail_pair(unsigned int w0, unsigned int w1, unsigned int coef, int (&s)[4])
{
	s[0] = tail_dot2(__builtin_amdgcn_perm(w1, w0, 0x0c040c00u), coef, s[0]);
	s[1] = tail_dot2(__builtin_amdgcn_perm(w1, w0, 0x0c050c01u), coef, s[1]);
	s[2] = tail_dot2(__builtin_amdgcn_perm(w1, w0, 0x0c060c02u), coef, s[2]);
	s[3] = tail_dot2(__builtin_amdgcn_perm(w1, w0, 0x0c070c03u), coef, s[3]);
}

static __device__ __forceinline__ unsigned int tail_pack(const int (&s)[4])
{
	return tail_fin(s[0]) | (tail_fin(s[1]) << 8) | (tail_fin(s[2]) << 16) | (tail_fin(s[3]) << 24);
}

// reducev of ROWS output rows of one dword column; NV taps (0 = a.n_v at run time)
template <int NV, int ROWS>
static __device__ __forceinline__ void tail_vrows(const TailArgs &a, unsigned int *tile, const unsigned int *cvp,
	const int *first, int col, int y, int ny)
{
	int s[ROWS][4];
	if constexpr (NV > 0) {
		constexpr int NP = (NV + 1) / 2;
		unsigned int w[ROWS][NV];
		unsigned int c[ROWS][NP];
#pragma unroll
		for (int j = 0; j < ROWS; j++) {
			const int yy = min(y + j, ny - 1);
			const unsigned int *p = tile + first[yy] * a.pitch + col;
#pragma unroll
			for (int k = 0; k < NV; k++)
				w[j][k] = p[k * a.pitch];
#pragma unroll
			for (int k = 0; k < NP; k++)
				c[j][k] = cvp[yy * a.np4 + k];
		}
#pragma unroll
		for (int j = 0; j < ROWS; j++) {
			s[j][0] = s[j][1] = s[j][2] = s[j][3] = 0;
#pragma unroll
			for (int k = 0; k < NP; k++)
				tail_pair(w[j][2 * k], w[j][2 * k + 1 < NV ? 2 * k + 1 : 2 * k], c[j][k], s[j]);
		}
	}
	else {
		const unsigned int *p[ROWS];
		const unsigned int *c[ROWS];
#pragma unroll
		for (int j = 0; j < ROWS; j++) {
			const int yy = min(y + j, ny - 1);
			p[j] = tile + first[yy] * a.pitch + col;
			c[j] = cvp + yy * a.np4;
			s[j][0] = s[j][1] = s[j][2] = s[j][3] = 0;
		}
		for (int k = 0; k < a.n_v; k += 2) {
			const int k1 = k + 1 < a.n_v ? k + 1 : k; // the odd tap out pairs with itself, coefficient 0
#pragma unroll
			for (int j = 0; j < ROWS; j++)
				tail_pair(p[j][k * a.pitch], p[j][k1 * a.pitch], c[j][k >> 1], s[j]);
		}
	}
#pragma unroll
	for (int j = 0; j < ROWS; j++)
		if (y + j < ny)
			tile[(y + j) * a.pitch + col] = tail_pack(s[j]);
}

template <int NV>
__global__ void __launch_bounds__(TAIL_NT)
resize_tail_u8(TailPtrs ptrs_by_value, TailArgs a, const ReducePos *__restrict__ posv, const short *__restrict__ tabv,
	const ReducePos *__restrict__ posh, const short *__restrict__ tabh)
{
	VH_DYNAMIC_LDS(unsigned int, tail_lds);
	(void) ptrs_by_value;
	typedef const unsigned long long __attribute__((address_space(4))) *KernargPtrs;
	const KernargPtrs kp = (KernargPtrs) __builtin_amdgcn_kernarg_segment_ptr();
	const unsigned char *in = reinterpret_cast<const unsigned char *>(kp[blockIdx.y]);
	// (a pointer made from an integer is generic to the compiler: the stores say they are global)
	unsigned char __attribute__((address_space(1))) *out = (unsigned char __attribute__((address_space(1))) *) kp[TAIL_MAXB + blockIdx.y];
	unsigned int *tile = tail_lds;
	unsigned char *shrunk = reinterpret_cast<unsigned char *>(tail_lds + a.max_rows * a.pitch);
	// the tile's coefficients and positions: vertical taps of output row y as (tap 2p, tap 2p + 1)
	// pairs at cvp[y * np4 + p], horizontal taps of output pixel x at ch[x * n_h], first tap
	// rows (relative to the staged rows) at first[y], first tap columns at first[th + x]
	unsigned int *cvp = reinterpret_cast<unsigned int *>(reinterpret_cast<unsigned char *>(tail_lds) + a.off_cv);
	short *ch = reinterpret_cast<short *>(reinterpret_cast<unsigned char *>(tail_lds) + a.off_ch);
	int *first = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(tail_lds) + a.off_first);
	const int t = threadIdx.x;
	const int B = a.bands;

	// block -> tile: XCD x takes tile rows [x * band, (x + 1) * band)
	const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
	const int by = xcd * a.band + local / a.nbx;
	const int bx = local % a.nbx;
	if (by >= a.nby)
		return;
	const int y0 = by * a.th, ny = min(a.th, a.out_height - y0);
	const int x0 = bx * a.tw, nx = min(a.tw, a.out_width - x0);

	// columns of the shrunk image the tile's taps touch, and the input bytes under them
	const int c_lo = min(max(posh[x0].first, 0), a.shrunk_width - 1);
	const int c_hi = min(max(posh[x0 + nx - 1].first + a.n_h - 1, 0), a.shrunk_width - 1);
	const int ncol = c_hi - c_lo + 1;
	const int byte_lo = c_lo * a.hshrink * B;
	const int byte_hi = (min(c_hi * a.hshrink + a.hshrink - 1, a.width - 1) + 1) * B; // exclusive
	const int start_al = byte_lo & ~3;
	const int ndw = (byte_hi - start_al + 3) >> 2;
	// rows
	const int r_lo = posv[y0].first;
	const int nrows = posv[y0 + ny - 1].first + a.n_v - r_lo;

	// ---- stage: wave w takes rows w, w + 4, ...
	{
		const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
		const int lane = t & 63;
		if (a.debug & 1) {
		}
		else if (a.aligned) {
			// a row is ceil(ndw / 64) LDS-DMA loads
			const unsigned int lds_tile = VH_LDS_ADDR(tile);
			for (int r = wv; r < nrows; r += TAIL_NT / 64) {
				const int row = min(max(r_lo + r, 0), a.height - 1);
				const unsigned char *src = in + (long long) row * a.in_stride + start_al;
				for (int k = 0; k < ndw; k += 64) {
					if (k + lane < ndw)
						tail_dma_dword(src, (unsigned int) (k + lane) * 4u, lds_tile + (unsigned int) (r * a.pitch + k) * 4u);
				}
			}
		}
		else {
			// rows of any alignment: dword loads at byte addresses, the last bytes of a row singly
			// (nothing is read beyond the image)
			const int nbytes = byte_hi - start_al;
			const int nfull = nbytes >> 2;
			for (int r = wv; r < nrows; r += TAIL_NT / 64) {
				const int row = min(max(r_lo + r, 0), a.height - 1);
				const unsigned char *src = in + (long long) row * a.in_stride + start_al;
				for (int k = lane; k < nfull; k += 64) {
					unsigned int v;
					__builtin_memcpy(&v, src + 4 * k, 4);
					tile[r * a.pitch + k] = v;
				}
				if (lane < (nbytes & 3))
					reinterpret_cast<unsigned char *>(tile + r * a.pitch + nfull)[lane] = src[4 * nfull + lane];
			}
		}
		// the tables, while the tile is in flight
		const int np = (a.n_v + 1) >> 1;
		for (int i = t; i < ny * np; i += TAIL_NT) {
			const int y = i / np;
			const int k = 2 * (i - y * np);
			const short *c = tabv + (size_t) posv[y0 + y].phase * a.n_v;
			const unsigned int lo = (unsigned short) c[k];
			const unsigned int hi = k + 1 < a.n_v ? (unsigned short) c[k + 1] : 0u;
			cvp[y * a.np4 + (k >> 1)] = lo | (hi << 16);
		}
		for (int i = t; i < nx * a.n_h; i += TAIL_NT) {
			const int x = i / a.n_h;
			ch[i] = tabh[(size_t) posh[x0 + x].phase * a.n_h + (i - x * a.n_h)];
		}
		if (t < ny)
			first[t] = posv[y0 + t].first - r_lo;
		for (int i = t; i < nx; i += TAIL_NT)
			first[a.th + i] = posh[x0 + i].first;
		VH_WAIT_VMCNT(0);
	}
	__syncthreads();

	// ---- 1. reducev, in place.  Several output rows at a time: their tap loops are independent,
	// so their LDS reads are in flight together (a lone loop waits the LDS latency on every tap),
	// and the result rows are stored after all the sums (rows y..y+3 < first[y + 4]).
	if (!(a.debug & 2)) {
		constexpr int ROWS = NV > 16 ? 2 : 4;
		for (int col = t; col < ndw; col += TAIL_NT)
			for (int y = 0; y < ny; y += ROWS)
				tail_vrows<NV, ROWS>(a, tile, cvp, first, col, y, ny);
	}
	__syncthreads();

	// ---- 2. shrinkh: thread = one band element of the shrunk row, four rows at a time
	if (!(a.debug & 4)) {
		const unsigned char *rows = reinterpret_cast<const unsigned char *>(tile);
		const int per_row = ncol * B;
		const unsigned int amend = (unsigned int) (a.hshrink / 2);
		const int row_bytes = a.pitch * 4;
		for (int e = t; e < per_row; e += TAIL_NT) {
			const int c = e / B;
			const int b = e - c * B;
			const int px0 = (c_lo + c) * a.hshrink;
			for (int y = 0; y < ny; y += 4) {
				const unsigned char *src[4];
				unsigned int sum[4];
#pragma unroll
				for (int j = 0; j < 4; j++) {
					src[j] = rows + min(y + j, ny - 1) * row_bytes - start_al + b;
					sum[j] = amend;
				}
#pragma unroll 4
				for (int k = 0; k < a.hshrink; k++) {
					const int off = min(px0 + k, a.width - 1) * B;
#pragma unroll
					for (int j = 0; j < 4; j++)
						sum[j] += src[j][off];
				}
#pragma unroll
				for (int j = 0; j < 4; j++)
					if (y + j < ny)
						shrunk[(y + j) * a.s_pitch + e] = (unsigned char) ((sum[j] * a.mult8) >> 24);
			}
		}
	}
	__syncthreads();

	// ---- 3. reduceh: thread = (band element of the output row, group of rows); four rows at a time
	if (!(a.debug & 8)) {
		const int per_row = nx * B; // <= 256
		const int groups = TAIL_NT / per_row;
		const int g = t / per_row;
		const int e = t - g * per_row;
		if (g < groups) {
			const int x = e / B;
			const int b = e - x * B;
			const int f = first[a.th + x];
			const short *c = ch + x * a.n_h;
			for (int yb = g; yb < ny; yb += 4 * groups) {
				const unsigned char *src[4];
				int sum[4];
#pragma unroll
				for (int j = 0; j < 4; j++) {
					src[j] = shrunk + min(yb + j * groups, ny - 1) * a.s_pitch + b;
					sum[j] = 0;
				}
#pragma unroll 4
				for (int k = 0; k < a.n_h; k++) {
					const int off = (min(max(f + k, 0), a.shrunk_width - 1) - c_lo) * B;
					const int ck = c[k];
#pragma unroll
					for (int j = 0; j < 4; j++)
						sum[j] += ck * (int) src[j][off];
				}
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const int y = yb + j * groups;
					if (y < ny)
						out[(long long) (y0 + y) * a.out_stride + (long long) (x0 + x) * B + b] =
							(unsigned char) tail_fin(sum[j]);
				}
			}
		}
	}
}

namespace {

struct TailPlan {
	bool ok;
	int tw, th, pitch, s_pitch, max_rows;
};

typedef std::tuple<int, double, int, int, double, int, double, int, int, double, int, int, int, int> TailKey;

std::mutex g_tail_mutex;
std::map<TailKey, TailPlan> g_tail_plans;

constexpr int TAIL_LDS_BUDGET = 52 * 1024; // three blocks per CU

int tail_np4(int n_v)
{
	return (((n_v + 1) >> 1) + 3) & ~3;
}

// LDS layout: staged rows | shrunk rows | vertical coefficient pairs | horizontal coefficients | firsts
long long tail_lds_bytes(int rows, int pitch, int th, int s_pitch, int tw, int n_v, int n_h, int *offsets)
{
	long long at = (long long) rows * pitch * 4 + (long long) th * s_pitch;
	at = (at + 15) & ~15LL;
	const long long off_cv = at;
	at += (long long) th * tail_np4(n_v) * 4;
	const long long off_ch = at;
	at += ((long long) tw * n_h * 2 + 3) & ~3LL;
	const long long off_first = at;
	at += (long long) (th + tw) * 4;
	if (offsets) {
		offsets[0] = (int) off_cv;
		offsets[1] = (int) off_ch;
		offsets[2] = (int) off_first;
	}
	return at;
}

// The tile: 16 output rows unless LDS says less; as wide as one thread per staged dword allows,
// and only when no such tile exists one with several dwords per thread.
TailPlan tail_plan(const _VipsHipReduce *rv, int hs, int W3, const _VipsHipReduce *rh, int width, int bands,
	int out_width, int out_height, int tile)
{
	TailPlan plan = { false, 0, 0, 0, 0, 0 };
	std::vector<ReducePos> pv, ph;
	reduce_positions(rv, 0, out_height, tile, pv);
	reduce_positions(rh, 0, out_width, 0, ph);
	for (int y = 1; y < out_height; y++)
		if (pv[y].first <= pv[y - 1].first)
			return plan;
	for (int x = 1; x < out_width; x++)
		if (ph[x].first < ph[x - 1].first)
			return plan;
	auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; };
	// experiments: $VIPS_HIP_TAIL_TH = tallest tile tried, $VIPS_HIP_TAIL_LDS = LDS budget in KiB
	const int th_max = getenv("VIPS_HIP_TAIL_TH") ? atoi(getenv("VIPS_HIP_TAIL_TH")) : 16;
	const long long budget = getenv("VIPS_HIP_TAIL_LDS") ? atoi(getenv("VIPS_HIP_TAIL_LDS")) * 1024LL : TAIL_LDS_BUDGET;
	for (int pass = 0; pass < 2 && !plan.ok; pass++) {
		const int max_pitch = pass == 0 ? TAIL_NT : 8 * TAIL_NT;
		const int min_tw = pass == 0 ? 12 : 2;
		for (int th = th_max; th >= 2 && !plan.ok; th /= 2) {
			int rows = 0;
			for (int y0 = 0; y0 < out_height; y0 += th) {
				const int ny = std::min(th, out_height - y0);
				rows = std::max(rows, pv[y0 + ny - 1].first + rv->n_point - pv[y0].first);
			}
			for (int tw = 64; tw >= min_tw; tw--) {
				int pitch = 0, ncol_max = 0;
				for (int x0 = 0; x0 < out_width; x0 += tw) {
					const int nx = std::min(tw, out_width - x0);
					const int c_lo = clampi(ph[x0].first, 0, W3 - 1);
					const int c_hi = clampi(ph[x0 + nx - 1].first + rh->n_point - 1, 0, W3 - 1);
					const long long byte_lo = (long long) c_lo * hs * bands;
					const long long byte_hi = (long long) (std::min(c_hi * hs + hs - 1, width - 1) + 1) * bands;
					const long long start_al = byte_lo & ~3LL;
					pitch = std::max(pitch, (int) ((byte_hi - start_al + 3) >> 2));
					ncol_max = std::max(ncol_max, c_hi - c_lo + 1);
				}
				if (pitch > max_pitch)
					continue;
				const int s_pitch = (ncol_max * bands + 3) & ~3;
				if (tail_lds_bytes(rows, pitch, th, s_pitch, tw, rv->n_point, rh->n_point, nullptr) > budget)
					continue;
				plan.ok = true;
				plan.tw = tw;
				plan.th = th;
				plan.pitch = pitch;
				plan.s_pitch = s_pitch;
				plan.max_rows = rows;
				break;
			}
		}
	}
	return plan;
}

template <int NV>
void tail_launch(const TailPtrs &p, int count, const TailArgs &a, unsigned int blocks, size_t lds, const ReducePos *posv,
	const short *tabv, const ReducePos *posh, const short *tabh)
{
	hipLaunchKernelGGL(resize_tail_u8<NV>, dim3(blocks, count, 1), dim3(TAIL_NT, 1, 1), lds, stream(), p, a, posv, tabv,
		posh, tabh);
}

} // namespace

// vips_reducev(rv) -> vips_shrinkh(hs, ceil) -> vips_reduceh(rh) of a whole uchar image in one
// kernel.  `in` is the image after the vertical box shrink, `out` the resized image; W3 the
// width after shrinkh.  1 = handled, 0 = not this kernel's case, -1 = error.
int resize_tail_u8_try(_VipsHipReduce *rv, int hs, int W3, _VipsHipReduce *rh, const VipsHipRegion *const *ins,
	const VipsHipRegion *const *outs, int n_images, int tile)
{
	if (getenv("VIPS_HIP_NO_RESIZE_TAIL") || n_images < 1)
		return 0;
	const VipsHipRegion *in = ins[0], *out = outs[0];
	bool aligned = true;
	for (int i = 0; i < n_images; i++) {
		const VipsHipRegion *ri = ins[i], *ro = outs[i];
		if (ri->format != in->format || ri->bands != in->bands || ri->left != in->left || ri->top != in->top ||
			ri->width != in->width || ri->height != in->height || ri->im_width != in->im_width ||
			ri->im_height != in->im_height || ri->stride != in->stride)
			return 0;
		if (ro->format != out->format || ro->bands != out->bands || ro->left != out->left || ro->top != out->top ||
			ro->width != out->width || ro->height != out->height || ro->im_width != out->im_width ||
			ro->im_height != out->im_height || ro->stride != out->stride)
			return 0;
		aligned = aligned && !(((uintptr_t) ri->data & 3) || (ri->stride & 3));
	}
	if (in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR || in->bands != out->bands ||
		in->bands < 1 || in->bands > 4)
		return 0;
	if (in->left != 0 || in->top != 0 || in->width != in->im_width || in->height != in->im_height ||
		out->left != 0 || out->top != 0 || out->width != out->im_width || out->height != out->im_height)
		return 0;
	if (rv->in_size != in->height || rv->out_size != out->height || rh->in_size != W3 || rh->out_size != out->width)
		return 0;
	if (hs < 1 || hs > 256 || rv->n_point > 64 || rh->n_point > 64 || out->width < 1 || out->height < 1)
		return 0;
	if ((long long) in->width * in->bands > 0x3fffffffLL)
		return 0;

	TailPlan plan;
	{
		const TailKey key(rv->kernel, rv->shrink, rv->in_size, rv->out_size, rv->offset, hs, rh->shrink,
			rh->in_size, rh->out_size, rh->offset, in->width, in->bands, tile, rh->kernel);
		std::lock_guard<std::mutex> lock(g_tail_mutex);
		auto it = g_tail_plans.find(key);
		if (getenv("VIPS_HIP_TAIL_TH") || getenv("VIPS_HIP_TAIL_LDS")) {
			g_tail_plans.clear();
			it = g_tail_plans.end();
		}
		if (it == g_tail_plans.end()) {
			if (g_tail_plans.size() > 256)
				g_tail_plans.clear();
			it = g_tail_plans.emplace(key, tail_plan(rv, hs, W3, rh, in->width, in->bands, out->width,
				out->height, tile)).first;
		}
		plan = it->second;
	}
	if (!plan.ok)
		return 0;

	const void *tabv, *tabh;
	if (reduce_tables(rv, false, &tabv) || reduce_tables(rh, false, &tabh))
		return -1;
	const ReducePos *posv = reduce_device_positions(rv, 0, out->height, tile);
	const ReducePos *posh = reduce_device_positions(rh, 0, out->width, 0);
	if (!posv || !posh)
		return -1;

	TailArgs a;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.width = in->width;
	a.height = in->height;
	a.bands = in->bands;
	a.hshrink = hs;
	a.shrunk_width = W3;
	a.mult8 = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) hs));
	a.n_v = rv->n_point;
	a.n_h = rh->n_point;
	a.out_width = out->width;
	a.out_height = out->height;
	a.tw = plan.tw;
	a.th = plan.th;
	a.nbx = (out->width + plan.tw - 1) / plan.tw;
	a.nby = (out->height + plan.th - 1) / plan.th;
	a.band = (a.nby + 7) / 8;
	a.pitch = plan.pitch;
	a.s_pitch = plan.s_pitch;
	a.max_rows = plan.max_rows;
	a.np4 = tail_np4(a.n_v);
	a.aligned = aligned;
	const long long blocks = (long long) a.nbx * a.band * 8;
	if (blocks > 0x7fffffffLL)
		return 0;
	int offsets[3];
	const size_t lds = (size_t) tail_lds_bytes(a.max_rows, a.pitch, a.th, a.s_pitch, a.tw, a.n_v, a.n_h, offsets);
	a.off_cv = offsets[0];
	a.off_ch = offsets[1];
	a.off_first = offsets[2];
	a.debug = getenv("VIPS_HIP_TAIL_DEBUG") ? atoi(getenv("VIPS_HIP_TAIL_DEBUG")) : 0;
	Gate gate("resize_tail_u8");
	for (int base = 0; base < n_images; base += TAIL_MAXB) {
		const int count = n_images - base < TAIL_MAXB ? n_images - base : TAIL_MAXB;
		TailPtrs p;
		memset(&p, 0, sizeof(p));
		for (int i = 0; i < count; i++) {
			p.in[i] = (const unsigned char *) ins[base + i]->data;
			p.out[i] = (unsigned char *) outs[base + i]->data;
		}
#define TAIL_CASE(N) \
	case N: \
		tail_launch<N>(p, count, a, (unsigned int) blocks, lds, posv, (const short *) tabv, posh, (const short *) tabh); \
		break;
		switch (a.n_v) {
			TAIL_CASE(3)
			TAIL_CASE(5)
			TAIL_CASE(7)
			TAIL_CASE(9)
			TAIL_CASE(11)
			TAIL_CASE(13)
			TAIL_CASE(15)
			TAIL_CASE(17)
			TAIL_CASE(19)
			TAIL_CASE(21)
			TAIL_CASE(23)
			TAIL_CASE(25)
		default:
			tail_launch<0>(p, count, a, (unsigned int) blocks, lds, posv, (const short *) tabv, posh, (const short *) tabh);
			break;
		}
#undef TAIL_CASE
		if (hipGetLastError() != hipSuccess) {
			error("resize", "kernel launch failed");
			return -1;
		}
	}
	return 1;
}

} // namespace vh
