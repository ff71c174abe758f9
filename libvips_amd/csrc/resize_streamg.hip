// vips_resize() downsizing of uchar images at ANY scale in one streaming kernel, a batch of
// same-sized images per launch: the general-phase sibling of resize_stream.hip.
//
//   vips_shrinkv(vs, ceil) -> vips_reducev(rv) -> vips_shrinkh(hs, ceil) -> vips_reduceh(rh)
//   (resample/resize.c:207-228 chains shrinkv.c:158-268, reducev.cpp:418-459, shrinkh.c:78-156 and
//   reduceh.cpp:216-255 through three intermediate images; each rounds to uchar).
//
// A residual reduce that is not exactly 2 steps 2 or 3 (or 1, or 4) rows per output with a
// coefficient phase of its own per output row (reducev.cpp:517-560: Y accumulates in double and is
// re-seeded every generate call), so the static rotation of resize_stream.hip does not apply.
// Here the HOST lays the vertical pass out as a schedule, one 32-byte record per row of the
// box-shrunk image: the coefficient that row contributes to each of 12 accumulator slots (output
// row y lives in slot y mod 12 while its taps pass by; 0 when the slot has no use for the row)
// and the slot whose output row completes with it.  The kernel streams down the input like
// resize_stream.hip -- a 512-thread block per (strip of output columns, segment of output rows,
// image), a lane owns 4 consecutive bytes of the strip's 2 KB row span -- and per shrunk row
//   * sums its vs input rows as two 16-bit lanes per dword and rounds them as shrinkv does;
//   * reads the row's record with scalar loads and issues 12 x 4 v_mad_i32_i24 (all slots, a
//     zero coefficient changes nothing);
//   * retires the completing slot, if any: (sum + 2048) >> 12 saturated and packed into an LDS
//     slab of 8 rows (the retiring slot is a run-time value: a switch whose arms differ by an
//     inline-asm marker, so that they are not merged into one arm with a dynamic register index).
// Every 8 retired rows: shrinkh box sums and the reduceh taps from LDS, with the strip's
// coefficient rows and first taps staged in LDS once per block (as resize_tail.hip does).
//
// The image is read once and only the result is written, for any factor: the tail kernel behind
// shrinkv_u8 (two launches, the box-shrunk image through HBM, 1.1 TB/s on it) remains for
// geometries this kernel does not take.  Bit-exact with the separate operations.
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

constexpr int RG_NT = 512;
constexpr int RG_SPAN = 4 * RG_NT; // bytes of a row a strip covers
constexpr int RG_MAXB = 64;        // images per launch
constexpr int RG_NS = 12;          // output rows in flight per column
constexpr int RG_K = 8;            // rows per slab
constexpr int RG_MAXH = 32;        // most horizontal taps

struct GenRow {
	short c[RG_NS];    // this row's coefficient for the output row in each slot (0: none)
	short retire_slot; // slot whose output row completes with this row, -1: none
	short pad;
	int retire_y;      // that output row
};
static_assert(sizeof(GenRow) == 32, "schedule record");

struct GenArgs {
	long long in_stride, out_stride;
	int width, height, bands; // input images
	int vs, hs, h1, w3;       // box shrinks; height after shrinkv, width after shrinkh
	int out_width, out_height;
	unsigned int mult_v, mult_h; // 2^32 / (256 * shrink), shrinkv.c:201 / shrinkh.c:141
	int n_v, n_h;
	int tw, seg;                 // output columns per strip, output rows per segment
	int s_pitch;                 // bytes per row of the shrinkh slab
	int nstrips, nsegs, n_images, grouped;
	int row0;                    // the shrunk row sched[0] describes
	int off_ch, off_fx;          // byte offsets in LDS of the strip's coefficient rows / first taps
	const GenRow *sched;
	const ReducePos *posv, *posh;
	const short *tabh;
};

struct GenPtrs {
	const unsigned char *in[RG_MAXB];
	unsigned char *out[RG_MAXB];
};

// two i32 sums, already shifted, as one dword of two saturated bytes (see resize_stream.hip)
static __device__ __forceinline__ unsigned int rg_sat2(int lo, int hi)
{
	const unsigned int both = __builtin_amdgcn_perm((unsigned int) hi, (unsigned int) lo, 0x05040100u);
	unsigned int r;
	VH_SAT_PK_U8_I16(r, both);
	return r;
}

static __device__ __forceinline__ unsigned int rg_fin(int s)
{
	s = (s + (INTERPOLATE_SCALE >> 1)) >> INTERPOLATE_SHIFT;
	VH_VECTOR1(s);
	return (unsigned int) min(max(s, 0), 255);
}

// NS = accumulator slots in use (output row y lives in slot y mod NS): 8 when no more than 8
// output rows are ever in flight, else 12; the schedule's records hold 12 either way
// GS = input rows per group of loads (1..4; a box of vs rows is ceil(vs / GS) groups)
template <int NS, int GS>
__global__ void __launch_bounds__(RG_NT)
resize_streamg_u8(GenArgs a, GenPtrs ptrs_by_value)
{
	VH_DYNAMIC_LDS(unsigned int, rg_lds);
	unsigned char *T = reinterpret_cast<unsigned char *>(rg_lds); // RG_K rows of RG_SPAN bytes
	unsigned char *S = T + RG_K * RG_SPAN;                        // RG_K rows of s_pitch bytes
	short *ch = reinterpret_cast<short *>(T + a.off_ch);          // [x][n_h] horizontal taps of the strip
	int *fx = reinterpret_cast<int *>(T + a.off_fx);              // [x] first tap of output column x0 + x
	(void) ptrs_by_value;
	typedef const unsigned long long __attribute__((address_space(4))) *KernargPtrs;
	static_assert(sizeof(GenArgs) % 8 == 0, "kernarg layout");
	const KernargPtrs kp = (KernargPtrs) ((const char __attribute__((address_space(4))) *)
											  __builtin_amdgcn_kernarg_segment_ptr() +
		sizeof(GenArgs));
	typedef const unsigned char __attribute__((address_space(1))) *GlobalIn;
	typedef unsigned char __attribute__((address_space(1))) *GlobalOut;
	typedef const unsigned int __attribute__((address_space(1))) *GlobalIn1;

	// block -> (strip, segment, image) as in resize_stream.hip
	const int wg = blockIdx.x;
	int strip, unit;
	if (a.grouped) {
		const int grp = (wg >> 3) / a.nstrips;
		strip = (wg >> 3) - grp * a.nstrips;
		unit = grp * 8 + (wg & 7);
	}
	else {
		unit = wg / a.nstrips;
		strip = wg - unit * a.nstrips;
	}
	if (unit >= a.nsegs * a.n_images)
		return;
	const int img = unit / a.nsegs;
	const int seg_i = unit - img * a.nsegs;
	const GlobalIn in = (GlobalIn) kp[img];
	const GlobalOut out = (GlobalOut) kp[RG_MAXB + img];

	const int t = threadIdx.x;
	const int B = a.bands;
	const int x0 = strip * a.tw, nx = min(a.tw, a.out_width - x0);
	const int y0 = seg_i * a.seg, ny = min(a.seg, a.out_height - y0);
	const int c_lo = min(max(a.posh[x0].first, 0), a.w3 - 1);
	const int c_hi = min(max(a.posh[x0 + nx - 1].first + a.n_h - 1, 0), a.w3 - 1);
	const int ncol = c_hi - c_lo + 1;
	const int row_bytes = a.width * B;
	const int start_al = max(min((c_lo * a.hs * B) & ~3, row_bytes - RG_SPAN), 0);
	const GlobalIn span = in + start_al;
	const unsigned int lane_off = (unsigned int) min(4 * t, row_bytes - 4 - start_al);

	// the strip's horizontal coefficient rows and first taps
	for (int i = t; i < nx * a.n_h; i += RG_NT) {
		const int x = i / a.n_h;
		ch[i] = a.tabh[(size_t) a.posh[x0 + x].phase * a.n_h + (i - x * a.n_h)];
	}
	for (int i = t; i < nx; i += RG_NT)
		fx[i] = a.posh[x0 + i].first;

	int acc[NS][4];
#pragma unroll
	for (int s = 0; s < NS; s++)
#pragma unroll
		for (int b = 0; b < 4; b++)
			acc[s][b] = 0;

	// the slab's rows are output rows ybase .. ybase + nr - 1
	auto hphase = [&](int ybase, int nr) __attribute__((always_inline)) {
		__syncthreads();
		int th = t;
		VH_VECTOR1(th); // (see resize_stream.hip: nothing of this phase is held through the row loop)
		// shrinkh: thread = one band element of the shrunk rows, all slab rows
		{
			const int per_row = ncol * B;
			const unsigned int magic = (65536u + B - 1) / B;
			for (int e = th; e < per_row; e += RG_NT) {
				const int c = (int) ((e * magic) >> 16);
				const int b = e - c * B;
				const int px0 = (c_lo + c) * a.hs;
				const unsigned char *src = T + b - start_al;
				unsigned int sum[RG_K];
#pragma unroll
				for (int r = 0; r < RG_K; r++)
					sum[r] = (unsigned int) (a.hs / 2);
#pragma unroll 4
				for (int k = 0; k < a.hs; k++) {
					const int off = min(px0 + k, a.width - 1) * B;
#pragma unroll
					for (int r = 0; r < RG_K; r++)
						sum[r] += src[r * RG_SPAN + off];
				}
#pragma unroll
				for (int r = 0; r < RG_K; r++)
					S[r * a.s_pitch + e] = (unsigned char) ((sum[r] * a.mult_h) >> 24);
			}
		}
		__syncthreads();
		// reduceh: thread = one band element of the output rows (nx * B <= 512), all slab rows
		if (th < nx * B) {
			const unsigned int magic = (65536u + B - 1) / B;
			const int x = (int) ((th * magic) >> 16);
			const int b = th - x * B;
			const int f = fx[x];
			const short *cx = ch + x * a.n_h;
			int sum[RG_K];
#pragma unroll
			for (int r = 0; r < RG_K; r++)
				sum[r] = 0;
#pragma unroll 4
			for (int k = 0; k < a.n_h; k++) {
				const int off = (min(max(f + k, 0), a.w3 - 1) - c_lo) * B + b;
				const int ck = cx[k];
#pragma unroll
				for (int r = 0; r < RG_K; r++)
					sum[r] += ck * (int) S[r * a.s_pitch + off];
			}
			// The stores are spelled out in assembly on purpose: gfx9 counts loads and stores in one
			// counter and they may finish out of order, so with a store the compiler knows of still
			// pending it drains EVERY outstanding load at the top of the row loop (vmcnt(0)) and the
			// prefetch is lost.  Stores it does not see only make its counted waits stricter.
			const GlobalOut dst = out + (long long) ybase * a.out_stride + (long long) x0 * B + th;
#pragma unroll
			for (int r = 0; r < RG_K; r++)
				if (r < nr) {
					const unsigned int v = rg_fin(sum[r]);
					const GlobalOut p = dst + (long long) r * a.out_stride;
					VH_STORE_BYTE(p, v);
				}
		}
	};

	typedef const unsigned int __attribute__((address_space(4))) *SchedWords;
	const int k_first = a.posv[y0].first;
	const int k_last = a.posv[y0 + ny - 1].first + a.n_v - 1;
	const unsigned int amend = (unsigned int) (a.vs / 2);
	int trow = 0, ybase = y0;
	// Input rows arrive in groups of GS (one box is ceil(vs / GS) groups; a group that runs past
	// its box re-reads the box's last row and does not add it).  The loads of the NEXT group --
	// of this box or of the next shrunk row -- are issued before the current group is summed, so
	// a lane always has two groups in flight (without that the kernel waited out a full memory
	// latency per shrunk row: 2.9 TB/s).  Every step issues exactly GS loads, unconditionally:
	// with loads under (even wave-uniform) branches the compiler waits for ALL outstanding loads.
	const int groups = (a.vs + GS - 1) / GS;
	auto issue = [&](int k, int g, unsigned int (&w)[4]) __attribute__((always_inline)) {
		const int kc = min(max(k, 0), a.h1 - 1);
#pragma unroll
		for (int j = 0; j < GS; j++) {
			const int row = min(kc * a.vs + min(GS * g + j, a.vs - 1), a.height - 1);
			const unsigned int off = (unsigned int) row * (unsigned int) a.in_stride + lane_off;
			w[j] = *(GlobalIn1) (span + off);
		}
	};
	unsigned int buf_a[4] = { 0, 0, 0, 0 }, buf_b[4] = { 0, 0, 0, 0 };
	unsigned int e = 0, o = 0; // the box sums of the shrunk row under way
	// what happens when the last group of shrunk row k has been summed
	auto finish_row = [&](int k) __attribute__((always_inline)) {
		int px[4];
		px[0] = (int) ((((e & 0xffffu) + amend) * a.mult_v) >> 24);
		px[1] = (int) ((((o & 0xffffu) + amend) * a.mult_v) >> 24);
		px[2] = (int) ((((e >> 16) + amend) * a.mult_v) >> 24);
		px[3] = (int) ((((o >> 16) + amend) * a.mult_v) >> 24);

		// ---- its record: a coefficient per slot, the slot to retire
		const SchedWords rec = (SchedWords) (a.sched + (k - a.row0));
		unsigned int cw[NS / 2];
#pragma unroll
		for (int q = 0; q < NS / 2; q++)
			cw[q] = rec[q];
		const unsigned int tail = rec[RG_NS / 2];
		const int retire_y = (int) rec[RG_NS / 2 + 1];
#pragma unroll
		for (int s = 0; s < NS; s++) {
			const int c = (s & 1) ? (int) cw[s >> 1] >> 16 : (int) (short) (cw[s >> 1] & 0xffffu);
#pragma unroll
			for (int b = 0; b < 4; b++)
				acc[s][b] += px[b] * c;
		}
		const int slot = (int) (short) (tail & 0xffffu);
		if (slot >= 0) {
			unsigned int packed = 0;
			// (each arm carries its own marker: identical arms would be merged into one with a
			// dynamic register index, i.e. the accumulators would live in scratch memory)
#define RG_ARM(SL) \
	case SL: \
		VH_ASM_MARK("retire slot " #SL); \
		packed = rg_sat2((acc[SL][0] + 2048) >> INTERPOLATE_SHIFT, (acc[SL][1] + 2048) >> INTERPOLATE_SHIFT) | \
			(rg_sat2((acc[SL][2] + 2048) >> INTERPOLATE_SHIFT, (acc[SL][3] + 2048) >> INTERPOLATE_SHIFT) << 16); \
		acc[SL][0] = acc[SL][1] = acc[SL][2] = acc[SL][3] = 0; \
		break;
			switch (slot) {
				RG_ARM(0)
				RG_ARM(1)
				RG_ARM(2)
				RG_ARM(3)
				RG_ARM(4)
				RG_ARM(5)
				RG_ARM(6)
			default:
				if constexpr (NS == 8) {
					VH_ASM_MARK("retire slot 7");
					packed = rg_sat2((acc[7][0] + 2048) >> INTERPOLATE_SHIFT, (acc[7][1] + 2048) >> INTERPOLATE_SHIFT) |
						(rg_sat2((acc[7][2] + 2048) >> INTERPOLATE_SHIFT, (acc[7][3] + 2048) >> INTERPOLATE_SHIFT) << 16);
					acc[7][0] = acc[7][1] = acc[7][2] = acc[7][3] = 0;
				}
				else {
					switch (slot) {
						RG_ARM(7)
						RG_ARM(8)
						RG_ARM(9)
						RG_ARM(10)
					default:
						VH_ASM_MARK("retire slot 11");
						packed = rg_sat2((acc[NS - 1][0] + 2048) >> INTERPOLATE_SHIFT, (acc[NS - 1][1] + 2048) >> INTERPOLATE_SHIFT) |
							(rg_sat2((acc[NS - 1][2] + 2048) >> INTERPOLATE_SHIFT, (acc[NS - 1][3] + 2048) >> INTERPOLATE_SHIFT) << 16);
						acc[NS - 1][0] = acc[NS - 1][1] = acc[NS - 1][2] = acc[NS - 1][3] = 0;
						break;
					}
				}
				break;
			}
#undef RG_ARM
			if (retire_y >= y0 && retire_y < y0 + ny) {
				*reinterpret_cast<unsigned int *>(T + trow * RG_SPAN + 4 * t) = packed;
				trow++;
				if (trow == RG_K) {
					hphase(ybase, RG_K);
					ybase += RG_K;
					trow = 0;
				}
			}
		}
	};
	// One step = one group.  Three buffers rotate (by unrolling: a register copy would wait for
	// the loads it copies): the group two steps ahead is requested before the current one is
	// summed, so a lane has 3 GS loads in flight (with two buffers the kernel ran at 3.7 TB/s,
	// the latency of one group's loads still showing).
	int k = k_first, g = 0; // the group being summed
	auto after = [&](int &kk, int &gg) __attribute__((always_inline)) {
		const bool last = gg + 1 == groups;
		kk = last ? kk + 1 : kk;
		gg = last ? 0 : gg + 1;
	};
	int k1 = k, g1 = g;
	after(k1, g1);
	int k2 = k1, g2 = g1; // the group the next step requests
	after(k2, g2);
	auto step = [&](unsigned int (&cur)[4], unsigned int (&fill)[4]) __attribute__((always_inline)) {
		issue(k2, g2, fill); // (past the last row: a clamped row nobody sums)
#pragma unroll
		for (int j = 0; j < GS; j++) {
			const unsigned int keep = GS * g + j < a.vs ? 0xffffffffu : 0u; // (scalar)
			e += cur[j] & (0x00ff00ffu & keep);
			o += __builtin_amdgcn_perm(0u, cur[j], 0x0c030c01u) & keep;
		}
		if (g + 1 == groups) {
			if (k <= k_last)
				finish_row(k);
			e = o = 0;
		}
		k = k1;
		g = g1;
		k1 = k2;
		g1 = g2;
		after(k2, g2);
	};
	unsigned int buf_c[4] = { 0, 0, 0, 0 };
	issue(k, g, buf_a);
	issue(k1, g1, buf_b);
	const int total = (k_last - k_first + 1) * groups;
	// (whole turns of three steps: a step under a branch makes its buffer a merge of loaded and
	// not loaded, and the merge waits for every load; the one or two steps past the end sum
	// clamped rows into nothing)
	for (int st = 0; st < total; st += 3) {
		step(buf_a, buf_c);
		step(buf_b, buf_a);
		step(buf_c, buf_b);
	}
	if (trow > 0)
		hphase(ybase, trow);
}

namespace {

struct GenPlan {
	bool ok;
	int ns; // accumulator slots: 8 or 12
	int tw, s_pitch, row0;
	GenRow *d_sched;
};

// (the last field is the device the schedule lives on)
typedef std::tuple<int, double, int, int, double, int, double, int, int, double, int, int, int, int, int> GenKey;

std::mutex g_gen_mutex;
std::map<GenKey, GenPlan> g_gen_plans;

// the schedule and the strip width of one geometry; ok = false when this kernel does not take it
GenPlan gen_plan(const _VipsHipReduce *rv, int hs, int W3, const _VipsHipReduce *rh, int width, int bands,
	int out_width, int out_height, int tile)
{
	GenPlan plan = { false, RG_NS, 0, 0, 0, nullptr };
	std::vector<ReducePos> pv, ph;
	reduce_positions(rv, 0, out_height, tile, pv);
	reduce_positions(rh, 0, out_width, 0, ph);
	for (int y = 1; y < out_height; y++)
		if (pv[y].first <= pv[y - 1].first)
			return plan;
	for (int x = 1; x < out_width; x++)
		if (ph[x].first < ph[x - 1].first)
			return plan;
	const int n_v = rv->n_point;
	const int row0 = pv[0].first;
	const long long nrows = (long long) pv[out_height - 1].first + n_v - row0;
	if (nrows < 1 || nrows > (1 << 24))
		return plan;
	// most output rows in flight at a row: 8 slots when that is enough
	{
		std::vector<int> starts((size_t) nrows + 1, 0);
		for (int y = 0; y < out_height; y++) {
			starts[(size_t) (pv[y].first - row0)]++;
			starts[(size_t) (pv[y].first + n_v - row0)]--;
		}
		int live = 0, most = 0;
		for (long long k = 0; k < nrows; k++) {
			live += starts[(size_t) k];
			most = std::max(most, live);
		}
		if (most > RG_NS)
			return plan;
		plan.ns = most <= 8 ? 8 : RG_NS;
	}
	const int NS = plan.ns;
	std::vector<GenRow> sched((size_t) nrows);
	for (GenRow &r : sched) {
		memset(&r, 0, sizeof(r));
		r.retire_slot = -1;
		r.retire_y = -1;
	}
	std::vector<int> owner((size_t) nrows * RG_NS, -1); // which output row uses (row, slot)
	for (int y = 0; y < out_height; y++) {
		const int slot = y % NS;
		const short *c = &rv->matrixs[(size_t) pv[y].phase * n_v];
		for (int j = 0; j < n_v; j++) {
			const size_t at = (size_t) (pv[y].first + j - row0);
			if (owner[at * RG_NS + slot] >= 0)
				return plan; // more than RG_NS output rows in flight
			owner[at * RG_NS + slot] = y;
			sched[at].c[slot] = c[j];
		}
		GenRow &last = sched[(size_t) (pv[y].first + n_v - 1 - row0)];
		if (last.retire_slot >= 0)
			return plan;
		last.retire_slot = (short) slot;
		last.retire_y = y;
	}
	// the widest strip whose span fits the block (one lane dword per 4 bytes) and whose outputs fit
	// one thread per band element
	auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; };
	const long long row_bytes = (long long) width * bands;
	for (int tw = RG_NT / bands < 128 ? RG_NT / bands : 128; tw >= 2 && !plan.ok; tw--) {
		int ncol_max = 0;
		bool fits = true;
		for (int x0 = 0; x0 < out_width && fits; x0 += tw) {
			const int nx = std::min(tw, out_width - x0);
			const int c_lo = clampi(ph[x0].first, 0, W3 - 1);
			const int c_hi = clampi(ph[x0 + nx - 1].first + rh->n_point - 1, 0, W3 - 1);
			const long long byte_lo = (long long) c_lo * hs * bands;
			const long long byte_hi = (long long) (std::min(c_hi * hs + hs - 1, width - 1) + 1) * bands;
			long long start_al = std::min(byte_lo & ~3LL, row_bytes - RG_SPAN);
			if (start_al < 0)
				start_al = 0;
			if (byte_hi - start_al > RG_SPAN)
				fits = false;
			ncol_max = std::max(ncol_max, c_hi - c_lo + 1);
		}
		if (!fits)
			continue;
		plan.tw = tw;
		plan.s_pitch = (ncol_max * bands + 3) & ~3;
		plan.ok = true;
	}
	if (!plan.ok)
		return plan;
	plan.row0 = row0;
	plan.d_sched = (GenRow *) upload(sched.data(), sched.size() * sizeof(GenRow));
	if (!plan.d_sched)
		plan.ok = false;
	return plan;
}

} // namespace

// The whole downsizing chain of vips_resize on n uchar images of one geometry, any residual
// reduce.  `rv` was built for the image after shrinkv(vs) (height h1), `rh` for the one after
// shrinkh(hs) (width w3).  1 = handled, 0 = not this kernel's case (nothing launched), -1 = error.
int resize_streamg_u8_try(_VipsHipReduce *rv, int vs, _VipsHipReduce *rh, int hs, int h1, int w3,
	const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n, int tile)
{
	if (getenv("VIPS_HIP_NO_RESIZE_STREAM") || getenv("VIPS_HIP_NO_RESIZE_STREAMG") || n < 1)
		return 0;
	const VipsHipRegion *i0 = in[0], *o0 = out[0];
	for (int i = 0; i < n; i++) {
		const VipsHipRegion *ri = in[i], *ro = out[i];
		if (ri->format != VIPS_HIP_FORMAT_UCHAR || ro->format != VIPS_HIP_FORMAT_UCHAR || ri->bands != ro->bands ||
			ri->bands < 1 || ri->bands > 4)
			return 0;
		if (ri->left != 0 || ri->top != 0 || ri->width != ri->im_width || ri->height != ri->im_height ||
			ro->left != 0 || ro->top != 0 || ro->width != ro->im_width || ro->height != ro->im_height)
			return 0;
		if (ri->width != i0->width || ri->height != i0->height || ri->bands != i0->bands || ri->stride != i0->stride ||
			ro->width != o0->width || ro->height != o0->height || ro->stride != o0->stride)
			return 0;
		if (((uintptr_t) ri->data & 3) || (ri->stride & 3))
			return 0;
	}
	const int B = i0->bands;
	const long long row_bytes = (long long) i0->width * B;
	if ((row_bytes & 3) || row_bytes < 8 || row_bytes > 0x3fffffffLL)
		return 0;
	if ((unsigned long long) i0->stride * (unsigned long long) i0->height > 0xffffffffULL)
		return 0;
	if (vs < 1 || vs > 255 || hs < 1 || hs > 64)
		return 0;
	if (rv->in_size != h1 || rv->out_size != o0->height || rh->in_size != w3 || rh->out_size != o0->width)
		return 0;
	if (rv->n_point < 1 || rv->n_point > 64 || rh->n_point < 1 || rh->n_point > RG_MAXH)
		return 0;
	// Without a vertical box shrink every INPUT row costs a schedule step (NS x 4 multiply-adds per
	// dword), and with few taps the fused tail's dot2 walk is cheaper: measured on 64 images of
	// 8192^2 x 3 at scale 0.45 (15 taps) 0.34 against 0.27 ms per image, at 0.3 (21 taps) 0.22
	// against 0.26 (profiles/r02_probes.txt).  $VIPS_HIP_STREAMG_ALWAYS=1 takes this kernel anyway.
	if (vs == 1 && rv->n_point < 19 && !getenv("VIPS_HIP_STREAMG_ALWAYS"))
		return 0;

	GenPlan plan;
	{
		const GenKey key(rv->kernel, rv->shrink, rv->in_size, rv->out_size, rv->offset, hs, rh->shrink, rh->in_size,
			rh->out_size, rh->offset, i0->width, B, tile, rh->kernel, current_device());
		std::lock_guard<std::mutex> lock(g_gen_mutex);
		auto it = g_gen_plans.find(key);
		if (it == g_gen_plans.end()) {
			// schedules stay on the device for the life of the process (kernels of other threads
			// may be reading them): once the table is full, new geometries take the older path
			if (g_gen_plans.size() >= 256)
				return 0;
			it = g_gen_plans.emplace(key, gen_plan(rv, hs, w3, rh, i0->width, B, o0->width, o0->height, tile)).first;
		}
		plan = it->second;
	}
	if (!plan.ok)
		return 0;
	const void *tabh;
	if (reduce_tables(rh, false, &tabh))
		return -1;
	const ReducePos *posv = reduce_device_positions(rv, 0, o0->height, tile);
	const ReducePos *posh = reduce_device_positions(rh, 0, o0->width, 0);
	if (!posv || !posh)
		return -1;

	GenArgs a;
	memset(&a, 0, sizeof(a));
	a.in_stride = (long long) i0->stride;
	a.out_stride = (long long) o0->stride;
	a.width = i0->width;
	a.height = i0->height;
	a.bands = B;
	a.vs = vs;
	a.hs = hs;
	a.h1 = h1;
	a.w3 = w3;
	a.out_width = o0->width;
	a.out_height = o0->height;
	a.mult_v = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) vs));
	a.mult_h = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) hs));
	a.n_v = rv->n_point;
	a.n_h = rh->n_point;
	a.tw = plan.tw;
	a.s_pitch = plan.s_pitch;
	a.row0 = plan.row0;
	a.sched = plan.d_sched;
	a.posv = posv;
	a.posh = posh;
	a.tabh = (const short *) tabh;
	const int nstrips = (o0->width + plan.tw - 1) / plan.tw;
	long long want = getenv("VIPS_HIP_STREAM_BLOCKS") ? atoll(getenv("VIPS_HIP_STREAM_BLOCKS")) : 2048;
	int nsegs = (int) ((want + (long long) nstrips * n - 1) / ((long long) nstrips * n));
	int seg = (o0->height + nsegs - 1) / nsegs;
	const int seg_min = getenv("VIPS_HIP_STREAM_SEG") ? atoi(getenv("VIPS_HIP_STREAM_SEG")) : 32;
	if (seg < seg_min)
		seg = seg_min;
	nsegs = (o0->height + seg - 1) / seg;
	a.seg = seg;
	size_t lds = (size_t) RG_K * RG_SPAN + (size_t) RG_K * a.s_pitch;
	lds = (lds + 3) & ~(size_t) 3;
	a.off_ch = (int) lds;
	lds += ((size_t) plan.tw * a.n_h * sizeof(short) + 3) & ~(size_t) 3;
	a.off_fx = (int) lds;
	lds += (size_t) plan.tw * sizeof(int);

	// rows per group of loads: the whole box when it has at most 4 rows, else the group size that
	// re-reads the fewest rows
	int gs = vs <= 4 ? vs : 4;
	if (vs > 4) {
		int best = 1 << 30;
		for (int cand = 4; cand >= 2; cand--) {
			const int waste = (vs + cand - 1) / cand * cand - vs;
			if (waste < best) {
				best = waste;
				gs = cand;
			}
		}
	}
	Gate gate("resize_streamg_u8");
	for (int base = 0; base < n; base += RG_MAXB) {
		const int count = n - base < RG_MAXB ? n - base : RG_MAXB;
		GenPtrs p;
		memset(&p, 0, sizeof(p));
		for (int i = 0; i < count; i++) {
			p.in[i] = (const unsigned char *) in[base + i]->data;
			p.out[i] = (unsigned char *) out[base + i]->data;
		}
		a.nstrips = nstrips;
		a.nsegs = nsegs;
		a.n_images = count;
		const long long units = (long long) nsegs * count;
		a.grouped = units >= 64;
		const long long blocks = (a.grouped ? (units + 7) / 8 * 8 : units) * nstrips;
		if (blocks > 0x7fffffffLL) {
			error("resize", "image too large");
			return -1;
		}
		const dim3 grid((unsigned int) blocks, 1, 1), block(RG_NT, 1, 1);
#define RG_GO(NSLOTS, G) hipLaunchKernelGGL((resize_streamg_u8<NSLOTS, G>), grid, block, lds, stream(), a, p)
#define RG_GO_GS(NSLOTS) \
	switch (gs) { \
	case 1: RG_GO(NSLOTS, 1); break; \
	case 2: RG_GO(NSLOTS, 2); break; \
	case 3: RG_GO(NSLOTS, 3); break; \
	default: RG_GO(NSLOTS, 4); break; \
	}
		if (plan.ns == 8) {
			RG_GO_GS(8)
		}
		else {
			RG_GO_GS(RG_NS)
		}
#undef RG_GO_GS
#undef RG_GO
		if (hipGetLastError() != hipSuccess) {
			error("resize", "kernel launch failed");
			return -1;
		}
	}
	return 1;
}

} // namespace vh
