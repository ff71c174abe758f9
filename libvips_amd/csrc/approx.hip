// precision=approximate: vips_conva (2-D box decomposition) and vips_convasep (1-D line
// decomposition) for gfx950.
//
// The reference slices the rint()ed mask into `layers` horizontal slabs, turns every slab row
// into a run of ones (an "hline"), clusters near-identical runs, and then evaluates the mask as
// a sum of box sums with rolling accumulators (convolution/conva.c, convolution/convasep.c).
// Its results therefore differ from an exact convolution, and a drop-in has to differ in exactly
// the same way.  Two things are needed for that:
//
//   host    the decomposition itself (Boxes / Lines below), which fixes the approximated mask,
//           the divisor and the rounding term.  Written from the reference's description of the
//           algorithm, including the behaviours that look like slips but shape the output: the
//           common factor enters the area twice (conva.c:728-744, convasep.c:272-290), vlines
//           keep their un-reduced factors, the edge list is cleaned with a live read of the edge
//           being merged (conva.c:531-546), merged end points truncate toward zero.
//   device  the arithmetic of the generate functions.  All integer arithmetic there is modular
//           (the rolling sums wrap in the intermediate type and in the accumulator), so the box
//           sums are evaluated directly per output element; the types are the reference's:
//             conva     intermediate ushort/short when the longest hline < 256 else uint/int,
//                       vertical sums in unsigned int / int / float / double, the final
//                       (sum + rounding) / divisor + offset in that same type -- so unsigned
//                       totals wrap instead of going negative (conva.c:1056-1198)
//             convasep  line sums in unsigned int / int (double for float images), weighted total
//                       in int64 (double), the horizontal pass clips to the format without the
//                       offset, the vertical pass adds it (convasep.c:428-514, 592-675)
//           Float images: the reference's rolling sums are order dependent as soon as a partial
//           sum is inexact; the direct sums here agree with it whenever they are exact.
//
// Fast path: for 8- and 16-bit images where no intermediate can wrap and no total can go
// negative, a pass IS an integer convolution with the approximated mask W (the per-element
// sum of line factors), scale = divisor, rounding = (divisor + 1) / 2 -- so it runs on the
// convi kernels (conv.hip) or, for convasep, the fused separable kernel (convsep_f32.hip).
// Float images take the same route for convasep (double sums; see fast_ok).  32-bit integer and
// double images, and conva on float (float sums), stay on the generic kernels.
// VIPS_HIP_NO_APPROX_FAST=1 forces the generic kernels everywhere.
#include "conv.h"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

using namespace vh;

namespace {

// ------------------------------------------------------------------ host: mask -> int mask

struct IntMask {
	std::vector<double> v; // rint()ed elements
	int width, height;
	double scale, offset;
};

// vips__image_intize, convi.c:860-923: the scale is nudged so that a constant image keeps the
// brightness the double mask would give it.
IntMask intize(const double *mask, int width, int height, double scale, double offset)
{
	IntMask m;
	m.width = width;
	m.height = height;
	const int n = width * height;
	m.v.resize(n);

	double want = 0;
	for (int i = 0; i < n; i++)
		want += mask[i];
	want /= scale;

	for (int i = 0; i < n; i++)
		m.v[i] = rint(mask[i]);
	double s = rint(scale);
	if (s == 0)
		s = 1;

	int got = 0;
	for (int i = 0; i < n; i++)
		got = (int) (got + m.v[i]);
	got = (int) (got / s);

	s = rint(s + (got - want));
	if (s == 0)
		s = 1;
	m.scale = s;
	m.offset = rint(offset);
	return m;
}

int gcd(int a, int b)
{
	while (b != 0) {
		const int t = a % b;
		a = b;
		b = t;
	}
	return abs(a);
}

// The slab arithmetic shared by both decompositions (conva.c:311-329, convasep.c:176-196):
// the zero plane must sit on a slab boundary.
struct Slabs {
	double max, depth;
	int count, above;
	bool ok;
	Slabs(const std::vector<double> &c, int layers)
	{
		max = 0;
		double min = 0;
		for (double e : c) {
			max = e > max ? e : max;
			min = e < min ? e : min;
		}
		ok = max > 0; // with no positive element the reference's arithmetic is undefined
		if (!ok)
			return;
		depth = (max - min) / layers;
		above = (int) ceil(max / depth);
		depth = max / above;
		const int below = (int) floor(min / depth);
		const long long span = (long long) above - below;
		count = span < 1 ? 1 : (span > 1000 ? 1000 : (int) span);
	}
	// the level an element has to reach to be inside slab z (its mid-plane)
	double level(int z) const { return max - (1 + z) * depth + depth / 2; }
	bool positive(int z) const { return z < above; }
	bool inside(int z, double c) const { return positive(z) ? c >= level(z) : c <= level(z); }
};

constexpr int MAX_LINES = 1000;
constexpr int MAX_EDGES = 1000;

// ------------------------------------------------------------------ host: conva boxes

struct Run {
	int start, end, weight;
};
struct Use { // one use of a run: which run, on which mask row, how often
	int band, row, factor;
};
struct Column { // uses of one run on consecutive rows with one factor
	int band, factor, start, end;
};
struct Edge {
	int a, b, d;
};

struct Boxes {
	std::vector<Run> runs;
	std::vector<Use> uses;
	std::vector<Column> columns;
	int divisor, rounding, offset, max_line;
};

int edge_order(const void *p, const void *q)
{
	return ((const Edge *) p)->d - ((const Edge *) q)->d;
}

int use_order(const void *p, const void *q)
{
	const Use *a = (const Use *) p, *b = (const Use *) q;
	if (a->band != b->band)
		return a->band - b->band;
	if (a->factor != b->factor)
		return a->factor - b->factor;
	return a->row - b->row;
}

// conva.c:445-549.  The candidate list is a fixed 1000-entry array sorted with the C library's
// qsort, exactly as in the reference, so equal distances come out in the same order.
bool cluster_once(Boxes &bx, int cluster, Edge *edge)
{
	for (int i = 0; i < MAX_EDGES; i++)
		edge[i] = Edge{ -1, -1, 99999 };
	int worst_at = 0, worst = edge[0].d;
	const int n = (int) bx.runs.size();
	for (int i = 0; i < n; i++) {
		if (!bx.runs[i].weight)
			continue;
		for (int j = i + 1; j < n; j++) {
			if (!bx.runs[j].weight)
				continue;
			const int d = abs(bx.runs[i].start - bx.runs[j].start) + abs(bx.runs[i].end - bx.runs[j].end);
			if (d < worst) {
				edge[worst_at] = Edge{ i, j, d };
				worst_at = 0;
				worst = edge[0].d;
				for (int k = 0; k < MAX_EDGES; k++)
					if (edge[k].d > worst) {
						worst = edge[k].d;
						worst_at = k;
					}
			}
		}
	}
	qsort(edge, MAX_EDGES, sizeof(Edge), edge_order);

	bool merged = false;
	for (int k = 0; k < MAX_EDGES; k++) {
		Edge &e = edge[k];
		if (e.d > cluster)
			break;
		if (e.a == -1)
			continue;
		// fold run b into run a, end points weighted by how many runs each already stands for
		Run &ra = bx.runs[e.a], &rb = bx.runs[e.b];
		const double w = (double) rb.weight / (ra.weight + rb.weight);
		ra.start = (int) (ra.start + w * (rb.start - ra.start));
		ra.end = (int) (ra.end + w * (rb.end - ra.end));
		ra.weight += rb.weight;
		for (Use &u : bx.uses)
			if (u.band == e.b)
				u.band = e.a;
		rb.weight = 0;
		merged = true;
		// e is the first entry this loop visits and is cleared there, so the tests against
		// e.a below see -1 from then on: only the edges touching run b really go
		for (int i = k; i < MAX_EDGES; i++)
			if (edge[i].a == e.a || edge[i].b == e.a || edge[i].a == e.b || edge[i].b == e.b)
				edge[i].a = -1;
	}
	return merged;
}

// conva.c:676-767.  Returns an error string, or nullptr.
const char *decompose_boxes(Boxes &bx, const IntMask &m, int layers, int cluster)
{
	const Slabs slabs(m.v, layers);
	if (!slabs.ok)
		return "mask has no positive element";

	// :294-395 every slab row becomes runs of elements that reach the slab's mid-plane
	for (int z = 0; z < slabs.count; z++)
		for (int y = 0; y < m.height; y++) {
			int open = -1;
			for (int x = 0; x <= m.width; x++) {
				const bool in = x < m.width && slabs.inside(z, m.v[x + y * m.width]);
				if (in && open < 0)
					open = x;
				else if (!in && open >= 0) {
					bx.runs.push_back(Run{ open, x, 1 });
					bx.uses.push_back(Use{ (int) bx.runs.size() - 1, y, slabs.positive(z) ? 1 : -1 });
					open = -1;
					if (bx.runs.size() >= MAX_LINES)
						return "mask too complex";
				}
			}
		}
	if (bx.uses.empty())
		return "mask too complex";

	{
		std::vector<Edge> edge(MAX_EDGES);
		while (cluster_once(bx, cluster, edge.data()))
			;
	}

	// :551-581 squeeze out the runs that were merged away
	{
		std::vector<int> renumber(bx.runs.size(), -1);
		std::vector<Run> kept;
		for (size_t i = 0; i < bx.runs.size(); i++)
			if (bx.runs[i].weight > 0) {
				renumber[i] = (int) kept.size();
				kept.push_back(bx.runs[i]);
			}
		for (Use &u : bx.uses)
			u.band = renumber[u.band];
		bx.runs.swap(kept);
	}

	// :583-674 identical uses pile up as a factor; uses on consecutive rows form a column
	qsort(bx.uses.data(), bx.uses.size(), sizeof(Use), use_order);
	{
		std::vector<Use> piled;
		for (size_t y = 0; y < bx.uses.size();) {
			size_t z = y + 1;
			while (z < bx.uses.size() && bx.uses[z].band == bx.uses[y].band && bx.uses[z].row == bx.uses[y].row)
				z++;
			Use u = bx.uses[y];
			u.factor = u.factor > 0 ? (int) (z - y) : -(int) (z - y);
			piled.push_back(u);
			y = z;
		}
		bx.uses.swap(piled);
	}
	for (size_t y = 0; y < bx.uses.size();) {
		Column c = { bx.uses[y].band, bx.uses[y].factor, bx.uses[y].row, 0 };
		size_t z = y + 1;
		while (z < bx.uses.size() && bx.uses[z].band == c.band && bx.uses[z].factor == c.factor &&
			bx.uses[z].row == c.start + (int) (z - y))
			z++;
		c.end = bx.uses[z - 1].row + 1;
		bx.columns.push_back(c);
		y = z;
	}

	// :693-744 the divisor: |area| of the boxes against |area| of the mask.  The columns keep
	// the un-reduced factors and the area takes the common factor a second time.
	double area = 0;
	bx.max_line = 0;
	for (const Use &u : bx.uses) {
		const int len = bx.runs[u.band].end - bx.runs[u.band].start;
		area += abs(u.factor * len);
		bx.max_line = len > bx.max_line ? len : bx.max_line;
	}
	int common = bx.uses[0].factor;
	for (size_t y = 1; y < bx.uses.size(); y++)
		common = gcd(common, bx.uses[y].factor);
	area *= common;
	double mask_area = 0;
	for (double e : m.v)
		mask_area += fabs(e);
	const double d = rint(area * m.scale / mask_area);
	bx.divisor = d > 1 ? (int) d : 1;
	bx.rounding = (bx.divisor + 1) / 2;
	bx.offset = (int) m.offset;

	if (bx.runs.size() > 150)
		return "mask too complex";
	return nullptr;
}

// ------------------------------------------------------------------ host: convasep lines

struct Line {
	int start, end, factor;
};
struct Lines {
	std::vector<Line> lines;
	int divisor, rounding, offset, width;
};

// convasep.c:152-330
const char *decompose_lines(Lines &ln, const IntMask &m, int layers)
{
	const int width = m.width * m.height;
	ln.width = width;
	const Slabs slabs(m.v, layers);
	if (!slabs.ok)
		return "mask has no positive element";

	std::vector<Line> &l = ln.lines;
	for (int z = 0; z < slabs.count; z++) {
		int open = -1;
		for (int x = 0; x <= width; x++) {
			const bool in = x < width && slabs.inside(z, m.v[x]);
			if (in && open < 0)
				open = x;
			else if (!in && open >= 0) {
				l.push_back(Line{ open, x, slabs.positive(z) ? 1 : -1 });
				open = -1;
				if (l.size() >= MAX_LINES)
					return "mask too complex";
			}
		}
	}
	if (l.empty())
		return "mask too complex";

	// :249-262 identical lines pile up as a factor
	for (size_t z = 0; z < l.size(); z++)
		for (size_t n = z + 1; n < l.size(); n++)
			if (l[z].start == l[n].start && l[z].end == l[n].end) {
				l[z].factor += l[n].factor;
				l[n].factor = 0;
			}
	// :264-275 dead lines are shifted out, but the slot that moves up is not looked at again, so
	// the second of two adjacent dead lines stays (with factor 0 it adds nothing)
	l.push_back(Line{ 0, 0, 0 }); // the zeroed slot behind the array the reference shifts in
	size_t n_lines = l.size() - 1;
	for (size_t z = 0; z < n_lines; z++)
		if (l[z].factor == 0) {
			for (size_t x = z; x < n_lines; x++)
				l[x] = l[x + 1];
			n_lines -= 1;
		}
	l.resize(n_lines);
	if (l.empty())
		return "mask too complex";

	double area = 0;
	for (const Line &e : l)
		area += e.factor * (e.end - e.start);
	int common = l[0].factor;
	for (size_t z = 1; z < l.size(); z++)
		common = gcd(common, l[z].factor);
	if (common == 0)
		return "mask too complex"; // the reference divides by zero
	for (Line &e : l)
		e.factor /= common;
	area *= common; // the common factor enters the area a second time here too

	double sum = 0;
	for (double e : m.v)
		sum += e;
	const double d = rint(sum * area / m.scale);
	ln.divisor = d > 1 ? (int) d : 1;
	ln.rounding = (ln.divisor + 1) / 2;
	ln.offset = (int) m.offset;
	return nullptr;
}

} // namespace

// ------------------------------------------------------------------ the plan object

struct _VipsHipConva {
	bool separable;
	int mask_width, mask_height; // separable: mask_width = n, mask_height = 1
	Boxes boxes;
	Lines lines;
	std::vector<int> table; // what the kernels read (layout at the kernels)
	int *d_table;
	// fast path: the pass as a convi plan (separable: [0] horizontal, [1] vertical)
	VipsHipConv *fast[2];
	long long w_abs_sum; // sum |W|
	bool w_negative;     // some line / column has a negative factor
	std::mutex mutex;
	mutable std::atomic<int> device{ -1 }; // where the device tables live (vh::plan_device)
};

namespace vh {

struct ApproxArgs {
	const unsigned char *in;
	unsigned char *out;
	long long in_stride, out_stride;
	int in_left, in_top, im_width, im_height;
	int out_left, out_top, out_width, out_height;
	int epp;
	int half_w, half_h;
	int n_runs, n_columns; // conva; convasep: n_columns = number of lines
	const int *table;
	int divisor, rounding, offset;
	int vertical; // convasep pass direction
};

enum { CLIP_NONE = 0, CLIP_UCHAR, CLIP_CHAR, CLIP_USHORT, CLIP_SHORT, CLIP_SHORT_ON_UINT };

// ---------------------------------------------------------------- conva kernel
//
// table: n_runs x {start, end}, then n_columns x {run, factor, first row, last row + 1}.
// Accumulator semantics per type (conva.c:1056-1097): integers wrap modulo 2^32 (computed in
// unsigned, reinterpreted for the signed formats), float / double round after every operation.
template <typename ACC>
struct ApproxAcc;
template <>
struct ApproxAcc<unsigned int> {
	typedef unsigned int work_t;
	static __device__ __forceinline__ work_t zero() { return 0u; }
	template <typename MID>
	static __device__ __forceinline__ work_t add(work_t s, MID v) { return s + (unsigned int) v; }
	static __device__ __forceinline__ work_t mad(work_t s, int f, work_t v) { return s + (unsigned int) f * v; }
	static __device__ __forceinline__ work_t fin(work_t s, const ApproxArgs &a)
	{
		return (s + (unsigned int) a.rounding) / (unsigned int) a.divisor + (unsigned int) a.offset;
	}
};
template <>
struct ApproxAcc<int> {
	typedef unsigned int work_t; // two's complement arithmetic without signed-overflow UB
	static __device__ __forceinline__ work_t zero() { return 0u; }
	template <typename MID>
	static __device__ __forceinline__ work_t add(work_t s, MID v) { return s + (unsigned int) (int) v; }
	static __device__ __forceinline__ work_t mad(work_t s, int f, work_t v) { return s + (unsigned int) f * v; }
	static __device__ __forceinline__ work_t fin(work_t s, const ApproxArgs &a)
	{
		const int q = (int) (s + (unsigned int) a.rounding) / a.divisor; // C division truncates
		return (unsigned int) q + (unsigned int) a.offset;
	}
};
template <>
struct ApproxAcc<float> {
	typedef float work_t;
	static __device__ __forceinline__ work_t zero() { return 0.f; }
	template <typename MID>
	static __device__ __forceinline__ work_t add(work_t s, MID v) { return __fadd_rn(s, (float) v); }
	static __device__ __forceinline__ work_t mad(work_t s, int f, work_t v)
	{
		return __fadd_rn(s, __fmul_rn((float) f, v));
	}
	static __device__ __forceinline__ work_t fin(work_t s, const ApproxArgs &a)
	{
		return __fadd_rn(__fdiv_rn(__fadd_rn(s, (float) a.rounding), (float) a.divisor), (float) a.offset);
	}
};
template <>
struct ApproxAcc<double> {
	typedef double work_t;
	static __device__ __forceinline__ work_t zero() { return 0.0; }
	template <typename MID>
	static __device__ __forceinline__ work_t add(work_t s, MID v) { return __dadd_rn(s, (double) v); }
	static __device__ __forceinline__ work_t mad(work_t s, int f, work_t v)
	{
		return __dadd_rn(s, __dmul_rn((double) f, v));
	}
	static __device__ __forceinline__ work_t fin(work_t s, const ApproxArgs &a)
	{
		return __dadd_rn(__ddiv_rn(__dadd_rn(s, (double) a.rounding), (double) a.divisor), (double) a.offset);
	}
};

// the intermediate of the horizontal pass: a sum that wraps in MID
template <typename MID>
struct ApproxMid {
	typedef unsigned int work_t;
	template <typename IN>
	static __device__ __forceinline__ work_t add(work_t s, IN v) { return s + (unsigned int) (int) v; }
	static __device__ __forceinline__ MID fin(work_t s) { return (MID) s; } // modular narrowing
};
template <>
struct ApproxMid<unsigned int> {
	typedef unsigned int work_t;
	template <typename IN>
	static __device__ __forceinline__ work_t add(work_t s, IN v) { return s + (unsigned int) v; }
	static __device__ __forceinline__ unsigned int fin(work_t s) { return s; }
};
template <>
struct ApproxMid<unsigned short> {
	typedef unsigned int work_t;
	template <typename IN>
	static __device__ __forceinline__ work_t add(work_t s, IN v) { return s + (unsigned int) v; }
	static __device__ __forceinline__ unsigned short fin(work_t s) { return (unsigned short) s; }
};
template <>
struct ApproxMid<float> {
	typedef float work_t;
	template <typename IN>
	static __device__ __forceinline__ work_t add(work_t s, IN v) { return __fadd_rn(s, (float) v); }
	static __device__ __forceinline__ float fin(work_t s) { return s; }
};
template <>
struct ApproxMid<double> {
	typedef double work_t;
	template <typename IN>
	static __device__ __forceinline__ work_t add(work_t s, IN v) { return __dadd_rn(s, (double) v); }
	static __device__ __forceinline__ double fin(work_t s) { return s; }
};

template <int CLIP, typename W>
static __device__ __forceinline__ W approx_clip(W v)
{
	if constexpr (CLIP == CLIP_UCHAR) // on unsigned int: `< 0` never holds
		return v > 255u ? 255u : v;
	else if constexpr (CLIP == CLIP_USHORT)
		return v > 65535u ? 65535u : v;
	else if constexpr (CLIP == CLIP_CHAR) {
		const int s = (int) v;
		return (W) (s < -128 ? -128 : (s > 127 ? 127 : s));
	}
	else if constexpr (CLIP == CLIP_SHORT) {
		const int s = (int) v;
		return (W) (s < -32768 ? -32768 : (s > 32767 ? 32767 : s));
	}
	else if constexpr (CLIP == CLIP_SHORT_ON_UINT) // the short limits compared as unsigned (conva.c:1130)
		return v < 0xFFFF8000u ? 0xFFFF8000u : 32767u;
	else
		return v;
}

template <typename IN, typename MID, typename ACC, int CLIP>
__global__ void __launch_bounds__(256)
conva_kernel(ApproxArgs a)
{
	typedef ApproxAcc<ACC> A;
	typedef ApproxMid<MID> M;
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.out_width * a.epp)
		return;
	const int x = e / a.epp;
	const int b = e - x * a.epp;
	const int gx = a.out_left + x - a.half_w;
	const int *runs = a.table;
	const int *cols = a.table + 2 * a.n_runs;
	for (int y = blockIdx.y; y < a.out_height; y += gridDim.y) {
		const int gy = a.out_top + y - a.half_h;
		typename A::work_t sum = A::zero();
		for (int c = 0; c < a.n_columns; c++) {
			const int run = cols[4 * c], factor = cols[4 * c + 1];
			const int x0 = runs[2 * run], x1 = runs[2 * run + 1];
			typename A::work_t vsum = A::zero();
			for (int k = cols[4 * c + 2]; k < cols[4 * c + 3]; k++) {
				const int row = min(max(gy + k, 0), a.im_height - 1) - a.in_top;
				const IN *src = (const IN *) (a.in + row * a.in_stride);
				typename M::work_t hsum = 0;
				for (int i = x0; i < x1; i++) {
					const int col = min(max(gx + i, 0), a.im_width - 1) - a.in_left;
					hsum = M::add(hsum, src[(long long) col * a.epp + b]);
				}
				vsum = A::add(vsum, M::fin(hsum));
			}
			sum = A::mad(sum, factor, vsum);
		}
		IN *dst = (IN *) (a.out + (long long) y * a.out_stride);
		dst[e] = (IN) approx_clip<CLIP>(A::fin(sum, a));
	}
}

// ---------------------------------------------------------------- convasep kernel
//
// table: n x {start, end, factor}.  One pass along x (vertical = 0) or y.
template <typename T>
struct SepClip {
	static __device__ __forceinline__ T run(long long v) { return (T) v; } // CLIP_NONE: plain assignment
};
#define SEP_CLIP(TYPE, LO, HI) \
	template <> \
	struct SepClip<TYPE> { \
		static __device__ __forceinline__ TYPE run(long long v) \
		{ \
			return (TYPE) (v < (LO) ? (LO) : (v > (HI) ? (HI) : v)); \
		} \
	};
SEP_CLIP(unsigned char, 0, UCHAR_MAX)
SEP_CLIP(signed char, SCHAR_MIN, SCHAR_MAX)
SEP_CLIP(unsigned short, 0, USHRT_MAX)
SEP_CLIP(short, SHRT_MIN, SHRT_MAX)
#undef SEP_CLIP

template <typename T, bool IS_SIGNED>
__global__ void __launch_bounds__(256)
convasep_int_kernel(ApproxArgs a)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.out_width * a.epp)
		return;
	const int x = e / a.epp;
	const int b = e - x * a.epp;
	for (int y = blockIdx.y; y < a.out_height; y += gridDim.y) {
		const int gx = a.out_left + x - (a.vertical ? 0 : a.half_w);
		const int gy = a.out_top + y - (a.vertical ? a.half_h : 0);
		long long sum = 0;
		for (int z = 0; z < a.n_columns; z++) {
			unsigned int isum = 0; // the line sum wraps in 32 bits
			for (int k = a.table[3 * z]; k < a.table[3 * z + 1]; k++) {
				const int col = min(max(gx + (a.vertical ? 0 : k), 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + (a.vertical ? k : 0), 0), a.im_height - 1) - a.in_top;
				const T *src = (const T *) (a.in + row * a.in_stride);
				if constexpr (IS_SIGNED)
					isum += (unsigned int) (int) src[(long long) col * a.epp + b];
				else
					isum += (unsigned int) src[(long long) col * a.epp + b];
			}
			const long long wide = IS_SIGNED ? (long long) (int) isum : (long long) isum;
			sum += (long long) a.table[3 * z + 2] * wide;
		}
		sum = (sum + a.rounding) / a.divisor + (a.vertical ? a.offset : 0);
		T *dst = (T *) (a.out + (long long) y * a.out_stride);
		dst[e] = SepClip<T>::run(sum);
	}
}

template <typename T>
__global__ void __launch_bounds__(256)
convasep_float_kernel(ApproxArgs a)
{
	const int e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= a.out_width * a.epp)
		return;
	const int x = e / a.epp;
	const int b = e - x * a.epp;
	for (int y = blockIdx.y; y < a.out_height; y += gridDim.y) {
		const int gx = a.out_left + x - (a.vertical ? 0 : a.half_w);
		const int gy = a.out_top + y - (a.vertical ? a.half_h : 0);
		double sum = 0;
		for (int z = 0; z < a.n_columns; z++) {
			double dsum = 0;
			for (int k = a.table[3 * z]; k < a.table[3 * z + 1]; k++) {
				const int col = min(max(gx + (a.vertical ? 0 : k), 0), a.im_width - 1) - a.in_left;
				const int row = min(max(gy + (a.vertical ? k : 0), 0), a.im_height - 1) - a.in_top;
				const T *src = (const T *) (a.in + row * a.in_stride);
				dsum = __dadd_rn(dsum, (double) src[(long long) col * a.epp + b]);
			}
			sum = __dadd_rn(sum, __dmul_rn((double) a.table[3 * z + 2], dsum));
		}
		sum = __ddiv_rn(sum, (double) a.divisor);
		if (a.vertical)
			sum = __dadd_rn(sum, (double) a.offset);
		T *dst = (T *) (a.out + (long long) y * a.out_stride);
		dst[e] = (T) sum;
	}
}

template <typename K>
static int approx_launch(K kernel, const ApproxArgs &a, const char *name)
{
	const int ne = a.out_width * a.epp;
	dim3 grid((ne + 255) / 256, a.out_height < 32768 ? a.out_height : 32768, 1);
	Gate gate(name);
	hipLaunchKernelGGL(kernel, grid, dim3(256, 1, 1), 0, stream(), a);
	VH_CHECK(hipGetLastError());
	return 0;
}

} // namespace vh

namespace {

int upload_table(_VipsHipConva *c)
{
	std::lock_guard<std::mutex> lock(c->mutex);
	if (c->d_table)
		return 0;
	c->d_table = (int *) upload(c->table.data(), c->table.size() * sizeof(int));
	return c->d_table ? 0 : -1;
}

// The approximated mask as plain integer taps: W[y][x] = sum of the factors of every box
// covering (x, y).
std::vector<double> effective_mask(const _VipsHipConva *c)
{
	std::vector<double> w((size_t) c->mask_width * c->mask_height, 0.0);
	if (c->separable) {
		for (const Line &l : c->lines.lines)
			for (int x = l.start; x < l.end; x++)
				w[x] += l.factor;
	}
	else {
		for (const Column &col : c->boxes.columns) {
			const Run &r = c->boxes.runs[col.band];
			for (int y = col.start; y < col.end; y++)
				for (int x = r.start; x < r.end; x++)
					w[(size_t) y * c->mask_width + x] += col.factor;
		}
	}
	return w;
}

void finish_plan(_VipsHipConva *c)
{
	c->d_table = nullptr;
	c->fast[0] = c->fast[1] = nullptr;
	c->w_negative = false;
	if (c->separable) {
		for (const Line &l : c->lines.lines) {
			c->table.push_back(l.start);
			c->table.push_back(l.end);
			c->table.push_back(l.factor);
			c->w_negative |= l.factor < 0;
		}
	}
	else {
		for (const Run &r : c->boxes.runs) {
			c->table.push_back(r.start);
			c->table.push_back(r.end);
		}
		for (const Column &col : c->boxes.columns) {
			c->table.push_back(col.band);
			c->table.push_back(col.factor);
			c->table.push_back(col.start);
			c->table.push_back(col.end);
			c->w_negative |= col.factor < 0;
		}
	}
	c->w_abs_sum = 0;
	if (c->separable) {
		for (const Line &l : c->lines.lines)
			c->w_abs_sum += (long long) abs(l.factor) * (l.end - l.start);
	}
	else {
		for (const Column &col : c->boxes.columns)
			c->w_abs_sum += (long long) abs(col.factor) * (col.end - col.start) *
				(c->boxes.runs[col.band].end - c->boxes.runs[col.band].start);
	}
}

// Is a pass over `format` pixels exactly the integer convolution with W?  True when no line
// sum, intermediate or total can wrap or change sign in the reference's types.
bool fast_ok(const _VipsHipConva *c, int format)
{
	if (getenv("VIPS_HIP_NO_APPROX_FAST"))
		return false;
	// float images, separable: the line sums are taken in double, so while they are exact (the
	// only regime in which the reference's rolling sums do not depend on tile geometry) the
	// weighted total equals the convolution with W summed in double -- convi on a float image
	// (convi.c:721-741), fused in convsep_f32.hip.  |W| is far below the 2^29 that kernel needs.
	if (format == VIPS_HIP_FORMAT_FLOAT)
		return c->separable && c->w_abs_sum < (1LL << 29);
	long long maxval;
	bool is_unsigned;
	switch (format) {
	case VIPS_HIP_FORMAT_UCHAR: maxval = 255; is_unsigned = true; break;
	case VIPS_HIP_FORMAT_CHAR: maxval = 128; is_unsigned = false; break;
	case VIPS_HIP_FORMAT_USHORT: maxval = 65535; is_unsigned = true; break;
	case VIPS_HIP_FORMAT_SHORT: maxval = 32768; is_unsigned = false; break;
	default: return false;
	}
	const int divisor = c->separable ? c->lines.divisor : c->boxes.divisor;
	const int offset = c->separable ? c->lines.offset : c->boxes.offset;
	if (c->w_abs_sum * maxval + divisor >= (1LL << 31) - 1)
		return false;
	if (offset <= -(1 << 30) || offset >= (1 << 30))
		return false;
	if (!c->separable) {
		// the horizontal intermediate is 16 bits wide for short lines (conva.c:992-1003)
		if (c->boxes.max_line < 256 && maxval * c->boxes.max_line >= (is_unsigned ? 65536 : 32768))
			return false;
		// unsigned totals wrap instead of going negative (conva.c:1099-1140)
		if (is_unsigned && (c->w_negative || offset < 0))
			return false;
	}
	return true;
}

// The convi plans of the fast path, made on first use.
int fast_plans(_VipsHipConva *c)
{
	std::lock_guard<std::mutex> lock(c->mutex);
	if (c->fast[0])
		return 0;
	const std::vector<double> w = effective_mask(c);
	const int divisor = c->separable ? c->lines.divisor : c->boxes.divisor;
	const int offset = c->separable ? c->lines.offset : c->boxes.offset;
	if (c->separable) {
		VipsHipConv *h = vips_hip_conv_new(w.data(), c->mask_width, 1, divisor, 0.0, VIPS_HIP_PRECISION_INTEGER);
		VipsHipConv *v = vips_hip_conv_new(w.data(), 1, c->mask_width, divisor, offset, VIPS_HIP_PRECISION_INTEGER);
		if (!h || !v) {
			vips_hip_conv_free(h);
			vips_hip_conv_free(v);
			return -1;
		}
		h->rounding = v->rounding = c->lines.rounding;
		h->no_vector = v->no_vector = true;
		c->fast[1] = v;
		c->fast[0] = h;
	}
	else {
		VipsHipConv *p = vips_hip_conv_new(w.data(), c->mask_width, c->mask_height, divisor, offset,
			VIPS_HIP_PRECISION_INTEGER);
		if (!p)
			return -1;
		p->rounding = c->boxes.rounding;
		p->no_vector = true;
		c->fast[0] = p;
	}
	return 0;
}

int check_pair(const char *domain, const VipsHipRegion *in, const VipsHipRegion *out, int grow_w, int grow_h)
{
	if (check_region(domain, in) || check_region(domain, out))
		return -1;
	if (in->bands != out->bands || in->format != out->format) {
		error(domain, "output region has the wrong bands or format");
		return -1;
	}
	if (in->im_width != out->im_width || in->im_height != out->im_height) {
		error(domain, "input and output images must have the same size");
		return -1;
	}
	// the window must cover the output rect grown by the mask, clipped to the image
	int x0 = out->left - grow_w / 2, x1 = out->left + out->width - 1 - grow_w / 2 + grow_w - 1;
	int y0 = out->top - grow_h / 2, y1 = out->top + out->height - 1 - grow_h / 2 + grow_h - 1;
	x0 = x0 < 0 ? 0 : x0;
	y0 = y0 < 0 ? 0 : y0;
	x1 = x1 > in->im_width - 1 ? in->im_width - 1 : x1;
	y1 = y1 > in->im_height - 1 ? in->im_height - 1 : y1;
	if (x0 < in->left || y0 < in->top || x1 >= in->left + in->width || y1 >= in->top + in->height) {
		error(domain, "input region too small");
		return -1;
	}
	return 0;
}

ApproxArgs make_args(const _VipsHipConva *c, const VipsHipRegion *in, const VipsHipRegion *out)
{
	ApproxArgs a;
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_left = in->left;
	a.in_top = in->top;
	a.im_width = in->im_width;
	a.im_height = in->im_height;
	a.out_left = out->left;
	a.out_top = out->top;
	a.out_width = out->width;
	a.out_height = out->height;
	a.epp = region_elems_per_pel(in);
	a.table = c->d_table;
	a.vertical = 0;
	if (c->separable) {
		a.half_w = a.half_h = c->lines.width / 2;
		a.n_runs = 0;
		a.n_columns = (int) c->lines.lines.size();
		a.divisor = c->lines.divisor;
		a.rounding = c->lines.rounding;
		a.offset = c->lines.offset;
	}
	else {
		a.half_w = c->mask_width / 2;
		a.half_h = c->mask_height / 2;
		a.n_runs = (int) c->boxes.runs.size();
		a.n_columns = (int) c->boxes.columns.size();
		a.divisor = c->boxes.divisor;
		a.rounding = c->boxes.rounding;
		a.offset = c->boxes.offset;
	}
	return a;
}

} // namespace

extern "C" {

VipsHipConva *vips_hip_conva_new(const double *mask, int mask_width, int mask_height, double scale,
	double offset, int layers, int cluster)
{
	if (!mask || mask_width <= 0 || mask_height <= 0) {
		error("conva", "bad mask");
		return nullptr;
	}
	if (mask_width > 100000 || mask_height > 100000 || (long long) mask_width * mask_height > (1 << 24)) {
		error("conva", "matrix image too large");
		return nullptr;
	}
	// conva.c:1304-1316
	if (layers < 1 || layers > 1000) {
		error("conva", "parameter layers not set");
		return nullptr;
	}
	if (cluster < 1 || cluster > 100) {
		error("conva", "parameter cluster not set");
		return nullptr;
	}
	std::unique_ptr<_VipsHipConva> c(new _VipsHipConva);
	c->separable = false;
	c->mask_width = mask_width;
	c->mask_height = mask_height;
	const IntMask m = intize(mask, mask_width, mask_height, scale, offset);
	if (const char *why = decompose_boxes(c->boxes, m, layers, cluster)) {
		error("conva", "%s", why);
		return nullptr;
	}
	finish_plan(c.get());
	return c.release();
}

VipsHipConva *vips_hip_convasep_new(const double *mask, int mask_n, double scale, double offset, int layers)
{
	if (!mask || mask_n <= 0) {
		error("convasep", "bad mask");
		return nullptr;
	}
	if (mask_n > 100000) {
		error("convasep", "matrix image too large");
		return nullptr;
	}
	if (layers < 1 || layers > 1000) {
		error("convasep", "parameter layers not set");
		return nullptr;
	}
	std::unique_ptr<_VipsHipConva> c(new _VipsHipConva);
	c->separable = true;
	c->mask_width = mask_n;
	c->mask_height = 1;
	const IntMask m = intize(mask, mask_n, 1, scale, offset);
	if (const char *why = decompose_lines(c->lines, m, layers)) {
		error("convasep", "%s", why);
		return nullptr;
	}
	finish_plan(c.get());
	return c.release();
}

void vips_hip_conva_free(VipsHipConva *c)
{
	if (!c)
		return;
	vips_hip_free(c->d_table);
	vips_hip_conv_free(c->fast[0]);
	vips_hip_conv_free(c->fast[1]);
	delete c;
}

// Host-side view of the decomposition (no device needed).
//   2-D:       info = {n_runs, n_columns, divisor, rounding, offset, max_line};
//              lines = n_runs x {start, end}, n_columns x {run, factor, first row, last row + 1}
//   separable: info = {n_lines, divisor, rounding, offset, 0, 0}; lines = n x {start, end, factor}
// Returns the number of ints in lines[], -1 when max_ints is too small.
int vips_hip_conva_get_lines(const VipsHipConva *c, int *info, int *lines, int max_ints)
{
	if (!c || !info || !lines) {
		error("conva", "null argument");
		return -1;
	}
	if ((int) c->table.size() > max_ints) {
		error("conva", "buffer too small for %d ints", (int) c->table.size());
		return -1;
	}
	memcpy(lines, c->table.data(), c->table.size() * sizeof(int));
	if (c->separable) {
		info[0] = (int) c->lines.lines.size();
		info[1] = c->lines.divisor;
		info[2] = c->lines.rounding;
		info[3] = c->lines.offset;
		info[4] = info[5] = 0;
	}
	else {
		info[0] = (int) c->boxes.runs.size();
		info[1] = (int) c->boxes.columns.size();
		info[2] = c->boxes.divisor;
		info[3] = c->boxes.rounding;
		info[4] = c->boxes.offset;
		info[5] = c->boxes.max_line;
	}
	return (int) c->table.size();
}

// vips_conva_hgenerate + vips_conva_vgenerate (conva.c:876-1020, 1099-1198) in one pass.
int vips_hip_conva_gen(const VipsHipConva *plan, const VipsHipRegion *in, const VipsHipRegion *out)
{
	if (ensure_init())
		return -1;
	if (!plan || plan->separable) {
		error("conva", "not a conva plan");
		return -1;
	}
	if (plan_device("conva", &plan->device))
		return -1;
	_VipsHipConva *c = const_cast<_VipsHipConva *>(plan);
	if (check_pair("conva", in, out, c->mask_width, c->mask_height))
		return -1;
	const int fmt = format_real(in->format);
	if (fast_ok(c, fmt)) {
		if (fast_plans(c))
			return -1;
		return vips_hip_conv_gen(c->fast[0], in, out);
	}
	if (upload_table(c))
		return -1;
	const ApproxArgs a = make_args(c, in, out);
	const bool small = c->boxes.max_line < 256;
	switch (fmt) {
	case VIPS_HIP_FORMAT_UCHAR:
		return small ? approx_launch(conva_kernel<unsigned char, unsigned short, unsigned int, CLIP_UCHAR>, a, "conva")
					 : approx_launch(conva_kernel<unsigned char, unsigned int, unsigned int, CLIP_UCHAR>, a, "conva");
	case VIPS_HIP_FORMAT_CHAR:
		return small ? approx_launch(conva_kernel<signed char, short, int, CLIP_CHAR>, a, "conva")
					 : approx_launch(conva_kernel<signed char, int, int, CLIP_CHAR>, a, "conva");
	case VIPS_HIP_FORMAT_USHORT:
		return small ? approx_launch(conva_kernel<unsigned short, unsigned short, unsigned int, CLIP_USHORT>, a, "conva")
					 : approx_launch(conva_kernel<unsigned short, unsigned int, unsigned int, CLIP_USHORT>, a, "conva");
	case VIPS_HIP_FORMAT_SHORT:
		return small ? approx_launch(conva_kernel<short, short, int, CLIP_SHORT>, a, "conva")
					 : approx_launch(conva_kernel<short, int, int, CLIP_SHORT>, a, "conva");
	case VIPS_HIP_FORMAT_UINT:
		return small ? approx_launch(conva_kernel<unsigned int, unsigned short, unsigned int, CLIP_SHORT_ON_UINT>, a, "conva")
					 : approx_launch(conva_kernel<unsigned int, unsigned int, unsigned int, CLIP_NONE>, a, "conva");
	case VIPS_HIP_FORMAT_INT:
		return small ? approx_launch(conva_kernel<int, short, int, CLIP_NONE>, a, "conva")
					 : approx_launch(conva_kernel<int, int, int, CLIP_NONE>, a, "conva");
	case VIPS_HIP_FORMAT_FLOAT:
		return approx_launch(conva_kernel<float, float, float, CLIP_NONE>, a, "conva");
	case VIPS_HIP_FORMAT_DOUBLE:
		return approx_launch(conva_kernel<double, double, double, CLIP_NONE>, a, "conva");
	default:
		break;
	}
	error("conva", "unsupported band format %d", in->format);
	return -1;
}

// vips_convasep_generate_horizontal / _vertical (convasep.c:516-590, 677-750): one pass.
int vips_hip_convasep_gen(const VipsHipConva *plan, const VipsHipRegion *in, const VipsHipRegion *out,
	int vertical)
{
	if (ensure_init())
		return -1;
	if (!plan || !plan->separable) {
		error("convasep", "not a convasep plan");
		return -1;
	}
	if (plan_device("convasep", &plan->device))
		return -1;
	_VipsHipConva *c = const_cast<_VipsHipConva *>(plan);
	vertical = vertical ? 1 : 0;
	if (check_pair("convasep", in, out, vertical ? 1 : c->lines.width, vertical ? c->lines.width : 1))
		return -1;
	const int fmt = format_real(in->format);
	if (fast_ok(c, fmt)) {
		if (fast_plans(c))
			return -1;
		return vips_hip_conv_gen(c->fast[vertical], in, out);
	}
	if (upload_table(c))
		return -1;
	ApproxArgs a = make_args(c, in, out);
	a.vertical = vertical;
	switch (fmt) {
	case VIPS_HIP_FORMAT_UCHAR: return approx_launch(convasep_int_kernel<unsigned char, false>, a, "convasep");
	case VIPS_HIP_FORMAT_CHAR: return approx_launch(convasep_int_kernel<signed char, true>, a, "convasep");
	case VIPS_HIP_FORMAT_USHORT: return approx_launch(convasep_int_kernel<unsigned short, false>, a, "convasep");
	case VIPS_HIP_FORMAT_SHORT: return approx_launch(convasep_int_kernel<short, true>, a, "convasep");
	case VIPS_HIP_FORMAT_UINT: return approx_launch(convasep_int_kernel<unsigned int, false>, a, "convasep");
	case VIPS_HIP_FORMAT_INT: return approx_launch(convasep_int_kernel<int, true>, a, "convasep");
	case VIPS_HIP_FORMAT_FLOAT: return approx_launch(convasep_float_kernel<float>, a, "convasep");
	case VIPS_HIP_FORMAT_DOUBLE: return approx_launch(convasep_float_kernel<double>, a, "convasep");
	default: break;
	}
	error("convasep", "unsupported band format %d", in->format);
	return -1;
}

} // extern "C"

namespace vh {

// A whole uchar image whose conva pass IS an integer convolution (the fast path above): the matrix-core 2-D kernel
// (conv_u8_mfma.hip) instead of the general convi one when its exactness bounds and its rounding ((sum + scale / 2) /
// scale: conva's (divisor + 1) / 2 is that for an even divisor) hold.  1: not covered (nothing launched).
int conva_fast_image(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConva *plan)
{
	_VipsHipConva *c = const_cast<_VipsHipConva *>(plan);
	if (c->separable || in->format != VIPS_HIP_FORMAT_UCHAR || !fast_ok(c, VIPS_HIP_FORMAT_UCHAR) || getenv("VIPS_HIP_NO_CONVA_MFMA"))
		return 1;
	if (plan_device("conva", &plan->device) || fast_plans(c))
		return -1;
	return conv_u8_mfma_2d_try(in, out, c->fast[0]);
}


// Image-level halves used by ops_colour_conv.cpp.

// Both passes of a convasep through the fused separable kernel; 1 when not covered.
int convasep_fused(const _VipsHipImage *in, _VipsHipImage *out, const _VipsHipConva *plan)
{
	_VipsHipConva *c = const_cast<_VipsHipConva *>(plan);
	if (!c->separable || !fast_ok(c, in->format))
		return 1;
	if (fast_plans(c))
		return -1;
	return convsep_f32_fused(in, out, c->fast[0], (double) c->lines.offset);
}

} // namespace vh
