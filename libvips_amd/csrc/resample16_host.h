// Host side of the ushort streaming resample kernels (resample16_body.h): the vertical reduce's
// schedule, geometry, launches.  Included by resample16.hip (kernel launches) and by
// tests/emul/resample16_emul.cpp (host fiber runs).
#pragma once

#include "reduce_u8.h"
#include "resample.h"
#include "resample16_body.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace vh {

// defined by the including file; 0 on success
static int r16_launch_v(int which, const R16VArgs &a, int gx, int gy);
static int r16_launch_h(int which, int bands, const R16HArgs &a, int gx, int gy, size_t lds);
static int r16_launch_boxc(int bands, int lanes_per_box, const R16HArgs &a, int gx, int gy);

namespace {

struct R16Sched {
	int r_base = 0, seg_rows = 0, segs = 0;
	const int *d_seg_pairs = nullptr;
	const R16Pair *d_sched = nullptr;
	bool ok = false;
};

// whole images only, rows of whole 8-byte groups, dword-aligned
bool r16_whole(const VipsHipRegion *in, const VipsHipRegion *out, bool vertical)
{
	if (in->format != VIPS_HIP_FORMAT_USHORT || out->format != VIPS_HIP_FORMAT_USHORT || in->bands != out->bands ||
		in->bands < 1 || in->bands > 4)
		return false;
	if (in->left || in->top || out->left || out->top || in->width != in->im_width || in->height != in->im_height ||
		out->width != out->im_width || out->height != out->im_height)
		return false;
	if (vertical ? in->width != out->width : in->height != out->height)
		return false;
	// rows read as dwords; written as dwords (vertical; horizontal with an even band count) or as ushorts
	if (((uintptr_t) in->data | in->stride) & 3)
		return false;
	if (((uintptr_t) out->data | out->stride) & ((vertical || in->bands % 2 == 0) ? 3 : 1))
		return false;
	return true;
}

int r16_counter(int **counter)
{
	*counter = (int *) vips_hip_malloc(sizeof(int));
	if (!*counter)
		return -1;
	if (hipMemsetAsync(*counter, 0, sizeof(int), stream()) != hipSuccess) {
		vips_hip_free(*counter);
		return hip_failed(hipErrorUnknown, "hipMemsetAsync");
	}
	return 0;
}

// the schedule of a vertical reduce to `out_height` rows in segments of seg_rows: built once per plan
// and segment height, kept with the plan's other device tables (pos_cache, freed with the plan)
// (bias: what a sample is stored less -- 32768 for the ushort kernel's signed 16-bit lanes, 0 for bytes)
// (returns the schedule BY VALUE: what the host keeps of it sits in the plan, r->blob_info, under the plan's
// lock -- no pointer into a shared map outlives the lock; false + an error message on failure)
static R16Sched r16_from_info(const void *blob, const std::vector<long long> &v)
{
	R16Sched s;
	s.r_base = (int) v[0];
	s.seg_rows = (int) v[1];
	s.segs = (int) v[2];
	s.ok = v[3] != 0;
	if (s.ok) {
		s.d_seg_pairs = (const int *) blob;
		s.d_sched = (const R16Pair *) ((const unsigned char *) blob + v[4]);
	}
	return s;
}

bool r16_schedule(_VipsHipReduce *r, int out_height, int tile, int seg_rows, R16Sched *out, unsigned int bias = 32768u)
{
	std::lock_guard<std::mutex> plan_lock(r->mutex);
	const auto key = std::make_tuple((bias ? -16 : -(1 << 24)) - seg_rows, out_height, tile);
	auto it = r->pos_cache.find(key);
	if (it != r->pos_cache.end()) {
		auto h = r->blob_info.find(it->second);
		if (h == r->blob_info.end()) {
			error("reduce", "a streaming schedule lost its host side");
			return false;
		}
		*out = r16_from_info(it->second, h->second);
		return true;
	}
	std::vector<ReducePos> pos;
	reduce_positions(r, 0, out_height, tile, pos);
	const int n = r->n_point;
	R16Sched s;
	s.seg_rows = seg_rows;
	s.segs = (out_height + seg_rows - 1) / seg_rows;
	int r_lo = pos[0].first, r_hi = pos[0].first + n - 1;
	for (int y = 0; y < out_height; y++) {
		r_lo = pos[y].first < r_lo ? pos[y].first : r_lo;
		r_hi = pos[y].first + n - 1 > r_hi ? pos[y].first + n - 1 : r_hi;
	}
	s.r_base = r_lo;
	const int npairs = (r_hi - r_lo) / 2 + 1;
	std::vector<R16Pair> sched((size_t) npairs + R16_PF + 1);
	memset(sched.data(), 0, sched.size() * sizeof(R16Pair));
	std::vector<int> busy_until(R16_SLOTS, -1); // last pair of the slot's current output
	std::vector<int> seg_pairs((size_t) 2 * s.segs);
	s.ok = true;
	for (int y = 0; y < out_height && s.ok; y++) {
		const int slot = y % R16_SLOTS;
		const int ps = (pos[y].first - r_lo) >> 1, pe = (pos[y].first + n - 1 - r_lo) >> 1;
		if (ps <= busy_until[slot]) {
			s.ok = false; // more than 8 outputs in flight (a shrink close to 1): the general kernel
			break;
		}
		busy_until[slot] = pe;
		const short *c = &r->matrixs[(size_t) pos[y].phase * n];
		int csum = 0;
		for (int k = 0; k < n; k++)
			csum += c[k];
		for (int p = ps; p <= pe; p++) {
			const int k0 = r_lo + 2 * p - pos[y].first; // tap of the pair's first row
			const unsigned int lo = k0 >= 0 && k0 < n ? (unsigned short) c[k0] : 0u;
			const unsigned int hi = k0 + 1 >= 0 && k0 + 1 < n ? (unsigned short) c[k0 + 1] : 0u;
			sched[p].c2[slot] = lo | (hi << 16);
		}
		sched[ps].start_mask |= 1u << slot;
		sched[ps].init[slot] = (int) (2048u + bias * (unsigned int) csum);
		sched[pe].ret_mask |= 1u << slot;
		sched[pe].yret[slot] = y;
		const int seg = y / seg_rows;
		if (y % seg_rows == 0)
			seg_pairs[2 * seg] = ps;
		if (y % seg_rows == 0 || pe > seg_pairs[2 * seg + 1])
			seg_pairs[2 * seg + 1] = pe;
		if (ps < seg_pairs[2 * seg])
			seg_pairs[2 * seg] = ps;
	}
	void *blob = nullptr;
	size_t head_bytes = 0;
	if (s.ok) {
		const size_t head = (seg_pairs.size() * sizeof(int) + 15) & ~(size_t) 15;
		head_bytes = head;
		std::vector<unsigned char> bytes(head + sched.size() * sizeof(R16Pair));
		memcpy(bytes.data(), seg_pairs.data(), seg_pairs.size() * sizeof(int));
		memcpy(bytes.data() + head, sched.data(), sched.size() * sizeof(R16Pair));
		blob = upload(bytes.data(), bytes.size());
		if (!blob)
			return false;
		s.d_seg_pairs = (const int *) blob;
		s.d_sched = (const R16Pair *) ((const unsigned char *) blob + head);
	}
	else {
		static const unsigned char placeholder[16] = { 0 };
		blob = upload(placeholder, sizeof(placeholder)); // (a block so that the refusal is cached with the plan too)
		if (!blob)
			return false;
	}
	r->pos_cache[key] = (ReducePos *) blob;
	r->blob_info[blob] = { (long long) s.r_base, (long long) s.seg_rows, (long long) s.segs, s.ok ? 1LL : 0LL, (long long) head_bytes };
	*out = s;
	return true;
}

void r16_v_geometry(R16VArgs *a, const VipsHipRegion *in, const VipsHipRegion *out, int seg_rows, int elem_bytes = 2)
{
	memset(a, 0, sizeof(*a));
	a->in = (const unsigned char *) in->data;
	a->out = (unsigned char *) out->data;
	a->in_stride = (long long) in->stride;
	a->out_stride = (long long) out->stride;
	a->row_bytes = in->width * in->bands * elem_bytes;
	a->in_height = in->height;
	a->out_height = out->height;
	a->strips = (a->row_bytes + R16_NT * 8 - 1) / (R16_NT * 8);
	a->seg_rows = seg_rows;
	a->segs = (out->height + seg_rows - 1) / seg_rows;
	a->off_slot = 0;
}

int r16_seg_rows(int out_height, int strips, int min_rows, int per_cu = 8)
{
	// ~per_cu blocks of 4 waves per CU
	int want = (256 * per_cu + strips - 1) / strips;
	int seg = (out_height + want - 1) / want;
	if (seg < min_rows)
		seg = min_rows;
	if (getenv("VIPS_HIP_R16_SEG"))
		seg = atoi(getenv("VIPS_HIP_R16_SEG"));
	if (seg < 1)
		seg = 1;
	return seg > out_height ? out_height : seg;
}

} // namespace

// 1 = handled, 0 = not these kernels' case (nothing launched), -1 = error
int reducev16_stream_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	if (getenv("VIPS_HIP_NO_STREAM16") || !r16_whole(in, out, true) || (in->width * in->bands * 2) % 8)
		return 0;
	const int strips = (in->width * in->bands * 2 + R16_NT * 8 - 1) / (R16_NT * 8);
	// a segment re-reads n_point - 1 rows: at least 4 times that many input rows per segment
	int min_rows = (int) (4.0 * r->n_point / r->shrink) + 1;
	const int seg_rows = r16_seg_rows(out->height, strips, min_rows);
	R16Sched sched_v;
	if (!r16_schedule(r, out->height, tile, seg_rows, &sched_v))
		return -1;
	const R16Sched *s = &sched_v;
	if (!s->ok)
		return 0;
	R16VArgs a;
	r16_v_geometry(&a, in, out, seg_rows);
	a.r_base = s->r_base;
	a.sched = s->d_sched;
	a.seg_pairs = s->d_seg_pairs;
	if (r16_counter(&a.counter))
		return -1;
	const int items = a.strips * a.segs;
	Gate gate("reducev_u16_stream");
	const int rc = r16_launch_v(0, a, items < 256 * 8 ? items : 256 * 8, 1);
	vips_hip_free(a.counter);
	return rc ? -1 : 1;
}

// the same walk for uchar images whose rows take a coefficient row each (a fractional shrink)
int reducev8_stream_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, int tile)
{
	if (getenv("VIPS_HIP_NO_STREAM8"))
		return 0;
	if (in->format != VIPS_HIP_FORMAT_UCHAR || out->format != VIPS_HIP_FORMAT_UCHAR || in->bands != out->bands ||
		in->bands < 1 || in->bands > 4)
		return 0;
	if (in->left || in->top || out->left || out->top || in->width != in->im_width || in->height != in->im_height ||
		out->width != out->im_width || out->height != out->im_height || in->width != out->width)
		return 0;
	if ((((uintptr_t) in->data | in->stride | (uintptr_t) out->data | out->stride) & 3) || (in->width * in->bands) % 8)
		return 0;
	const int strips = (in->width * in->bands + R16_NT * 8 - 1) / (R16_NT * 8);
	// about three blocks per CU (one round of what the kernel's registers let a CU hold), segments not
	// shorter than twice the rows a segment re-reads: on 8192^2 x 3 by 7.3 segments of 26 / 18 / 13 / 9
	// / 6 rows ran 0.111 / 0.092 / 0.116 / 0.105 / 0.118 ms (profiles/NOTES.md R4.8)
	int min_rows = (int) (2.0 * r->n_point / r->shrink) + 1;
	const int seg_rows = r16_seg_rows(out->height, strips, min_rows, 3);
	R16Sched sched_v;
	if (!r16_schedule(r, out->height, tile, seg_rows, &sched_v, 0u))
		return -1;
	const R16Sched *s = &sched_v;
	if (!s->ok)
		return 0;
	R16VArgs a;
	r16_v_geometry(&a, in, out, seg_rows, 1);
	a.r_base = s->r_base;
	a.sched = s->d_sched;
	a.seg_pairs = s->d_seg_pairs;
	if (r16_counter(&a.counter))
		return -1;
	const int items = a.strips * a.segs;
	Gate gate("reducev_u8_stream");
	const int rc = r16_launch_v(2, a, items < 256 * 8 ? items : 256 * 8, 1);
	vips_hip_free(a.counter);
	return rc ? -1 : 1;
}

int shrinkv16_stream_try(int vshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	if (getenv("VIPS_HIP_NO_STREAM16") || !r16_whole(in, out, true) || (in->width * in->bands * 2) % 8 || vshrink < 1 ||
		vshrink > 32768)
		return 0;
	R16VArgs a;
	r16_v_geometry(&a, in, out, out->height);
	a.vshrink = vshrink;
	a.mult = (unsigned int) (((1ULL << 32) + vshrink - 1) / vshrink); // (vshrink 1 is not multiplied)
	const int gx = (a.row_bytes + R16_NT * 16 - 1) / (R16_NT * 16);
	int gy = 32768 / gx;
	gy = gy < 1 ? 1 : gy > out->height ? out->height : gy;
	Gate gate("shrinkv_u16_stream");
	const int rc = r16_launch_v(1, a, gx, gy);
	return rc ? -1 : 1;
}

int reduceh16_stream_try(_VipsHipReduce *r, const VipsHipRegion *in, const VipsHipRegion *out, const ReducePos *pos,
	const short *table)
{
	if (getenv("VIPS_HIP_NO_STREAM16") || !r16_whole(in, out, false) || (in->width * in->bands * 2) % 4)
		return 0;
	// the staged span of a block must fit
	const int PB = in->bands * 2;
	const long long span = (long long) ((R16H_PX + 1) * r->shrink + r->n_point + 2) * PB + 8;
	const long long padded = span + ((span >> 6) << 3) + 16;
	if (padded * R16H_ROWS > 60 * 1024)
		return 0;
	R16HArgs a;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_width = in->width;
	a.out_width = out->width;
	a.height = out->height;
	a.bands = in->bands;
	a.n_point = r->n_point;
	a.span_dwords = (int) ((padded + 3) >> 2);
	a.pos = (const R16Pos *) pos;
	a.table = table;
	const int gx = (out->width + R16H_PX - 1) / R16H_PX;
	int gy = 8192 / gx;
	gy = gy < 1 ? 1 : gy;
	const int groups = (out->height + R16H_ROWS - 1) / R16H_ROWS;
	gy = groups < gy ? groups : gy;
	Gate gate("reduceh_u16_lds");
	const int rc = r16_launch_h(0, in->bands, a, gx, gy, (size_t) a.span_dwords * 4 * R16H_ROWS);
	return rc ? -1 : 1;
}

// vips_shrink on a whole ushort image in one kernel (shrinkbox16_body): 1 launched, 0 not its case, -1 error
int shrinkbox16_try(int hshrink, int vshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	if (getenv("VIPS_HIP_NO_STREAM16") || getenv("VIPS_HIP_NO_SHRINKBOX16") || hshrink < 2 || vshrink < 2 || hshrink > 4096 ||
		vshrink > 4096)
		return 0;
	if (in->format != VIPS_HIP_FORMAT_USHORT || out->format != VIPS_HIP_FORMAT_USHORT || in->bands != out->bands ||
		in->bands < 1 || in->bands > 4)
		return 0;
	if (in->left || in->top || out->left || out->top || in->width != in->im_width || in->height != in->im_height ||
		out->width != out->im_width || out->height != out->im_height)
		return 0;
	// (the pixel stores: dwords for an even band count; the wide loads want rows that start on 16 bytes)
	if ((((uintptr_t) in->data | in->stride) & 1) || (((uintptr_t) out->data | out->stride) & (in->bands % 2 == 0 ? 3 : 1)))
		return 0;
	R16HArgs a;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_width = in->width;
	a.in_height = in->height;
	a.out_width = out->width;
	a.height = out->height;
	a.bands = in->bands;
	a.hshrink = hshrink;
	a.vshrink = vshrink;
	a.aligned16 = !(((uintptr_t) in->data | in->stride) & 15);
	a.mult = (unsigned int) (((1ULL << 32) + hshrink - 1) / hshrink);
	a.multv = (unsigned int) (((1ULL << 32) + vshrink - 1) / vshrink);
	Gate gate("shrinkbox_u16");
	// boxes of 16 / 32 / 64 bytes on 16-byte rows: the lanes on consecutive 16-byte groups (shrinkbox16c), the ragged
	// last column of a ceil shrink by the thread-per-pixel form
	const int box_bytes = hshrink * in->bands * 2;
	a.x_first = 0;
	a.x_full = in->width / hshrink;
	if (a.aligned16 && in->bands != 3 && (box_bytes == 16 || box_bytes == 32 || box_bytes == 64) && a.x_full > 0 &&
		!getenv("VIPS_HIP_NO_SHRINKBOX16C")) {
		const int xf = a.x_full < out->width ? a.x_full : out->width;
		a.x_full = xf;
		const long long lanes = (long long) xf * (box_bytes / 16);
		const int gx = (int) ((lanes + R16_NT - 1) / R16_NT);
		int gy = 32768 / gx;
		gy = gy < 1 ? 1 : gy > out->height ? out->height : gy;
		if (r16_launch_boxc(in->bands, box_bytes / 16, a, gx, gy))
			return -1;
		if (xf >= out->width)
			return 1;
		a.x_first = xf;
	}
	const int cols = out->width - a.x_first;
	const int gx = (cols + R16_NT - 1) / R16_NT;
	int gy = 16384 / gx;
	gy = gy < 1 ? 1 : gy;
	gy = out->height < gy ? out->height : gy;
	const int rc = r16_launch_h(2, in->bands, a, gx, gy, 0);
	return rc ? -1 : 1;
}

int shrinkh16_stream_try(int hshrink, const VipsHipRegion *in, const VipsHipRegion *out)
{
	if (getenv("VIPS_HIP_NO_STREAM16") || !r16_whole(in, out, false) || hshrink < 1 || hshrink > 32768)
		return 0;
	if (in->bands % 2 == 0 && (in->width * in->bands * 2) % 4)
		return 0;
	R16HArgs a;
	memset(&a, 0, sizeof(a));
	a.in = (const unsigned char *) in->data;
	a.out = (unsigned char *) out->data;
	a.in_stride = (long long) in->stride;
	a.out_stride = (long long) out->stride;
	a.in_width = in->width;
	a.out_width = out->width;
	a.height = out->height;
	a.bands = in->bands;
	a.hshrink = hshrink;
	a.mult = (unsigned int) (((1ULL << 32) + hshrink - 1) / hshrink);
	const int gx = (out->width + R16_NT - 1) / R16_NT;
	int gy = 8192 / gx;
	gy = gy < 1 ? 1 : gy;
	const int groups = (out->height + R16H_ROWS - 1) / R16H_ROWS;
	gy = groups < gy ? groups : gy;
	Gate gate("shrinkh_u16_stream");
	const int rc = r16_launch_h(1, in->bands, a, gx, gy, 0);
	return rc ? -1 : 1;
}

} // namespace vh
