// Host side of the fused resize + sharpen kernel: tables, geometry, LDS layout, launches.
// Included by resize_sharpen.hip (which defines rsh_launch() as the kernel launch) and by
// tests/emul/resize_sharpen_emul.cpp (which defines it as the host fiber run of the same body).
#pragma once

#include "colour.h"
#include "reduce_u8.h"
#include "resample.h"
#include "resize_sharpen_body.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace vh {

// defined by the including file: launch `blocks` blocks of the <VS, 7> body; 0 on success
static int rsh_launch(int vs, const RshArgs &a, const RshPtrs &p, unsigned int blocks, size_t lds);

namespace {

// the per-device tables of the kernel, made once from the reference's own table code
// (colour.hip: calcul_tables = LabQ2sRGB.c:130-160, table_init = XYZ2Lab.c:92-106)
struct RshTables {
	float *v2Y = nullptr;
	RshPair *Y2v = nullptr;
};

const RshTables *rsh_tables()
{
	static std::mutex mutex;
	static std::map<int, RshTables> by_device;
	std::lock_guard<std::mutex> lock(mutex);
	const int device = current_device();
	auto it = by_device.find(device);
	if (it != by_device.end())
		return it->second.Y2v ? &it->second : nullptr;
	RshTables &tb = by_device[device];
	std::vector<float> v2Y, cb;
	std::vector<int> Y2v;
	colour_tables_host(v2Y, Y2v, cb);
	std::vector<RshPair> y2(256);
	for (int i = 0; i < 256; i++) {
		y2[i].x = (float) Y2v[i];
		y2[i].y = (float) (Y2v[i + 1] - Y2v[i]);
	}
	tb.v2Y = (float *) upload(v2Y.data(), 256 * sizeof(float));
	RshPair *yp = (RshPair *) upload(y2.data(), y2.size() * sizeof(RshPair));
	if (!tb.v2Y || !yp)
		return nullptr;
	tb.Y2v = yp;
	return &tb;
}

// the part of a sharpen LUT that is not constant, as shorts on the device (kept per LUT contents)
struct RshLut {
	int lo = 0, n = 0, below = 0, above = 0;
	short *d = nullptr;
	bool ok = false;
};

const RshLut *rsh_lut(const int *lut)
{
	static std::mutex mutex;
	static std::map<std::pair<int, std::vector<int>>, RshLut> cache; // (device, signature)
	int lo = 0, hi = 65535;
	while (lo < 65536 && lut[lo] == lut[0])
		lo++;
	while (hi >= 0 && lut[hi] == lut[65535])
		hi--;
	RshLut l;
	l.below = lut[0];
	l.above = lut[65535];
	l.lo = lo < 65536 ? lo : 0;
	l.n = lo < 65536 && hi >= lo ? hi - lo + 1 : 0;
	// the LUT is a function of five doubles: its signature is where it bends and what it holds there
	std::vector<int> sig = { l.lo, l.n, l.below, l.above };
	for (int k = 0; k < l.n; k += l.n / 16 + 1)
		sig.push_back(lut[l.lo + k]);
	std::lock_guard<std::mutex> lock(mutex);
	const auto key = std::make_pair(current_device(), sig);
	auto it = cache.find(key);
	if (it != cache.end())
		return &it->second;
	if (cache.size() > 64)
		return nullptr; // (a caller cycling through LUTs: the unfused path)
	std::vector<short> s((size_t) l.n + 1, 0);
	l.ok = l.n <= 6144;
	for (int k = 0; k < l.n && l.ok; k++) {
		if (lut[l.lo + k] < -32768 || lut[l.lo + k] > 32767)
			l.ok = false;
		s[k] = (short) lut[l.lo + k];
	}
	if (l.ok) {
		l.d = (short *) upload(s.data(), s.size() * sizeof(short));
		if (!l.d)
			return nullptr;
	}
	return &(cache[key] = l);
}

bool rsh_regular(const std::vector<ReducePos> &pos, int *first0, int *phase)
{
	if (pos.empty())
		return false;
	*first0 = pos[0].first;
	*phase = pos[0].phase;
	for (size_t k = 0; k < pos.size(); k++)
		if (pos[k].first != *first0 + 2 * (int) k || pos[k].phase != *phase)
			return false;
	return true;
}

int rsh_env(const char *name, int fallback)
{
	const char *v = getenv(name);
	return v ? atoi(v) : fallback;
}

} // namespace

// vips_resize's downsizing chain then vips_sharpen on n 3-band uchar sRGB images of one geometry.
// `rv` was built for the image after shrinkv(vs) (height h1), `rh` for the one after shrinkh(hs)
// (width w3); coef / scale: the blur mask as convi's integers; lut: sharpen.c's 65536 ints (host).
// 1 = handled, 0 = not this kernel's case (nothing launched), -1 = error.
int resize_sharpen_stream_u8_try(_VipsHipReduce *rv, int vs, _VipsHipReduce *rh, int hs, int h1, int w3,
	const VipsHipRegion *const *in, const VipsHipRegion *const *out, int n, int tile, const int *coef, int ncoef,
	int scale, const int *lut)
{
	if (n < 1)
		return 0;
	const VipsHipRegion *i0 = in[0], *o0 = out[0];
	for (int i = 0; i < n; i++) {
		const VipsHipRegion *ri = in[i], *ro = out[i];
		if (ri->format != VIPS_HIP_FORMAT_UCHAR || ro->format != VIPS_HIP_FORMAT_UCHAR || ri->bands != 3 || ro->bands != 3)
			return 0;
		if (ri->left != 0 || ri->top != 0 || ri->width != ri->im_width || ri->height != ri->im_height ||
			ro->left != 0 || ro->top != 0 || ro->width != ro->im_width || ro->height != ro->im_height)
			return 0;
		if (ri->width != i0->width || ri->height != i0->height || ri->stride != i0->stride || ro->width != o0->width ||
			ro->height != o0->height || ro->stride != o0->stride)
			return 0;
		if (((uintptr_t) ri->data & 3) || (ri->stride & 3))
			return 0;
	}
	constexpr int B = 3, NP = RSH_NP;
	const long long row_bytes = (long long) i0->width * B;
	if ((row_bytes & 3) || row_bytes < RSH_SPAN || row_bytes > 0x3fffffffLL)
		return 0;
	if ((unsigned long long) i0->stride * (unsigned long long) i0->height > 0xffffffffULL)
		return 0;
	if ((unsigned long long) o0->stride * (unsigned long long) o0->height > 0xffffffffULL)
		return 0;
	if (vs != 4 && vs != 8)
		return 0;
	// the dword form of the horizontal pass: boxes of whole dwords, none clipped by the image's edge
	if (hs < 1 || (hs * B) % 4 != 0 || hs * B > 32 || i0->width % hs != 0)
		return 0;
	if (rv->in_size != h1 || rv->out_size != o0->height || rh->in_size != w3 || rh->out_size != o0->width)
		return 0;
	if (rv->n_point != 13 || rh->n_point > 13 || rh->n_point < 1)
		return 0;
	if (ncoef < 1 || ncoef > 5 || !(ncoef & 1) || scale <= 0)
		return 0;
	const int half = ncoef / 2;
	if (half < 1)
		return 0;
	long long abs_sum = 0;
	for (int k = 0; k < ncoef; k++)
		abs_sum += coef[k] < 0 ? -(long long) coef[k] : coef[k];
	if (abs_sum * 32768 + scale >= (1LL << 31)) // 32-bit sums
		return 0;
	std::vector<ReducePos> pv, ph;
	reduce_positions(rv, 0, o0->height, tile, pv);
	reduce_positions(rh, 0, o0->width, 0, ph);
	int fv, fh, phase_v, phase_h;
	if (!rsh_regular(pv, &fv, &phase_v) || !rsh_regular(ph, &fh, &phase_h))
		return 0;
	const RshLut *l = rsh_lut(lut);
	if (!l)
		return -1;
	if (!l->ok)
		return 0;
	const RshTables *tb = rsh_tables();
	if (!tb)
		return -1;
	const CbrtExact *cx = cbrt_exact_tables();
	if (!cx) {
		vips_hip_error_clear();
		return 0; // (this host's cbrtf does not fit cbrt_exact.h: the separate kernels read the table itself)
	}

	RshArgs a;
	memset(&a, 0, sizeof(a));
	a.in_stride = (long long) i0->stride;
	a.out_stride = (long long) o0->stride;
	a.width = i0->width;
	a.height = i0->height;
	a.h1 = h1;
	a.w3 = w3;
	a.out_width = o0->width;
	a.out_height = o0->height;
	a.hs = hs;
	a.mult_v = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) vs));
	a.mult_h = (unsigned int) ((1LL << 32) / ((1 << 8) * (long long) hs));
	a.fv = fv;
	a.fh = fh;
	a.n_h = rh->n_point;
	a.half = half;
	for (int k = 0; k < 5; k++)
		a.coef[k] = k < ncoef ? coef[k] : 0;
	a.scale = scale;
	a.rounding = scale / 2;
	if (scale > 1) {
		int lg = 0;
		while ((1LL << lg) < scale)
			lg++;
		a.magic = (unsigned int) ((((1ULL << 32) * ((1ULL << lg) - (unsigned long long) scale)) / (unsigned long long) scale) + 1);
		a.shift = lg - 1;
	}
	a.lut_lo = l->lo;
	a.lut_n = l->n;
	a.lut_below = l->below;
	a.lut_above = l->above;
	a.lut = l->d;
	a.v2Y = tb->v2Y;
	a.Y2v = tb->Y2v;
	a.cbrt = *cx;

	// the widest strip whose span fits: (2 (tw + 2 half) + n_h - 1) shrunk columns of hs pixels, + 3 bytes
	// of alignment; then strips of equal width
	const int nxr_max = (int) (((RSH_SPAN - 3) / ((long long) hs * B) - (a.n_h - 1)) / 2);
	int tw = nxr_max - 2 * half;
	if (tw > o0->width)
		tw = o0->width;
	if (tw < 8)
		return 0;
	int nstrips = (o0->width + tw - 1) / tw;
	tw = (o0->width + nstrips - 1) / nstrips;
	tw = rsh_env("VIPS_HIP_RSH_TW", tw);
	if (tw < 8 || tw > nxr_max - 2 * half)
		return 0;
	nstrips = (o0->width + tw - 1) / tw;
	a.tw = tw;
	a.nstrips = nstrips;
	const int nxr = tw + 2 * half;
	a.s_len = ((2 * nxr + a.n_h - 1 + 1 + 1) / 2) | 1;
	a.s_pitch = B * a.s_len * 4;
	a.o_pitch = (tw * B + 3 + 3) & ~3;
	a.r_pitch = (nxr * B + 3) & ~3;
	a.l_pitch = (nxr + 4 + 1) & ~1;
	a.ab_pitch = tw;
	a.h_pitch = (tw + 1) & ~1;
	a.window = rsh_env("VIPS_HIP_STREAM_WINDOW", 13);
	if (a.window < 4 || a.window > 40)
		a.window = 13;
	auto align16 = [](size_t v) { return (v + 15) & ~(size_t) 15; };
	size_t off = align16((size_t) NP * RSH_SPAN);
	a.off_S = (int) off;
	off = align16(off + (size_t) NP * a.s_pitch);
	a.off_HM = (int) off;
	off = align16(off + 36 * sizeof(unsigned int));
	a.off_R = (int) off;
	off = align16(off + (size_t) 2 * NP * a.r_pitch);
	a.off_L = (int) off;
	off = align16(off + (size_t) RSH_RING * a.l_pitch * sizeof(short));
	a.off_AB = (int) off;
	off = align16(off + (size_t) RSH_RING * a.ab_pitch * sizeof(unsigned int));
	a.off_H = (int) off;
	off = align16(off + (size_t) RSH_RING * a.h_pitch * sizeof(short));
	a.off_v2Y = (int) off;
	off = align16(off + 256 * sizeof(float));
	a.off_Y2v = (int) off;
	off = align16(off + 256 * sizeof(RshPair));
	a.off_lut = (int) off;
	off = align16(off + (size_t) (a.lut_n + 1) * sizeof(short));
	a.off_cres = (int) off;
	off = align16(off + (size_t) CBRT_RES_WORDS * sizeof(unsigned int));
	a.off_cbd = (int) off;
	off = align16(off + (size_t) (CBRT_BLOCKS + 1) * sizeof(CbrtBlockD));
	a.off_cbi = (int) off;
	off = align16(off + (size_t) (CBRT_BLOCKS + 1) * sizeof(CbrtBlockI));
	a.off_O = (int) off;
	// the stage takes what is left of half a CU's LDS (two blocks per CU), at most 14 slabs
	const size_t budget = (size_t) rsh_env("VIPS_HIP_RSH_LDS", 80 * 1024);
	int stage_rows = rsh_env("VIPS_HIP_STREAM_BURST", 14) * NP;
	if (stage_rows < 2 * NP)
		stage_rows = 2 * NP;
	while (stage_rows > 2 * NP && off + (size_t) stage_rows * a.o_pitch > budget)
		stage_rows -= NP;
	if (off + (size_t) stage_rows * a.o_pitch > budget)
		return 0;
	a.stage_rows = stage_rows;
	const size_t lds = off + (size_t) stage_rows * a.o_pitch;

	// segments: enough blocks to fill the chip several times, but tall (a segment re-reads 6 pairs
	// and makes 2 half rows more than it writes)
	long long want = getenv("VIPS_HIP_STREAM_BLOCKS") ? atoll(getenv("VIPS_HIP_STREAM_BLOCKS")) : 2048;
	int nsegs = (int) ((want + (long long) nstrips * n - 1) / ((long long) nstrips * n));
	if (nsegs < 1)
		nsegs = 1;
	int seg = (o0->height + nsegs - 1) / nsegs;
	const int seg_min = rsh_env("VIPS_HIP_STREAM_SEG", 21);
	if (seg < seg_min)
		seg = seg_min;
	seg = (seg + NP - 1) / NP * NP;
	nsegs = (o0->height + seg - 1) / seg;
	a.seg = seg;
	a.nsegs = nsegs;
	const short *cvs = &rv->matrixs[(size_t) phase_v * rv->n_point];
	for (int q = 0; q < NP; q++) {
		const unsigned int lo = (unsigned short) cvs[2 * q];
		const unsigned int hi = 2 * q + 1 < rv->n_point ? (unsigned short) cvs[2 * q + 1] : 0u;
		a.cv[q] = lo | (hi << 16);
	}
	const short *chs = &rh->matrixs[(size_t) phase_h * rh->n_point];
	for (int k = 0; k < rh->n_point; k++)
		a.ch[k] = chs[k];

	Gate gate("resize_sharpen_u8");
	for (int base = 0; base < n; base += RSH_MAXB) {
		const int count = n - base < RSH_MAXB ? n - base : RSH_MAXB;
		RshPtrs p;
		memset(&p, 0, sizeof(p));
		for (int i = 0; i < count; i++) {
			p.in[i] = (const unsigned char *) in[base + i]->data;
			p.out[i] = (unsigned char *) out[base + i]->data;
		}
		a.n_images = count;
		const long long units = (long long) nsegs * count;
		a.grouped = units >= 64;
		const long long blocks = (a.grouped ? (units + 7) / 8 * 8 : units) * nstrips;
		if (blocks > 0x7fffffffLL) {
			error("resize", "image too large");
			return -1;
		}
		if (rsh_launch(vs, a, p, (unsigned int) blocks, lds)) {
			error("resize", "kernel launch failed");
			return -1;
		}
	}
	return 1;
}

} // namespace vh
