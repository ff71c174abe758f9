// vips_resize(1 / (2 k)) -> vips_sharpen() of 3-band uchar sRGB images in ONE streaming kernel
// (BASELINE config 4, the batched thumbnail pipeline): the body of the kernel, written against
// gcn.h so that tests/emul can run it on the host.
//
//   resize  shrinkv(vs) -> reducev(2) -> shrinkh(hs) -> reduceh(2), resample/resize.c:207-228 --
//           the streaming block of resize_stream.hip: a 512-thread block owns a strip of output
//           columns and a segment of output rows of one image and walks down the input rows, one
//           dword of every row per lane, NP output rows ("a slab") retiring into LDS at a time
//   sharpen sRGB -> LabS, L blurred with the 3- or 5-tap integer gaussian (convsep, C path of
//           convi.c:698-716), the LUT on L - blur (sharpen.c:116-168), LabS -> sRGB
//           (sharpen.c:171-302; colour/sRGB2scRGB.c, scRGB2XYZ.c, XYZ2Lab.c, Lab2LabS.c and back
//           LabS2Lab.c, Lab2XYZ.c, LabQ2sRGB.c:263-360)
//
// The thumbnail never exists in memory: a strip carries `half` more resized columns on either
// side and a segment `half` more resized rows above and below (the blur's halo; at the image's
// edges columns and rows are clamped, the embed(COPY) of the convolution), and the sharpen's
// three stages trail the resize by a few slabs, IN THE SAME WAVES, between the row loads:
//
//   step s:  V(s)   the vertical pass of slab s: 7 pairs of shrunk rows, 8 row loads in flight
//            F(s-2) one resized pixel per thread -> LabS into 16-row rings (L; a, b)
//            B(s-3) one output pixel per thread: vertical blur of the horizontally blurred L,
//                   LUT, LabS -> sRGB, into the output stage
//            then, between two barriers: shrinkh(s); hblur(s-2) (L ring -> H ring); the output
//            stage's burst when the chip-wide write window says so; then reduceh(s) -> the
//            resized slab R[s & 1]
//
// F and B read NO table from global memory: a wave's gather from an L2-resident table costs ~146
// cycles of its CU's L1 fill path (tools/gather_probe.hip) -- the path the row loads need -- and
// the first build of this kernel, with XYZ2Lab's cube-root table and a per-L table gathered, ran
// 0.066 ms per image against 0.041 for the resize alone.  Everything the sharpen looks up is in
// LDS: the two 8-bit sRGB tables, the bending part of sharpen's LUT, and the cube-root table in
// the exact 30 KB form of cbrt_exact.h.  A block never stops loading while it sharpens; the ALU
// work of the sharpen runs in what was the block's wait for its loads.
//
// Arithmetic: every rounding is the separate operations' own (the result is theirs bit for bit),
// with the per-pixel work cut to what is not a function of one small integer:
//   * a / 500 with a = A / 256: A / 128000 (the same real number, correctly rounded once);
//   * X / 100.0 in double then float == the correctly rounded FLOAT quotient (no double rounding:
//     100 m for a float midpoint m needs 29 bits; tools/c4_identities.c checks every float);
//   * the 8-bit sRGB tables as {l0, l1 - l0} float pairs, the cube-root table as {t, dt} pairs.
#pragma once

#include "gcn.h"

#include "cbrt_exact.h"

namespace vh {

constexpr int RSH_SPAN = 2048; // bytes of a row a strip covers
constexpr int RSH_NT = 512;
constexpr int RSH_MAXB = 64;   // images per launch
constexpr int RSH_NP = 7;      // coefficient pairs of the vertical reduce = rows per slab
constexpr int RSH_RING = 16;   // rows of the LabS and blur rings

struct RshPair {
	float x, y;
};

struct RshArgs {
	long long in_stride, out_stride;
	int width, height; // input images (3 bands)
	int h1, w3;        // height after shrinkv, width after shrinkh
	int out_width, out_height;
	int hs;
	unsigned int mult_v, mult_h; // 2^32 / (256 * shrink), shrinkv.c:201 / shrinkh.c:141
	int fv, fh;                  // first tap of output row 0 / output column 0
	int n_h;                     // horizontal taps (<= 13)
	int tw, seg;                 // output columns per strip, output rows per segment
	int nstrips, nsegs, n_images;
	int grouped; // the strips of a (segment, image) on one XCD
	unsigned int cv[RSH_NP]; // vertical taps (2q, 2q + 1) as i16 pairs
	short ch[16];            // horizontal taps (those past n_h are 0)
	int s_len;               // dwords per band and row of the shrinkh slab
	int o_pitch, stage_rows; // bytes per staged output row, rows the stage holds
	int window;              // the staged rows leave when the 100 MHz clock crosses a multiple of 2^window ticks
	// ---- sharpen
	int half;          // blur taps / 2: 1 or 2
	int coef[5];       // the blur mask, zero beyond its taps
	int scale, rounding;
	unsigned int magic; // n / scale = (t + ((n - t) >> 1)) >> shift, t = mulhi(magic, n)
	int shift;
	int lut_lo, lut_n;        // sharpen.c's LUT is constant below index lut_lo and from lut_lo + lut_n on
	int lut_below, lut_above; // ... these constants
	const short *lut;         // lut_n entries between
	const float *v2Y;         // 256: sRGB2scRGB
	const RshPair *Y2v;       // 256: scRGB2sRGB as {l0, l1 - l0}
	CbrtExact cbrt;           // XYZ2Lab's table, cbrt_exact.h (copied to LDS)
	// ---- LDS layout, byte offsets (T at 0)
	int off_S, off_HM, off_O, off_R, off_L, off_AB, off_H, off_v2Y, off_Y2v, off_lut, off_cres, off_cbd, off_cbi;
	int s_pitch;  // bytes per row of S
	int r_pitch;  // bytes per row of R (and per buffer NP rows)
	int l_pitch;  // shorts per row of the L ring
	int ab_pitch; // dwords per row of the (a, b) ring
	int h_pitch;  // shorts per row of the H ring
};

struct RshPtrs {
	const unsigned char *in[RSH_MAXB];
	unsigned char *out[RSH_MAXB];
};

// (sum + 2048) >> 12, clip (templates.h:152-157)
VH_DEV unsigned int rsh_fin(int s)
{
	s = (s + 2048) >> 12;
	opaque(s);
	return (unsigned int) min(max(s, 0), 255);
}

// two i32 sums, already shifted, as one dword of two saturated bytes: {0, 0, sat_u8(hi), sat_u8(lo)}
VH_DEV unsigned int rsh_sat2(int lo, int hi)
{
	return sat_pk_u8_i16(perm((unsigned int) hi, (unsigned int) lo, 0x05040100u));
}

// shrinkv's rounding on two 16-bit sums in one dword: ((sum + vs/2) * (2^32 / (256 vs))) >> 24, shrinkv.c:158-165
template <int VS>
VH_DEV unsigned int rsh_box2(unsigned int sums, unsigned int mult)
{
	if constexpr ((VS & (VS - 1)) == 0) {
		constexpr int SH = VS == 1 ? 0 : VS == 2 ? 1 : VS == 4 ? 2 : VS == 8 ? 3 : 4;
		constexpr unsigned int RND = (unsigned int) (VS / 2) * 0x00010001u;
		return ((sums + RND) >> SH) & 0x00ff00ffu;
	}
	else {
		const unsigned int lo = (((sums & 0xffffu) + VS / 2) * mult) >> 24;
		const unsigned int hi = (((sums >> 16) + VS / 2) * mult) >> 24;
		return lo | (hi << 16);
	}
}

// a / y for a constant y, correctly rounded, finite a far from overflow: q0 = a * RN(1 / y), one
// FMA residual, one FMA correction (Markstein; see colour_device.h div_const)
VH_DEV double rsh_div(double a, double y, double r)
{
	const double q0 = a * r;
	const double e = __builtin_fma(-y, q0, a);
	return __builtin_fma(e, r, q0);
}
#define RSH_DIV(A, Y) rsh_div((A), (Y), 1.0 / (Y))
// (float) ((double) a / c) for the XYZ of a uchar pixel: see colour_device.h quant_div_finite
#define rsh_quant(A, C) rsh_quant_((A), (float) (1.0 / (C)), (float) (1.0 / (C) - (double) (float) (1.0 / (C))))
VH_DEV float rsh_quant_(float a, float rhi, float rlo)
{
	return __builtin_fmaf(a, rhi, a * rlo);
}
VH_DEV float rsh_divf100(float a)
{
	const float r = 1.0f / 100.0f;
	const float q0 = a * r;
	const float e = __builtin_fmaf(-100.0f, q0, a);
	return __builtin_fmaf(e, r, q0);
}

// ((sum + rounding) / scale), C division, clip to short (convi.c:698-716)
VH_DEV int rsh_convi_fin(int sum, const RshArgs &a)
{
	const int x = sum + a.rounding;
	const unsigned int n = (unsigned int) (x < 0 ? -x : x);
	unsigned int m = n;
	if (a.scale != 1) {
		const unsigned int t = umulhi(a.magic, n);
		m = (t + ((n - t) >> 1)) >> a.shift;
	}
	const int q = x < 0 ? -(int) m : (int) m;
	return min(max(q, -32768), 32767);
}

// Lab2LabS.c:59-73 for a finite value: double multiply, clip, truncate
VH_DEV int rsh_labs(float v, double scale, double lo)
{
	const double d = (double) v * scale;
	return vh::cvt_i32(__builtin_fmax(lo, __builtin_fmin(d, 32767.0)));
}

// one channel of vips_col_scRGB2sRGB (LabQ2sRGB.c:290-360) for a finite value, into byte k of `old`
VH_DEV unsigned int rsh_channel(const RshPair *Y2v, float v, unsigned int k, unsigned int old)
{
	float Yf = v * 255.0f;
	Yf = Yf < 0.0f ? 0.0f : Yf; // (NaN cannot happen: finite input)
	Yf = Yf > 255.0f ? 255.0f : Yf;
	const int Yi = vh::cvt_i32(Yf);
	const RshPair e = Y2v[Yi];
	const float r = e.x + e.y * fract(Yf);
	return cvt_pk_u8(rne(r), k, old);
}

template <int VS, int NP, class Words>
static __device__ __forceinline__ void resize_sharpen_body(const RshArgs &a, const Words &kp, int wg, unsigned int *lds)
{
	constexpr int NT = RSH_NT;
	unsigned char *const lds8 = reinterpret_cast<unsigned char *>(lds);
	unsigned char *const T = lds8;                                                  // NP rows of RSH_SPAN bytes
	unsigned char *const S = lds8 + a.off_S;                                        // NP rows of s_pitch bytes
	unsigned int *const HM = reinterpret_cast<unsigned int *>(lds8 + a.off_HM);     // byte masks [band][dword of a box]
	unsigned int *const CLK = HM + 32;                                              // the clock slot the block acts on
	unsigned char *const O = lds8 + a.off_O;                                        // the output stage
	unsigned char *const R = lds8 + a.off_R;                                        // 2 x NP resized rows (sRGB bytes)
	short *const LR = reinterpret_cast<short *>(lds8 + a.off_L);                    // L ring
	unsigned int *const ABR = reinterpret_cast<unsigned int *>(lds8 + a.off_AB);    // (a, b) ring
	short *const HR = reinterpret_cast<short *>(lds8 + a.off_H);                    // horizontally blurred L ring
	float *const v2Y = reinterpret_cast<float *>(lds8 + a.off_v2Y);
	RshPair *const Y2v = reinterpret_cast<RshPair *>(lds8 + a.off_Y2v);
	short *const LUT = reinterpret_cast<short *>(lds8 + a.off_lut);
	unsigned int *const CRES = reinterpret_cast<unsigned int *>(lds8 + a.off_cres);
	CbrtBlockD *const CBD = reinterpret_cast<CbrtBlockD *>(lds8 + a.off_cbd);
	CbrtBlockI *const CBI = reinterpret_cast<CbrtBlockI *>(lds8 + a.off_cbi);
	// (the table's single-precision form when the host made one -- cbrt_exact.h: the blocks' floats
	// take the place of the doubles, the residuals are that form's)
	const bool c32 = a.cbrt.bf != nullptr;
	CbrtBlockF *const CBF = reinterpret_cast<CbrtBlockF *>(lds8 + a.off_cbd);
	const CbrtExact cbrt = { CBD, CBI, CRES, nullptr, CBF, CRES };

	// block -> (strip, segment, image), the strips of one (segment, image) on one XCD (see resize_stream.hip)
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
	const gptr_in in = gptr_in_of(kp[img]);
	const gptr_out out = gptr_out_of(kp[RSH_MAXB + img]);

	const int t = tid();
	constexpr int B = 3;
	// tables into LDS (ordered before their readers by the first slab's barriers)
	if (t < 32) {
		const int b = t >> 3, j = t & 7;
		unsigned int m = 0;
#pragma unroll
		for (int i = 0; i < 4; i++)
			if ((4 * j + i) % B == b)
				m |= 1u << (8 * i);
		HM[t] = m;
	}
	if (t < 256) {
		v2Y[t] = a.v2Y[t];
		Y2v[t] = a.Y2v[t];
	}
	for (int i = t; i < a.lut_n; i += NT)
		LUT[i] = a.lut[i];
	for (int i = t; i < CBRT_RES_WORDS; i += NT)
		CRES[i] = c32 ? a.cbrt.res32[i] : a.cbrt.res[i];
	if (t <= CBRT_BLOCKS) {
		if (c32)
			CBF[t] = a.cbrt.bf[t];
		else
			CBD[t] = a.cbrt.bd[t];
		CBI[t] = a.cbrt.bi[t];
	}

	const int h = a.half;
	const int x0 = strip * a.tw, nx = min(a.tw, a.out_width - x0);
	const int xr0 = x0 - h, nxr = nx + 2 * h; // resized columns the strip makes: xr0 + u, clamped into the image
	const int y0 = seg_i * a.seg, ny = min(a.seg, a.out_height - y0);
	const int ylo = max(y0 - h, 0), yhi = min(y0 + ny + h, a.out_height);
	const int nyr = yhi - ylo;                // resized rows the segment makes: ylo ...
	const int ns = (nyr + NP - 1) / NP;       // slabs
	const int rho0 = y0 - ylo;                // the first output row, relative to ylo
	const int nb = (rho0 + ny - 1 + h) / NP + 1; // output batches: rows rho in [NP b - h, NP b + NP - h)
	const int last_sg = max(ns - 1, nb + 2);

	// columns of the shrinkh image the strip's taps touch, and the input bytes under them
	const int cb = 2 * xr0 + a.fh; // shrunk column of S index 0 (unclamped)
	const int c_lo = min(max(cb, 0), a.w3 - 1);
	const int row_bytes = a.width * B;
	const int start_al = max(min((c_lo * a.hs * B) & ~3, row_bytes - RSH_SPAN), 0);
	const gptr_in span = in + start_al;
	const unsigned int lane_off = (unsigned int) min(4 * t, row_bytes - 4 - start_al);

	auto load = [&](int r, int k) -> unsigned int {
		const int rc = min(max(r, 0), a.h1 - 1);
		const int row = min(rc * VS + k, a.height - 1);
		const unsigned int row_off = (unsigned int) row * (unsigned int) a.in_stride;
		return gload32(span, row_off + lane_off);
	};

	int acc[NP][4];
	int half_rnd = 2048;
	opaque(half_rnd);
#pragma unroll
	for (int s = 0; s < NP; s++)
#pragma unroll
		for (int b = 0; b < 4; b++)
			acc[s][b] = 0;

	const int r0 = 2 * ylo + a.fv;
	unsigned int ring[2][VS];
#pragma unroll
	for (int hh = 0; hh < 2; hh++)
#pragma unroll
		for (int k = 0; k < VS; k++)
			ring[hh][k] = load(r0 - 2 + hh, k);

	// idx / d for idx < 2048, d <= 128
	const unsigned int magic_nxr = ((1u << 20) + nxr - 1) / nxr;
	const unsigned int magic_nx = ((1u << 20) + nx - 1) / nx;
	const unsigned int out_lo = gptr_low(out + (long long) x0 * B);

	// ---- F: resized pixel idx of slab sf -> LabS
	auto f_stage = [&](int sf, int idx) {
		const int rows = min(NP, nyr - NP * sf);
		const int r = (int) (((unsigned int) idx * magic_nxr) >> 20);
		const int u = idx - r * nxr;
		if (r >= rows)
			return;
		const unsigned char *px = R + ((sf & 1) * NP + r) * a.r_pitch + 3 * u;
		const float Rl = v2Y[px[0]] * 100.0f, Gl = v2Y[px[1]] * 100.0f, Bl = v2Y[px[2]] * 100.0f;
		// scRGB2XYZ.c:58-82
		const float X = (0.4124F * Rl + 0.3576F * Gl) + 0.1805F * Bl;
		const float Y = (0.2126F * Rl + 0.7152F * Gl) + 0.0722F * Bl;
		const float Z = (0.0193F * Rl + 0.1192F * Gl) + 0.9505F * Bl;
		// XYZ2Lab.c:109-138: nX = QUANT_ELEMENTS * X / X0 in double, to float; index, fraction, lerp
		// (the double quotient in two float operations: colour_device.h quant_div_finite)
		const float n0 = rsh_quant(100000.0f * X, 95.0470);
		const float n1 = rsh_quant(100000.0f * Y, 100.0);
		const float n2 = rsh_quant(100000.0f * Z, 108.8827);
		const int i0 = min(max(vh::cvt_i32(n0), 0), CBRT_N - 2);
		const int i1 = min(max(vh::cvt_i32(n1), 0), CBRT_N - 2);
		const int i2 = min(max(vh::cvt_i32(n2), 0), CBRT_N - 2);
		const int slot = (NP * sf + r) & (RSH_RING - 1);
		float t0, dt;
		if (c32)
			cbrt_pair32(cbrt, i1, &t0, &dt);
		else
			cbrt_pair(cbrt, i1, &t0, &dt);
		const float cby = t0 + (n1 - (float) i1) * dt;
		const int L = rsh_labs(116.0F * cby - 16.0F, 32767.0 / 100.0, 0.0);
		LR[slot * a.l_pitch + u] = (short) L;
		const int x = u - h;
		if (x >= 0 && x < nx) {
			if (c32)
				cbrt_pair32(cbrt, i0, &t0, &dt);
			else
				cbrt_pair(cbrt, i0, &t0, &dt);
			const float cbx = t0 + (n0 - (float) i0) * dt;
			if (c32)
				cbrt_pair32(cbrt, i2, &t0, &dt);
			else
				cbrt_pair(cbrt, i2, &t0, &dt);
			const float cbz = t0 + (n2 - (float) i2) * dt;
			const int A = rsh_labs(500.0F * (cbx - cby), 32768.0 / 128.0, -32768.0);
			const int Bv = rsh_labs(200.0F * (cby - cbz), 32768.0 / 128.0, -32768.0);
			ABR[slot * a.ab_pitch + x] = ((unsigned int) A & 0xffffu) | ((unsigned int) Bv << 16);
		}
	};

	// ---- B: output pixel idx of batch bb -> the stage
	int staged = 0, flushed = 0; // rows in the stage; output rows already written
	auto b_stage = [&](int bb, int idx) {
		const int r = (int) (((unsigned int) idx * magic_nx) >> 20);
		const int x = idx - r * nx;
		const int first = max(NP * bb - h, rho0);
		const int rho = NP * bb - h + r;
		if (r >= NP || rho < first || rho >= rho0 + ny)
			return;
		const int y = ylo + rho;
		int sum = 0;
#pragma unroll
		for (int k = 0; k < 5; k++) {
			int rr = min(max(y - h + k, 0), a.out_height - 1) - ylo;
			rr = min(rr, nyr - 1);
			sum += a.coef[k] * (int) HR[(rr & (RSH_RING - 1)) * a.h_pitch + x];
		}
		const int blur = rsh_convi_fin(sum, a);
		const int slot = rho & (RSH_RING - 1);
		const int v1 = LR[slot * a.l_pitch + x + h];
		// sharpen.c:116-168
		const int d = (v1 & 0x7fff) - (blur & 0x7fff) + 32768 - a.lut_lo;
		int lv = d < 0 ? a.lut_below : a.lut_above;
		if ((unsigned int) d < (unsigned int) a.lut_n)
			lv = LUT[d];
		const int sharp = min(max(v1 + lv, 0), 32767);
		const unsigned int ab = ABR[slot * a.ab_pitch + x];
		const int A = (int) (short) (ab & 0xffffu), Bv = (int) ab >> 16;
		// LabS2Lab.c:55-69 and Lab2XYZ.c:84-109 (a = A / 256, b = B / 256: exact)
		const float Lf = (float) RSH_DIV((double) sharp, 32767.0 / 100.0);
		double fy = RSH_DIV((double) Lf + 16.0, 116.0);
		float Y = (float) (((100.0 * fy) * fy) * fy);
		if (Lf < 8.0f) {
			Y = (float) RSH_DIV((double) Lf * 100.0, 903.3);
			fy = 7.787 * RSH_DIV((double) Y, 100.0) + 16.0 / 116.0;
		}
		const double fx = RSH_DIV((double) A, 128000.0) + fy;
		const double fz = fy - RSH_DIV((double) Bv, 51200.0);
		float X = (float) (((95.0470 * fx) * fx) * fx);
		float Z = (float) (((108.8827 * fz) * fz) * fz);
		if (fx < 0.2069)
			X = (float) RSH_DIV(95.0470 * (fx - 0.13793), 7.787);
		if (fz < 0.2069)
			Z = (float) RSH_DIV(108.8827 * (fz - 0.13793), 7.787);
		// LabQ2sRGB.c:263-283
		const float Xn = rsh_divf100(X), Yn = rsh_divf100(Y), Zn = rsh_divf100(Z);
		const float rl = (3.240625F * Xn + -1.537208F * Yn) + -0.498629F * Zn;
		const float gl = (-0.968931F * Xn + 1.875756F * Yn) + 0.041518F * Zn;
		const float bl = (0.055710F * Xn + -0.204021F * Yn) + 1.056996F * Zn;
		unsigned int px = rsh_channel(Y2v, rl, 0, 0u);
		px = rsh_channel(Y2v, gl, 1, px);
		px = rsh_channel(Y2v, bl, 2, px);
		const unsigned int mis = (out_lo + (unsigned int) y * (unsigned int) a.out_stride) & 3u;
		unsigned char *o = O + (staged + rho - first) * a.o_pitch + (int) mis + 3 * x;
		o[0] = (unsigned char) px;
		o[1] = (unsigned char) (px >> 8);
		o[2] = (unsigned char) (px >> 16);
	};

	unsigned int slot_clk = 0;
	for (int n = 0; n <= last_sg + 1; n++) {
		const int sg = n - 1;       // the slab this step's vertical pass makes
		const int sf = sg - 2;      // F's slab
		const int bb = sg - 3;      // B's batch
		const bool do_f = sf >= 0 && sf < ns, do_b = bb >= 0 && bb < nb;
		const int nfpix = do_f ? nxr * min(NP, nyr - NP * sf) : 0;
		const int nbpix = do_b ? nx * NP : 0;
		if (n <= ns) {
#pragma unroll
			for (int p = 0; p < NP; p++) {
				const int j = NP * n + p - 1;
				// the sharpen stages of this step, each behind one pair's row work
				const bool f_now = (p == 1 || p == NP - 2), b_now = (p == 2 || p == NP - 1);
				const int rd = p >= NP - 2 ? 1 : 0; // second round: pixels 512 ...
				unsigned int sb[2][2];
#pragma unroll
				for (int hh = 0; hh < 2; hh++) {
					unsigned int e = 0, o = 0;
#pragma unroll
					for (int k = 0; k < VS; k++) {
						const unsigned int w = ring[hh][k];
						e += w & 0x00ff00ffu;
						o += perm(0u, w, 0x0c030c01u);
						ring[hh][k] = load(r0 + 2 * (j + 1) + hh, k);
					}
					sb[hh][0] = rsh_box2<VS>(e, a.mult_v);
					sb[hh][1] = rsh_box2<VS>(o, a.mult_v);
				}
#pragma unroll
				for (int b = 0; b < 4; b++) {
					const int w = b & 1;
					const unsigned int pk = perm(sb[1][w], sb[0][w], (b & 2) ? 0x07060302u : 0x05040100u);
#pragma unroll
					for (int q = 0; q < NP; q++) {
						const int slot = (p - 1 - q + 2 * NP) % NP;
						if (q == 0)
							acc[slot][b] = dot2_s(pk, a.cv[0], half_rnd);
						else
							acc[slot][b] = dot2(pk, a.cv[q], acc[slot][b]);
					}
				}
				const unsigned int packed = rsh_sat2(acc[p][0] >> 12, acc[p][1] >> 12) |
					(rsh_sat2(acc[p][2] >> 12, acc[p][3] >> 12) << 16);
				*reinterpret_cast<unsigned int *>(T + p * RSH_SPAN + 4 * t) = packed;
				sched_fence();
				if (f_now && rd * NT < nfpix)
					f_stage(sf, rd * NT + t);
				if (b_now && rd * NT < nbpix)
					b_stage(bb, rd * NT + t);
				sched_fence();
			}
		}
		else {
			// the segment's rows are all made: the stages still in flight, without row work
			for (int rd = 0; rd * NT < nfpix; rd++)
				f_stage(sf, rd * NT + t);
			for (int rd = 0; rd * NT < nbpix; rd++)
				b_stage(bb, rd * NT + t);
		}
		if (n == 0)
			continue;

		// ---- between the steps
		int th = t;
		opaque(th);
		if (th == 0)
			*CLK = (unsigned int) (realtime() >> a.window);
		barrier();
		if (do_b)
			staged += min(NP * bb + NP - h, rho0 + ny) - max(NP * bb - h, rho0);
		{
			// the stage's burst: when the chip-wide clock has crossed a window boundary, when the
			// stage could not take another batch, and at the end (see resize_stream.hip)
			const unsigned int now = *CLK;
			if (staged > 0 && (now != slot_clk || staged + NP > a.stage_rows || sg == last_sg)) {
				slot_clk = now;
				const int nbytes = nx * B;
				for (int row = th >> 6; row < staged; row += NT / 64) {
					const long long first = (long long) (y0 + flushed + row) * a.out_stride + (long long) x0 * B;
					const int mis = (int) (gptr_low(out + first) & 3u);
					const gptr_out g = out + first - mis;
					const unsigned char *src = O + row * a.o_pitch;
					for (int d = th & 63; 4 * d < mis + nbytes; d += 64) {
						const int b0 = 4 * d;
						if (b0 >= mis && b0 + 4 <= mis + nbytes)
							gstore32(g + b0, *reinterpret_cast<const unsigned int *>(src + b0));
						else
							for (int k = 0; k < 4; k++)
								if (b0 + k >= mis && b0 + k < mis + nbytes)
									gstore8(g + b0 + k, src[b0 + k]);
					}
				}
				flushed += staged;
				staged = 0;
			}
		}
		const bool do_h = sg >= 0 && sg < ns;
		const int len = 2 * nxr + a.n_h - 1;
		if (do_h) {
			// shrinkh: thread = one band of one shrunk column u (column cb + u of the shrunk image,
			// clamped into it), band-major; all slab rows
			const int total = len * B;
			const int ndw = (a.hs * B) >> 2;
			unsigned short *S16 = reinterpret_cast<unsigned short *>(S);
			const int band_pitch = 2 * a.s_len;
			for (int e = th; e < total; e += NT) {
				const int b = (e >= len) + (e >= 2 * len);
				const int u = e - b * len;
				const int col = min(max(cb + u, 0), a.w3 - 1);
				const unsigned int *src = reinterpret_cast<const unsigned int *>(T + col * a.hs * B - start_al);
				unsigned int sum[NP];
#pragma unroll
				for (int r = 0; r < NP; r++)
					sum[r] = (unsigned int) (a.hs / 2);
#pragma unroll
				for (int j = 0; j < 8; j++)
					if (j < ndw) {
						const unsigned int m = HM[b * 8 + j];
#pragma unroll
						for (int r = 0; r < NP; r++)
							sum[r] = udot4(src[r * (RSH_SPAN / 4) + j], m, sum[r]);
					}
#pragma unroll
				for (int r = 0; r < NP; r++)
					S16[(r * B + b) * band_pitch + u] = (unsigned short) ((sum[r] * a.mult_h) >> 24);
			}
		}
		if (sf >= 0 && sf < ns) {
			// hblur: the rows F made in this step, L ring -> H ring (convi.c:698-716 on a short image)
			const int rows = min(NP, nyr - NP * sf);
			for (int e = th; e < rows * nx; e += NT) {
				const int r = (int) (((unsigned int) e * magic_nx) >> 20);
				const int x = e - r * nx;
				const int slot = (NP * sf + r) & (RSH_RING - 1);
				const short *row = LR + slot * a.l_pitch + x;
				int sum = 0;
#pragma unroll
				for (int k = 0; k < 5; k++)
					sum += a.coef[k] * (int) row[k];
				HR[slot * a.h_pitch + x] = (short) rsh_convi_fin(sum, a);
			}
		}
		barrier();
		if (do_h) {
			// reduceh: thread = one band element of the resized rows; taps (2 q, 2 q + 1) of column
			// xr0 + u are the two lanes of dword (clamped column - xr0) + q of the band's row
			if (th < nxr * B) {
				const int u = (int) (((unsigned int) th * 21846u) >> 16);
				const int b = th - u * B;
				const int xc = min(max(xr0 + u, 0), a.out_width - 1) - xr0;
				const unsigned int *row = reinterpret_cast<const unsigned int *>(S) + b * a.s_len + xc;
				const int nq = (a.n_h + 1) >> 1;
				int sum[NP];
#pragma unroll
				for (int r = 0; r < NP; r++)
					sum[r] = 0;
#pragma unroll
				for (int q = 0; q < 7; q++)
					if (q < nq) {
						const unsigned int ck = (unsigned int) (unsigned short) a.ch[2 * q] |
							((unsigned int) (unsigned short) a.ch[2 * q + 1] << 16);
#pragma unroll
						for (int r = 0; r < NP; r++)
							sum[r] = dot2(row[r * B * a.s_len + q], ck, sum[r]);
					}
				unsigned char *dst = R + (sg & 1) * NP * a.r_pitch + th;
#pragma unroll
				for (int r = 0; r < NP; r++)
					dst[r * a.r_pitch] = (unsigned char) rsh_fin(sum[r]);
			}
		}
	}
}

} // namespace vh
