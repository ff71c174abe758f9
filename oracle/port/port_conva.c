/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's approximate convolutions
 * (precision=approximate: the box-decomposition blur).
 *
 * Follows libvips 8.19.0 (/root/reference/libvips):
 *   mask -> int mask            convolution/convi.c:860-923 (vips__image_intize)
 *   conva, 2-D                  convolution/conva.c:294-395 (layers -> hlines), :397-549
 *                               (clustering), :551-581 (renumber), :583-674 (vlines),
 *                               :676-767 (area / divisor), :840-873 (HCONV), :1056-1097 (VCONV),
 *                               :1099-1198 (type dispatch), :1231-1280 (build)
 *   convasep, 1-D               convolution/convasep.c:152-330 (decompose), :428-474
 *                               (HCONV_INT), :475-514 (HCONV_FLOAT), :592-675 (VCONV_*),
 *                               :775-828 (build)
 *   edge handling               vips_embed(VIPS_EXTEND_COPY): conversion/embed.c:226-341
 *
 * The reference keeps rolling sums along each tile row / column.  For the integer formats all
 * of its arithmetic is modular, so the sums are evaluated directly here; for float / double the
 * rolling sums are only order-independent while every partial sum is exact (e.g. integer-valued
 * pixels), and the parity tests use such data.
 *
 * Several quirks of the reference are kept on purpose, because "bit-identical" is the bar:
 *   - the divisor is computed from an area that is multiplied by the common factor twice
 *     (conva.c:728-744, convasep.c:272-290), so box-like masks come out darker;
 *   - conva's vertical pass for unsigned formats sums in `unsigned int`, so negative totals wrap
 *     and clip to the maximum (conva.c:1099-1140);
 *   - conva on uint images with short lines clips with the `short` limits (conva.c:1130-1134).
 *
 * Parity status: PINNED by tests/test_oracle_conva.py against oracle/_ref directly and against
 * golden vectors made by it (tests/golden/conva.npz).
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "port.h"

#define CA_MAX_LINES 1000
#define CA_MAX_EDGES 1000

static int
ca_clampi(int v, int lo, int hi)
{
	return v < lo ? lo : (v > hi ? hi : v);
}

/* convi.c:860-923.  mask_out holds rint()ed elements; *scale_out / *offset_out the adjusted
 * scale and rounded offset. */
static void
ca_intize(const double *mask, int n, double scale, double offset, double *mask_out,
	double *scale_out, double *offset_out)
{
	double double_result = 0;
	for (int i = 0; i < n; i++)
		double_result += mask[i];
	double_result /= scale;

	for (int i = 0; i < n; i++)
		mask_out[i] = rint(mask[i]);

	double out_scale = rint(scale);
	if (out_scale == 0)
		out_scale = 1;

	int int_result = 0;
	for (int i = 0; i < n; i++)
		int_result += mask_out[i]; /* int += double, as the reference writes it */
	int_result /= out_scale;

	out_scale = rint(out_scale + (int_result - double_result));
	if (out_scale == 0)
		out_scale = 1;

	*scale_out = out_scale;
	*offset_out = rint(offset);
}

static int
ca_gcd(int a, int b)
{
	return b == 0 ? abs(a) : ca_gcd(b, a % b);
}

/* ------------------------------------------------------------------ conva: decomposition */

typedef struct {
	int a, b, d;
} CaEdge;

typedef struct {
	int band, row, factor;
} CaElement;

typedef struct {
	int n_hline;
	int hstart[CA_MAX_LINES], hend[CA_MAX_LINES], hweight[CA_MAX_LINES];
	int n_el;
	CaElement el[CA_MAX_LINES];
	int n_vline;
	int vband[CA_MAX_LINES], vfactor[CA_MAX_LINES], vstart[CA_MAX_LINES], vend[CA_MAX_LINES];
	CaEdge edge[CA_MAX_EDGES];
	int divisor, rounding, offset, max_line;
} CaBoxes;

static int
ca_close_hline(CaBoxes *bx, int x, int y, int factor)
{
	bx->hend[bx->n_hline] = x;
	bx->el[bx->n_el].row = y;
	bx->el[bx->n_el].band = bx->n_hline;
	bx->el[bx->n_el].factor = factor;
	if (bx->n_hline >= CA_MAX_LINES - 1)
		return -1;
	bx->n_hline += 1;
	if (bx->n_el >= CA_MAX_LINES - 1)
		return -1;
	bx->n_el += 1;
	return 0;
}

/* conva.c:294-395 */
static int
ca_slice_hlines(CaBoxes *bx, const double *coeff, int mw, int mh, int n_layers)
{
	double max = 0, min = 0;
	for (int i = 0; i < mw * mh; i++) {
		if (coeff[i] > max)
			max = coeff[i];
		if (coeff[i] < min)
			min = coeff[i];
	}
	/* the reference's layer arithmetic is undefined without a positive element */
	if (!(max > 0))
		return -1;

	double depth = (max - min) / n_layers;
	const int layers_above = ceil(max / depth);
	depth = max / layers_above;
	const int layers_below = floor(min / depth);
	int64_t span = (int64_t) layers_above - layers_below;
	const int layers = span < 1 ? 1 : (span > 1000 ? 1000 : (int) span);

	for (int z = 0; z < layers; z++) {
		const double z_ph = max - (1 + z) * depth + depth / 2;
		const int positive = z < layers_above;

		for (int y = 0; y < mh; y++) {
			int inside = 0;
			for (int x = 0; x < mw; x++) {
				const double c = coeff[x + y * mw];
				if ((positive && c >= z_ph) || (!positive && c <= z_ph)) {
					if (!inside) {
						bx->hstart[bx->n_hline] = x;
						bx->hweight[bx->n_hline] = 1;
						inside = 1;
					}
				}
				else if (inside) {
					if (ca_close_hline(bx, x, y, positive ? 1 : -1))
						return -1;
					inside = 0;
				}
			}
			if (inside && ca_close_hline(bx, mw, y, positive ? 1 : -1))
				return -1;
		}
	}
	return 0;
}

static int
ca_edge_cmp(const void *p1, const void *p2)
{
	return ((const CaEdge *) p1)->d - ((const CaEdge *) p2)->d;
}

/* conva.c:397-443 */
static void
ca_merge(CaBoxes *bx, int a, int b)
{
	const int fa = bx->hweight[a], fb = bx->hweight[b];
	const double w = (double) fb / (fa + fb);

	bx->hstart[a] += w * (bx->hstart[b] - bx->hstart[a]);
	bx->hend[a] += w * (bx->hend[b] - bx->hend[a]);
	bx->hweight[a] += bx->hweight[b];
	for (int i = 0; i < bx->n_el; i++)
		if (bx->el[i].band == b)
			bx->el[i].band = a;
	bx->hweight[b] = 0;
}

/* conva.c:445-549: one clustering sweep; non-zero when something merged */
static int
ca_cluster_sweep(CaBoxes *bx, int cluster)
{
	CaEdge *edge = bx->edge;
	for (int i = 0; i < CA_MAX_EDGES; i++) {
		edge[i].a = -1;
		edge[i].b = -1;
		edge[i].d = 99999;
	}
	int worst_i = 0;
	int worst = edge[0].d;

	for (int i = 0; i < bx->n_hline; i++) {
		if (bx->hweight[i] == 0)
			continue;
		for (int j = i + 1; j < bx->n_hline; j++) {
			if (bx->hweight[j] == 0)
				continue;
			const int distance =
				abs(bx->hstart[i] - bx->hstart[j]) + abs(bx->hend[i] - bx->hend[j]);
			if (distance < worst) {
				edge[worst_i].a = i;
				edge[worst_i].b = j;
				edge[worst_i].d = distance;

				worst_i = 0;
				worst = edge[0].d;
				for (int k = 0; k < CA_MAX_EDGES; k++)
					if (edge[k].d > worst) {
						worst = edge[k].d;
						worst_i = k;
					}
			}
		}
	}

	/* the same libc qsort as the reference: ties between equal distances resolve alike */
	qsort(edge, CA_MAX_EDGES, sizeof(CaEdge), ca_edge_cmp);

	int merged = 0;
	for (int k = 0; k < CA_MAX_EDGES; k++) {
		CaEdge *e = &edge[k];
		if (e->d > cluster)
			break;
		if (e->a == -1)
			continue;
		ca_merge(bx, e->a, e->b);
		merged = 1;
		/* e itself is the first entry visited and loses its `a` there, so the later
		 * comparisons against e->a see -1: written as the reference reads it */
		for (int i = k; i < CA_MAX_EDGES; i++) {
			CaEdge *ei = &edge[i];
			if (ei->a == e->a || ei->b == e->a || ei->a == e->b || ei->b == e->b)
				ei->a = -1;
		}
	}
	return merged;
}

static int
ca_element_cmp(const void *p1, const void *p2)
{
	const CaElement *a = p1, *b = p2;
	if (a->band != b->band)
		return a->band - b->band;
	if (a->factor != b->factor)
		return a->factor - b->factor;
	return a->row - b->row;
}

/* conva.c:676-767 */
static int
ca_decompose_boxes(CaBoxes *bx, const double *coeff, int mw, int mh, double scale, double offset,
	int n_layers, int cluster)
{
	memset(bx, 0, sizeof(*bx));
	if (ca_slice_hlines(bx, coeff, mw, mh, n_layers))
		return -1;
	if (bx->n_el == 0)
		return -1;

	while (ca_cluster_sweep(bx, cluster))
		;

	/* :551-581 drop the merged-away hlines */
	for (int i = 0; i < bx->n_hline;) {
		if (bx->hweight[i] > 0) {
			i++;
			continue;
		}
		for (int j = 0; j < bx->n_el; j++)
			if (bx->el[j].band > i)
				bx->el[j].band -= 1;
		for (int j = i; j + 1 < bx->n_hline; j++) {
			bx->hstart[j] = bx->hstart[j + 1];
			bx->hend[j] = bx->hend[j + 1];
			bx->hweight[j] = bx->hweight[j + 1];
		}
		bx->n_hline -= 1;
	}

	/* :583-674 */
	qsort(bx->el, bx->n_el, sizeof(CaElement), ca_element_cmp);
	for (int y = 0; y < bx->n_el; y++) {
		int z;
		for (z = y + 1; z < bx->n_el; z++)
			if (bx->el[z].band != bx->el[y].band || bx->el[z].row != bx->el[y].row)
				break;
		bx->el[y].factor = bx->el[y].factor > 0 ? z - y : y - z;
		memmove(bx->el + y + 1, bx->el + z, sizeof(CaElement) * (bx->n_el - z));
		bx->n_el -= z - y - 1;
	}
	bx->n_vline = 0;
	for (int y = 0; y < bx->n_el;) {
		const int n = bx->n_vline;
		int z;
		bx->vband[n] = bx->el[y].band;
		bx->vfactor[n] = bx->el[y].factor;
		bx->vstart[n] = bx->el[y].row;
		for (z = y + 1; z < bx->n_el; z++)
			if (bx->el[z].band != bx->vband[n] || bx->el[z].factor != bx->vfactor[n] ||
				bx->el[z].row != bx->vstart[n] + z - y)
				break;
		bx->vend[n] = bx->el[z - 1].row + 1;
		bx->n_vline += 1;
		y = z;
	}

	double area = 0;
	bx->max_line = 0;
	for (int y = 0; y < bx->n_el; y++) {
		const int b = bx->el[y].band;
		const int len = bx->hend[b] - bx->hstart[b];
		area += abs(bx->el[y].factor * len);
		if (len > bx->max_line)
			bx->max_line = len;
	}
	/* the vlines above keep the un-reduced factors; only the area sees the common factor */
	int x = bx->el[0].factor;
	for (int y = 1; y < bx->n_el; y++)
		x = ca_gcd(x, bx->el[y].factor);
	for (int y = 0; y < bx->n_el; y++)
		bx->el[y].factor /= x;
	area *= x;

	double sum = 0;
	for (int i = 0; i < mw * mh; i++)
		sum += fabs(coeff[i]);

	const double d = rint(area * scale / sum);
	bx->divisor = d > 1 ? d : 1;
	bx->rounding = (bx->divisor + 1) / 2;
	bx->offset = offset;

	if (bx->n_hline > 150)
		return -1;
	return 0;
}

/* Decomposition only, for the host-logic tests: lines[] receives n_hline (start, end) pairs then
 * n_vline (band, factor, start, end) quads; info = {n_hline, n_vline, divisor, rounding, offset,
 * max_line}.  Returns the number of ints written, -1 on error. */
int
port_conva_decompose(const double *mask, int mw, int mh, double scale, double offset, int layers,
	int cluster, int *info, int *lines, int max_ints)
{
	CaBoxes *bx = malloc(sizeof(CaBoxes));
	double *im = malloc(sizeof(double) * mw * mh);
	double iscale, ioffset;
	int n = -1;

	ca_intize(mask, mw * mh, scale, offset, im, &iscale, &ioffset);
	if (ca_decompose_boxes(bx, im, mw, mh, iscale, ioffset, layers, cluster) == 0 &&
		2 * bx->n_hline + 4 * bx->n_vline <= max_ints) {
		n = 0;
		for (int i = 0; i < bx->n_hline; i++) {
			lines[n++] = bx->hstart[i];
			lines[n++] = bx->hend[i];
		}
		for (int i = 0; i < bx->n_vline; i++) {
			lines[n++] = bx->vband[i];
			lines[n++] = bx->vfactor[i];
			lines[n++] = bx->vstart[i];
			lines[n++] = bx->vend[i];
		}
		info[0] = bx->n_hline;
		info[1] = bx->n_vline;
		info[2] = bx->divisor;
		info[3] = bx->rounding;
		info[4] = bx->offset;
		info[5] = bx->max_line;
	}
	free(im);
	free(bx);
	return n;
}

/* ------------------------------------------------------------------ conva: pixels */

/* One output element: `IN` pixels, `MID` the type of the horizontal intermediate, `ACC` the type the
 * vertical pass sums in (conva.c:1099-1198). */
#define CONVA_PIXELS(IN, MID, ACC, CLIP) \
	for (int y = 0; y < height; y++) \
		for (int x = 0; x < width; x++) \
			for (int b = 0; b < bands; b++) { \
				ACC sum = 0; \
				for (int v = 0; v < bx->n_vline; v++) { \
					const int hl = bx->vband[v]; \
					ACC vsum = 0; \
					for (int k = bx->vstart[v]; k < bx->vend[v]; k++) { \
						const int yy = ca_clampi(y + k - mh / 2, 0, height - 1); \
						MID hsum = 0; \
						for (int i = bx->hstart[hl]; i < bx->hend[hl]; i++) { \
							const int xx = ca_clampi(x + i - mw / 2, 0, width - 1); \
							hsum += ((const IN *) in)[((size_t) yy * width + xx) * bands + b]; \
						} \
						vsum += hsum; \
					} \
					sum += bx->vfactor[v] * vsum; \
				} \
				sum = (sum + bx->rounding) / bx->divisor + bx->offset; \
				CLIP(sum); \
				((IN *) out)[((size_t) y * width + x) * bands + b] = sum; \
			}

#define CA_CLIP_UCHAR(V) \
	{ \
		if ((V) < 0) \
			(V) = 0; \
		else if ((V) > UCHAR_MAX) \
			(V) = UCHAR_MAX; \
	}
#define CA_CLIP_CHAR(V) \
	{ \
		if ((V) < SCHAR_MIN) \
			(V) = SCHAR_MIN; \
		else if ((V) > SCHAR_MAX) \
			(V) = SCHAR_MAX; \
	}
#define CA_CLIP_USHORT(V) \
	{ \
		if ((V) < 0) \
			(V) = 0; \
		else if ((V) > USHRT_MAX) \
			(V) = USHRT_MAX; \
	}
#define CA_CLIP_SHORT(V) \
	{ \
		if ((V) < SHRT_MIN) \
			(V) = SHRT_MIN; \
		else if ((V) > SHRT_MAX) \
			(V) = SHRT_MAX; \
	}
#define CA_CLIP_NONE(V) \
	{ \
	}

int
port_conva(const void *in, int width, int height, int bands, int format, const double *mask,
	int mw, int mh, double scale, double offset, int layers, int cluster, void *out)
{
	CaBoxes *bx = malloc(sizeof(CaBoxes));
	double *im = malloc(sizeof(double) * mw * mh);
	double iscale, ioffset;
	int result = 0;

	ca_intize(mask, mw * mh, scale, offset, im, &iscale, &ioffset);
	if (ca_decompose_boxes(bx, im, mw, mh, iscale, ioffset, layers, cluster)) {
		free(im);
		free(bx);
		return -1;
	}
	const int small = bx->max_line < 256;

	/* unsigned totals wrap, and `V < 0` never holds for them: kept as the reference has it */
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wtype-limits"
#pragma GCC diagnostic ignored "-Wsign-compare"
	switch (format) {
	case PORT_FORMAT_UCHAR:
		if (small)
			CONVA_PIXELS(unsigned char, unsigned short, unsigned int, CA_CLIP_UCHAR)
		else
			CONVA_PIXELS(unsigned char, unsigned int, unsigned int, CA_CLIP_UCHAR)
		break;
	case PORT_FORMAT_CHAR:
		if (small)
			CONVA_PIXELS(signed char, signed short, signed int, CA_CLIP_CHAR)
		else
			CONVA_PIXELS(signed char, signed int, signed int, CA_CLIP_CHAR)
		break;
	case PORT_FORMAT_USHORT:
		if (small)
			CONVA_PIXELS(unsigned short, unsigned short, unsigned int, CA_CLIP_USHORT)
		else
			CONVA_PIXELS(unsigned short, unsigned int, unsigned int, CA_CLIP_USHORT)
		break;
	case PORT_FORMAT_SHORT:
		if (small)
			CONVA_PIXELS(signed short, signed short, signed int, CA_CLIP_SHORT)
		else
			CONVA_PIXELS(signed short, signed int, signed int, CA_CLIP_SHORT)
		break;
	case PORT_FORMAT_UINT:
		if (small)
			CONVA_PIXELS(unsigned int, unsigned short, unsigned int, CA_CLIP_SHORT)
		else
			CONVA_PIXELS(unsigned int, unsigned int, unsigned int, CA_CLIP_NONE)
		break;
	case PORT_FORMAT_INT:
		if (small)
			CONVA_PIXELS(signed int, signed short, signed int, CA_CLIP_NONE)
		else
			CONVA_PIXELS(signed int, signed int, signed int, CA_CLIP_NONE)
		break;
	case PORT_FORMAT_FLOAT:
		CONVA_PIXELS(float, float, float, CA_CLIP_NONE)
		break;
	case PORT_FORMAT_DOUBLE:
		CONVA_PIXELS(double, double, double, CA_CLIP_NONE)
		break;
	default:
		result = -1;
	}
#pragma GCC diagnostic pop

	free(im);
	free(bx);
	return result;
}

/* ------------------------------------------------------------------ convasep */

typedef struct {
	int n_lines;
	int start[CA_MAX_LINES + 1], end[CA_MAX_LINES + 1], factor[CA_MAX_LINES + 1];
	int divisor, rounding, offset, width;
} CaLines;

static int
ca_close_line(CaLines *ln, int x)
{
	ln->end[ln->n_lines] = x;
	if (ln->n_lines >= CA_MAX_LINES - 1)
		return -1;
	ln->n_lines += 1;
	return 0;
}

/* convasep.c:152-330 */
static int
ca_decompose_lines(CaLines *ln, const double *coeff, int width, double scale, double offset,
	int n_layers)
{
	memset(ln, 0, sizeof(*ln));
	ln->width = width;

	double max = 0, min = 0;
	for (int x = 0; x < width; x++) {
		if (coeff[x] > max)
			max = coeff[x];
		if (coeff[x] < min)
			min = coeff[x];
	}
	if (!(max > 0))
		return -1;

	double depth = (max - min) / n_layers;
	const int layers_above = ceil(max / depth);
	depth = max / layers_above;
	const int layers_below = floor(min / depth);
	int64_t span = (int64_t) layers_above - layers_below;
	const int layers = span < 1 ? 1 : (span > 1000 ? 1000 : (int) span);

	for (int z = 0; z < layers; z++) {
		const double y = max - (1 + z) * depth;
		const double y_ph = y + depth / 2;
		const int positive = z < layers_above;
		int inside = 0;

		for (int x = 0; x < width; x++) {
			if ((positive && coeff[x] >= y_ph) || (!positive && coeff[x] <= y_ph)) {
				if (!inside) {
					ln->start[ln->n_lines] = x;
					ln->factor[ln->n_lines] = positive ? 1 : -1;
					inside = 1;
				}
			}
			else if (inside) {
				if (ca_close_line(ln, x))
					return -1;
				inside = 0;
			}
		}
		if (inside && ca_close_line(ln, width))
			return -1;
	}
	if (ln->n_lines == 0)
		return -1;

	/* :249-262 common up identical lines */
	for (int z = 0; z < ln->n_lines; z++)
		for (int n = z + 1; n < ln->n_lines; n++)
			if (ln->start[z] == ln->start[n] && ln->end[z] == ln->end[n]) {
				ln->factor[z] += ln->factor[n];
				ln->factor[n] = 0;
			}
	/* :264-275 drop factor-0 lines; z is not re-examined after a shift, so the second of two
	 * adjacent dead lines survives (it contributes nothing) */
	for (int z = 0; z < ln->n_lines; z++)
		if (ln->factor[z] == 0) {
			for (int x = z; x < ln->n_lines; x++) {
				ln->start[x] = ln->start[x + 1];
				ln->end[x] = ln->end[x + 1];
				ln->factor[x] = ln->factor[x + 1];
			}
			ln->n_lines -= 1;
		}

	double area = 0;
	for (int z = 0; z < ln->n_lines; z++)
		area += ln->factor[z] * (ln->end[z] - ln->start[z]);

	int x = ln->factor[0];
	for (int z = 1; z < ln->n_lines; z++)
		x = ca_gcd(x, ln->factor[z]);
	if (x == 0)
		return -1; /* the reference divides by zero here */
	for (int z = 0; z < ln->n_lines; z++)
		ln->factor[z] /= x;
	area *= x;

	double sum = 0;
	for (int z = 0; z < width; z++)
		sum += coeff[z];

	const double d = rint(sum * area / scale);
	ln->divisor = d > 1 ? d : 1;
	ln->rounding = (ln->divisor + 1) / 2;
	ln->offset = offset;
	return 0;
}

/* lines[] receives n_lines (start, end, factor) triples; info = {n_lines, divisor, rounding, offset}. */
int
port_convasep_decompose(const double *mask, int n, double scale, double offset, int layers,
	int *info, int *lines, int max_ints)
{
	CaLines *ln = malloc(sizeof(CaLines));
	double *im = malloc(sizeof(double) * n);
	double iscale, ioffset;
	int count = -1;

	ca_intize(mask, n, scale, offset, im, &iscale, &ioffset);
	if (ca_decompose_lines(ln, im, n, iscale, ioffset, layers) == 0 && 3 * ln->n_lines <= max_ints) {
		count = 0;
		for (int i = 0; i < ln->n_lines; i++) {
			lines[count++] = ln->start[i];
			lines[count++] = ln->end[i];
			lines[count++] = ln->factor[i];
		}
		info[0] = ln->n_lines;
		info[1] = ln->divisor;
		info[2] = ln->rounding;
		info[3] = ln->offset;
	}
	free(im);
	free(ln);
	return count;
}

/* One pass along x (vertical == 0) or y.  The vertical pass adds the offset (convasep.c:452-455). */
#define CASEP_INT(ACC, TYPE, CLIP) \
	for (int y = 0; y < height; y++) \
		for (int x = 0; x < width; x++) \
			for (int b = 0; b < bands; b++) { \
				int64_t sum = 0; \
				for (int z = 0; z < ln->n_lines; z++) { \
					ACC isum = 0; \
					for (int k = ln->start[z]; k < ln->end[z]; k++) { \
						const int xx = vertical ? x : ca_clampi(x + k - ln->width / 2, 0, width - 1); \
						const int yy = vertical ? ca_clampi(y + k - ln->width / 2, 0, height - 1) : y; \
						isum += ((const TYPE *) in)[((size_t) yy * width + xx) * bands + b]; \
					} \
					sum += (int64_t) ln->factor[z] * isum; \
				} \
				sum = (sum + ln->rounding) / ln->divisor + (vertical ? ln->offset : 0); \
				CLIP(sum); \
				((TYPE *) out)[((size_t) y * width + x) * bands + b] = sum; \
			}

#define CASEP_FLOAT(TYPE) \
	for (int y = 0; y < height; y++) \
		for (int x = 0; x < width; x++) \
			for (int b = 0; b < bands; b++) { \
				double sum = 0; \
				for (int z = 0; z < ln->n_lines; z++) { \
					double dsum = 0; \
					for (int k = ln->start[z]; k < ln->end[z]; k++) { \
						const int xx = vertical ? x : ca_clampi(x + k - ln->width / 2, 0, width - 1); \
						const int yy = vertical ? ca_clampi(y + k - ln->width / 2, 0, height - 1) : y; \
						dsum += ((const TYPE *) in)[((size_t) yy * width + xx) * bands + b]; \
					} \
					sum += ln->factor[z] * dsum; \
				} \
				if (vertical) \
					sum = sum / ln->divisor + ln->offset; \
				else \
					sum = sum / ln->divisor; \
				((TYPE *) out)[((size_t) y * width + x) * bands + b] = sum; \
			}

static int
ca_sep_pass(const CaLines *ln, const void *in, int width, int height, int bands, int format,
	int vertical, void *out)
{
	switch (format) {
	case PORT_FORMAT_UCHAR:
		CASEP_INT(unsigned int, unsigned char, CA_CLIP_UCHAR)
		break;
	case PORT_FORMAT_CHAR:
		CASEP_INT(signed int, signed char, CA_CLIP_CHAR)
		break;
	case PORT_FORMAT_USHORT:
		CASEP_INT(unsigned int, unsigned short, CA_CLIP_USHORT)
		break;
	case PORT_FORMAT_SHORT:
		CASEP_INT(signed int, signed short, CA_CLIP_SHORT)
		break;
	case PORT_FORMAT_UINT:
		CASEP_INT(unsigned int, unsigned int, CA_CLIP_NONE)
		break;
	case PORT_FORMAT_INT:
		CASEP_INT(signed int, signed int, CA_CLIP_NONE)
		break;
	case PORT_FORMAT_FLOAT:
		CASEP_FLOAT(float)
		break;
	case PORT_FORMAT_DOUBLE:
		CASEP_FLOAT(double)
		break;
	default:
		return -1;
	}
	return 0;
}

int
port_convasep(const void *in, int width, int height, int bands, int format, const double *mask,
	int n, double scale, double offset, int layers, void *out)
{
	static const int size_of[] = { 1, 1, 2, 2, 4, 4, 4, 8, 8, 16 };
	CaLines *ln = malloc(sizeof(CaLines));
	double *im = malloc(sizeof(double) * n);
	double iscale, ioffset;
	int result = -1;

	ca_intize(mask, n, scale, offset, im, &iscale, &ioffset);
	if (format >= 0 && format <= 9 && ca_decompose_lines(ln, im, n, iscale, ioffset, layers) == 0) {
		void *mid = malloc((size_t) width * height * bands * size_of[format]);
		result = ca_sep_pass(ln, in, width, height, bands, format, 0, mid);
		if (result == 0)
			result = ca_sep_pass(ln, mid, width, height, bands, format, 1, out);
		free(mid);
	}
	free(im);
	free(ln);
	return result;
}
