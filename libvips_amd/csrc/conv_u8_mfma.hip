// vips_convsep / vips_gaussblur (precision integer) on uchar images, both passes on the matrix cores: the
// __global__ wrapper and launch of conv_u8_mfma_body.h (see there); host side conv_u8_mfma_host.h (both
// shared with tests/emul).
#include "conv_u8_mfma_body.h"

#include <cstdio>

namespace vh {

// blocks are dealt to the 8 XCDs round-robin by the hardware: give an XCD a contiguous range of items
// (neighbouring strips of a segment share their halo columns in its L2)
template <int B, bool WIDE, int MODE>
// (3 waves per SIMD: the 45 KB of LDS a block takes at up to 3 bands allow 3 blocks per CU; the separable kernel on
// 4 bands takes 56-64 KB -- two blocks a CU whatever the registers -- and spilled 10 / 24 vector registers at 168:
// two waves per SIMD there, profiles/r05_kernel_resources.txt)
__global__ void __launch_bounds__(256, (B == 4 && MODE == 0) ? 2 : 3)
conv_u8_mfma_sep(CmArgs a, int items)
{
	extern __shared__ __attribute__((aligned(16))) unsigned int cm_lds[];
	const int per = (items + 7) >> 3;
	const int item = (int) (blockIdx.x & 7) * per + (int) (blockIdx.x >> 3);
	if ((int) (blockIdx.x >> 3) < per && item < items)
		conv_u8_mfma_item<B, WIDE, MODE>(a, item, cm_lds);
}

// ushort: B byte planes (2 x bands); 75 KB of LDS a block at 3 bands and 33 taps: two blocks a CU (one at 4 bands)
template <int B, bool WIDE>
__global__ void __launch_bounds__(256, B <= 6 ? 2 : 1)
conv_u16_mfma_sep(CmArgs a, int items)
{
	extern __shared__ __attribute__((aligned(16))) unsigned int cm_lds[];
	const int per = (items + 7) >> 3;
	const int item = (int) (blockIdx.x & 7) * per + (int) (blockIdx.x >> 3);
	if ((int) (blockIdx.x >> 3) < per && item < items)
		conv_u8_mfma_item<B, WIDE, 0, true>(a, item, cm_lds);
}

} // namespace vh

#include "conv_u8_mfma_host.h"

namespace vh {

template <typename K>
static int cm_go(K kernel, const CmArgs &a, int items, size_t lds)
{
	if (lds > 64 * 1024)
		VH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	if (getenv("VIPS_HIP_CONV_MFMA_DEBUG")) {
		int nb = -1;
		(void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, CM_NT, lds);
		fprintf(stderr, "conv_u8_mfma: %d items, %zu bytes of LDS, %d blocks per CU\n", items, lds, nb);
	}
	const int grid = 8 * ((items + 7) >> 3);
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(CM_NT), lds, stream(), a, items);
	VH_CHECK(hipGetLastError());
	return 0;
}

static int cm_launch16(int bands, bool wide, const CmArgs &a, int grid, size_t lds)
{
#define CM_CASE16(NB) \
	case NB: \
		return wide ? cm_go(conv_u16_mfma_sep<2 * NB, true>, a, grid, lds) : cm_go(conv_u16_mfma_sep<2 * NB, false>, a, grid, lds);
	switch (bands) {
		CM_CASE16(1)
		CM_CASE16(2)
		CM_CASE16(3)
		CM_CASE16(4)
	}
#undef CM_CASE16
	return 1;
}

static int cm_launch(int bands, bool wide, bool twod, const CmArgs &a, int grid, size_t lds)
{
#define CM_MODE(B, M) (wide ? cm_go(conv_u8_mfma_sep<B, true, M>, a, grid, lds) : cm_go(conv_u8_mfma_sep<B, false, M>, a, grid, lds))
#define CM_CASE(B) \
	case B: \
		if (twod && a.ksteps == 3 && a.mh == 3) \
			return CM_MODE(B, 3); \
		if (twod && a.ksteps == 3 && a.mh == 5) \
			return CM_MODE(B, 5); \
		if (twod) \
			return CM_MODE(B, -1); \
		return CM_MODE(B, 0);
	switch (bands) {
		CM_CASE(1)
		CM_CASE(2)
		CM_CASE(3)
		CM_CASE(4)
	}
#undef CM_CASE
#undef CM_MODE
	return 1;
}

} // namespace vh
