// TEST INFRASTRUCTURE: the matrix-core separable convolution (libvips_amd/csrc/conv_u8_mfma_body.h) on host
// fibers; takes the place of conv_u8_mfma.hip in libvipship_emul.so (see conv_u8_emul.cpp).
#include "gcn.h"

#include "conv_u8_mfma_body.h"

#include "conv_u8_mfma_host.h"

#include <atomic>
#include <thread>

namespace vh {

template <int B, bool WIDE, int MODE, bool U16 = false>
static void cm_run(const CmArgs &a, int items, size_t lds)
{
	(void) hipStreamSynchronize(stream());
	std::atomic<int> next(0);
	auto worker = [&]() {
		std::vector<unsigned int> buf(lds / 4 + 4);
		for (;;) {
			const int item = next.fetch_add(1);
			if (item >= items)
				break;
			for (size_t i = 0; i < buf.size(); i++)
				buf[i] = 0xdeadbeefu + (unsigned int) i * 2654435761u;
			emul::run_block(CM_NT, [&]() { conv_u8_mfma_item<B, WIDE, MODE, U16>(a, item, buf.data()); });
		}
	};
	unsigned int nthreads = std::thread::hardware_concurrency();
	nthreads = nthreads < 1 ? 1 : nthreads > (unsigned int) items ? (unsigned int) items : nthreads;
	std::vector<std::thread> pool;
	for (unsigned int i = 0; i < nthreads; i++)
		pool.emplace_back(worker);
	for (std::thread &t : pool)
		t.join();
}

static int cm_launch16(int bands, bool wide, const CmArgs &a, int grid, size_t lds)
{
#define CM_CASE16(NB) \
	case NB: \
		if (wide) \
			cm_run<2 * NB, true, 0, true>(a, grid, lds); \
		else \
			cm_run<2 * NB, false, 0, true>(a, grid, lds); \
		return 0;
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
#define CM_MODE(B, M) \
	do { \
		if (wide) \
			cm_run<B, true, M>(a, grid, lds); \
		else \
			cm_run<B, false, M>(a, grid, lds); \
	} while (0)
#define CM_CASE(B) \
	case B: \
		if (twod && a.ksteps == 3 && a.mh == 3) \
			CM_MODE(B, 3); \
		else if (twod && a.ksteps == 3 && a.mh == 5) \
			CM_MODE(B, 5); \
		else if (twod) \
			CM_MODE(B, -1); \
		else \
			CM_MODE(B, 0); \
		return 0;
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
