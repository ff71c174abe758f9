// TEST INFRASTRUCTURE: the matrix-core separable convolution (libvips_amd/csrc/conv_u8_mfma_body.h) on host
// fibers; takes the place of conv_u8_mfma.hip in libvipship_emul.so (see conv_u8_emul.cpp).
#include "gcn.h"

#include "conv_u8_mfma_body.h"

#include "conv_u8_mfma_host.h"

#include <atomic>
#include <thread>

namespace vh {

template <int B, bool WIDE>
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
			emul::run_block(CM_NT, [&]() { conv_u8_mfma_item<B, WIDE>(a, item, buf.data()); });
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

static int cm_launch(int bands, bool wide, const CmArgs &a, int grid, size_t lds)
{
#define CM_CASE(B) \
	case B: \
		if (wide) \
			cm_run<B, true>(a, grid, lds); \
		else \
			cm_run<B, false>(a, grid, lds); \
		return 0;
	switch (bands) {
		CM_CASE(1)
		CM_CASE(2)
		CM_CASE(3)
		CM_CASE(4)
	}
#undef CM_CASE
	return 1;
}

} // namespace vh
