// TEST INFRASTRUCTURE: the streaming integer convolution on ushort (libvips_amd/csrc/conv_u16_body.h)
// on host fibers; takes the place of conv_u16.hip in libvipship_emul.so.
#include "gcn.h"

#include "conv_u16_body.h"

#include "conv_u16_host.h"

#include <atomic>
#include <thread>
#include <vector>

namespace vh {

template <typename F>
static void cu16_run(int grid, size_t lds, F block)
{
	(void) hipStreamSynchronize(stream());
	std::atomic<int> next(0);
	auto worker = [&]() {
		std::vector<unsigned int> buf(lds / 4 + 4);
		for (;;) {
			const int wg = next.fetch_add(1);
			if (wg >= grid)
				break;
			emul::run_block(CU16_NT, [&]() { block(buf.data()); });
		}
	};
	unsigned int nthreads = std::thread::hardware_concurrency();
	nthreads = nthreads < 1 ? 1 : nthreads > (unsigned int) grid ? (unsigned int) grid : nthreads;
	std::vector<std::thread> pool;
	for (unsigned int i = 0; i < nthreads; i++)
		pool.emplace_back(worker);
	for (std::thread &t : pool)
		t.join();
}

#define CU16_2D(B, MH, H) \
	if (bands == B && mh == MH && h == H) { \
		cu16_run(grid, lds, [&](unsigned int *l) { conv_u16_2d_block<B, MH, H>(a, l); }); \
		return 0; \
	}
#define CU16_2D_B(B) CU16_2D(B, 1, 1) CU16_2D(B, 1, 2) CU16_2D(B, 3, 1) CU16_2D(B, 3, 2) CU16_2D(B, 5, 1) CU16_2D(B, 5, 2)

static int cu16_launch(int bands, int mh, int h, const Cu16Args &a, int grid, size_t lds)
{
	CU16_2D_B(1) CU16_2D_B(3) CU16_2D_B(4)
	return 1;
}

} // namespace vh
