// TEST INFRASTRUCTURE: the packed-byte integer convolutions (libvips_amd/csrc/conv_u8_body.h) on
// host fibers; takes the place of conv_u8.hip in libvipship_emul.so (see resize_sharpen_emul.cpp).
#include "gcn.h"

#include "conv_u8_body.h"

#include "conv_u8_host.h"

#include <atomic>
#include <thread>

namespace vh {

template <typename F>
static void cu8_run(const Cu8Args &a, int grid, size_t lds, F block)
{
	(void) hipStreamSynchronize(stream());
	std::atomic<int> next(0);
	auto worker = [&]() {
		std::vector<unsigned int> buf(lds / 4 + 4);
		for (;;) {
			const int wg = next.fetch_add(1);
			if (wg >= grid)
				break;
			for (size_t i = 0; i < buf.size(); i++)
				buf[i] = 0xdeadbeefu + (unsigned int) i * 2654435761u;
			emul::run_block(CU8_NT, [&]() { block(buf.data()); });
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

#define CU8_SEP(B, ND, H) \
	if (bands == B && nd == ND && h == H) { \
		if (regs || ND == 3) \
			cu8_run(a, grid, lds, [&](unsigned int *l) { conv_u8_sep_block<B, ND, H, true>(a, l); }); \
		else \
			cu8_run(a, grid, lds, [&](unsigned int *l) { conv_u8_sep_block<B, ND, H, false>(a, l); }); \
		return 0; \
	}
#define CU8_SEP_B(B) \
	CU8_SEP(B, 3, 1) CU8_SEP(B, 3, 2) CU8_SEP(B, 3, -1) CU8_SEP(B, 5, -1) CU8_SEP(B, 7, -1) CU8_SEP(B, 9, -1)
#define CU8_2D(B, MH, H) \
	if (bands == B && mh == MH && h == H) { \
		cu8_run(a, grid, lds, [&](unsigned int *l) { conv_u8_2d_block<B, MH, H>(a, l); }); \
		return 0; \
	}
#define CU8_2D_B(B) \
	CU8_2D(B, 3, 1) CU8_2D(B, 3, 2) CU8_2D(B, 3, -1) CU8_2D(B, 5, 1) CU8_2D(B, 5, 2) CU8_2D(B, 5, -1) \
	CU8_2D(B, 7, 1) CU8_2D(B, 7, 2) CU8_2D(B, 7, -1)

static int cu8_launch_sep(int bands, int nd, bool regs, const Cu8Args &a, int grid, size_t lds)
{
	const int h = nd == 3 && a.half <= 2 ? a.half : -1;
	CU8_SEP_B(1) CU8_SEP_B(3) CU8_SEP_B(4)
	return 1;
}

static int cu8_launch_2d(int bands, int mh, const Cu8Args &a, int grid, size_t lds)
{
	const int h = a.half < 1 ? 1 : a.half > 2 ? -1 : a.half;
	CU8_2D_B(1) CU8_2D_B(3) CU8_2D_B(4)
	return 1;
}

} // namespace vh
