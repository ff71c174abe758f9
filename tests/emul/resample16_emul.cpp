// TEST INFRASTRUCTURE: the ushort streaming resample kernels (libvips_amd/csrc/resample16_body.h)
// on host fibers; takes the place of resample16.hip in libvipship_emul.so.
#include "gcn.h"

#include "resample16_body.h"

#include "resample16_host.h"

#include <atomic>
#include <thread>

namespace vh {

template <typename F>
static void r16_run(int blocks, size_t lds, F block)
{
	(void) hipStreamSynchronize(stream());
	std::atomic<int> next(0);
	auto worker = [&]() {
		std::vector<unsigned int> buf(lds / 4 + 8);
		for (;;) {
			const int wg = next.fetch_add(1);
			if (wg >= blocks)
				break;
			for (size_t i = 0; i < buf.size(); i++)
				buf[i] = 0xdeadbeefu + (unsigned int) i * 2654435761u;
			emul::run_block(R16_NT, [&]() { block(wg, buf.data()); });
		}
	};
	unsigned int nthreads = std::thread::hardware_concurrency();
	nthreads = nthreads < 1 ? 1 : nthreads > (unsigned int) blocks ? (unsigned int) blocks : nthreads;
	std::vector<std::thread> pool;
	for (unsigned int i = 0; i < nthreads; i++)
		pool.emplace_back(worker);
	for (std::thread &t : pool)
		t.join();
}

static int r16_launch_v(int which, const R16VArgs &a, int gx, int gy)
{
	r16_run(gx * gy, 64, [&](int wg, unsigned int *l) {
		if (which == 0)
			reducev16_block(a, l);
		else if (which == 2)
			reducev8_block(a, l);
		else
			shrinkv16_body(a, wg % gx, wg / gx, gy);
	});
	return 0;
}

static int r16_launch_h(int which, int bands, const R16HArgs &a, int gx, int gy, size_t lds)
{
#define R16_H(B) \
	if (bands == B) { \
		r16_run(gx * gy, lds, [&](int wg, unsigned int *l) { \
			if (which == 0) \
				reduceh16_body<B>(a, wg % gx, wg / gx, gy, l); \
			else if (which == 2) \
				shrinkbox16_body<B>(a, wg % gx, wg / gx, gy); \
			else \
				shrinkh16_body<B>(a, wg % gx, wg / gx, gy); \
		}); \
		return 0; \
	}
	R16_H(1) R16_H(2) R16_H(3) R16_H(4)
#undef R16_H
	return -1;
}

static int r16_launch_boxc(int bands, int lanes_per_box, const R16HArgs &a, int gx, int gy)
{
#define R16_C(B, L) \
	if (bands == B && lanes_per_box == L) { \
		r16_run(gx * gy, 64, [&](int wg, unsigned int *) { shrinkbox16c_body<B, L>(a, wg % gx, wg / gx, gy); }); \
		return 0; \
	}
	R16_C(1, 1) R16_C(1, 2) R16_C(1, 4) R16_C(2, 1) R16_C(2, 2) R16_C(2, 4) R16_C(4, 1) R16_C(4, 2) R16_C(4, 4)
#undef R16_C
	return -1;
}

} // namespace vh
