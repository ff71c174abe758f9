// TEST INFRASTRUCTURE: the banded-matrix vertical reduce (libvips_amd/csrc/reduce_band_body.h) on host
// fibers; takes the place of reduce_band.hip in libvipship_emul.so (see conv_u8_emul.cpp).
#include "gcn.h"

#include "reduce_band_body.h"

#include "reduce_band_host.h"

#include <atomic>
#include <thread>

namespace vh {

static int rb_launch(const RbArgs &a, int grid)
{
	(void) hipStreamSynchronize(stream());
	const int groups = grid / a.nblocks;
	std::atomic<int> next(0);
	auto worker = [&]() {
		for (;;) {
			const int id = next.fetch_add(1);
			if (id >= grid)
				break;
			emul::run_block(RB_NT, [&]() {
				const int g = id / groups, grp = id - g * groups;
				const int strip = 4 * grp + wave_index();
				if (strip < a.strips)
					reducev_band_wave(a, strip, g);
			});
		}
	};
	unsigned int nthreads = std::thread::hardware_concurrency();
	nthreads = nthreads < 1 ? 1 : nthreads > (unsigned int) grid ? (unsigned int) grid : nthreads;
	std::vector<std::thread> pool;
	for (unsigned int i = 0; i < nthreads; i++)
		pool.emplace_back(worker);
	for (std::thread &t : pool)
		t.join();
	return 0;
}

template <int B>
static void rbh_run(const RbhArgs &a, int grid)
{
	(void) hipStreamSynchronize(stream());
	const int groups = (a.xtiles + 3) / 4;
	std::atomic<int> next(0);
	auto worker = [&]() {
		for (;;) {
			const int id = next.fetch_add(1);
			if (id >= grid)
				break;
			const int yt = id / groups, grp = id - yt * groups;
			emul::run_block(RB_NT, [&]() {
				const int xt = 4 * grp + wave_index();
				if (xt < a.xtiles)
					reduceh_band_wave<B>(a, xt, yt);
			});
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

static int rbh_launch(int bands, const RbhArgs &a, int grid)
{
	switch (bands) {
	case 1:
		rbh_run<1>(a, grid);
		return 0;
	case 2:
		rbh_run<2>(a, grid);
		return 0;
	case 3:
		rbh_run<3>(a, grid);
		return 0;
	case 4:
		rbh_run<4>(a, grid);
		return 0;
	}
	return 1;
}

} // namespace vh
