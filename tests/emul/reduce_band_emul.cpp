// TEST INFRASTRUCTURE: the banded-matrix vertical reduce (libvips_amd/csrc/reduce_band_body.h) on host
// fibers; takes the place of reduce_band.hip in libvipship_emul.so (see conv_u8_emul.cpp).
#include "gcn.h"

#include "reduce_band_body.h"

#include "reduce_band_host.h"

#include <atomic>
#include <thread>

namespace vh {

static int rb_launch(const RbArgs &a, int groups, int wblocks, bool u16)
{
	(void) hipStreamSynchronize(stream());
	const int grid = groups * wblocks;
	std::atomic<int> next(0);
	auto worker = [&]() {
		for (;;) {
			const int id = next.fetch_add(1);
			if (id >= grid)
				break;
			emul::run_block(RB_NT, [&]() {
				const int g = id / groups, grp = id - g * groups;
				const int strip = 4 * grp + wave_index();
				if (strip < a.strips) {
					if (a.vs > 1) {
						switch (a.vs) {
#define RB_BOX(VS) \
	case VS: \
		reducev_box_band_wave<VS>(a, strip, g, rb_bottom_up(a, g)); \
		break;
							RB_BOX(2) RB_BOX(3) RB_BOX(4) RB_BOX(5) RB_BOX(6) RB_BOX(7) RB_BOX(8) RB_BOX(9) RB_BOX(10) RB_BOX(11)
							RB_BOX(12) RB_BOX(13) RB_BOX(14) RB_BOX(15) RB_BOX(16)
#undef RB_BOX
						}
					}
					else if (u16)
						reducev_band_wave<true>(a, strip, g, rb_bottom_up(a, g));
					else
						reducev_band_wave<false>(a, strip, g, rb_bottom_up(a, g));
				}
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

template <int B, bool U16>
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
				if (xt < a.xtiles) {
					if constexpr (U16)
						reduceh16_band_wave<B>(a, xt, yt);
					else
						reduceh_band_wave<B>(a, xt, yt);
				}
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

static int rbh_launch(int bands, const RbhArgs &a, int grid, bool u16)
{
#define RBH_CASE(B) \
	case B: \
		if (u16) \
			rbh_run<B, true>(a, grid); \
		else \
			rbh_run<B, false>(a, grid); \
		return 0;
	switch (bands) {
		RBH_CASE(1)
		RBH_CASE(2)
		RBH_CASE(3)
		RBH_CASE(4)
	}
#undef RBH_CASE
	return 1;
}

} // namespace vh
