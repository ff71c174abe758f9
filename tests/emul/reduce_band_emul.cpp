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
	const int groups = (a.strips + 3) / 4;
	std::atomic<int> next(0);
	auto worker = [&]() {
		for (;;) {
			const int id = next.fetch_add(1);
			if (id >= grid)
				break;
			const int g = id / groups, grp = id - g * groups;
			emul::run_block(RB_NT, [&]() {
				const int strip = 4 * grp + wave_index();
				// (every fiber of a wave takes the same branch; a wave past the last strip meets nobody)
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

} // namespace vh
