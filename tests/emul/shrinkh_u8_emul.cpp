// TEST INFRASTRUCTURE: the packed-byte vips_shrinkh on uchar (libvips_amd/csrc/shrinkh_u8_body.h) on
// host fibers; takes the place of shrinkh_u8.hip in libvipship_emul.so.
#include "gcn.h"

#include "shrinkh_u8_body.h"

#include "shrinkh_u8_host.h"

#include <atomic>
#include <thread>
#include <vector>

namespace vh {

template <typename F>
static void sh8_run(int blocks, F block)
{
	(void) hipStreamSynchronize(stream());
	std::atomic<int> next(0);
	auto worker = [&]() {
		for (;;) {
			const int wg = next.fetch_add(1);
			if (wg >= blocks)
				break;
			emul::run_block(SH8_NT, [&]() { block(wg); });
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

template <int B>
static int sh8_launch_b(int hs, const Sh8Args &a, int gx, int gy)
{
#define SH8_CASE(HS) \
	case HS: \
		sh8_run(gx * gy, [&](int wg) { shrinkh_u8_body<B, HS>(a, wg % gx, wg / gx, gy); }); \
		return 0;
	switch (hs) {
		SH8_CASE(0) SH8_CASE(2) SH8_CASE(3) SH8_CASE(4) SH8_CASE(5) SH8_CASE(6) SH8_CASE(7) SH8_CASE(8)
	default:
		return -1;
	}
#undef SH8_CASE
}

static int sh8_launch(int bands, int hs_template, const Sh8Args &a, int gx, int gy)
{
	switch (bands) {
	case 1: return sh8_launch_b<1>(hs_template, a, gx, gy);
	case 2: return sh8_launch_b<2>(hs_template, a, gx, gy);
	case 3: return sh8_launch_b<3>(hs_template, a, gx, gy);
	case 4: return sh8_launch_b<4>(hs_template, a, gx, gy);
	default: return -1;
	}
}

} // namespace vh
