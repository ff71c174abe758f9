// TEST INFRASTRUCTURE: the fused resize + sharpen kernel's body on host fibers.  This file takes
// the place of libvips_amd/csrc/resize_sharpen.hip in tests/emul/_build/libvipship_emul.so: the
// same body (resize_sharpen_body.h) and the same host side (resize_sharpen_host.h), the launch
// replaced by emul::run_block per workgroup.  Run under the mock HIP runtime (device memory is
// host memory), the library's batch entry point then produces real thumbnails on the CPU, which
// tests/test_emul_resize_sharpen.py compares with the reference.
#include "gcn.h"

#include "resize_sharpen_body.h"

#include "resize_sharpen_host.h"

#include <atomic>
#include <thread>

namespace vh {

template <int VS>
static void rsh_run_blocks(const RshArgs &a, const RshPtrs &p, unsigned int blocks, size_t lds)
{
	std::vector<unsigned long long> words(2 * RSH_MAXB);
	for (int i = 0; i < RSH_MAXB; i++) {
		words[i] = (unsigned long long) (uintptr_t) p.in[i];
		words[RSH_MAXB + i] = (unsigned long long) (uintptr_t) p.out[i];
	}
	std::atomic<unsigned int> next(0);
	auto worker = [&]() {
		// LDS starts as garbage on the device: make reads of unwritten LDS stand out
		std::vector<unsigned int> buf(lds / 4 + 4);
		for (;;) {
			const unsigned int wg = next.fetch_add(1);
			if (wg >= blocks)
				break;
			for (size_t i = 0; i < buf.size(); i++)
				buf[i] = 0xdeadbeefu + (unsigned int) i * 2654435761u;
			const KernargWords kp = { words.data() };
			emul::run_block(RSH_NT, [&]() { resize_sharpen_body<VS, RSH_NP>(a, kp, (int) wg, buf.data()); });
		}
	};
	unsigned int nthreads = std::thread::hardware_concurrency();
	if (nthreads < 1)
		nthreads = 1;
	if (nthreads > blocks)
		nthreads = blocks;
	std::vector<std::thread> pool;
	for (unsigned int i = 0; i < nthreads; i++)
		pool.emplace_back(worker);
	for (std::thread &t : pool)
		t.join();
}

static int rsh_launch(int vs, const RshArgs &a, const RshPtrs &p, unsigned int blocks, size_t lds)
{
	(void) hipStreamSynchronize(stream());
	if (vs == 4)
		rsh_run_blocks<4>(a, p, blocks, lds);
	else
		rsh_run_blocks<8>(a, p, blocks, lds);
	return 0;
}

} // namespace vh
