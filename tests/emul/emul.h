// TEST INFRASTRUCTURE: a workgroup on host fibers.  run_block() runs `threads` copies of a
// function, each on its own ucontext stack, switching only at barrier(); a block is correct here
// exactly when it is correct under any interleaving that respects its barriers only if it has no
// data race between barriers -- which is what the bodies promise.  To shake out ordering
// assumptions the fibers of a block are resumed in a different order after every barrier.
#ifndef VH_EMUL_H
#define VH_EMUL_H

#include <cstddef>
#include <functional>

namespace emul {

int current_tid();
void barrier();
unsigned long long clock_ticks();
// every thread of the block hands in a value and gets the one thread `src` handed in (two barriers)
unsigned int exchange(unsigned int value, int src);
// v_cmp + s_mov of the lane mask: the lanes of the calling thread's wave (64 consecutive threads) that make
// this call before their next barrier (or their end) vote; a lane that goes to the barrier without voting is
// a lane the branch had masked off.  Bit i = lane i's predicate.
unsigned long long ballot(bool pred);
// the same meeting with a payload (at most 16 bytes): what every voting lane of the wave handed in, for the
// cross-lane instructions (DPP moves, the matrix instructions' operands)
struct WaveData {
	unsigned long long voters;   // lanes that took part
	unsigned long long mask;     // their predicates
	unsigned char data[64][16];
};
const WaveData &wave_share(const void *payload, int bytes, bool pred);
// the block's dynamic LDS (set by whoever runs the block; convsep_stream's LDS-DMA addresses are offsets into it)
void set_lds_base(unsigned char *base, size_t bytes = 0);
size_t lds_bytes();
unsigned char *lds_base();
// run `fn` as `threads` fibers (one block); fn reads current_tid()
void run_block(int threads, const std::function<void()> &fn);

} // namespace emul

#endif
