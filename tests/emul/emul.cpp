// TEST INFRASTRUCTURE: see emul.h.
#include "emul.h"

#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace emul {

namespace {

// $EMUL_BACKTRACE=1: the native frames of a crash inside a fiber (addresses + /proc/self/maps for addr2line)
void crash_handler(int sig)
{
	void *frames[48];
	const int n = backtrace(frames, 48);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
struct CrashHook {
	CrashHook()
	{
		if (getenv("EMUL_BACKTRACE")) {
			static char altstack[1 << 16];
			stack_t ss;
			ss.ss_sp = altstack;
			ss.ss_size = sizeof(altstack);
			ss.ss_flags = 0;
			sigaltstack(&ss, nullptr);
			struct sigaction sa;
			sa.sa_handler = crash_handler;
			sigemptyset(&sa.sa_mask);
			sa.sa_flags = SA_ONSTACK;
			sigaction(SIGSEGV, &sa, nullptr);
		}
	}
} g_crash_hook;


struct Fiber {
	ucontext_t ctx;
	char *stack = nullptr;
	bool done = false;
	bool voting = false; // yielded inside ballot()
	bool pred = false;
	unsigned char payload[16] = {};
	const WaveData *shared = nullptr;
};

thread_local std::vector<Fiber> *g_fibers = nullptr;
thread_local ucontext_t g_main;
thread_local int g_tid = 0;
thread_local const std::function<void()> *g_fn = nullptr;
thread_local unsigned long long g_ticks = 0;
thread_local std::vector<unsigned int> *g_exchange = nullptr;
thread_local unsigned char *g_lds = nullptr;
thread_local size_t g_lds_bytes = 0;
// fiber stacks are kept by the OS thread from block to block (a grid of one-thread-per-pixel kernels
// runs tens of thousands of blocks)
struct StackPool {
	std::vector<char *> stacks;
	~StackPool()
	{
		for (char *p : stacks)
			free(p);
	}
};
thread_local StackPool g_stacks;

void trampoline()
{
	(*g_fn)();
	(*g_fibers)[g_tid].done = true;
	swapcontext(&(*g_fibers)[g_tid].ctx, &g_main);
}

} // namespace

int current_tid() { return g_tid; }

unsigned long long clock_ticks() { return g_ticks; }

void barrier()
{
	Fiber &f = (*g_fibers)[g_tid];
	swapcontext(&f.ctx, &g_main);
}

const WaveData &wave_share(const void *payload, int bytes, bool pred)
{
	Fiber &f = (*g_fibers)[g_tid];
	f.voting = true;
	f.pred = pred;
	if (bytes > 0)
		memcpy(f.payload, payload, (size_t) bytes);
	swapcontext(&f.ctx, &g_main);
	return *f.shared;
}

unsigned long long ballot(bool pred) { return wave_share(nullptr, 0, pred).mask; }

void set_lds_base(unsigned char *base, size_t bytes)
{
	g_lds = base;
	g_lds_bytes = bytes;
	// A block starts on zeroed LDS: kernels may read LDS nobody wrote (lanes whose result is thrown away)
	// and turn it into a table index through a cast that saturates on the device and is undefined here.
	if (base && bytes)
		memset(base, 0, bytes);
}
size_t lds_bytes() { return g_lds_bytes; }
unsigned char *lds_base() { return g_lds; }

unsigned int exchange(unsigned int value, int src)
{
	(*g_exchange)[g_tid] = value;
	barrier();
	const unsigned int got = (*g_exchange)[src];
	barrier();
	return got;
}

void run_block(int threads, const std::function<void()> &fn)
{
	std::vector<unsigned int> xchg(threads, 0u);
	g_exchange = &xchg;
	const size_t stack_bytes = 256 * 1024;
	std::vector<Fiber> fibers(threads);
	g_fibers = &fibers;
	g_fn = &fn;
	for (int t = 0; t < threads; t++) {
		Fiber &f = fibers[t];
		if ((int) g_stacks.stacks.size() <= t)
			g_stacks.stacks.push_back((char *) malloc(stack_bytes));
		f.stack = g_stacks.stacks[t];
		getcontext(&f.ctx);
		f.ctx.uc_stack.ss_sp = f.stack;
		f.ctx.uc_stack.ss_size = stack_bytes;
		f.ctx.uc_link = &g_main;
		makecontext(&f.ctx, trampoline, 0);
	}
	std::vector<WaveData> waves((threads + 63) / 64);
	unsigned int round = 0;
	for (;;) {
		int alive = 0, finished = 0;
		// a different order every round: forwards, backwards, odd-even
		for (int k = 0; k < threads; k++) {
			int t = k;
			if (round % 3 == 1)
				t = threads - 1 - k;
			else if (round % 3 == 2 && threads % 2 == 0)
				t = k < threads / 2 ? 2 * k + 1 : 2 * (k - threads / 2); // odd lanes first
			Fiber &f = fibers[t];
			if (f.done) {
				finished++;
				continue;
			}
			g_tid = t;
			swapcontext(&g_main, &f.ctx);
			alive++;
		}
		// votes: every fiber has now run to its next barrier, its end, or a ballot; the voters of a wave get
		// their mask and run on (to the next ballot or the barrier) until nobody waits for a vote
		for (;;) {
			bool any = false;
			for (int w0 = 0; w0 < threads; w0 += 64) {
				const int w1 = w0 + 64 < threads ? w0 + 64 : threads;
				WaveData &wd = waves[w0 / 64];
				std::vector<int> voters;
				for (int t = w0; t < w1; t++)
					if (fibers[t].voting)
						voters.push_back(t);
				if (voters.empty())
					continue;
				wd.voters = wd.mask = 0;
				for (int t : voters) {
					wd.voters |= 1ull << (t - w0);
					if (fibers[t].pred)
						wd.mask |= 1ull << (t - w0);
					memcpy(wd.data[t - w0], fibers[t].payload, 16);
					fibers[t].voting = false;
					fibers[t].shared = &wd;
				}
				for (int t : voters) {
					any = true;
					g_tid = t;
					swapcontext(&g_main, &fibers[t].ctx);
				}
			}
			if (!any)
				break;
		}
		g_ticks += 997; // (the 100 MHz clock moves between barriers)
		round++;
		if (!alive)
			break;
		if (round > 100000000u) {
			fprintf(stderr, "emul: block does not terminate\n");
			abort();
		}
	}
	g_fibers = nullptr;
}

} // namespace emul
