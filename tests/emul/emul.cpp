// TEST INFRASTRUCTURE: see emul.h.
#include "emul.h"

#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace emul {

namespace {

struct Fiber {
	ucontext_t ctx;
	char *stack = nullptr;
	bool done = false;
};

thread_local std::vector<Fiber> *g_fibers = nullptr;
thread_local ucontext_t g_main;
thread_local int g_tid = 0;
thread_local const std::function<void()> *g_fn = nullptr;
thread_local unsigned long long g_ticks = 0;
thread_local std::vector<unsigned int> *g_exchange = nullptr;

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
		f.stack = (char *) malloc(stack_bytes);
		getcontext(&f.ctx);
		f.ctx.uc_stack.ss_sp = f.stack;
		f.ctx.uc_stack.ss_size = stack_bytes;
		f.ctx.uc_link = &g_main;
		makecontext(&f.ctx, trampoline, 0);
	}
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
		g_ticks += 997; // (the 100 MHz clock moves between barriers)
		round++;
		if (!alive)
			break;
		if (round > 100000000u) {
			fprintf(stderr, "emul: block does not terminate\n");
			abort();
		}
	}
	for (Fiber &f : fibers)
		free(f.stack);
	g_fibers = nullptr;
}

} // namespace emul
