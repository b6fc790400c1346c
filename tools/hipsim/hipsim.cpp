// hipsim.cpp -- see hipsim.h.  TEST INFRASTRUCTURE ONLY.
#include "hipsim.h"

#include <omp.h>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipsim {

thread_local Block* g_block = nullptr;
static thread_local Block t_block;
static const size_t kStack = 128 * 1024;

static void yield_to_sched()
{
    Block* B = g_block;
    swapcontext(&B->ctx[B->current], &B->sched);
}

void sync_block()
{
    Block* B = g_block;
    const unsigned g = B->bar_gen;
    if (++B->bar_count >= (unsigned)B->alive) {
        B->bar_count = 0;
        B->bar_gen++;
        return;
    }
    while (B->bar_gen == g) yield_to_sched();
}

void quad_sync()
{
    Block* B = g_block;
    const int q = B->current >> 2;
    const unsigned g = B->quad_gen[q];
    if (++B->quad_count[q] == 4) {
        B->quad_count[q] = 0;
        B->quad_gen[q]++;
        return;
    }
    while (B->quad_gen[q] == g) yield_to_sched();
}

void group16_sync()
{
    Block* B = g_block;
    const int q = B->current >> 4;
    const unsigned g = B->g16_gen[q];
    if (++B->g16_count[q] == 16) {
        B->g16_count[q] = 0;
        B->g16_gen[q]++;
        return;
    }
    while (B->g16_gen[q] == g) yield_to_sched();
}

void group_sync(int log2size)
{
    if (log2size == 4) { group16_sync(); return; }
    Block* B = g_block;
    const int q = B->current >> 6;
    const int members = std::min(64, B->nthreads - q * 64);
    const unsigned g = B->g64_gen[q];
    if (++B->g64_count[q] == (unsigned)members) {
        B->g64_count[q] = 0;
        B->g64_gen[q]++;
        return;
    }
    while (B->g64_gen[q] == g) yield_to_sched();
}

static void fiber_entry()
{
    Block* B = g_block;
    B->body();
    B->done[B->current] = 1;
    B->alive--;
    // a thread that exits releases a barrier the remaining threads are all waiting at
    if (B->alive > 0 && B->bar_count >= (unsigned)B->alive) {
        B->bar_count = 0;
        B->bar_gen++;
    }
    // returning resumes uc_link (the scheduler)
}

static void run_block(const std::function<void()>& body, dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz)
{
    Block* B = &t_block;
    g_block = B;
    const int n = (int)(block.x * block.y * block.z);
    if (B->nthreads != n) {
        B->nthreads = n;
        B->ctx.resize(n);
        B->done.resize(n);
        B->stacks.resize((size_t)n * kStack);
        B->quad_count.resize((n + 3) / 4);
        B->quad_gen.resize((n + 3) / 4);
        B->exch.resize(n);
        B->g16_count.resize((n + 15) / 16);
        B->g16_gen.resize((n + 15) / 16);
        B->g64_count.resize((n + 63) / 64);
        B->g64_gen.resize((n + 63) / 64);
    }
    B->body = body;
    B->alive = n;
    B->bar_count = 0;
    B->bar_gen = 0;
    std::fill(B->quad_count.begin(), B->quad_count.end(), 0u);
    std::fill(B->quad_gen.begin(), B->quad_gen.end(), 0u);
    std::fill(B->g16_count.begin(), B->g16_count.end(), 0u);
    std::fill(B->g16_gen.begin(), B->g16_gen.end(), 0u);
    std::fill(B->g64_count.begin(), B->g64_count.end(), 0u);
    std::fill(B->g64_gen.begin(), B->g64_gen.end(), 0u);
    std::fill(B->done.begin(), B->done.end(), 0);
    gridDim = grid;
    blockDim = block;
    blockIdx = dim3(bx, by, bz);
    for (int i = 0; i < n; i++) {
        getcontext(&B->ctx[i]);
        B->ctx[i].uc_stack.ss_sp = &B->stacks[(size_t)i * kStack];
        B->ctx[i].uc_stack.ss_size = kStack;
        B->ctx[i].uc_link = &B->sched;
        makecontext(&B->ctx[i], fiber_entry, 0);
    }
    while (B->alive > 0) {
        for (int i = 0; i < n; i++) {
            if (B->done[i]) continue;
            B->current = i;
            threadIdx = dim3((unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y));
            swapcontext(&B->sched, &B->ctx[i]);
        }
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
    const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(dynamic, 1)
    for (long b = 0; b < nblocks; b++)
        run_block(body, grid, block, (unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
}

}  // namespace hipsim
