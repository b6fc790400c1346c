// hipsim.cpp -- see hipsim.h.  TEST INFRASTRUCTURE ONLY.
#include "hipsim.h"

#include <omp.h>
#include <sys/mman.h>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipsim {

thread_local Block* g_block = nullptr;
static thread_local Block t_block;
static const size_t kStack = 128 * 1024;

// Fiber switch.  glibc's swapcontext saves and restores the signal mask with a system call per switch (measured: 18 of the 57 CPU-minutes of the CPU test
// suite were kernel time); the work-items of a workgroup never touch signal masks, so on x86-64 the switch is the callee-saved registers, the
// SSE / x87 control words and the stack pointer -- nothing else survives a function call in the SysV ABI.
#if defined(__x86_64__)
extern "C" void hipsim_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipsim_switch
    .type hipsim_switch,@function
hipsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipsim_switch,.-hipsim_switch
)");
static void fiber_entry();
static void fiber_trampoline()
{
    fiber_entry();
    Block* B = g_block;
    void* dead;
    hipsim_switch(&dead, B->sched_sp);      // a finished work-item is never resumed
    abort();
}
// a fresh stack on which hipsim_switch "returns" into fiber_trampoline
static void* fiber_prepare(char* stack, size_t size)
{
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    uint64_t* sp = reinterpret_cast<uint64_t*>(top - 8);      // (top - 8) = 8 mod 16: what a callee sees right after a call
    *--sp = (uint64_t)(uintptr_t)&fiber_trampoline;            // return address
    for (int i = 0; i < 6; i++) *--sp = 0;                     // rbp rbx r12 r13 r14 r15
    --sp;
    uint32_t cw[2] = {0, 0};
    asm volatile("stmxcsr %0" : "=m"(cw[0]));
    uint16_t fcw;
    asm volatile("fnstcw %0" : "=m"(fcw));
    cw[1] = fcw;
    memcpy(sp, cw, 8);
    return sp;
}
static inline void switch_to_sched(Block* B) { hipsim_switch(&B->sp[B->current], B->sched_sp); }
static inline void switch_to_fiber(Block* B, int i) { hipsim_switch(&B->sched_sp, B->sp[i]); }
#else
static inline void switch_to_sched(Block* B) { swapcontext(&B->ctx[B->current], &B->sched); }
static inline void switch_to_fiber(Block* B, int i) { swapcontext(&B->sched, &B->ctx[i]); }
#endif

static void yield_to_sched()
{
    switch_to_sched(g_block);
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
#if defined(__x86_64__)
        B->sp.resize(n);
#else
        B->ctx.resize(n);
#endif
        B->done.resize(n);
        // stacks: mapped lazily and only ever grown (a std::vector would zero-fill 128 MB whenever a launch changes the block size; a work-item touches a few KB)
        if ((size_t)n * kStack > B->stack_bytes) {
            if (B->stack_base) munmap(B->stack_base, B->stack_bytes);
            B->stack_bytes = (size_t)n * kStack;
            void* m = mmap(nullptr, B->stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (m == MAP_FAILED) { perror("hipsim: mmap of the work-item stacks"); abort(); }
            B->stack_base = static_cast<char*>(m);
        }
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
#if defined(__x86_64__)
        B->sp[i] = fiber_prepare(B->stack_base + (size_t)i * kStack, kStack);
#else
        getcontext(&B->ctx[i]);
        B->ctx[i].uc_stack.ss_sp = B->stack_base + (size_t)i * kStack;
        B->ctx[i].uc_stack.ss_size = kStack;
        B->ctx[i].uc_link = &B->sched;
        makecontext(&B->ctx[i], fiber_entry, 0);
#endif
    }
    while (B->alive > 0) {
        for (int i = 0; i < n; i++) {
            if (B->done[i]) continue;
            B->current = i;
            threadIdx = dim3((unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y));
            switch_to_fiber(B, i);
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
