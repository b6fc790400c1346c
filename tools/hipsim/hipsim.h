// hipsim.h -- a tiny CPU "SIMT" simulator for the HIP subset used by localexpstereo_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  It lets the kernel and launch logic of les_hip.hip / les_kernels.h be
// executed on the build container (which has no GPU) and compared against the oracle.  It is never
// linked into, loaded by, or a fallback for the product library liblocalexp_hip.so; the product path
// fails loudly without a HIP device (include/localexp_hip.h).
//
// Model: every workgroup runs on one OS thread; its work-items are ucontext fibers scheduled
// round-robin; __syncthreads() and the quad exchange are cooperative yield points.  Workgroups are
// distributed over OS threads with OpenMP.  `__shared__` becomes `static thread_local`.
#pragma once

#include <ucontext.h>

#include <algorithm>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

static inline const char* hipGetErrorString(hipError_t) { return "hipsim error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = nullptr; return posix_memalign(p, 256, n ? n : 1) == 0 ? hipSuccess : hipErrorUnknown; }     // (device allocations are 256-byte aligned)
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)1 << 40; return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t i = 0; i < h; i++) memcpy((char*)d + i * dp, (const char*)s + i * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }

using std::max;
using std::min;
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipsim {

struct Block {
    int nthreads = 0, current = 0, alive = 0;
    std::vector<ucontext_t> ctx;      // (architectures without the hand-written switch of hipsim.cpp)
    std::vector<void*> sp;            // x86-64: saved stack pointer of every work-item ...
    void* sched_sp = nullptr;         // ... and of the scheduler
    std::vector<char> done;
    ucontext_t sched;
    char* stack_base = nullptr;       // work-item stacks (mmap, grown on demand, never zero-filled)
    size_t stack_bytes = 0;
    unsigned bar_count = 0, bar_gen = 0;
    std::vector<unsigned> quad_count, quad_gen;
    std::vector<unsigned> g16_count, g16_gen;
    std::vector<unsigned> g64_count, g64_gen;
    std::vector<uint64_t> exch;
    std::function<void()> body;
};
extern thread_local Block* g_block;

void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_block();
void quad_sync();
void group16_sync();
void group_sync(int log2size);     // 4: DPP row of 16 work-items, 6: wave of 64

template <typename T>
inline void quad_allgather(T v, T out[4])
{
    static_assert(sizeof(T) <= 8, "quad exchange of <= 8 byte values");
    Block* B = g_block;
    const int tid = B->current;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    B->exch[tid] = bits;
    quad_sync();
    for (int j = 0; j < 4; j++) memcpy(&out[j], &B->exch[(tid & ~3) + j], sizeof(T));
    quad_sync();
}

// value of v in work-item (lane ^ mask) of this wave of 64 (every work-item of the wave must call it)
template <typename T>
inline T wave_xor(T v, int mask)
{
    static_assert(sizeof(T) <= 8, "wave exchange of <= 8 byte values");
    Block* B = g_block;
    const int tid = B->current;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    B->exch[tid] = bits;
    group_sync(6);
    T r;
    memcpy(&r, &B->exch[((tid & ~63) | ((tid & 63) ^ mask))], sizeof(T));
    group_sync(6);
    return r;
}

// exchange among the 16 work-items of a DPP row (lanes 16r .. 16r+15): every one of them must call it
template <typename T>
inline void group16_allgather(T v, T out[16])
{
    static_assert(sizeof(T) <= 8, "row exchange of <= 8 byte values");
    Block* B = g_block;
    const int tid = B->current;
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    B->exch[tid] = bits;
    group16_sync();
    for (int j = 0; j < 16; j++) memcpy(&out[j], &B->exch[(tid & ~15) + j], sizeof(T));
    group16_sync();
}

// value of v in work-item `l` of this wave of 64 (every work-item of the wave must call it)
inline int wave_readlane(int v, int l)
{
    Block* B = g_block;
    const int tid = B->current;
    B->exch[tid] = (uint64_t)(uint32_t)v;
    group_sync(6);
    const int r = (int)(uint32_t)B->exch[(tid & ~63) + l];
    group_sync(6);
    return r;
}

}  // namespace hipsim

static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned atomicMax(unsigned* p, unsigned v)
{
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

static inline void __syncthreads() { hipsim::sync_block(); }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipsim::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
