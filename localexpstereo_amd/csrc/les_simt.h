// les_simt.h -- the (small) set of SIMT primitives the kernels use.
//
// Product build: hipcc --offload-arch=gfx950, wave64, quad exchanges lower to DPP quad_perm moves.
// LES_SIM build: tools/hipsim (CPU fiber simulator, TEST INFRASTRUCTURE ONLY) provides the same
// names so the kernel logic can be checked against the oracle without a GPU.  The product library
// is never built with LES_SIM.
#pragma once

#if defined(LES_SIM)
#include "hipsim.h"
#else
#include <hip/hip_runtime.h>
#endif

#include <stdint.h>

namespace les {

#if defined(LES_SIM)

template <typename T>
__device__ inline void quad_allgather(T v, T out[4]) { hipsim::quad_allgather(v, out); }
template <int K, typename T>
__device__ inline T quad_bcast(T v)
{
    T o[4];
    hipsim::quad_allgather(v, o);
    return o[K];
}
// sum over the 4 lanes of the quad, as the butterfly (l ^ 1) then (l ^ 2) evaluates it
template <typename T>
__device__ inline T quad_sum(T v)
{
    T o[4];
    hipsim::quad_allgather(v, o);
    const int l = hipsim::g_block->current & 3;
    T a = o[l] + o[l ^ 1], b = o[l ^ 2] + o[l ^ 3];
    return a + b;
}

// ---- primitives of the march kernel (les_march.h)
struct alignas(16) int4 { int x, y, z, w; };
#define LES_MARCH_SCHED_FENCE() ((void)0)
__device__ inline int readfirstlane_i32(int v) { return v; }
// floor(x + 0.5), saturating (v_cvt_rpi_i32_f32)
__device__ inline int cvt_rpi_i32(float x)
{
    if (!(x == x)) return 0;
    const float f = floorf(x + 0.5f);
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)0x80000000;
    return (int)f;
}
// DPP row_shr:N within rows of 16 lanes, lanes shifted in read 0 (every lane of the row must call it)
template <int N>
__device__ inline int dpp_row_shr(int v)
{
    int o[16];
    hipsim::group16_allgather(v, o);
    const int l = hipsim::g_block->current & 15;
    return l >= N ? o[l - N] : 0;
}
__device__ inline void wave_sync() { hipsim::group_sync(6); }
// v_min3_f32 / v_max3_f32: IEEE minNum / maxNum of three (a NaN operand is ignored unless all are NaN)
__device__ inline float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ inline float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// value of `v` in lane `l` of this wave (every lane of the wave must call it)
__device__ inline int readlane_i32(int v, int l) { return hipsim::wave_readlane(v, l); }

#else

// dpp_ctrl for quad_perm:[k,k,k,k]
template <int K>
__device__ __forceinline__ int quad_bcast_i32(int v)
{
    return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, true);
}
template <int K>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(quad_bcast_i32<K>(__float_as_int(v)));
}
template <int K>
__device__ __forceinline__ double quad_bcast(double v)
{
    int lo = quad_bcast_i32<K>(__double2loint(v));
    int hi = quad_bcast_i32<K>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int quad_perm(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v)
{
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum over the 4 lanes of the quad: butterfly quad_perm:[1,0,3,2] then [2,3,0,1]
template <typename T>
__device__ __forceinline__ T quad_sum(T v)
{
    v = v + quad_perm<0xB1>(v);
    v = v + quad_perm<0x4E>(v);
    return v;
}
// out[j] = value held by lane j of this lane's quad (lanes 4q..4q+3 of the wave)
template <typename T>
__device__ __forceinline__ void quad_allgather(T v, T out[4])
{
    out[0] = quad_bcast<0>(v);
    out[1] = quad_bcast<1>(v);
    out[2] = quad_bcast<2>(v);
    out[3] = quad_bcast<3>(v);
}

// ---- primitives of the march kernel (les_march.h)
// the row groups are fully unrolled; without a fence the scheduler hoists the LDS reads of every group to the top of the block
// and the register footprint is set by that alone
#define LES_MARCH_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ int readfirstlane_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
// floor(x + 0.5) in one instruction (saturating): round-half-up keeps the quantisation of a, b unbiased
__device__ __forceinline__ int cvt_rpi_i32(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
template <int N>
__device__ __forceinline__ int dpp_row_shr(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xf, 0xf, true);       // row_shr:N, lanes shifted in read 0
}
// LDS traffic of one wave is processed in order; the fence only keeps the compiler from moving accesses across it
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int readlane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }    // folds to v_min3_f32
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

#endif

__device__ __forceinline__ float readlane_f32(float v, int l) { return __int_as_float(readlane_i32(__float_as_int(v), l)); }

}  // namespace les
