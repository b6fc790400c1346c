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

#endif

}  // namespace les
