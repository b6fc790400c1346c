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

// sum over the 64 lanes of the wave as the butterfly (l ^ 1), (l ^ 2), ... (l ^ 32) evaluates it: a fixed balanced tree, every lane gets the same value
template <typename T>
__device__ inline T wave_sum_tree(T v)
{
    for (int m = 1; m < 64; m <<= 1) v = v + hipsim::wave_xor(v, m);
    return v;
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
// round-4 primitives of the march kernel: signed byte B of a packed word times a 24-bit integer (v_mul_i32_i24 with an SDWA byte
// operand), float of signed byte B (v_cvt_f32_i32 with an SDWA byte operand), min without NaN handling (finite operands only)
template <int B>
__device__ inline int mul24_sbyte(uint32_t g, int v) { return (((int)(g << (24 - 8 * B))) >> 24) * v; }
template <int B>
__device__ inline float cvt_f32_sbyte(uint32_t g) { return (float)(((int)(g << (24 - 8 * B))) >> 24); }
__device__ inline float min_f32_finite(float a, float b) { return b < a ? b : a; }
// median of three (v_med3_f32): clamps x to [lo, hi] for lo <= hi; a NaN x gives lo here (the callers exclude NaN)
__device__ inline float med3_f32(float x, float lo, float hi) { return !(x > lo) ? lo : (x > hi ? hi : x); }
__device__ inline void acc64_add_i32(long long& acc, int d) { acc += (long long)d; }
// raw buffer access (march kernel, round 4): base + per-lane byte offset + wave-uniform byte offset
struct BufRsrc { const char* base; uint32_t bytes; };
__device__ inline BufRsrc make_buf(const void* p, uint32_t bytes) { return BufRsrc{(const char*)p, bytes}; }
template <class T>
__device__ inline T buf_load(const BufRsrc& r, uint32_t voff, uint32_t soff)
{
    T v = T(0);                                                     // the descriptor's range check: out-of-range offsets read 0
    const unsigned long long o = (unsigned long long)voff + (unsigned long long)soff;
    if (o + sizeof(T) <= r.bytes) memcpy(&v, r.base + o, sizeof(T));
    return v;
}
// low 24 bits of a times low 24 bits of b, plus c (v_mad_u32_u24)
__device__ inline uint32_t mad_u24(int a, int b, int c) { return ((uint32_t)a & 0xffffffu) * ((uint32_t)b & 0xffffffu) + (uint32_t)c; }
template <class T>
__device__ inline void buf_store(const BufRsrc& r, uint32_t voff, uint32_t soff, T v) { memcpy(const_cast<char*>(r.base) + (size_t)voff + (size_t)soff, &v, sizeof(T)); }
// bit i = flag of lane i, for the first n lanes of the wave (every lane of the wave must call it)
__device__ inline uint32_t ballot_low(bool f, int n)
{
    uint32_t m = 0;
    for (int i = 0; i < n; i++) m |= (uint32_t)(hipsim::wave_readlane(f ? 1 : 0, i) & 1) << i;
    return m;
}
// sign-extended bit `bit` of a wave-uniform word: all ones or zero
__device__ inline int sbfe1(uint32_t w, int bit) { return -(int)((w >> bit) & 1u); }
// one 16-byte LDS store
__device__ inline void lds_store4(int4* p, int x, int y, int z, int w) { *p = int4{x, y, z, w}; }

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

// sum over the 64 lanes of the wave as the butterfly (l ^ 1), (l ^ 2), ... (l ^ 32) evaluates it: a fixed balanced tree, every lane gets the same value
// (IEEE addition is commutative, so both partners of an exchange compute the same sum)
__device__ __forceinline__ int wave_sum_tree(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_tree(double v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}

// ---- primitives of the march kernel (les_march.h)
// the row groups are fully unrolled; without a fence the scheduler hoists the LDS reads of every group to the top of the block
// and the register footprint is set by that alone
#define LES_MARCH_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ int readfirstlane_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
// floor(x + 0.5) in one instruction (saturating): round-half-up keeps the quantisation of a, b unbiased
#if defined(LES_SIMT_PLAIN)
// -DLES_SIMT_PLAIN (the check build the GPU tests compare bit for bit with the product, tests/test_gpu_parity.py): the arithmetic primitives below
// as plain C++ -- the definitions the simulator uses -- instead of inline assembly
__device__ __forceinline__ int cvt_rpi_i32(float x)
{
    if (!(x == x)) return 0;
    const float f = floorf(x + 0.5f);
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)0x80000000;
    return (int)f;
}
#else
__device__ __forceinline__ int cvt_rpi_i32(float x)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
#endif
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
// round-4 primitives of the march kernel (tools/ubench/valu_rates.hip: an SDWA byte operand costs nothing on top of the operation,
// against a separate v_bfe_i32 at 4.2 cycles):
//   mul24_sbyte<B>(g, v)   sign-extended byte B of g times the low 24 bits of v        (v_mul_i32_i24_sdwa)
//   cvt_f32_sbyte<B>(g)    float of the sign-extended byte B of g                       (v_cvt_f32_i32_sdwa)
//   min_f32_finite(a, b)   one v_min_f32 (fminf() adds a canonicalising v_max_f32 in IEEE mode; the operands here are finite)
#if defined(LES_SIMT_PLAIN)
template <int B>
__device__ __forceinline__ int mul24_sbyte(uint32_t g, int v) { return (((int)(g << (24 - 8 * B))) >> 24) * v; }
template <int B>
__device__ __forceinline__ float cvt_f32_sbyte(uint32_t g) { return (float)(((int)(g << (24 - 8 * B))) >> 24); }
__device__ __forceinline__ float min_f32_finite(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float med3_f32(float x, float lo, float hi) { return !(x > lo) ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ void acc64_add_i32(long long& acc, int d) { acc += (long long)d; }
#else
template <int B>
__device__ __forceinline__ int mul24_sbyte(uint32_t g, int v)
{
    int r;
    static_assert(B >= 0 && B < 3, "guide byte");
    if constexpr (B == 0) asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(g), "v"(v));
    else if constexpr (B == 1) asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(g), "v"(v));
    else asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(g), "v"(v));
    return r;
}
template <int B>
__device__ __forceinline__ float cvt_f32_sbyte(uint32_t g)
{
    float r;
    static_assert(B >= 0 && B < 3, "guide byte");
    if constexpr (B == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(g));
    else if constexpr (B == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(r) : "v"(g));
    else asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(g));
    return r;
}
__device__ __forceinline__ float min_f32_finite(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float med3_f32(float x, float lo, float hi)
{
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}
// acc += sign-extended d in ONE instruction (v_mad_i64_i32 with the inline constant 1; the compiler's own lowering of a 64-bit add
// of a sign-extended word is v_ashrrev + v_add_co + v_addc)
__device__ __forceinline__ void acc64_add_i32(long long& acc, int d)
{
    asm("v_mad_i64_i32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(d) : "vcc");
}
#endif
// Raw buffer access (march kernel, round 4): address = descriptor base + per-lane byte offset (VGPR) + wave-uniform byte offset (SGPR),
// i.e. a row of an image costs NO scalar address arithmetic (the global_load `saddr` form needs a 64-bit scalar add per row and
// access); out-of-range offsets read 0.  num_records is in bytes (stride 0).
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_buf(const void* p, uint32_t bytes)
{
    // The descriptor must be PROVABLY wave-uniform (base and size go through v_readfirstlane): a descriptor the compiler takes for
    // divergent -- e.g. a base computed from a job field -- makes every access a "waterfall" loop (4 x v_readfirstlane, two 64-bit
    // compares, s_and_saveexec, the access, a branch back).  All callers pass wave-uniform values.
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    void* q = (void*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00027000);
}
template <class T>
__device__ __forceinline__ T buf_load(BufRsrc r, uint32_t voff, uint32_t soff)
{
    static_assert(sizeof(T) == 4, "dword access");
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
template <class T>
__device__ __forceinline__ void buf_store(BufRsrc r, uint32_t voff, uint32_t soff, T v)
{
    static_assert(sizeof(T) == 4, "dword access");
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ uint32_t mad_u24(int a, int b, int c) { return (uint32_t)__umul24((unsigned)a, (unsigned)b) + (uint32_t)c; }    // folds to v_mad_u32_u24
// bit i = flag of lane i (one v_cmp into a scalar pair; the row flags of a block are then tested with scalar bit operations)
__device__ __forceinline__ uint32_t ballot_low(bool f, int) { return (uint32_t)__builtin_amdgcn_ballot_w64(f); }
// sign-extended bit `bit` of a wave-uniform word: all ones or zero (s_bfe_i32)
__device__ __forceinline__ int sbfe1(uint32_t w, int bit) { return __builtin_amdgcn_sbfe((int)w, (unsigned)bit, 1u); }
// ONE ds_write_b128.  (An int4 struct store is four scalar stores to the optimiser, which the load / store merger re-pairs as it sees fit:
// under register pressure role A's row stores came out as two ds_write2_b32 each -- twice the LDS instructions, with bank conflicts.)
__device__ __forceinline__ void lds_store4(int4* p, int x, int y, int z, int w)
{
    typedef int v4i __attribute__((ext_vector_type(4)));
    *reinterpret_cast<v4i*>(p) = v4i{x, y, z, w};
}
__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }    // folds to v_min3_f32
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

#endif

__device__ __forceinline__ float readlane_f32(float v, int l) { return __int_as_float(readlane_i32(__float_as_int(v), l)); }

}  // namespace les
