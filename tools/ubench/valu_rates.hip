// valu_rates.hip -- issue-rate micro-benchmark for the instruction classes the strip kernel is built from (gfx950).
// Every kernel runs ITER x 32 independent instructions of one kind per wave; 4 waves per SIMD are resident
// (grid = 256 CUs x 4 workgroups of 256 threads).  Reported: shader cycles per wave-instruction per SIMD,
// derived from the s_memtime span of a wave divided by the instructions all co-resident waves of its SIMD issued.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define ITER 2000
typedef float f4v __attribute__((ext_vector_type(4)));

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// 32-bit register chains: v[0..7] accumulators, independent
#define KERNEL32(NAME, ASM)                                                                         \
    __global__ void __launch_bounds__(256) NAME(unsigned long long* out, float seed)                \
    {                                                                                               \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 0.5f, c = seed * 0.25f;                                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int i = 0; i < ITER; i++) {                                                            \
            asm volatile(REP8(ASM(%0) ASM(%1) ASM(%2) ASM(%3)) REP8(ASM(%4) ASM(%5) ASM(%6) ASM(%7)) \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s10", "s11", "s12"); \
        }                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;            \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[0] = 0;                          \
    }

#define KERNEL64(NAME, ASM)                                                                         \
    __global__ void __launch_bounds__(256) NAME(unsigned long long* out, float seedf)               \
    {                                                                                               \
        double seed = seedf;                                                                        \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        double b = seed * 0.5, c = seed * 0.25;                                                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int i = 0; i < ITER; i++) {                                                            \
            asm volatile(REP8(ASM(%0) ASM(%1) ASM(%2) ASM(%3)) REP8(ASM(%4) ASM(%5) ASM(%6) ASM(%7)) \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;            \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456) out[0] = 0;                           \
    }

// mixed: 64-bit destination from a 32-bit source (and the other way round)
#define KERNEL_CVT(NAME, ASM, DT, ST)                                                               \
    __global__ void __launch_bounds__(256) NAME(unsigned long long* out, float seedf)               \
    {                                                                                               \
        DT a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                          \
        ST s0 = (ST)seedf, s1 = (ST)(seedf + 1), s2 = (ST)(seedf + 2), s3 = (ST)(seedf + 3);        \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int i = 0; i < ITER; i++) {                                                            \
            asm volatile(REP8(ASM(%0, %8) ASM(%1, %9) ASM(%2, %10) ASM(%3, %11)) REP8(ASM(%4, %8) ASM(%5, %9) ASM(%6, %10) ASM(%7, %11)) \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0), "v"(s1), "v"(s2), "v"(s3)); \
        }                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;            \
        if ((double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 == 123.456) out[0] = 0; \
    }

#define A_ADD_F32(r) "v_add_f32 " #r ", " #r ", %8\n"
#define A_FMA_F32(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define A_MUL_F32(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define A_ADD_U32(r) "v_add_u32 " #r ", " #r ", %8\n"
#define A_MAD_U24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9\n"
#define A_MUL_LO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n"
#define A_BFE(r) "v_bfe_i32 " #r ", " #r ", 3, 10\n"
#define A_CVT_F32_I32(r) "v_cvt_f32_i32 " #r ", " #r "\n"
#define A_DPP(r) "v_mov_b32_dpp " #r ", " #r " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define A_ADD_DPP(r) "v_add_f32_dpp " #r ", " #r ", %8 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define A_MIN_F32(r) "v_min_f32 " #r ", " #r ", %8\n"
#define A_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define A_CNDMASK64(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, s[10:11]\n"
#define A_CMP_CND(r) "v_cmp_lt_f32 vcc, " #r ", %8\n v_cndmask_b32 " #r ", " #r ", %9, vcc\n"
#define A_MOV_B32(r) "v_mov_b32 " #r ", %8\n"
#define A_AND_B32(r) "v_and_b32 " #r ", " #r ", %8\n"
#define A_SUB_U32(r) "v_sub_u32 " #r ", " #r ", %8\n"
#define A_ASHR(r) "v_ashrrev_i32 " #r ", 3, " #r "\n"
#define A_LSHL_ADD_U32(r) "v_lshl_add_u32 " #r ", " #r ", 2, %8\n"
#define A_ALIGNBIT(r) "v_alignbit_b32 " #r ", " #r ", %8, 9\n"
#define A_CVT_RPI(r) "v_cvt_rpi_i32_f32 " #r ", " #r "\n"
#define A_CVT_I32_F32(r) "v_cvt_i32_f32 " #r ", " #r "\n"
#define A_FMAC_F32(r) "v_fmac_f32 " #r ", %8, %9\n"
#define A_MIN3(r) "v_min3_f32 " #r ", " #r ", %8, %9\n"
#define A_CMP_ONLY(r) "v_cmp_lt_f32 vcc, " #r ", %8\n"
#define A_READLANE(r) "v_readlane_b32 s12, " #r ", 3\n"

#define A_ADD_F64(r) "v_add_f64 " #r ", " #r ", %8\n"
#define A_FMA_F64(r) "v_fma_f64 " #r ", " #r ", %8, %9\n"
#define A_MUL_F64(r) "v_mul_f64 " #r ", " #r ", %8\n"
#define A_PK_FMA_F32(r) "v_pk_fma_f32 " #r ", " #r ", %8, %9\n"
#define A_PK_ADD_F32(r) "v_pk_add_f32 " #r ", " #r ", %8\n"
#define A_PK_MUL_F32(r) "v_pk_mul_f32 " #r ", " #r ", %8\n"
#define A_LSHL_ADD_U64(r) "v_lshl_add_u64 " #r ", " #r ", 0, %8\n"
#define A_MOV_B64(r) "v_mov_b64 " #r ", %8\n"

#define A_CVT_F64_F32(d, s) "v_cvt_f64_f32 " #d ", " #s "\n"
#define A_CVT_F64_I32(d, s) "v_cvt_f64_i32 " #d ", " #s "\n"
#define A_CVT_F32_F64(d, s) "v_cvt_f32_f64 " #d ", " #s "\n"
#define A_MAD_U64_U32(d, s) "v_mad_u64_u32 " #d ", vcc, " #s ", " #s ", " #d "\n"
#define A_MAD_I64_I32(d, s) "v_mad_i64_i32 " #d ", vcc, " #s ", " #s ", " #d "\n"

KERNEL32(k_add_f32, A_ADD_F32)
KERNEL32(k_fma_f32, A_FMA_F32)
KERNEL32(k_mul_f32, A_MUL_F32)
KERNEL32(k_add_u32, A_ADD_U32)
KERNEL32(k_mad_u24, A_MAD_U24)
KERNEL32(k_mul_lo, A_MUL_LO)
KERNEL32(k_bfe, A_BFE)
KERNEL32(k_cvt_f32_i32, A_CVT_F32_I32)
KERNEL32(k_dpp, A_DPP)
KERNEL32(k_add_dpp, A_ADD_DPP)
KERNEL32(k_min_f32, A_MIN_F32)
KERNEL32(k_cndmask, A_CNDMASK)
KERNEL32(k_cndmask64, A_CNDMASK64)
KERNEL32(k_cmp_cnd, A_CMP_CND)
KERNEL32(k_mov_b32, A_MOV_B32)
KERNEL32(k_and_b32, A_AND_B32)
KERNEL32(k_sub_u32, A_SUB_U32)
KERNEL32(k_ashr, A_ASHR)
KERNEL32(k_lshl_add_u32, A_LSHL_ADD_U32)
KERNEL32(k_alignbit, A_ALIGNBIT)
KERNEL32(k_cvt_rpi, A_CVT_RPI)
KERNEL32(k_cvt_i32_f32, A_CVT_I32_F32)
KERNEL32(k_fmac_f32, A_FMAC_F32)
KERNEL32(k_min3, A_MIN3)
KERNEL32(k_cmp_only, A_CMP_ONLY)
KERNEL32(k_readlane, A_READLANE)
KERNEL64(k_add_f64, A_ADD_F64)
KERNEL64(k_fma_f64, A_FMA_F64)
KERNEL64(k_mul_f64, A_MUL_F64)
KERNEL64(k_pk_fma_f32, A_PK_FMA_F32)
KERNEL64(k_pk_add_f32, A_PK_ADD_F32)
KERNEL64(k_pk_mul_f32, A_PK_MUL_F32)
KERNEL64(k_lshl_add_u64, A_LSHL_ADD_U64)
KERNEL64(k_mov_b64, A_MOV_B64)
KERNEL_CVT(k_cvt_f64_f32, A_CVT_F64_F32, double, float)
KERNEL_CVT(k_cvt_f64_i32, A_CVT_F64_I32, double, int)
KERNEL_CVT(k_cvt_f32_f64, A_CVT_F32_F64, float, double)
KERNEL_CVT(k_mad_u64_u32, A_MAD_U64_U32, unsigned long long, unsigned)
KERNEL_CVT(k_mad_i64_i32, A_MAD_I64_I32, long long, int)


// ---- round 4: candidates for the instruction diet of the march kernel
#define A_MUL_I24_SDWA(r) "v_mul_i32_i24_sdwa " #r ", sext(" #r "), %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define A_ADD_U32_SDWA(r) "v_add_u32_sdwa " #r ", sext(" #r "), %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
#define A_MAD_I24(r) "v_mad_i32_i24 " #r ", " #r ", %8, %9\n"
#define A_MUL_I24(r) "v_mul_i32_i24 " #r ", " #r ", %8\n"
#define A_ADD3_U32(r) "v_add3_u32 " #r ", " #r ", %8, %9\n"
#define A_PERM_B32(r) "v_perm_b32 " #r ", " #r ", %8, %9\n"
#define A_CVT_F32_UBYTE1(r) "v_cvt_f32_ubyte1 " #r ", " #r "\n"
#define A_RNDNE_F32(r) "v_rndne_f32 " #r ", " #r "\n"
#define A_FLOOR_F32(r) "v_floor_f32 " #r ", " #r "\n"
#define A_FRACT_F32(r) "v_fract_f32 " #r ", " #r "\n"
#define A_MUL_HI_I32(r) "v_mul_hi_i32 " #r ", " #r ", %8\n"
#define A_MUL_HI_U24(r) "v_mul_hi_u32_u24 " #r ", " #r ", %8\n"
#define A_BFE_U32(r) "v_bfe_u32 " #r ", " #r ", 8, 8\n"
#define A_LSHRREV(r) "v_lshrrev_b32 " #r ", 8, " #r "\n"
#define A_MAX_F32(r) "v_max_f32 " #r ", " #r ", %8\n"
#define A_MED3_F32(r) "v_med3_f32 " #r ", " #r ", %8, %9\n"
#define A_MIN_I32(r) "v_min_i32 " #r ", " #r ", %8\n"
#define A_AND_OR(r) "v_and_or_b32 " #r ", " #r ", %8, %9\n"
#define A_XAD(r) "v_xad_u32 " #r ", " #r ", %8, %9\n"
#define A_LSHL_OR(r) "v_lshl_or_b32 " #r ", " #r ", 3, %8\n"
#define A_ADD_CO(r) "v_add_co_u32 " #r ", vcc, " #r ", %8\n"
#define A_ADDC_CO(r) "v_addc_co_u32 " #r ", vcc, " #r ", %8, vcc\n"
#define A_DOT4_I8(r) "v_dot4_i32_i8 " #r ", " #r ", %8, %9\n"
#define A_DOT2_I16(r) "v_dot2_i32_i16 " #r ", " #r ", %8, %9\n"
#define A_PK_ADD_U16(r) "v_pk_add_u16 " #r ", " #r ", %8\n"
#define A_PK_MAD_I16(r) "v_pk_mad_i16 " #r ", " #r ", %8, %9\n"
#define A_CVT_F32_I32_SDWA(r) "v_cvt_f32_i32_sdwa " #r ", sext(" #r ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
#define A_CVT_FLR(r) "v_cvt_flr_i32_f32 " #r ", " #r "\n"
#define A_LDEXP(r) "v_ldexp_f32 " #r ", " #r ", 3\n"
KERNEL32(k_mul_i24_sdwa, A_MUL_I24_SDWA)
KERNEL32(k_add_u32_sdwa, A_ADD_U32_SDWA)
KERNEL32(k_mad_i24, A_MAD_I24)
KERNEL32(k_mul_i24, A_MUL_I24)
KERNEL32(k_add3_u32, A_ADD3_U32)
KERNEL32(k_perm_b32, A_PERM_B32)
KERNEL32(k_cvt_f32_ubyte1, A_CVT_F32_UBYTE1)
KERNEL32(k_rndne_f32, A_RNDNE_F32)
KERNEL32(k_floor_f32, A_FLOOR_F32)
KERNEL32(k_fract_f32, A_FRACT_F32)
KERNEL32(k_mul_hi_i32, A_MUL_HI_I32)
KERNEL32(k_mul_hi_u24, A_MUL_HI_U24)
KERNEL32(k_bfe_u32, A_BFE_U32)
KERNEL32(k_lshrrev, A_LSHRREV)
KERNEL32(k_max_f32, A_MAX_F32)
KERNEL32(k_med3_f32, A_MED3_F32)
KERNEL32(k_min_i32, A_MIN_I32)
KERNEL32(k_and_or, A_AND_OR)
KERNEL32(k_xad, A_XAD)
KERNEL32(k_lshl_or, A_LSHL_OR)
KERNEL32(k_add_co, A_ADD_CO)
KERNEL32(k_addc_co, A_ADDC_CO)
KERNEL32(k_dot4_i8, A_DOT4_I8)
KERNEL32(k_dot2_i16, A_DOT2_I16)
KERNEL32(k_pk_add_u16, A_PK_ADD_U16)
KERNEL32(k_pk_mad_i16, A_PK_MAD_I16)
KERNEL32(k_cvt_f32_i32_sdwa, A_CVT_F32_I32_SDWA)
KERNEL32(k_cvt_flr, A_CVT_FLR)
KERNEL32(k_ldexp, A_LDEXP)

// ---- LDS: 32 reads (or writes) per iteration, conflict-free lane-linear addresses
#define KERNEL_LDS(NAME, BODY, BYTES)                                                               \
    __global__ void __launch_bounds__(256) NAME(unsigned long long* out, float seed)                \
    {                                                                                               \
        __shared__ __attribute__((aligned(16))) unsigned char lds[256 * BYTES * 2];                  \
        unsigned addr = threadIdx.x * BYTES;   /* the only LDS object: offset 0 */                                 \
        for (int i = threadIdx.x; i < 256 * BYTES * 2 / 4; i += 256) ((float*)lds)[i] = seed;       \
        __syncthreads();                                                                            \
        float acc = 0;                                                                              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int i = 0; i < ITER; i++) { BODY }                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;            \
        if (acc == 123.456f) out[0] = 0;                                                            \
    }

#define LDS_R32 { float r0, r1, r2, r3; asm volatile(REP8("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4\n ds_read_b32 %3, %4 offset:1024\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory"); acc += r0 + r1 + r2 + r3; }
#define LDS_R64 { double r0, r1, r2, r3; asm volatile(REP8("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:2048\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4 offset:2048\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory"); acc += (float)(r0 + r1 + r2 + r3); }
#define LDS_R128 { f4v r0, r1, r2, r3; asm volatile(REP8("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4 offset:4096\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory"); acc += r0.x + r1.y + r2.z + r3.w; }
#define LDS_W32 { float v = seed; asm volatile(REP8("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:1024\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v) : "memory"); }
#define LDS_W64 { double v = seed; asm volatile(REP8("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:2048\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v) : "memory"); }
#define LDS_W128 { f4v v = {seed, seed, seed, seed}; asm volatile(REP8("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:4096\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v) : "memory"); }

KERNEL_LDS(k_lds_r32, LDS_R32, 4)
KERNEL_LDS(k_lds_r64, LDS_R64, 8)
KERNEL_LDS(k_lds_r128, LDS_R128, 16)
KERNEL_LDS(k_lds_w32, LDS_W32, 4)
KERNEL_LDS(k_lds_w64, LDS_W64, 8)
KERNEL_LDS(k_lds_w128, LDS_W128, 16)

// ---- co-issue: fp64 adds in one half of the waves, LDS reads in the other half (do the pipes overlap?)
__global__ void __launch_bounds__(256) k_mix_f64_lds(unsigned long long* out, float seed)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 8 * 2];
    unsigned addr = threadIdx.x * 8;
    for (int i = threadIdx.x; i < 256 * 8 * 2 / 4; i += 256) ((float*)lds)[i] = seed;
    __syncthreads();
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5;
    float acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; i++) {
        double r0, r1, r2, r3;
        // 16 LDS b64 reads + 16 fp64 adds per iteration, interleaved in every wave
        asm volatile(REP8("ds_read_b64 %4, %8\n v_add_f64 %0, %0, %9\n ds_read_b64 %5, %8 offset:2048\n v_add_f64 %1, %1, %9\n")
                     REP8("ds_read_b64 %6, %8\n v_add_f64 %2, %2, %9\n ds_read_b64 %7, %8 offset:2048\n v_add_f64 %3, %3, %9\n") "s_waitcnt lgkmcnt(0)\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr), "v"(b) : "memory");
        acc += (float)(r0 + r1 + r2 + r3);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc + (float)(a0 + a1 + a2 + a3) == 123.456f) out[0] = 0;
}

typedef void (*Kern)(unsigned long long*, float);
struct Entry { const char* name; Kern fn; int per_iter; };

int main(int argc, char** argv)
{
    int wg_per_cu = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const int grid = ncu * wg_per_cu;
    unsigned long long* d_out;
    (void)hipMalloc(&d_out, sizeof(unsigned long long) * grid * 4);
    std::vector<unsigned long long> h(grid * 4);
    Entry es[] = {
        {"v_add_f32", k_add_f32, 64}, {"v_fma_f32", k_fma_f32, 64}, {"v_mul_f32", k_mul_f32, 64}, {"v_min_f32", k_min_f32, 64},
        {"v_cndmask_b32 (vcc)", k_cndmask, 64}, {"v_cndmask_b32_e64 (sgpr mask)", k_cndmask64, 64}, {"v_cmp_lt_f32 + v_cndmask (pair)", k_cmp_cnd, 64},
        {"v_cmp_lt_f32 vcc", k_cmp_only, 64}, {"v_mov_b32", k_mov_b32, 64}, {"v_and_b32", k_and_b32, 64}, {"v_sub_u32", k_sub_u32, 64}, {"v_ashrrev_i32", k_ashr, 64},
        {"v_lshl_add_u32", k_lshl_add_u32, 64}, {"v_alignbit_b32", k_alignbit, 64}, {"v_cvt_rpi_i32_f32", k_cvt_rpi, 64}, {"v_cvt_i32_f32", k_cvt_i32_f32, 64},
        {"v_fmac_f32", k_fmac_f32, 64}, {"v_min3_f32", k_min3, 64}, {"v_readlane_b32", k_readlane, 64},
        {"v_add_u32", k_add_u32, 64}, {"v_mad_u32_u24", k_mad_u24, 64}, {"v_mul_lo_u32", k_mul_lo, 64}, {"v_bfe_i32", k_bfe, 64},
        {"v_cvt_f32_i32", k_cvt_f32_i32, 64}, {"v_mov_b32_dpp", k_dpp, 64}, {"v_add_f32_dpp", k_add_dpp, 64},
        {"v_add_f64", k_add_f64, 64}, {"v_fma_f64", k_fma_f64, 64}, {"v_mul_f64", k_mul_f64, 64},
        {"v_pk_fma_f32", k_pk_fma_f32, 64}, {"v_pk_add_f32", k_pk_add_f32, 64}, {"v_pk_mul_f32", k_pk_mul_f32, 64},
        {"v_lshl_add_u64", k_lshl_add_u64, 64}, {"v_mov_b64", k_mov_b64, 64},
        {"v_cvt_f64_f32", k_cvt_f64_f32, 64}, {"v_cvt_f64_i32", k_cvt_f64_i32, 64}, {"v_cvt_f32_f64", k_cvt_f32_f64, 64},
        {"v_mad_u64_u32", k_mad_u64_u32, 64}, {"v_mad_i64_i32", k_mad_i64_i32, 64},
        {"v_mul_i32_i24_sdwa (sext byte)", k_mul_i24_sdwa, 64}, {"v_add_u32_sdwa (sext byte)", k_add_u32_sdwa, 64}, {"v_mad_i32_i24", k_mad_i24, 64}, {"v_mul_i32_i24", k_mul_i24, 64},
        {"v_add3_u32", k_add3_u32, 64}, {"v_perm_b32", k_perm_b32, 64}, {"v_cvt_f32_ubyte1", k_cvt_f32_ubyte1, 64}, {"v_rndne_f32", k_rndne_f32, 64}, {"v_floor_f32", k_floor_f32, 64},
        {"v_fract_f32", k_fract_f32, 64}, {"v_mul_hi_i32", k_mul_hi_i32, 64}, {"v_mul_hi_u32_u24", k_mul_hi_u24, 64}, {"v_bfe_u32", k_bfe_u32, 64}, {"v_lshrrev_b32", k_lshrrev, 64},
        {"v_max_f32", k_max_f32, 64}, {"v_med3_f32", k_med3_f32, 64}, {"v_min_i32", k_min_i32, 64}, {"v_and_or_b32", k_and_or, 64}, {"v_xad_u32", k_xad, 64}, {"v_lshl_or_b32", k_lshl_or, 64},
        {"v_add_co_u32", k_add_co, 64}, {"v_addc_co_u32", k_addc_co, 64}, {"v_dot4_i32_i8", k_dot4_i8, 64}, {"v_dot2_i32_i16", k_dot2_i16, 64}, {"v_pk_add_u16", k_pk_add_u16, 64},
        {"v_pk_mad_i16", k_pk_mad_i16, 64}, {"v_cvt_f32_i32_sdwa (sext byte)", k_cvt_f32_i32_sdwa, 64}, {"v_cvt_flr_i32_f32", k_cvt_flr, 64}, {"v_ldexp_f32", k_ldexp, 64},
        {"ds_read_b32", k_lds_r32, 32}, {"ds_read_b64", k_lds_r64, 32}, {"ds_read_b128", k_lds_r128, 32},
        {"ds_write_b32", k_lds_w32, 32}, {"ds_write_b64", k_lds_w64, 32}, {"ds_write_b128", k_lds_w128, 32},
        {"mix 16 ds_read_b64 + 16 v_add_f64", k_mix_f64_lds, 32},
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("device %s, %d CUs, clock %d kHz, %d workgroups (x256 threads) per CU\n", prop.name, ncu, prop.clockRate, wg_per_cu);
    printf("%-36s %10s %14s %12s\n", "instruction", "ms", "cyc/inst/SIMD", "(eff. GHz)");
    for (auto& e : es) {
        hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, d_out, 1.0f);     // warm-up
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, d_out, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double span = sum / h.size();                                   // s_memtime ticks of a wave (100 MHz constant clock on gfx9)
        const double insts_per_simd = (double)ITER * e.per_iter * wg_per_cu;  // one wave of each workgroup per SIMD
        // wall-clock based: cycles at nominal 2.4 GHz
        const double cyc_wall = ms * 1e-3 * 2.4e9 / insts_per_simd;
        printf("%-36s %10.3f %14.2f   memtime span %.0f ticks\n", e.name, ms, cyc_wall, span);
    }
    return 0;
}
