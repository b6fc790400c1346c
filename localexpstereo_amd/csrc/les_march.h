// les_march.h -- the fixed-point "column march" kernel: gather + guided-filter aggregation with exact integer box sums
// (gfx950 / CDNA4, wave64).  It computes the same quantity as les_strip_kernel (les_kernels.h):
//
//     p(s)  = min( lerp_d vol[a*x+b*y+c][y][x], th_col )                       LES/CostVolumeEnergy.h:70-98
//     q     = guided filter of p with the colour guide, radius R, eps           LES/GuidedFilter.h:142-266, 301-326
//     q    -> 1e6 where the label is invalid                                    LES/CostVolumeEnergy.h:176-183
//
// but is organised around what the MI355X micro-benchmarks say is cheap (tools/ubench/valu_rates.hip, profiles/round4_valu_rates_w3.log):
// only fp32 add / mul / fma, 32-bit integer add / sub, logic, shifts and moves issue at ~2.9 cycles per wave-instruction with the
// three waves a SIMD holds here; EVERY other VALU instruction -- fp64, conversions, min / max, selects, 24- and 32-bit integer
// multiplies, three-operand integer adds, DPP, bit-field extracts -- takes ~4.2, an SDWA byte operand is free on top of that, an
// LDS ds_write_b128 costs ~13.7 CU cycles and a ds_read_b128 4.1, a wave issues at most one instruction per ~4 cycles, a taken
// branch costs ~26, and the arbiter prefers the oldest wave.  So the kernel minimises instruction COUNT: all four box-filter passes
// are EXACT INTEGER sums (round 4: 32-bit everywhere except the four stage-2 vertical sums of role D, which are exact 64-bit integers by one
// v_mad_i64_i32 each -- acc64_add_i32; no fp64 anywhere in the loops), computed by three wave-specialised
// roles (A, C, D below; the prefix passes B1 / B2 run on the waves of role A) that work on consecutive row blocks at the same time:
//
//   A  (lane = image column of a job, marching down the rows; NJ jobs per workgroup)
//        p -> centred fixed point  pi = rint(p * sp) + c0  in [-2^(PB-1), 2^(PB-1)]   (PB = 20: one v_fma_f32 onto 1.5 * 2^23 + c0, the
//        integer is the low mantissa of the result; p lives in [vmin, th_col], vmin = min of the volume)
//        vertical 2R+1 running sums over a register ring of (-pi, packed guide):  Sp = sum pi            (|Sp| < 2^24)
//                                                                                Sc = sum Iq_c * pi     (Iq = u8 - 128; |Sc| < 2^30.4:
//        v_mul_i32_i24 with the guide byte as an SDWA operand, v_add3_u32; int32, exact)
//        -> LDS T1[row][col] = {Sp, (Sc + 2^4) >> 5}                 (the only rounding of stage 1: 2^-26 of full scale)
//   B1 (lane = (row, 8-column segment)) in-place prefix sums along x, modulo 2^32 (the 2R+1-column differences are exact)
//   C  (lane = column)  box sums s, t_c = P(x+R) - P(x-R-1);  N cov_c = t_c + hi32(M_c * 8 s)  (M_c = MINUS mean_c in 2^-24 u8 units:
//        one v_mad_i64_i32, the cancellation is exact);  a = inv * cov,  b = mean_p - a . mean   in fp32 (as les_strip_kernel);
//        a, b -> int32 with a scale derived from a rigorous bound on |a|, |b|  -> LDS T2
//   B2 prefix sums of T2
//   D  (lane = column)  horizontal box = prefix difference (int32, < 2^30), vertical running sums over an int32 register ring in
//        64-bit integers (one v_mad_i64_i32 each), rounded once to 2^4:  q = (Sb * 255 + sum_c Sa_c * Iq_c) * 2^4 / (255 scale N) + p(0)  in fp32.
//
// Error budget against the double-precision reference (tools/fixedpoint_probe.py, DESIGN.md "Numerics"): the fixed-point cost
// (2.4e-7 of the range per pixel, unbiased, before a 441-pixel average), the 2^-26 rounding of the stage-1 vertical sums, M_c
// (2^-25 u8), the fp32 3x3 algebra (shared with les_strip_kernel), the stage-2 quantisation (resolution ~1e-6 of |a|max before a
// 441-pixel average), the 2^4 rounding of the stage-2 window sums (2^-30 of their full scale) and the fp32 combination.
//
// The kernel is only launched when the host has established its preconditions (les_hip.hip: build_march_view per context, build_march_jobs per batch): a finite volume,
// th_col - vmin <= 8 |th_col|, and every target at least 2R away from clip borders that are not image borders (so that every
// consumed stage-1 window is a true covariance window and the bound on |a| holds).  Everything else runs les_strip_kernel.
#pragma once

#include "les_kernels.h"

// Measurement switches (role masks, role order, ablations, cache policies, per-role clocks) live in les_march_lab.h and exist only
// in -DLES_MARCH_LAB builds; here every hook is a no-op.
#if defined(LES_MARCH_LAB)
#include "les_march_lab.h"
#else
#define LES_MARCH_ROLE_ORDER 5
#define LES_STATS_POLICY ""
#define LES_LAB_ROLE_ON(bit) true
#define LES_LAB_ABLATE(bit) false
#endif
#ifndef LES_TICK_BEGIN
#define LES_TICK_BEGIN() ((void)0)
#define LES_TICK_MARK() ((void)0)
#define LES_TICK_BARRIER() __syncthreads()
#define LES_TICK_END(role_) ((void)0)
#endif

namespace les {

constexpr int kMarchSH = 5;       // right shift of the vertical sums of Iq * pi before the horizontal pass
constexpr int kMarchMB = 24;      // fraction bits of M_c (centred guide mean in u8 units): |M_c| <= 2^31
constexpr int kMarchSL = 3;       // left shift of the window sum s before the product with M_c:  SH + MB + SL = 32
constexpr int kMarchS2 = 4;       // right shift of the stage-2 horizontal box sums before the vertical pass
constexpr int kMarchMagicBits = 0x4B400000;     // float bits of 1.5 * 2^23: fma(p, sp, 1.5 * 2^23 + c0) has the integer rint(p sp) + c0 in its low mantissa
static_assert(kMarchSH + kMarchMB + kMarchSL == 32, "N cov = t - hi32(M * (s << SL)) needs the scales to meet at 2^32");
// Bits of the centred fixed-point cost for radius R: the stage-1 window sum s << SL and the vertical sums of Iq * pi must fit int32
// ((2R+1)^2 2^(PB-1) 2^SL < 2^31 and (2R+1) 2^7 2^(PB-1) < 2^31).
__host__ __device__ constexpr int march_pb(int R) { return (2 * R + 1) * (2 * R + 1) < 512 ? 20 : 19; }
// words per pixel of the guide statistics: 9 = {inv00 inv01 inv02 inv11} {inv12 inv22 M0 M1} {M2} (36 bytes: two 16-byte loads and one
// 4-byte load per pixel and hypothesis; the float means are derived from M_c), 12 = the round-3 record with mu_c stored as well
#ifndef LES_MARCH_STAT_WORDS
#define LES_MARCH_STAT_WORDS 9
#endif
constexpr int kMarchStatWords = LES_MARCH_STAT_WORDS;
static_assert(kMarchStatWords == 9 || kMarchStatWords == 12, "statistics record of 36 or 48 bytes");

struct MarchView {
    const float* vol;             // [D][H][W]
    const uint32_t* ipk8;         // [H*W] guide pixel as three signed bytes u8 - 128 (byte 3 = 0)
    const float* mstats;          // [H*W][kMarchStatWords]: {inv00, inv01, inv02, inv11} {inv12, inv22, M0, M1} {M2 [, mu0, mu1, mu2]}
                                  //   inv = (Sigma + eps U)^-1 of the guide in [0,1] units (LES/GuidedFilter.h:87-101),
                                  //   M_c = -rint((255 mean_I_c - 128) 2^MB) (int32 bits: MINUS the centred mean), mu_c = mean_I_c - 128/255 = -M_c 2^-MB / 255
    float sp;                     // counts per cost unit: (2^PB - 1) / (th_col - vmin)
    float pmagic;                 // 1.5 * 2^23 + c0,  c0 = rint(-vmin sp) - 2^(PB-1)  (an integer: exact in fp32)
    float poff;                   // the cost that count 0 stands for: -c0 / sp
    float kapS;                   // 2^SH * u_p / 255 * scale        (u_p = 1 / sp, scale = stage-2 counts per unit of a, b)
    float upS;                    // u_p * scale
    float qscale;                 // 2^S2 / (255 * scale)
    float kmu;                    // 2^-MB / 255
    // image-based energy (les_hip_create_naive): vol is the raw-cost scratch les_naive_raw_kernel has just filled, raw_off[i] the
    // float offset of call i's filterRect patch in it (row stride = filterRect width), vmin = 0, th_col -> th_color + th_grad.
    // Null for a cost-volume context.
    const long long* raw_off;
    // the same volume once more in a TILED layout, [H][ceil(W/8)][D][8] (8 columns x all slices contiguous: 32 bytes per slice), or null:
    // what the taps of a slanted plane read (role A's KIND 5).  In [D][H][W] the two taps of 8 neighbouring pixels lie in 8|a| + 2 different slices,
    // i.e. in as many 128-byte lines of which 4 .. 32 bytes are used; here they lie within (8|a| + 2) x 32 contiguous bytes.
    const float* vol_t;
};
// planes with |a| (disparity change per column) at or above this take their taps from the tiled copy: whole-image slabs and wide cells (one job
// per workgroup) / the two-job geometry of the small cells
#ifndef LES_TILED_MIN_SLOPE_WIDE
#define LES_TILED_MIN_SLOPE_WIDE 0.05f
#endif
#ifndef LES_TILED_MIN_SLOPE
#define LES_TILED_MIN_SLOPE 0.125f
#endif

template <int R, int WGC, int NJ, int BY>
struct MarchCfg {
    static constexpr int KS = 2 * R + 1;
    static constexpr int RS = 3 * BY;                      // ring length: three blocks (the tick loop is unrolled by 3, so ring slots and stage-2 buffers are compile-time); >= KS
    static constexpr int TW = WGC - 4 * R;                 // output columns per job
    static constexpr int HWV = WGC / 64;                   // waves per role and job slot
    static constexpr int NW = 3 * HWV * NJ;                // waves per workgroup: roles A, C, D
    static constexpr int NT = 64 * NW;
    static constexpr int SEGL = 8;                         // columns per prefix segment
    // Physical columns: a leading zero element (P(-1)) + one pad element per PADW columns; with PADW = 16 the row length is rounded
    // up to 4 mod 8 elements.  The prefix pass reads, per 16 lanes, the 8 segments (stride 8 columns = 32 banks, shifted by one
    // element = 4 banks per pad) of two rows (shifted by +-16 banks when the row length is 4 mod 8): all 64 banks once.  The
    // consumers' 16 consecutive columns then straddle one pad (one bank collision per 16 lanes) instead of two.
    static constexpr int PADW = 16;
    static constexpr int PCOLS0 = 1 + WGC + WGC / PADW;
    static constexpr int PCOLS = PCOLS0 + ((4 - PCOLS0 % 8) + 8) % 8;      // 4 or 12 mod 16: the second row lands 16 banks off either way
    static_assert(RS >= KS, "the ring must hold a whole window (the row that leaves it is the one written KS rows ago)");
    static_assert(RS - KS < BY, "a ring longer than the window by a block or more wastes registers: pick the smallest block height");
    static_assert(WGC % 64 == 0, "a job slot is a whole number of waves");
    static_assert(BY * (64 / SEGL) <= 64, "one wave prefixes its own tile");
    static_assert(TW > 0, "job too narrow for this radius");
    static_assert(2 * R + 1 < 64, "a window crosses at most one wave boundary");
    static_assert((long long)KS * KS * (1ll << (march_pb(R) - 1 + kMarchSL)) < (1ll << 31), "stage-1 window sums (shifted for the product with M) must fit int32");
    static_assert((long long)KS * 128 * (1ll << (march_pb(R) - 1)) < (1ll << 31), "vertical sums of Iq * pi must fit int32");
    static_assert((long long)KS * (1ll << (30 - kMarchS2)) < (1ll << 31), "the stage-2 window sums, rounded to 2^S2, must fit int32");
    __host__ __device__ static constexpr int pcol(int ci) { return 1 + ci + ci / PADW; }
};

// ---------------------------------------------------------------------------------------------------
// Wave-specialised pipeline.  A workgroup owns NJ jobs; each job slot has 3 x (WGC/64) waves:
//   role A (wave = 64 columns of the job)   block k   : gather, fixed point, vertical sums -> T1[k&1], prefix of its own tile;
//                                            block k-2 : prefix of its tile of T2 (this role has the shortest row loop)
//   role C                                   block k-1 : box sums from T1, algebra -> T2[(k-1)%3]
//   role D                                   block k-3 : box sums from T2, vertical sums, output
// and ONE workgroup barrier per block ("tick").  Every role issues the global loads of its next block before it waits at
// the barrier, so memory latency is covered by the other two roles' work; role A / D keep their rings in registers, role C
// has no state and can hold the statistics words (9 per pixel: the 36-byte record) of all BY rows in flight.  The prefix sums are wave-local (each wave
// prefixes the 64 columns it wrote, no cross-wave synchronisation): a window that crosses a wave boundary adds the total of
// the left neighbour's tile (its last prefix element).
// ---------------------------------------------------------------------------------------------------
// prefix sums along x of the BY x 64 tile this wave wrote (rows i, physical columns of ci0 .. ci0+63), in place, modulo 2^32.
// A lane owns one 8-column segment of one row.  The two rows that share a DPP row of 16 lanes are INTERLEAVED (lane = 16 (row / 2)
// + 2 seg + (row & 1)), so the scan of the segment totals shifts by 2, 4, 8 lanes, never crosses from one image row into the other,
// and the zero fill of row_shr is exactly the scan boundary: three v_add_u32_dpp per component, no masks.
// (scan of the 8 elements a lane holds: local inclusive prefix, exclusive scan of the segment totals over the row's 8 lanes, offset)
__device__ __forceinline__ void march_prefix_scan8(int4 (&v)[8])
{
#pragma unroll
    for (int j = 1; j < 8; j++) { v[j].x += v[j - 1].x; v[j].y += v[j - 1].y; v[j].z += v[j - 1].z; v[j].w += v[j - 1].w; }
    int4 inc = v[7];
#if defined(LES_SIM) || defined(LES_MARCH_SCAN_PLAIN)
#define LES_SCAN_STEP(N) { inc.x += dpp_row_shr<N>(inc.x); inc.y += dpp_row_shr<N>(inc.y); inc.z += dpp_row_shr<N>(inc.z); inc.w += dpp_row_shr<N>(inc.w); }
    LES_SCAN_STEP(2) LES_SCAN_STEP(4) LES_SCAN_STEP(8)
#undef LES_SCAN_STEP
#else
    // one v_add_u32_dpp per step and component (the compiler's lowering of the line above is v_mov_b32_dpp + v_add_u32: twice the
    // instructions).  A DPP operand must have been written at least two wait states earlier: the four components are interleaved, so
    // only the first step needs the s_nop.
#define LES_SCAN_ROW(N) "v_add_u32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                        "v_add_u32_dpp %1, %1, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                        "v_add_u32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
                        "v_add_u32_dpp %3, %3, %3 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
    asm("s_nop 1\n" LES_SCAN_ROW(2) LES_SCAN_ROW(4) LES_SCAN_ROW(8) : "+v"(inc.x), "+v"(inc.y), "+v"(inc.z), "+v"(inc.w));
#undef LES_SCAN_ROW
#endif
    const int4 off = int4{inc.x - v[7].x, inc.y - v[7].y, inc.z - v[7].z, inc.w - v[7].w};
#pragma unroll
    for (int j = 0; j < 8; j++) { v[j].x += off.x; v[j].y += off.y; v[j].z += off.z; v[j].w += off.w; }
}

template <int BY, int PCOLS>
__device__ __forceinline__ void march_prefix_tile(int4 (*T)[PCOLS], int ci0, int lane)
{
    constexpr int SEGL = 8;
    const int row = 2 * (lane >> 4) + (lane & 1), seg = (lane & 15) >> 1;
    const bool act = row < BY;
    const int c0 = ci0 + seg * SEGL;                                           // first column of the segment; a segment never contains a pad
    int4* p = &T[act ? row : 0][1 + c0 + c0 / 16];                // (the lanes of the unused 8th row read row 0 and write nothing)
    int4 v[SEGL];
#pragma unroll
    for (int j = 0; j < SEGL; j++) v[j] = p[j];
    march_prefix_scan8(v);
    if (act) {
#pragma unroll
        for (int j = 0; j < SEGL; j++) lds_store4(&p[j], v[j].x, v[j].y, v[j].z, v[j].w);
    }
}

// The same for two tiles at once: TB holds data older than a workgroup barrier, TA was written by this wave just now.  The reads
// of TB are issued before the wave waits for its own stores, the reads of TA are in flight during the arithmetic of TB: one
// exposed LDS round trip instead of two (costs 32 more registers).
template <int BY, int PCOLS>
__device__ __forceinline__ void march_prefix_pair(int4 (*TA)[PCOLS], int4 (*TB)[PCOLS], int ci0, int lane)
{
    constexpr int SEGL = 8;
    const int row = 2 * (lane >> 4) + (lane & 1), seg = (lane & 15) >> 1;
    const bool act = row < BY;
    const int c0 = ci0 + seg * SEGL;
    const int pc = 1 + c0 + c0 / 16;
    int4* pa = &TA[act ? row : 0][pc];
    int4* pb = &TB[act ? row : 0][pc];
    int4 va[SEGL], vb[SEGL];
#pragma unroll
    for (int j = 0; j < SEGL; j++) vb[j] = pb[j];
    wave_sync();
#pragma unroll
    for (int j = 0; j < SEGL; j++) va[j] = pa[j];
    LES_MARCH_SCHED_FENCE();
    march_prefix_scan8(vb);
    if (act) {
#pragma unroll
        for (int j = 0; j < SEGL; j++) lds_store4(&pb[j], vb[j].x, vb[j].y, vb[j].z, vb[j].w);
    }
    LES_MARCH_SCHED_FENCE();
    march_prefix_scan8(va);
    if (act) {
#pragma unroll
        for (int j = 0; j < SEGL; j++) lds_store4(&pa[j], va[j].x, va[j].y, va[j].z, va[j].w);
    }
}

// Memory access of the roles (round 4): raw buffer loads / stores (les_simt.h: buf_load / buf_store) -- descriptor base + the lane's
// 32-bit byte offset in a row + a wave-uniform row offset that v_readlane hands over from a lane-computed table.  A row costs one
// v_readlane and no scalar address arithmetic (the global_load `saddr` form took a 64-bit scalar shift-add per row and access, and the
// flag bits packed into the row word another four scalar instructions; 22 % of the kernel's instructions were scalar).  The row
// flags of a block travel as ONE ballot of the lane table (bit i = row i) instead.
// The statistics rows of role C are loop-carried register tuples that are reloaded in place, one row at a time, while the rest of
// the block is still being consumed.  Written as plain C++ loads, the register allocator lands every reload in a fresh tuple and
// copies it to the loop-carried one right away, i.e. waits for the load it has just issued.  The loads are therefore issued as
// inline assembly with the destination TIED to the variable, and the vmcnt bookkeeping for them is done by hand (this role issues
// no other vector-memory instruction): march_stats_wait<n>(rows...) waits until at most n of this wave's loads are outstanding and,
// by naming the rows as read-write operands, keeps their uses behind the wait.  A row is three loads: 16 + 16 + 4 bytes (16 + 16 +
// 16 with the 48-byte record).
#if defined(LES_SIM)
typedef float4 mstat4;
#else
typedef float mstat4 __attribute__((ext_vector_type(4)));
#endif
#if LES_MARCH_STAT_WORDS == 9
typedef float mstat_tail;                       // {M2}
#else
typedef mstat4 mstat_tail;                      // {M2, mu0, mu1, mu2}
#endif
struct MarchStatRow { mstat4 a, b; mstat_tail c; };
#if defined(LES_SIM) || defined(LES_MARCH_STATS_PLAIN)
// Plain C++ loads (the simulator, and the -DLES_MARCH_STATS_PLAIN check build whose outputs the GPU tests compare bit for bit with the
// product's): a range-checked record read -- offsets beyond the table read 0, as the buffer descriptor does -- and no hand-kept waits.
struct MarchStatDesc { const char* base; uint32_t bytes; };
__device__ inline MarchStatDesc march_stats_desc(const float* base, uint32_t bytes) { return MarchStatDesc{(const char*)base, bytes}; }
__device__ inline void march_stats_load(MarchStatRow& r, const MarchStatDesc& d, uint32_t voff, uint32_t soff)
{
    const unsigned long long o = (unsigned long long)voff + (unsigned long long)soff;
    float w[8 + sizeof(mstat_tail) / 4];
    for (unsigned i = 0; i < sizeof w / 4; i++) w[i] = (o + 4ull * i + 4 <= d.bytes) ? *reinterpret_cast<const float*>(d.base + o + 4ull * i) : 0.0f;   // 36-byte records: dword aligned only
    memcpy(&r.a, w, 16); memcpy(&r.b, w + 4, 16); memcpy(&r.c, w + 8, sizeof r.c);
}
template <int N>
__device__ inline void march_stats_wait(MarchStatRow& r) { (void)r; }
template <int N>
__device__ inline void march_stats_wait(MarchStatRow& r, MarchStatRow& q) { (void)r; (void)q; }
#else
typedef int MarchStatDesc __attribute__((ext_vector_type(4)));      // buffer descriptor words (raw, stride 0, num_records in bytes)
__device__ __forceinline__ MarchStatDesc march_stats_desc(const float* base, uint32_t bytes)
{
    const unsigned long long a = (unsigned long long)base;
    MarchStatDesc d;
    d.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    d.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    d.z = __builtin_amdgcn_readfirstlane((int)bytes);
    d.w = 0x00027000;
    return d;
}
// soff must be an SGPR that a SCALAR instruction wrote (the caller masks the table word): a v_readlane result used by a vector-memory
// instruction within five wait states is a hazard the compiler cannot see inside inline assembly
__device__ __forceinline__ void march_stats_load(MarchStatRow& r, const MarchStatDesc& d, uint32_t voff, uint32_t soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" LES_STATS_POLICY : "+v"(r.a) : "v"(voff), "s"(d), "s"(soff));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" LES_STATS_POLICY : "+v"(r.b) : "v"(voff), "s"(d), "s"(soff));
#if LES_MARCH_STAT_WORDS == 9
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:32" LES_STATS_POLICY : "+v"(r.c) : "v"(voff), "s"(d), "s"(soff));
#else
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:32" LES_STATS_POLICY : "+v"(r.c) : "v"(voff), "s"(d), "s"(soff));
#endif
}
template <int N>
__device__ __forceinline__ void march_stats_wait(MarchStatRow& r) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r.a), "+v"(r.b), "+v"(r.c) : "n"(N)); }
template <int N>
__device__ __forceinline__ void march_stats_wait(MarchStatRow& r, MarchStatRow& q)
{
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r.a), "+v"(r.b), "+v"(r.c), "+v"(q.a), "+v"(q.b), "+v"(q.c) : "n"(N));
}
#endif
__device__ __forceinline__ int march_stat_m2(const MarchStatRow& r)
{
#if LES_MARCH_STAT_WORDS == 9
    return __float_as_int(r.c);
#else
    return __float_as_int(r.c.x);
#endif
}

// LES/StereoEnergy.h:560-610 for a whole target rectangle at once: true when the plane is certainly a valid label at every pixel of
// [x0, x1] x [y0, y1] (the five disparities ds, ds +- 5a +- 5b lie inside [mind, maxd] by a margin far above the rounding of the
// reference's float expression).  Role D then skips the per-pixel test; anything not proven runs the exact per-pixel code.
__device__ __forceinline__ bool march_label_surely_valid(const Geom& g, float4 pl, int x0, int x1, int y0, int y1)
{
    const float ax0 = pl.x * (float)x0, ax1 = pl.x * (float)x1, by0 = pl.y * (float)y0, by1 = pl.y * (float)y1;
    const float spread = 5.0f * (fabsf(pl.x) + fabsf(pl.y));
    const float lo = (fminf(ax0, ax1) + fminf(by0, by1)) + pl.z - spread;
    const float hi = (fmaxf(ax0, ax1) + fmaxf(by0, by1)) + pl.z + spread;
    const float mag = fmaxf(fabsf(ax0), fabsf(ax1)) + fmaxf(fabsf(by0), fabsf(by1)) + fabsf(pl.z) + spread;
    const float margin = 1e-5f * mag + 1e-30f;
    // (a non-finite coefficient makes lo / hi / mag non-finite or NaN and fails the comparisons; 0 * v must be 0 for the reference's ds)
    return (0.0f * pl.w == 0.0f) && mag < 1e30f && (lo - margin >= g.mind) && (hi + margin <= g.maxd);
}

// The SHORT GATHER of a general plane (role A's KIND 4): with MIN_DISPARITY = 0 and MAX_DISPARITY = D - 1 (the reference's own setting,
// LES/main.cpp:341) the three branches of LES/CostVolumeEnergy.h:78-92 are ONE expression of the clamped disparity dc = min(max(d, 0), D - 1):
//     d0 = floor(dc),  f1 = dc - d0,  C = (1 - f1) vol[d0] + f1 vol[d0 + (f1 > 0)]
// -- below the range d0 = 0, f1 = 0: C = vol[0] exactly; at or above it d0 = D - 1, f1 = 0: C = vol[D - 1] exactly; inside it the reference's
// own interpolation (int(d) = floor(d) for d >= 0); a tap beyond the last slice never happens (d0 = D - 1 only with f1 = 0).  It needs a plane with finite coefficients of moderate size (no NaN
// disparity: a NaN takes the reference's invalid branch) and taps that fit a 32-bit byte offset from one descriptor base: the whole volume
// (below 2^30 floats), or -- for larger volumes -- a plane that is TAME over the job, i.e. whose disparity stays inside the range, by a margin far
// above float rounding, at every pixel the job gathers, so that its slices span less than 2^30 bytes from the lowest one (`slice_lo`).
// Everything else (NaN / infinite planes, other disparity ranges, images of 2^24 pixels or more) takes the general per-pixel path (KIND 2).
__device__ __forceinline__ bool march_plane_short_gather(const Geom& g, float4 pl, const Job& job, int R, int* slice_lo)
{
    *slice_lo = 0;
    const unsigned long long HW = (unsigned long long)g.H * (unsigned long long)g.W;
    const bool moderate = fabsf(pl.x) < 1e30f && fabsf(pl.y) < 1e30f && fabsf(pl.z) < 1e30f;      // (false for NaN)
    if (!(moderate && g.mind == 0.0f && g.D0 == 0 && g.maxd == (float)(g.D - 1) && g.D >= 2 && HW < (1ull << 24))) return false;
    if ((unsigned long long)g.D * HW < (1ull << 30)) return true;                                   // every tap within 4 GB of slice 0
    const int x0 = max(job.tx0 - 2 * R, job.cx0), x1 = min(job.tx0 + job.tw + 2 * R, job.cx1) - 1;
    const int y0 = max(job.ty0 - 2 * R, job.cy0), y1 = min(job.ty0 + job.th + 2 * R, job.cy1) - 1;
    if (x1 < x0 || y1 < y0) return false;
    const float ax0 = pl.x * (float)x0, ax1 = pl.x * (float)x1, by0 = pl.y * (float)y0, by1 = pl.y * (float)y1;
    const float lo = (fminf(ax0, ax1) + fminf(by0, by1)) + pl.z, hi = (fmaxf(ax0, ax1) + fmaxf(by0, by1)) + pl.z;
    const float mag = fmaxf(fabsf(ax0), fabsf(ax1)) + fmaxf(fabsf(by0), fabsf(by1)) + fabsf(pl.z);
    const float margin = 1e-5f * mag + 1e-30f;
    if (!((lo - margin >= 0.0f) && (hi + margin < g.maxd))) return false;
    const int s0 = (int)floorf(lo - margin), s1 = (int)floorf(hi + margin) + 1;                     // lowest / highest slice touched
    *slice_lo = s0;
    return s0 >= 0 && s1 < g.D && (unsigned long long)(s1 - s0 + 1) * HW < (1ull << 28);
}

template <int R, int WGC, int NJ, int BY>
__global__ void __launch_bounds__(3 * WGC * NJ)
les_march_kernel(Geom g, MarchView view, const Job* __restrict__ jobs, const float4* __restrict__ planes,
                 float* __restrict__ out, int ngroups, int check)
{
    using Cfg = MarchCfg<R, WGC, NJ, BY>;
    constexpr int KS = Cfg::KS, RS = Cfg::RS, NT = Cfg::NT, HWV = Cfg::HWV, PCOLS = Cfg::PCOLS;
    constexpr int UN = 3;                     // blocks per ring = ticks per unrolled loop iteration = stage-2 buffers

    __shared__ int4 s_T1[2][NJ][BY][PCOLS];  // stage 1: vertical sums, then (in place) their prefix sums along x; double buffered over blocks
    __shared__ int4 s_T2[3][NJ][BY][PCOLS];  // stage 2: quantised (a_0, a_1, a_2, b), then their prefix sums; three blocks in flight (written, prefixed, consumed)
    __shared__ float s_rtab[KS + 1];         // 1/n, n = 0..2R+1 (0 for n = 0)

    // XCD-aware group order (cf. les_strip_kernel): consecutive groups (same strip, consecutive planes) share an XCD's L2
    int grp;
    {
        const int nwg = (int)gridDim.x, orig = (int)blockIdx.x;
        const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        grp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    }
    if (grp >= ngroups) return;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = readfirstlane_i32(tid >> 6);
    // Which role gets the oldest waves of a job slot.  The SIMD arbiter strictly prefers older waves (tools/ubench/mix_issue.hip),
    // so the youngest role only fills the issue slots the other two leave.  Measured on the headline workload (ms per pass, roles
    // listed oldest first): A D C 2.53 | D A C 2.58 | A C D 3.00 | C A D 3.20 | C D A 3.29 | D C A 3.40 -- the two roles that are
    // chains of LDS round trips (A, D) want the priority, the role with the most arithmetic per row (C) fills the gaps.
    constexpr int kRoleOf[6][3] = {{0, 1, 2}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {1, 0, 2}, {0, 2, 1}};
    const int slot = wave / (3 * HWV), role = kRoleOf[LES_MARCH_ROLE_ORDER][(wave % (3 * HWV)) / HWV], half = wave % HWV;
    const int ci0 = half * 64, ci = ci0 + lane;
    const Job job = jobs[grp * NJ + slot];
    int th_max = 0;
#pragma unroll
    for (int s = 0; s < NJ; s++) th_max = max(th_max, jobs[grp * NJ + s].th);
    const int nblk = (th_max + 4 * R + BY - 1) / BY;
    const int nticks = nblk + 3;
    const int Ttot = job.th > 0 ? job.th + 4 * R : 0;     // an empty slot (padding of the last group) never passes a row test
    const float4 plane = planes[job.plane_idx];
    const int cy1m = max(job.cy1 - 1, job.cy0);

    if (tid <= KS) s_rtab[tid] = tid > 0 ? (float)(1.0 / (double)tid) : 0.0f;
    // element 0 of every row is P(-1) = 0 and stays untouched; every other element is written before it is read (a block is
    // written for all its columns before the next role reads it), the pads are never read
    for (int k = tid; k < 3 * NJ * BY; k += NT) {
        if (k < 2 * NJ * BY) (&s_T1[0][0][0][0])[k * PCOLS] = int4{0, 0, 0, 0};
        (&s_T2[0][0][0][0])[k * PCOLS] = int4{0, 0, 0, 0};
    }
    __syncthreads();

    // ---- per-lane column constants
    const int gx = job.tx0 - 2 * R + ci;                                  // image column of this lane (p, stage-1 and output column alike)
    const bool col_in = gx >= job.cx0 && gx < job.cx1 && job.th > 0;
    const int sx = min(max(gx, job.cx0), max(job.cx1 - 1, job.cx0));      // clamped: addresses stay inside the image
    const uint32_t sx4 = (uint32_t)sx * 4u;                               // its byte offset in a row of floats / packed pixels
    const uint32_t rowB = (uint32_t)g.W * 4u;                             // bytes per image row of floats / packed pixels
    const uint32_t imgB = (uint32_t)g.H * rowB;                           // bytes per image plane (the host keeps images of 2^26 pixels or more on the strip kernel)
    const BufRsrc rs_guide = make_buf(view.ipk8, imgB);
    const int nx = window_count(gx, R, job.cx0, job.cx1);                 // the same count serves stage 1 and stage 2 (same column)
    // physical columns of P(x+R), P(x-R-1) and, when the window crosses a wave boundary, of the left neighbour's tile total
    const int cP = min(ci + R, WGC - 1), cM = ci - R - 1;
    const int pcP = Cfg::pcol(cP);
    const int pcM = cM >= 0 ? Cfg::pcol(cM) : 0;
    const bool cross = cM >= 0 && (cP >> 6) != (cM >> 6);
    const int pcX = cross ? Cfg::pcol((cP >> 6) * 64 - 1) : 0;           // element 0 is the constant zero
    const int pcS = Cfg::pcol(ci);

    if (role == 0 && LES_LAB_ROLE_ON(1)) {
        // ================================================= role A =================================================
        const uint32_t HWu = (uint32_t)g.H * (uint32_t)g.W;
        const float g_ax = plane.x * (float)sx;                           // a * x, LES/CostVolumeEnergy.h:76
        // fronto-parallel planes (a = b = 0): d = c for every pixel, so taps / weight / mode are per-job constants
        const bool fronto = plane.x == 0.0f && plane.y == 0.0f;
        GatherPrep gpc = gather_prepare(g, 0.0f, plane.z, 0u, HWu, true);
        if (gpc.f1 == 0.0f) gpc.i1 = gpc.i0;                              // weight 0 on a finite volume: the second tap is never needed
        int ringN[RS];                   // MINUS the fixed-point cost of the last RS p-rows of this column (the sign the row leaves the sums with)
        uint32_t ringG[RS];              // their guide pixels
#pragma unroll
        for (int k = 0; k < RS; k++) { ringN[k] = 0; ringG[k] = 0u; }
        int Sp = 0;
        int Sc[3] = {1 << (kMarchSH - 1), 1 << (kMarchSH - 1), 1 << (kMarchSH - 1)};   // rounding bias of the >> SH folded in
        const uint32_t i0s = (uint32_t)readfirstlane_i32((int)gpc.i0), i1s = (uint32_t)readfirstlane_i32((int)gpc.i1);
        const float f1s = __int_as_float(readfirstlane_i32(__float_as_int(gpc.f1)));
        const int modes = readfirstlane_i32(gpc.mode);
        const float f0s = 1.0f - f1s;
        // fronto-parallel plane with an invalid label (LES/CostVolumeEnergy.h:80-88: C = 1e6 at every pixel, so p = th_col): the
        // count is a per-job constant -- multiply the loaded cost by 0 and put the count of th_col into the addend
        const bool inv_job = fronto && modes == 2 && !view.raw_off;
        int slice_lo = 0;
        const bool tame = !fronto && !view.raw_off && march_plane_short_gather(g, plane, job, R, &slice_lo);
        const uint32_t W8 = ((uint32_t)g.W + 7u) >> 3;
        // (tiled taps: 32-bit byte offsets from the tiles of the first image row the job gathers -- round 5: the descriptor starts at that row, so the
        //  copy may be of any size (configs[4]: 12.3 GB); what has to stay below 2^32 bytes is the job's own span of rows, (th + 4R + padding) x W8 x D x 32)
        const uint32_t trowB = W8 * (uint32_t)g.D * 32u;                  // bytes of one image row of tiles
        const int trow_lo = max(job.ty0 - 2 * R, job.cy0);
        const int trow_hi = min(job.ty0 - 2 * R + (nblk + 2) * BY, job.cy1);          // one past the last row a (clamped) address can name
        const unsigned long long tspanB = (unsigned long long)max(trow_hi - trow_lo, 1) * (unsigned long long)trowB;
        const bool tiled = tame && view.vol_t && fabsf(plane.x) >= (NJ == 1 ? LES_TILED_MIN_SLOPE_WIDE : LES_TILED_MIN_SLOPE) && tspanB < 0xfffffff0ull &&
                           (unsigned long long)W8 * (unsigned long long)g.D * 32ull < (1ull << 31);
        // Columns outside the clip contribute count 0: their factor is 0 and their addend the bare 1.5 * 2^23 (per-lane constants, so
        // the column half of the clip test costs nothing per row; the row half is one v_and with a scalar mask)
        const float spj = !col_in ? 0.0f : (inv_job ? 0.0f : view.sp);
        const float pmj = !col_in ? __int_as_float(kMarchMagicBits) : (inv_job ? fmaf(g.th_col, view.sp, view.pmagic) : view.pmagic);
        // The march itself exists in four specialisations, selected once per job (a wave-uniform test per row costs the wave a
        // taken branch and splits the row code into blocks the scheduler cannot interleave):
        //   KIND 0  fronto-parallel plane, one volume tap  (integer disparity, clamped or invalid label)
        //   KIND 1  fronto-parallel plane, two taps
        //   KIND 2  general plane: taps / weight / mode per lane and row
        //   KIND 3  image-based energy: one tap into the call's raw-cost patch (any plane; no truncation, no invalid mode)
        //   KIND 4  general plane with the short gather (march_plane_short_gather): two taps of the clamped disparity per pixel, no special cases
        //   KIND 5  the same from the tiled copy of the volume (MarchView::vol_t): planes steep along x
        // For fronto-parallel planes everything but the clip test is per-job, the row bases are scalars and the loads need no
        // address arithmetic.
        auto march_a = [&](auto kind_tag) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_tag)::value;
        GatherPrep gp[BY];
        float v0[BY], v1[BY];
        float f1r[BY];                   // KIND 4, 5: interpolation weight of the second tap
        uint32_t gw[BY];
        uint32_t rowbits = 0, rowbits_nx = 0;   // bit i = p-row i of the block (in flight / being loaded) is inside the clip and the march
        // Row scalars of block b: lane i < BY computes those of p-row b*BY + i, v_readlane hands them to the wave as scalars.
        // Loads are issued one row at a time, right after the same row of the previous block has been consumed ("rolling"
        // prefetch: a whole tick of latency cover without a second set of registers).  Rows beyond the march are clamped to a
        // valid address and flagged off, so the issue needs no branch.
        int nx_rowB = 0;                 // byte offset of the (clamped) image row
        float nx_dbase = 0.0f;
        int nx_rowraw = 0;               // KIND 3: byte offset of the row in the call's patch
        int nx_rowT = 0;                 // KIND 5: byte offset of the (clamped) image row's tiles in the tiled copy
        const int fw = job.cx1 - job.cx0;
        // descriptors: the volume slice(s) of a fronto-parallel plane / the call's raw-cost patch (its column 0 = image column cx0)
        const float* v0base = view.vol + (size_t)i0s;
        if constexpr (KIND == 3) v0base = view.vol + view.raw_off[job.plane_idx] - job.cx0;
        if constexpr (KIND == 4) v0base = view.vol + (size_t)slice_lo * (size_t)HWu;          // the lowest slice the job touches: every tap lies within 2^30 bytes of it
        if constexpr (KIND == 5) v0base = view.vol_t + (size_t)trow_lo * (size_t)(trowB >> 2);      // the tiles of the first row this job gathers
        // (KIND 4: masked rows beyond the march compute addresses from a disparity outside the tame range -- the descriptor must end where
        //  the volume ends, so that whatever passes its range check is inside the allocation)
        const unsigned long long rest4 = (unsigned long long)(g.D - slice_lo) * (unsigned long long)imgB;
        const unsigned long long trestB = (unsigned long long)(g.H - trow_lo) * (unsigned long long)trowB;      // from there to the end of the copy
        const uint32_t tiledB = (uint32_t)(trestB < 0xfffffffcull ? trestB : 0xfffffffcull);   // (whatever passes the range check lies inside the allocation)
        const BufRsrc rs_v0 = make_buf(v0base, KIND == 3 ? 0xfffffffcu : (KIND == 4 ? (uint32_t)(rest4 < 0xfffffffcull ? rest4 : 0xfffffffcull) : (KIND == 5 ? tiledB : imgB)));
        // KIND 5: byte offset of (slice 0, image row 0, this lane's column) in the tiled copy, bytes per image row of tiles, the slice the range starts at
        const uint32_t tcol = (((uint32_t)sx >> 3) * (uint32_t)g.D * 8u + ((uint32_t)sx & 7u)) * 4u;
        const int dsub = g.D0 - slice_lo;                                 // KIND 4: slice index relative to the descriptor base
        const BufRsrc rs_v1 = make_buf(view.vol + (size_t)i1s, imgB);
        auto prep = [&](int b) __attribute__((always_inline)) {
            const int t = b * BY + lane;
            const int gy = job.ty0 - 2 * R + t;
            const int sy = min(max(gy, job.cy0), cy1m);
            nx_rowB = (int)((uint32_t)sy * rowB);
            rowbits_nx = ballot_low(t < Ttot && gy >= job.cy0 && gy < job.cy1, BY);
            if constexpr (KIND == 3) nx_rowraw = (sy - job.cy0) * fw * 4;
            else nx_dbase = plane.y * (float)sy + plane.z;              // b*y + c, LES/CostVolumeEnergy.h:73
            if constexpr (KIND == 5) nx_rowT = (int)((uint32_t)(sy - trow_lo) * trowB);          // (sy >= max(ty0 - 2R, cy0) = trow_lo)
        };
        auto issue_row = [&](auto itag) __attribute__((always_inline)) {
            constexpr int i = decltype(itag)::value;
            const uint32_t ro = (uint32_t)readlane_i32(nx_rowB, i);
            if (LES_LAB_ABLATE(64)) { v0[i] = (float)lane; v1[i] = 0.0f; gw[i] = (uint32_t)lane; return; }
            if constexpr (KIND == 3) {
                v0[i] = buf_load<float>(rs_v0, sx4, (uint32_t)readlane_i32(nx_rowraw, i));
            } else if constexpr (KIND < 2) {
                v0[i] = buf_load<float>(rs_v0, sx4, ro);
                if constexpr (KIND == 1) v1[i] = buf_load<float>(rs_v1, sx4, ro);
            } else if constexpr (KIND == 4) {
                // LES/CostVolumeEnergy.h:73-92 as one expression of the clamped disparity (see march_plane_short_gather): taps d0 and d0 + 1.
                // Lanes / rows outside the clip compute on the clamped pixel and are masked when the row is consumed.
                const float d = g_ax + readlane_f32(nx_dbase, i);
                const float dc = med3_f32(d, 0.0f, g.maxd);
                const float df = floorf(dc);
                f1r[i] = dc - df;
                // (the second tap coincides with the first where its weight is zero -- integer and clamped disparities, the top slice
                //  included: df = D - 1 only with f1 = 0 -- so clamped regions cost one slice of traffic, not two)
                const uint32_t e = mad_u24((int)df + dsub, (int)HWu, sx);
                v0[i] = buf_load<float>(rs_v0, e << 2, ro);
                v1[i] = buf_load<float>(rs_v0, (e << 2) + (f1r[i] > 0.0f ? imgB : 0u), ro);
            } else if constexpr (KIND == 5) {
                // the same two taps at their addresses in the tiled copy: slice d0 of this pixel's tile is 32 d0 bytes into the tile, slice d0 + 1 the next 32
                const float d = g_ax + readlane_f32(nx_dbase, i);
                const float dc = med3_f32(d, 0.0f, g.maxd);
                const float df = floorf(dc);
                f1r[i] = dc - df;
                const uint32_t e = (((uint32_t)(int)df + (uint32_t)g.D0) << 5) + tcol;
                const uint32_t rt = (uint32_t)readlane_i32(nx_rowT, i);
                v0[i] = buf_load<float>(rs_v0, e, rt);
                v1[i] = buf_load<float>(rs_v0, e + (f1r[i] > 0.0f ? 32u : 0u), rt);
            } else {
                const float d_base = readlane_f32(nx_dbase, i);
                const bool inside = col_in && ((rowbits_nx >> i) & 1u);
                gp[i] = gather_prepare(g, g_ax, d_base, (ro >> 2) + (uint32_t)sx, HWu, inside);
                if (gp[i].f1 == 0.0f) gp[i].i1 = gp[i].i0;
                v0[i] = view.vol[gp[i].i0];
                v1[i] = view.vol[gp[i].i1];
            }
            gw[i] = buf_load<uint32_t>(rs_guide, sx4, ro);
        };
        prep(0);
        rowbits = rowbits_nx;
        prep(0);
        static_for<BY>([&](auto itag) { issue_row(itag); });
        LES_TICK_BEGIN();
        // three ticks per loop iteration: the ring slot of a block's first row, (k * BY) mod RS, is then a compile-time constant
        // and the rings stay in fixed registers (a branch per block on the slot base made the allocator spill half of them)
        for (int k0 = 0; k0 < nticks; k0 += UN) {
            static_for<UN>([&](auto utag) {
                constexpr int BASE = decltype(utag)::value * BY;
                const int k = k0 + decltype(utag)::value;
                if (k < nticks) {
                    if (k < nblk) {
                        int4 (*T)[PCOLS] = s_T1[k & 1][slot];
                        prep(k + 1);
                        static_for<BY>([&](auto itag) {
                            constexpr int i = decltype(itag)::value;
                            constexpr int SLOT = BASE + i;                   // ring slot of p-row k*BY + i
                            constexpr int OLD = (SLOT + RS - KS) % RS;       // slot of the row that leaves the window (2R+1 rows ago); == SLOT when RS == KS
                            int pi;
                            if constexpr (KIND == 3) {
                                pi = (__float_as_int(fmaf(v0[i], spj, pmj)) - kMarchMagicBits) & sbfe1(rowbits, i);
                            } else if constexpr (KIND < 2) {
                                // LES/CostVolumeEnergy.h:78-96 with per-job taps: clamped / interpolated / invalid, then min(C, th_col)
                                // (one tap: the weight of the second is zero and the volume is finite, so f0 v0 + 0 v1 = v0)
                                float C = v0[i];
                                if constexpr (KIND == 1) C = f0s * v0[i] + f1s * v1[i];
                                const float p = min_f32_finite(C, g.th_col);
                                pi = (__float_as_int(fmaf(p, spj, pmj)) - kMarchMagicBits) & sbfe1(rowbits, i);
                            } else if constexpr (KIND == 4 || KIND == 5) {
                                const float f0 = 1.0f - f1r[i];
                                const float C = f0 * v0[i] + f1r[i] * v1[i];
                                const float p = min_f32_finite(C, g.th_col);
                                pi = (__float_as_int(fmaf(p, spj, pmj)) - kMarchMagicBits) & sbfe1(rowbits, i);
                            } else {
                                const float p = gather_finish(g, gp[i], v0[i], v1[i]);
                                pi = __float_as_int(fmaf(p, view.sp, view.pmagic)) - kMarchMagicBits;
                                pi = gp[i].mode == 3 ? 0 : pi;
                            }
                            const uint32_t gi = gw[i];
                            const int no = ringN[OLD];
                            const uint32_t go = ringG[OLD];
                            ringN[SLOT] = -pi;
                            ringG[SLOT] = gi;
                            Sp = Sp + pi + no;
                            Sc[0] = Sc[0] + mul24_sbyte<0>(gi, pi) + mul24_sbyte<0>(go, no);
                            Sc[1] = Sc[1] + mul24_sbyte<1>(gi, pi) + mul24_sbyte<1>(go, no);
                            Sc[2] = Sc[2] + mul24_sbyte<2>(gi, pi) + mul24_sbyte<2>(go, no);
                            lds_store4(&T[i][pcS], Sp, Sc[0] >> kMarchSH, Sc[1] >> kMarchSH, Sc[2] >> kMarchSH);
                            issue_row(itag);                     // the same row of block k + 1
                        });
                        rowbits = rowbits_nx;                    // (prep(k + 1) above replaced the table; the rows of block k used the old flags)
                        LES_TICK_MARK();
                    }
                    // The prefix sums of this block of stage 1 and of block k - 2 of stage 2 (written by the role-C wave of the same
                    // tile at the previous tick): this role has the shortest row loop, so it does the prefix work of both stages.
                    {
                        int4 (*TB)[PCOLS] = s_T2[(decltype(utag)::value + UN - 2 % UN) % UN][slot];
                        // (general planes in the two-job geometry: role C prefixes its own block one tick later instead, see there)
                        const bool da = k < nblk, db = !(NJ > 1 && KIND == 2) && k >= 2 && k < nblk + 2;
                        constexpr bool kPair = KIND != 2 && KIND != 4 && KIND != 5;       // (the general-plane marches hold two taps and a weight per row in flight: the paired pass, 32 more registers, would spill)
                        if (kPair && da && db) march_prefix_pair<BY, PCOLS>(s_T1[k & 1][slot], TB, ci0, lane);
                        else {
                            if (da) { wave_sync(); march_prefix_tile<BY, PCOLS>(s_T1[k & 1][slot], ci0, lane); }
                            if (db) march_prefix_tile<BY, PCOLS>(TB, ci0, lane);
                        }
                    }
                    LES_TICK_BARRIER();
                }
            });
        }
        LES_TICK_END(0);
        };
        if (view.raw_off) march_a(std::integral_constant<int, 3>{});
        else if (fronto && f1s == 0.0f) march_a(std::integral_constant<int, 0>{});
        else if (fronto) march_a(std::integral_constant<int, 1>{});
        else if (tiled) march_a(std::integral_constant<int, 5>{});
        else if (tame) march_a(std::integral_constant<int, 4>{});
        else march_a(std::integral_constant<int, 2>{});
    } else if (role == 1 && LES_LAB_ROLE_ON(2)) {
        // ================================================= role C =================================================
        const bool s1_col = col_in && ci >= R && ci < WGC - R;            // stage-1 column with a complete horizontal window
        int slice_lo_unused = 0;
        const bool general_plane = !view.raw_off && !(plane.x == 0.0f && plane.y == 0.0f) && !march_plane_short_gather(g, plane, job, R, &slice_lo_unused);  // (role A's KIND 2)
        // a, b are zero outside the clip and before the march is primed: the column part of that rule is folded into the lane's
        // normalisation factors, the row part into the row's 1/count_y (0 * finite = 0, and v_cvt_rpi(+-0) = 0)
        const float kap_x = s1_col ? view.kapS * s_rtab[nx] : 0.0f;
        const float up_x = s1_col ? view.upS * s_rtab[nx] : 0.0f;
        const uint32_t sxS = (uint32_t)sx * (uint32_t)(4 * kMarchStatWords);
        const uint32_t rowS = (uint32_t)g.W * (uint32_t)(4 * kMarchStatWords);      // bytes per image row of statistics records
        const MarchStatDesc ds_stats = march_stats_desc(view.mstats, (uint32_t)g.H * rowS);
        MarchStatRow st[BY];
#pragma unroll
        for (int i = 0; i < BY; i++) {
            st[i].a = st[i].b = mstat4{0.0f, 0.0f, 0.0f, 0.0f};
#if LES_MARCH_STAT_WORDS == 9
            st[i].c = 0.0f;
#else
            st[i].c = mstat4{0.0f, 0.0f, 0.0f, 0.0f};
#endif
        }
        float rny[BY];                                                    // 1 / count_y of the rows of the block in flight, 0 for rows outside the clip or before the march is primed (wave-uniform: scalar registers)
        int nx_srow = 0;                                                  // byte offset of the (clamped) statistics row
        float nx_rny = 0.0f;
        auto prep = [&](int b) __attribute__((always_inline)) {
            const int t = b * BY + lane;
            const int gy1 = job.ty0 - 3 * R + t;                          // centre of the vertical window that ends at p-row t
            const bool on = gy1 >= job.cy0 && gy1 < job.cy1 && t >= 2 * R && t < Ttot;
            nx_rny = on ? s_rtab[window_count(gy1, R, job.cy0, job.cy1)] : 0.0f;
            nx_srow = (int)((uint32_t)min(max(gy1, job.cy0), cy1m) * rowS);
        };
        auto issue_row = [&](auto itag) __attribute__((always_inline)) {      // rolling prefetch, see role A
            constexpr int i = decltype(itag)::value;
            rny[i] = readlane_f32(nx_rny, i);
            uint32_t soff = (uint32_t)readlane_i32(nx_srow, i) & 0xfffffffcu;           // (the mask makes it a scalar-written register: see march_stats_load)
            if (LES_LAB_ABLATE(4)) soff &= 0x3u;                                        // lab: every row loads the statistics of image row 0 (always cached)
            if (LES_LAB_ABLATE(1)) return;
            march_stats_load(st[i], ds_stats, sxS, soff);
        };
        prep(0);
        static_for<BY>([&](auto itag) { issue_row(itag); });
        LES_TICK_BEGIN();
        for (int k = 0; k < nticks; k++) {
            if (k >= 1 && k <= nblk) {
                const int b = k - 1;
                const int4 (*T1)[PCOLS] = s_T1[b & 1][slot];
                int4 (*T2)[PCOLS] = s_T2[b % 3][slot];
                // Rows are processed in stages of GC; the LDS reads of stage s + 1 are issued before the arithmetic of stage s (two
                // register buffers), so that only the first read of a tick waits for the LDS.  The fences keep the scheduler from
                // hoisting all reads to the top (the register footprint would be set by that alone).
                constexpr int GC = 2, NS = (BY + GC - 1) / GC;
                prep(k);
                int4 pp[2][GC], pm[2][GC], px[2][GC];
                auto lds_stage = [&](auto stag) __attribute__((always_inline)) {
                    constexpr int S = decltype(stag)::value;
#pragma unroll
                    for (int j = 0; j < GC; j++)
                        if (S * GC + j < BY) {
                            if (LES_LAB_ABLATE(2)) { pp[S & 1][j] = int4{lane, k, 2, 3}; pm[S & 1][j] = int4{3, 2, k, lane}; px[S & 1][j] = int4{0, 0, 0, 0}; }
                            else { pp[S & 1][j] = T1[S * GC + j][pcP]; pm[S & 1][j] = T1[S * GC + j][pcM]; px[S & 1][j] = T1[S * GC + j][pcX]; }
                        }
                };
                lds_stage(std::integral_constant<int, 0>{});
                static_for<NS>([&](auto gtag) {
                    constexpr int S = decltype(gtag)::value, CB = S & 1;
                    constexpr int LO = S * GC;
                    constexpr int N = (BY - LO) < GC ? (BY - LO) : GC;
                    if constexpr (S + 1 < NS) lds_stage(std::integral_constant<int, S + 1>{});
                    LES_MARCH_SCHED_FENCE();
                    // loads are in flight in issue order: the BY rows of the previous tick (or of the initial issue), then the rows
                    // of this tick's earlier stages.  Younger than the last row of this stage: 3 loads for each of the other BY - N rows.
                    static_assert(GC == 2, "vmcnt bookkeeping below is written for stages of two rows");
                    if constexpr (N == 2) march_stats_wait<3 * (BY - 2)>(st[LO], st[LO + 1]);
                    else march_stats_wait<3 * (BY - 1)>(st[LO]);
                    static_for<N>([&](auto jtag) {
                        constexpr int j = decltype(jtag)::value;
                        constexpr int i = LO + j;
                        const int s = pp[CB][j].x - pm[CB][j].x + px[CB][j].x;                         // sum of pi over the window (exact)
                        const mstat4 q0 = st[i].a, q1 = st[i].b;
                        // the record holds MINUS M_c, so that N cov_c = t_c - mean_c s is a three-operand add of the prefix difference, the
                        // neighbour tile's total and the high word of (-M_c)(8 s) + 2^31 (the product rounded at 2^-32 of its own scale);
                        // units: 2^SH (u8 * count)
                        const int M0 = __float_as_int(q1.z), M1 = __float_as_int(q1.w), M2 = march_stat_m2(st[i]);
                        const long long s8 = (long long)(s << kMarchSL);
                        const float d0 = (float)((pp[CB][j].y - pm[CB][j].y) + px[CB][j].y + (int)(((long long)M0 * s8 + (1ll << 31)) >> 32));
                        const float d1 = (float)((pp[CB][j].z - pm[CB][j].z) + px[CB][j].z + (int)(((long long)M1 * s8 + (1ll << 31)) >> 32));
                        const float d2 = (float)((pp[CB][j].w - pm[CB][j].w) + px[CB][j].w + (int)(((long long)M2 * s8 + (1ll << 31)) >> 32));
                        // LES/GuidedFilter.h:204-221 with the scale of the integer stage 2 folded into the normalisation
                        const float ka = kap_x * rny[i];
                        const float a0 = fmaf(q0.z, d2, fmaf(q0.y, d1, q0.x * d0)) * ka;     // inv00 inv01 inv02
                        const float a1 = fmaf(q1.x, d2, fmaf(q0.w, d1, q0.y * d0)) * ka;     // inv01 inv11 inv12
                        const float a2 = fmaf(q1.y, d2, fmaf(q1.x, d1, q0.z * d0)) * ka;     // inv02 inv12 inv22
                        const float mp = (float)s * (up_x * rny[i]);
#if LES_MARCH_STAT_WORDS == 9
                        // b = mean_p - a . mu with mu_c = M_c 2^-MB / 255 (the float of M_c carries the mean to 2^-24 relative, as a stored float would)
                        const float bb = fmaf(view.kmu, fmaf(a2, (float)M2, fmaf(a1, (float)M1, a0 * (float)M0)), mp);      // (-a . mu: the sign sits in M)
#else
                        const float bb = fmaf(-a2, st[i].c.w, fmaf(-a1, st[i].c.z, fmaf(-a0, st[i].c.y, mp)));
#endif
                        lds_store4(&T2[i][pcS], cvt_rpi_i32(a0), cvt_rpi_i32(a1), cvt_rpi_i32(a2), cvt_rpi_i32(bb));
                    });
                    LES_MARCH_SCHED_FENCE();
                    static_for<N>([&](auto jtag) { issue_row(std::integral_constant<int, LO + decltype(jtag)::value>{}); });   // the same rows of block k
                });
                LES_TICK_MARK();
            }
            // Two-job geometry with a general plane: role A's row loop (per-pixel taps and weights) is then the longest of the three,
            // and this role prefixes the block it wrote at the previous tick itself.  Measured (ms per pass): cell-batched optimiser
            // geometry 10.14 -> 9.49; in the one-job geometry the same switch costs 1 % (slopes <= 0.05) to 6 % (bench H2) and is off.
            if (NJ > 1 && general_plane && k >= 2 && k < nblk + 2) march_prefix_tile<BY, PCOLS>(s_T2[(k - 2) % 3][slot], ci0, lane);
            LES_TICK_BARRIER();
        }
        LES_TICK_END(1);
    } else if (role == 2 && LES_LAB_ROLE_ON(4)) {
        // ================================================= role D =================================================
        const bool out_col = ci >= 2 * R && ci < 2 * R + job.tw && job.th > 0;
        const uint32_t oc4 = (uint32_t)max(ci - 2 * R, 0) * 4u;           // byte offset of the lane's output column in a row of the output tile
        // IsValiLabel (LES/StereoEnergy.h:560-610) without branches: ds = ((x a + y b) + 1 c) + 0 v and the four corner values
        // ds +- 5a +- 5b must all lie in [MIN, MAX]  <=>  min of the five >= MIN and max <= MAX (a NaN only arises next to an
        // infinity, which fails the range test; an all-NaN set fails the comparison itself)
        const float vl_xa = (float)gx * plane.x, vl_c = 1.0f * plane.z, vl_zv = 0.0f * plane.w, vl_a5 = plane.x * 5, vl_b5 = plane.y * 5;
        const float c_lane = view.qscale * s_rtab[nx];                     // 2^S2 / (255 scale count_x)
        int ring2[4][RS];                // rounded horizontal box sums of (a_0, a_1, a_2, b) of the last RS stage-1 rows
#pragma unroll
        for (int k = 0; k < RS; k++) ring2[0][k] = ring2[1][k] = ring2[2][k] = ring2[3][k] = 0;
        // their sums over the last 2R+1 rows: exact 64-bit integers (< 2^34.4), one v_mad_i64_i32 per row and quantity (acc64_add_i32); rounded once, to 2^-S2, when a row is
        // written -- rounding every horizontal sum instead (int32 accumulators) triples the error of the full-size linearity property
        // (tools/fixedpoint_probe.py --full: 6.6e-7 against 3.0e-7; the test bound is 5e-7)
        long long S2[4] = {1ll << (kMarchS2 - 1), 1ll << (kMarchS2 - 1), 1ll << (kMarchS2 - 1), 1ll << (kMarchS2 - 1)};
        uint32_t gq[BY];
        float rny2[BY];                                                   // 1 / count_y of the block's output rows
        uint32_t okbits = 0;                                              // bit i = row i of the block in flight is an output row
        int nx_grow = 0;                                                  // byte offset of the (clamped) guide row
        float nx_rny = 0.0f;
        const BufRsrc rs_out = make_buf(out + job.out_off, 0xfffffffcu);  // the job's output tile: row r at r * out_stride floats
        const uint32_t ostrideB = (uint32_t)job.out_stride * 4u;
        auto prep = [&](int b) __attribute__((always_inline)) {
            const int t = b * BY + lane;
            const int gy2 = job.ty0 - 4 * R + t;
            nx_rny = s_rtab[window_count(gy2, R, job.cy0, job.cy1)];
            nx_grow = (int)((uint32_t)min(max(gy2, job.cy0), cy1m) * rowB);
            okbits = ballot_low(t >= 4 * R && t < Ttot, BY);
        };
        auto issue_row = [&](auto itag) __attribute__((always_inline)) {
            constexpr int i = decltype(itag)::value;
            rny2[i] = readlane_f32(nx_rny, i);
            if (LES_LAB_ABLATE(128)) { gq[i] = (uint32_t)lane; return; }
            gq[i] = buf_load<uint32_t>(rs_guide, sx4, (uint32_t)readlane_i32(nx_grow, i));
        };
        // two specialisations (label check on / off), selected once per job -- see role A
        auto march_d = [&](auto check_tag) __attribute__((always_inline)) {
        constexpr bool CHECK = decltype(check_tag)::value != 0;
        LES_TICK_BEGIN();
        for (int k0 = 0; k0 < nticks; k0 += UN) {
            static_for<UN>([&](auto utag) {
                constexpr int U = decltype(utag)::value;
                constexpr int BASE = ((U + UN - 3 % UN) % UN) * BY;     // block k - 3: its ring slot base is a compile-time constant (see role A)
                const int k = k0 + U;
                if (k < nticks) {
                    if (k >= 3) {
                        const int b = k - 3;
                        const int4 (*T2)[PCOLS] = s_T2[(U + UN - 3 % UN) % UN][slot];
                        constexpr int GD = 1, NS = (BY + GD - 1) / GD;   // as in role C: the reads of the next row are in flight during the arithmetic of this one
                        int4 pp[2][GD], pm[2][GD], px[2][GD];
                        auto lds_stage = [&](auto stag) __attribute__((always_inline)) {
                            constexpr int S = decltype(stag)::value;
#pragma unroll
                            for (int j = 0; j < GD; j++)
                                if (S * GD + j < BY) {
                                    if (LES_LAB_ABLATE(16)) { pp[S & 1][j] = int4{lane, b, 2, 3}; pm[S & 1][j] = int4{3, 2, b, lane}; px[S & 1][j] = int4{0, 0, 0, 0}; }
                                    else { pp[S & 1][j] = T2[S * GD + j][pcP]; pm[S & 1][j] = T2[S * GD + j][pcM]; px[S & 1][j] = T2[S * GD + j][pcX]; }
                                }
                        };
                        lds_stage(std::integral_constant<int, 0>{});
                        static_for<NS>([&](auto gtag) {
                            constexpr int S = decltype(gtag)::value, CB = S & 1;
                            constexpr int LO = S * GD;
                            constexpr int N = (BY - LO) < GD ? (BY - LO) : GD;
                            if constexpr (S + 1 < NS) lds_stage(std::integral_constant<int, S + 1>{});
                            LES_MARCH_SCHED_FENCE();
                            static_for<N>([&](auto jtag) {
                                constexpr int j = decltype(jtag)::value;
                                constexpr int i = LO + j;
                                constexpr int SLOT = BASE + i, OLD = (SLOT + RS - KS) % RS;
                                // horizontal box sums (exact, |h| < 2^30 by the stage-2 scale)
                                const int h0 = pp[CB][j].x - pm[CB][j].x + px[CB][j].x, h1 = pp[CB][j].y - pm[CB][j].y + px[CB][j].y;
                                const int h2 = pp[CB][j].z - pm[CB][j].z + px[CB][j].z, h3 = pp[CB][j].w - pm[CB][j].w + px[CB][j].w;
                                const int o0 = ring2[0][OLD], o1 = ring2[1][OLD], o2 = ring2[2][OLD], o3 = ring2[3][OLD];
                                ring2[0][SLOT] = h0; ring2[1][SLOT] = h1; ring2[2][SLOT] = h2; ring2[3][SLOT] = h3;
                                acc64_add_i32(S2[0], h0 - o0); acc64_add_i32(S2[1], h1 - o1); acc64_add_i32(S2[2], h2 - o2); acc64_add_i32(S2[3], h3 - o3);
                                if (out_col && ((okbits >> i) & 1u)) {
                                    const int t = b * BY + i;
                                    const uint32_t gi = gq[i];
                                    // LES/GuidedFilter.h:243: (b + a . I) / N on the centred guide (fp32: each window sum carries 2^-24 relative)
                                    const float f0 = (float)(int)(S2[0] >> kMarchS2), f1 = (float)(int)(S2[1] >> kMarchS2), f2 = (float)(int)(S2[2] >> kMarchS2), f3 = (float)(int)(S2[3] >> kMarchS2);
                                    const float acc = fmaf(f2, cvt_f32_sbyte<2>(gi), fmaf(f1, cvt_f32_sbyte<1>(gi), fmaf(f0, cvt_f32_sbyte<0>(gi), f3 * 255.0f)));
                                    float q = fmaf(acc * c_lane, rny2[i], view.poff);
                                    if constexpr (CHECK) {
                                        const int gy2 = job.ty0 + t - 4 * R;
                                        const float ds = ((vl_xa + (float)gy2 * plane.y) + vl_c) + vl_zv;
                                        const float dp = ds + vl_a5, dm = ds - vl_a5;
                                        const float d1 = dp + vl_b5, d2 = dp - vl_b5, d3 = dm + vl_b5, d4 = dm - vl_b5;
                                        const float mn = fmin3(fmin3(ds, d1, d2), d3, d4), mx = fmax3(fmax3(ds, d1, d2), d3, d4);
                                        if (!(mn >= g.mind && mx <= g.maxd)) q = LES_COST_INVALID;
                                    }
                                    if (!(LES_LAB_ABLATE(128) && q != 12345.0f))
                                    buf_store<float>(rs_out, oc4, (uint32_t)(t - 4 * R) * ostrideB, q);
                                }
                            });
                            LES_MARCH_SCHED_FENCE();
                        });
                    }
                    if (k >= 2 && k <= nblk + 1) {                // guide rows of the block this role handles at the next tick (BY dwords per
                        prep(k - 2);                              // lane; this role is never the last to arrive at the barrier, and it has no
                        static_for<BY>([&](auto itag) { issue_row(itag); });   // registers to spare for a rolling issue)
                    }
                    LES_TICK_BARRIER();
                }
            });
        }
        LES_TICK_END(2);
        };
        // The per-pixel label test is only compiled in where it can matter: not for a plane that is provably valid on the whole target
        // strip (every fronto-parallel plane with c inside the range, and the optimiser's proposals away from the disparity limits).
        const bool surely_valid = march_label_surely_valid(g, plane, job.tx0, job.tx0 + max(job.tw, 1) - 1, job.ty0, job.ty0 + max(job.th, 1) - 1);
        if (check && !surely_valid) march_d(std::integral_constant<int, 1>{});
        else march_d(std::integral_constant<int, 0>{});
    }
}

// ---------------------------------------------------------------------------------------------------
// One-time preparation for the march kernel.
// ---------------------------------------------------------------------------------------------------
// guide as signed bytes + statistics in the march format (from the same fp64 horizontal sums as les_stats_finish_kernel);
// *inv_diag_max (float bits, positive) collects max over pixels of the diagonal of the inverse covariance
__global__ void les_march_stats_kernel(const double* __restrict__ hs, const uint32_t* __restrict__ ipk, uint32_t* __restrict__ ipk8,
                                       float* __restrict__ mstats, unsigned* __restrict__ inv_diag_max, int H, int W, int R, double eps)
{
    int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), y = (int)blockIdx.y;
    if (x >= W) return;
    size_t P = (size_t)H * W;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int dy = -R; dy <= R; dy++) {
        int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int k = 0; k < 9; k++) s[k] += hs[k * P + (size_t)yy * W + x];
    }
    // LES/GuidedFilter.h:69-101 (identical to les_stats_finish_kernel)
    double N = (double)(window_count(x, R, 0, W) * window_count(y, R, 0, H));
    double m0 = s[0] / N, m1 = s[1] / N, m2 = s[2] / N;
    double rr = s[3] / N - m0 * m0 + eps, rg = s[4] / N - m0 * m1, rb = s[5] / N - m0 * m2;
    double gg = s[6] / N - m1 * m1 + eps, gb = s[7] / N - m1 * m2, bb = s[8] / N - m2 * m2 + eps;
    double irr = gg * bb - gb * gb, irg = gb * rb - rg * bb, irb = rg * gb - gg * rb;
    double igg = rr * bb - rb * rb, igb = rb * rg - rr * gb, ibb = rr * gg - rg * rg;
    double det = irr * rr + irg * rg + irb * rb;
    irr /= det; irg /= det; irb /= det; igg /= det; igb /= det; ibb /= det;
    size_t px = (size_t)y * W + x;
    // centred means in u8 units, [-128, 127], as integers with MB fraction bits (|M| <= 2^31: the clamp only guards the rounding of a
    // mean of exactly 255)
    const double fs = (double)(1ll << kMarchMB);
    const double c0 = m0 * 255.0 - 128.0, c1 = m1 * 255.0 - 128.0, c2 = m2 * 255.0 - 128.0;
    // stored NEGATED (role C adds -mean_c s): -c in [-127, 128], and +128 * 2^24 = 2^31 is clamped to 2^31 - 1 (an error of 2^-24 u8 on a black window)
    const int M0 = (int)fmin(fmax(rint(-c0 * fs), -2147483648.0), 2147483647.0);
    const int M1 = (int)fmin(fmax(rint(-c1 * fs), -2147483648.0), 2147483647.0);
    const int M2 = (int)fmin(fmax(rint(-c2 * fs), -2147483648.0), 2147483647.0);
    float* o = mstats + px * kMarchStatWords;
    o[0] = (float)irr; o[1] = (float)irg; o[2] = (float)irb; o[3] = (float)igg;
    o[4] = (float)igb; o[5] = (float)ibb; o[6] = __int_as_float(M0); o[7] = __int_as_float(M1);
    o[8] = __int_as_float(M2);
    if (kMarchStatWords == 12) { o[9] = (float)(c0 / 255.0); o[10] = (float)(c1 / 255.0); o[11] = (float)(c2 / 255.0); }
    const uint32_t v = ipk[px];
    const uint32_t b0 = ((v & 0xffu) - 128u) & 0xffu, b1 = (((v >> 8) & 0xffu) - 128u) & 0xffu, b2 = (((v >> 16) & 0xffu) - 128u) & 0xffu;
    ipk8[px] = b0 | (b1 << 8) | (b2 << 16);
    const float dmax = fmaxf(fmaxf((float)irr, (float)igg), (float)ibb);
    if (dmax > 0.0f) atomicMax(inv_diag_max, __float_as_uint(dmax));
}

// the tiled copy of the volume (MarchView::vol_t): dst[((y W8 + x / 8) D + d) 8 + x % 8] = src[(d H + y) W + x]; columns beyond W are zero.
// One workgroup = one image row x 64 columns (8 tiles) x all slices: reads 256 contiguous bytes per slice, writes 8 x 32 bytes (the tiles'
// slices are filled in order, so the 32-byte pieces of a line meet in the L2 before it is written back).
__global__ void les_tile_volume_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H, int W)
{
    const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)), y = (int)blockIdx.y;
    const int W8 = (W + 7) >> 3;
    if (x >= W8 * 8) return;
    const size_t HW = (size_t)H * W;
    float* o = dst + (((size_t)y * W8 + (size_t)(x >> 3)) * D) * 8 + (x & 7);
    for (int d = (int)(threadIdx.x >> 6); d < D; d += (int)(blockDim.x >> 6))
        o[(size_t)d * 8] = x < W ? src[(size_t)d * HW + (size_t)y * W + x] : 0.0f;
}

// min of a float array and whether every element is finite: part[2*b] = min bits of block b, part[2*b+1] = 1 if a non-finite value was seen
__global__ void les_range_kernel(const float* __restrict__ v, size_t n, float* __restrict__ part_min, int* __restrict__ part_bad)
{
    __shared__ float s_min[256];
    __shared__ int s_bad[256];
    float m = INFINITY;
    int bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = v[i];
        if (!(fabsf(x) < INFINITY)) bad = 1;
        else m = fminf(m, x);
    }
    s_min[threadIdx.x] = m; s_bad[threadIdx.x] = bad;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            s_min[threadIdx.x] = fminf(s_min[threadIdx.x], s_min[threadIdx.x + k]);
            s_bad[threadIdx.x] |= s_bad[threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part_min[blockIdx.x] = s_min[0]; part_bad[blockIdx.x] = s_bad[0]; }
}

}  // namespace les
