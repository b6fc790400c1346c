// les_march.h -- the fixed-point "column march" kernel: gather + guided-filter aggregation with exact integer box sums
// (gfx950 / CDNA4, wave64).  It computes the same quantity as les_strip_kernel (les_kernels.h):
//
//     p(s)  = min( lerp_d vol[a*x+b*y+c][y][x], th_col )                       LES/CostVolumeEnergy.h:70-98
//     q     = guided filter of p with the colour guide, radius R, eps           LES/GuidedFilter.h:142-266, 301-326
//     q    -> 1e6 where the label is invalid                                    LES/CostVolumeEnergy.h:176-183
//
// but is organised around what the MI355X micro-benchmarks say is cheap (tools/ubench/valu_rates.hip, mix_issue.hip): 32-bit
// integer / fp32 adds issue in 2 cycles per wave, every fp64 operation, conversion, DPP move or 64-bit integer MAD in 4 on a pipe
// the waves of a SIMD share, an LDS ds_write_b128 costs ~13 CU cycles and a ds_read_b128 4, a wave issues at most one
// instruction per ~4 cycles, a taken branch costs ~26, and the arbiter prefers the oldest wave.  All four box-filter passes are
// therefore EXACT INTEGER sums, computed by three wave-specialised roles (A, C, D below; the prefix passes B1 / B2 run on the
// waves of role A) that work on consecutive row blocks at the same time:
//
//   A  (lane = image column of a job, marching down the rows; NJ jobs per workgroup)
//        p -> 22-bit fixed point  pi = round((p - vmin) * sp)      (p lives in [vmin, th_col]; vmin = min of the volume)
//        vertical 2R+1 running sums over a register ring of (pi, packed guide):  Sp = sum pi   (int32, exact)
//                                                                                Sc = sum Iq_c * pi (int64, exact; Iq = u8 - 128)
//        -> LDS T1[row][col] = {Sp, (Sc + 2^8) >> 9}                 (the only rounding of stage 1: 2^-31 of full scale)
//   B1 (lane = (row, 8-column segment)) in-place prefix sums along x, modulo 2^32 (the 21-column differences are exact)
//   C  (lane = column)  box sums s, t_c = P(x+R) - P(x-R-1);  N cov_c = t_c - hi32(M_c * s)  (M_c = mean_c in 2^-23 u8 units:
//        one v_mad_i64_i32, the cancellation is exact);  a = inv * cov,  b = mean_p - a . mean   in fp32 (as les_strip_kernel);
//        a, b -> int32 with a scale derived from a rigorous bound on |a|, |b|  -> LDS T2
//   B2 prefix sums of T2
//   D  (lane = column)  horizontal box = prefix difference (int32), vertical running sums over an int32 register ring,
//        accumulated in fp64 (integers < 2^53: exact),  q = (Sb * 255 + sum_c Sa_c * Iq_c) * rn / (255 scale) + vmin.
//
// Error budget against the double-precision reference (tools/fixedpoint_probe.py, DESIGN.md "Numerics"): the fixed-point
// cost (6e-8 of the range), the 2^-31 rounding of the stage-1 vertical sums, M_c (2^-31), the fp32 3x3 algebra (shared with
// les_strip_kernel) and the stage-2 quantisation (resolution ~1e-6 of |a|max before a 441-pixel average): measured 6e-8 ..
// 4e-7 absolute on costs in [0, 0.5] for eps = 1e-4.
//
// The kernel is only launched when the host has established its preconditions (les_hip.hip: march_usable): a finite volume,
// th_col - vmin <= 8 |th_col|, and every target at least 2R away from clip borders that are not image borders (so that every
// consumed stage-1 window is a true covariance window and the bound on |a| holds).  Everything else runs les_strip_kernel.
#pragma once

#include "les_kernels.h"

namespace les {

constexpr int kMarchPB = 22;      // bits of the fixed-point cost
constexpr int kMarchSH = 9;       // right shift of the vertical sums of Iq * pi before the horizontal pass (M_c has 23 fraction bits: SH + 23 = 32)

struct MarchView {
    const float* vol;             // [D][H][W]
    const uint32_t* ipk8;         // [H*W] guide pixel as three signed bytes u8 - 128 (byte 3 = 0)
    const float4* mstats;         // [H*W][3]: {inv00, inv01, inv02, inv11} {inv12, inv22, mu0, mu1} {mu2, M0, M1, M2}
                                  //   inv = (Sigma + eps U)^-1 of the guide in [0,1] units (LES/GuidedFilter.h:87-101),
                                  //   mu_c = mean_I_c - 128/255 (fp32), M_c = rint((255 mean_I_c - 128) 2^23) (int32 bits)
    float vmin;                   // lower end of the cost range (min of the volume, <= th_col)
    float sp;                     // (2^PB - 1) / (th_col - vmin)
    float kapS;                   // 2^SH * u_p / 255 * scale        (u_p = (th_col - vmin) / (2^PB - 1))
    float upS;                    // u_p * scale
    double qscale;                // 1 / (255 * scale)
    // image-based energy (les_hip_create_naive): vol is the raw-cost scratch les_naive_raw_kernel has just filled, raw_off[i] the
    // float offset of call i's filterRect patch in it (row stride = filterRect width), vmin = 0, th_col -> th_color + th_grad.
    // Null for a cost-volume context.
    const long long* raw_off;
};

template <int R, int WGC, int NJ, int BY>
struct MarchCfg {
    static constexpr int KS = 2 * R + 1;
    static constexpr int TW = WGC - 4 * R;                 // output columns per job
    static constexpr int HWV = WGC / 64;                   // waves per role and job slot
    static constexpr int NW = 3 * HWV * NJ;                // waves per workgroup: roles A, C, D
    static constexpr int NT = 64 * NW;
    static constexpr int SEGL = 8;                         // columns per prefix segment
#ifndef LES_MARCH_PADW
#define LES_MARCH_PADW 16
#endif
    // Physical columns: a leading zero element (P(-1)) + one pad element per PADW columns; with PADW = 16 the row length is rounded
    // up to 4 mod 8 elements.  The prefix pass reads, per 16 lanes, the 8 segments (stride 8 columns = 32 banks, shifted by one
    // element = 4 banks per pad) of two rows (shifted by +-16 banks when the row length is 4 mod 8): all 64 banks once.  The
    // consumers' 16 consecutive columns then straddle one pad (one bank collision per 16 lanes) instead of two.
    static constexpr int PADW = LES_MARCH_PADW;
    static constexpr int PCOLS0 = 1 + WGC + WGC / PADW;
    static constexpr int PCOLS = PADW == 16 ? PCOLS0 + ((4 - PCOLS0 % 8) + 8) % 8 : PCOLS0;      // 4 or 12 mod 16: the second row lands 16 banks off either way
    static_assert(KS % BY == 0, "the block height must divide the ring length (compile-time ring slots)");
    static_assert(KS / BY == 3, "three blocks per ring: the ring slots and the stage-2 buffer of a block are compile-time constants in a loop unrolled by 3");
    static_assert(WGC % 64 == 0, "a job slot is a whole number of waves");
    static_assert(BY * (64 / SEGL) <= 64, "one wave prefixes its own tile");
    static_assert(TW > 0, "job too narrow for this radius");
    static_assert(2 * R + 1 < 64, "a window crosses at most one wave boundary");
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
// has no state and can hold the 12 statistics words of all BY rows in flight.  The prefix sums are wave-local (each wave
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
#define LES_SCAN_STEP(N) { inc.x += dpp_row_shr<N>(inc.x); inc.y += dpp_row_shr<N>(inc.y); inc.z += dpp_row_shr<N>(inc.z); inc.w += dpp_row_shr<N>(inc.w); }
    LES_SCAN_STEP(2) LES_SCAN_STEP(4) LES_SCAN_STEP(8)
#undef LES_SCAN_STEP
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
    int4* p = &T[act ? row : 0][1 + c0 + c0 / LES_MARCH_PADW];                // (the lanes of the unused 8th row read row 0 and write nothing)
    int4 v[SEGL];
#pragma unroll
    for (int j = 0; j < SEGL; j++) v[j] = p[j];
    march_prefix_scan8(v);
    if (act) {
#pragma unroll
        for (int j = 0; j < SEGL; j++) p[j] = v[j];
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
    const int pc = 1 + c0 + c0 / LES_MARCH_PADW;
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
        for (int j = 0; j < SEGL; j++) pb[j] = vb[j];
    }
    LES_MARCH_SCHED_FENCE();
    march_prefix_scan8(va);
    if (act) {
#pragma unroll
        for (int j = 0; j < SEGL; j++) pa[j] = va[j];
    }
}

// -DLES_PHASE_TIMING: lane 0 of every wave accumulates the cycles it computes per tick and the cycles it waits at the tick
// barrier: les_dbg[2 role] += compute, les_dbg[2 role + 1] += wait, les_dbg[6 + role] += ticks, les_dbg[9 + role] += the part of
// `compute` before LES_TICK_MARK (the row loop; the rest is the prefix pass of the tile)            (tools/phase_probe.py)
#if defined(LES_PHASE_TIMING) && !defined(LES_SIM)
#define LES_TICK_BEGIN() unsigned long long tk_c_ = 0, tk_w_ = 0, tk_n_ = 0, tk_r_ = 0, tk_m_ = 0, tk_t_ = clock64()
#define LES_TICK_MARK() (tk_m_ = clock64())
#define LES_TICK_BARRIER() do { const unsigned long long a_ = clock64(); __syncthreads(); const unsigned long long b_ = clock64(); tk_c_ += a_ - tk_t_; tk_r_ += (tk_m_ > tk_t_ ? tk_m_ : a_) - tk_t_; tk_w_ += b_ - a_; tk_t_ = b_; tk_n_++; } while (0)
#define LES_TICK_END(role_) do { if (lane == 0) { atomicAdd(&les_dbg[2 * (role_)], tk_c_); atomicAdd(&les_dbg[2 * (role_) + 1], tk_w_); atomicAdd(&les_dbg[6 + (role_)], tk_n_); atomicAdd(&les_dbg[9 + (role_)], tk_r_); } } while (0)
#else
#define LES_TICK_BEGIN() ((void)0)
#define LES_TICK_MARK() ((void)0)
#define LES_TICK_BARRIER() __syncthreads()
#define LES_TICK_END(role_) ((void)0)
#endif

// wave-uniform base + unsigned 32-bit per-lane byte offset: compiles to the `saddr` form of global_load / global_store, i.e. no
// 64-bit address arithmetic on the VALU.  (The empty asm pins the offset as a 32-bit value in the block of the access; otherwise its
// zero-extension is hoisted out of the loops and instruction selection falls back to a 64-bit VGPR address + v_lshl_add_u64.)
#if defined(LES_SIM)
#define LES_PIN_U32(x) ((void)0)
#else
#define LES_PIN_U32(x) asm volatile("" : "+v"(x))
#endif
template <class T>
__device__ __forceinline__ T ld_sbase(const T* base, uint32_t byte_off)
{
    LES_PIN_U32(byte_off);
#if defined(LES_VOL_NT) && !defined(LES_SIM)
    return __builtin_nontemporal_load((const T*)((const char*)base + byte_off));      // measurement switch: streaming loads of volume / guide rows
#else
    return *(const T*)((const char*)base + byte_off);
#endif
}
template <class T>
__device__ __forceinline__ void st_sbase(T* base, uint32_t byte_off, T v) { LES_PIN_U32(byte_off); *(T*)((char*)base + byte_off) = v; }

// The statistics rows of role C are loop-carried 16-byte register tuples that are reloaded in place, one row at a time, while
// the rest of the block is still being consumed.  Written as plain C++ loads, the register allocator lands every reload in a
// fresh tuple and copies it to the loop-carried one right away, i.e. waits for the load it has just issued.  The loads are
// therefore issued as inline assembly with the destination TIED to the variable, and the vmcnt bookkeeping for them is done
// by hand (this role issues no other vector-memory instruction): march_stats_wait3/6<n>(rows...) waits until at most n of this
// wave's loads are outstanding and, by naming the rows as read-write operands, keeps their uses behind the wait.
#if defined(LES_SIM)
typedef float4 mstat4;
#define LES_STATS_LOAD(dst, base, off, IMM) ((dst) = *(const float4*)((const char*)(base) + (off) + (IMM)))
template <int N>
__device__ inline void march_stats_wait3(float4 (&r)[3]) { (void)r; }
template <int N>
__device__ inline void march_stats_wait6(float4 (&r)[3], float4 (&q)[3]) { (void)r; (void)q; }
#else
typedef float mstat4 __attribute__((ext_vector_type(4)));
#ifndef LES_STATS_POLICY
#define LES_STATS_POLICY ""        // cache-policy bits of the statistics loads (measurement switch: " nt", " sc0", " sc1", " sc0 sc1")
#endif
#define LES_STATS_LOAD(dst, base, off, IMM) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #IMM LES_STATS_POLICY : "+v"(dst) : "v"(off), "s"(base))
template <int N>
__device__ __forceinline__ void march_stats_wait3(mstat4 (&r)[3]) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]) : "n"(N)); }
template <int N>
__device__ __forceinline__ void march_stats_wait6(mstat4 (&r)[3], mstat4 (&q)[3])
{
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]) : "n"(N));
}
#endif

// measurement switches (ablations: the output is meaningless, the time shows what a resource costs): 1 role C issues no statistics
// loads, 2 role C reads no LDS, 16 role D reads no LDS, 64 role A loads nothing, 128 role D loads / stores nothing
#ifndef LES_MARCH_EXP
#define LES_MARCH_EXP 0
#endif
template <int R, int WGC, int NJ, int BY>
__global__ void __launch_bounds__(3 * WGC * NJ)
les_march_kernel(Geom g, MarchView view, const Job* __restrict__ jobs, const float4* __restrict__ planes,
                 float* __restrict__ out, int ngroups, int check)
{
    using Cfg = MarchCfg<R, WGC, NJ, BY>;
    constexpr int KS = Cfg::KS, NT = Cfg::NT, HWV = Cfg::HWV, NW = Cfg::NW, PCOLS = Cfg::PCOLS;

    __shared__ int4 s_T1[2][NJ][BY][PCOLS];  // stage 1: vertical sums, then (in place) their prefix sums along x; double buffered over blocks
    __shared__ int4 s_T2[3][NJ][BY][PCOLS];  // stage 2: quantised (a_0, a_1, a_2, b), then their prefix sums; three blocks in flight (written, prefixed, consumed)
    __shared__ double s_rtab[KS + 1];        // 1/n, n = 0..2R+1

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
#ifndef LES_MARCH_ROLE_ORDER
#define LES_MARCH_ROLE_ORDER 5
#endif
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

    if (tid <= KS) s_rtab[tid] = tid > 0 ? 1.0 / (double)tid : 0.0;
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
    const int nx = window_count(gx, R, job.cx0, job.cx1);                 // the same count serves stage 1 and stage 2 (same column)
    // physical columns of P(x+R), P(x-R-1) and, when the window crosses a wave boundary, of the left neighbour's tile total
    const int cP = min(ci + R, WGC - 1), cM = ci - R - 1;
    const int pcP = Cfg::pcol(cP);
    const int pcM = cM >= 0 ? Cfg::pcol(cM) : 0;
    const bool cross = cM >= 0 && (cP >> 6) != (cM >> 6);
    const int pcX = cross ? Cfg::pcol((cP >> 6) * 64 - 1) : 0;           // element 0 is the constant zero
    const int pcS = Cfg::pcol(ci);

    // measurement switch (tools/role_time.sh): bit 0 / 1 / 2 = compile role A / C / D in; the waves of the other roles exit at once
#ifndef LES_MARCH_ROLE_MASK
#define LES_MARCH_ROLE_MASK 7
#endif
    if (role == 0 && (LES_MARCH_ROLE_MASK & 1)) {
        // ================================================= role A =================================================
        const uint32_t HWu = (uint32_t)g.H * (uint32_t)g.W;
        const float g_ax = plane.x * (float)sx;                           // a * x, LES/CostVolumeEnergy.h:76
        // fronto-parallel planes (a = b = 0): d = c for every pixel, so taps / weight / mode are per-job constants
        const bool fronto = plane.x == 0.0f && plane.y == 0.0f;
        GatherPrep gpc = gather_prepare(g, 0.0f, plane.z, 0u, HWu, true);
        if (gpc.f1 == 0.0f) gpc.i1 = gpc.i0;                              // weight 0 on a finite volume: the second tap is never needed
        int ringP[KS];                   // fixed-point cost of the last 2R+1 p-rows of this column
        uint32_t ringG[KS];              // their guide pixels
#pragma unroll
        for (int k = 0; k < KS; k++) { ringP[k] = 0; ringG[k] = 0u; }
        int Sp = 0;
        long long Sc[3] = {1ll << (kMarchSH - 1), 1ll << (kMarchSH - 1), 1ll << (kMarchSH - 1)};   // rounding bias of the >> SH folded in
        const uint32_t i0s = (uint32_t)readfirstlane_i32((int)gpc.i0), i1s = (uint32_t)readfirstlane_i32((int)gpc.i1);
        const float f1s = __int_as_float(readfirstlane_i32(__float_as_int(gpc.f1)));
        const int modes = readfirstlane_i32(gpc.mode);
        const float pbias = fmaf(-view.vmin, view.sp, 0.5f);             // pi = trunc(p * sp + (0.5 - vmin * sp))
        const float f0s = 1.0f - f1s;
        const bool inv_job = modes == 2;
        // The march itself exists in three specialisations, selected once per job (a wave-uniform test per row costs the wave a
        // taken branch and splits the row code into blocks the scheduler cannot interleave):
        //   KIND 0  fronto-parallel plane, one volume tap  (integer disparity, clamped or invalid label)
        //   KIND 1  fronto-parallel plane, two taps
        //   KIND 2  general plane: taps / weight / mode per lane and row
        //   KIND 3  image-based energy: one tap into the call's raw-cost patch (any plane; no truncation, no invalid mode)
        // For fronto-parallel planes everything but the clip test is per-job, the row bases are scalars and the loads need no
        // address arithmetic.
        auto march_a = [&](auto kind_tag) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_tag)::value;
        GatherPrep gp[BY];
        float v0[BY], v1[BY];
        uint32_t gw[BY];
        uint32_t rowbits = 0;            // fronto path: bit i = p-row i of the block is inside the clip and the march
        // Row scalars of block b: lane i < BY computes those of p-row b*BY + i, v_readlane hands them to the wave as scalars.
        // Loads are issued one row at a time, right after the same row of the previous block has been consumed ("rolling"
        // prefetch: a whole tick of latency cover without a second set of registers).  Rows beyond the march are clamped to a
        // valid address and flagged off, so the issue needs no branch.
        int nx_rowpx = 0;
        float nx_dbase = 0.0f;
        int nx_rowraw = 0;               // KIND 3: float offset of the row in the call's patch
        const int fw = job.cx1 - job.cx0;
        const float* rawbase = nullptr;
        if constexpr (KIND == 3) rawbase = view.vol + view.raw_off[job.plane_idx] - job.cx0;
        auto prep = [&](int b) __attribute__((always_inline)) {
            const int t = b * BY + lane;
            const int gy = job.ty0 - 2 * R + t;
            const int sy = min(max(gy, job.cy0), cy1m);
            nx_rowpx = (int)(((uint32_t)sy * (uint32_t)g.W) | ((t < Ttot && gy >= job.cy0 && gy < job.cy1) ? 0x80000000u : 0u));
            if constexpr (KIND == 3) nx_rowraw = (sy - job.cy0) * fw;
            else nx_dbase = plane.y * (float)sy + plane.z;              // b*y + c, LES/CostVolumeEnergy.h:73
        };
        auto issue_row = [&](auto itag) __attribute__((always_inline)) {
            constexpr int i = decltype(itag)::value;
            const uint32_t rowpx = (uint32_t)readlane_i32(nx_rowpx, i);
            const uint32_t ro = rowpx & 0x7fffffffu;
            if (LES_MARCH_EXP & 64) { rowbits = 0x7f; v0[i] = (float)lane; v1[i] = 0.0f; gw[i] = (uint32_t)lane; return; }
            if constexpr (KIND == 3) {
                rowbits = (rowbits & ~(1u << i)) | ((rowpx >> 31) << i);
                const float* r0 = rawbase + (size_t)(uint32_t)readlane_i32(nx_rowraw, i);
                v0[i] = ld_sbase(r0, sx4);
            } else if constexpr (KIND < 2) {
                rowbits = (rowbits & ~(1u << i)) | ((rowpx >> 31) << i);
                const float* r0 = view.vol + (size_t)(i0s + ro);          // scalar bases: the loads take them + the lane's column
                v0[i] = ld_sbase(r0, sx4);
                if constexpr (KIND == 1) { const float* r1 = view.vol + (size_t)(i1s + ro); v1[i] = ld_sbase(r1, sx4); }
            } else {
                const float d_base = readlane_f32(nx_dbase, i);
                const bool inside = col_in && (rowpx >> 31);
                gp[i] = gather_prepare(g, g_ax, d_base, ro + (uint32_t)sx, HWu, inside);
                if (gp[i].f1 == 0.0f) gp[i].i1 = gp[i].i0;
                v0[i] = view.vol[gp[i].i0];
                v1[i] = view.vol[gp[i].i1];
            }
            const uint32_t* rg = view.ipk8 + (size_t)ro;
            gw[i] = ld_sbase(rg, sx4);
        };
        prep(0);
        static_for<BY>([&](auto itag) { issue_row(itag); });
        LES_TICK_BEGIN();
        // three ticks per loop iteration: the ring slot of a block's first row, (k * BY) mod KS, is then a compile-time constant
        // and the rings stay in fixed registers (a branch per block on the slot base made the allocator spill half of them)
        constexpr int UN = KS / BY;
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
                            constexpr int SLOT = BASE + i;       // ring slot of p-row k*BY + i; it holds the row that leaves the window (2R+1 rows ago)
                            int pi;
                            if constexpr (KIND == 3) {
                                pi = (col_in && ((rowbits >> i) & 1u)) ? (int)fmaf(v0[i], view.sp, pbias) : 0;
                            } else if constexpr (KIND < 2) {
                                // LES/CostVolumeEnergy.h:78-96 with per-job taps: clamped / interpolated / invalid, then min(C, th_col)
                                // (one tap: the weight of the second is zero and the volume is finite, so f0 v0 + 0 v1 = v0)
                                float C = v0[i];
                                if constexpr (KIND == 1) C = f0s * v0[i] + f1s * v1[i];
                                C = inv_job ? LES_COST_INVALID : C;
                                const float p = (g.th_col < C) ? g.th_col : C;
                                pi = (col_in && ((rowbits >> i) & 1u)) ? (int)fmaf(p, view.sp, pbias) : 0;
                            } else {
                                const float p = gather_finish(g, gp[i], v0[i], v1[i]);
                                pi = gp[i].mode == 3 ? 0 : (int)fmaf(p, view.sp, pbias);
                            }
                            const uint32_t gi = gw[i];
                            const int po = ringP[SLOT];
                            const uint32_t go = ringG[SLOT];
                            ringP[SLOT] = pi;
                            ringG[SLOT] = gi;
                            Sp += pi - po;
                            const int npo = -po;
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                const int qi = ((int)(gi << (24 - 8 * c))) >> 24, qo = ((int)(go << (24 - 8 * c))) >> 24;
                                Sc[c] += (long long)qi * (long long)pi;
                                Sc[c] += (long long)qo * (long long)npo;
                            }
                            T[i][pcS] = int4{Sp, (int)(Sc[0] >> kMarchSH), (int)(Sc[1] >> kMarchSH), (int)(Sc[2] >> kMarchSH)};
                            issue_row(itag);                     // the same row of block k + 1
                        });
                        LES_TICK_MARK();
                    }
                    // The prefix sums of this block of stage 1 and of block k - 2 of stage 2 (written by the role-C wave of the same
                    // tile at the previous tick): this role has the shortest row loop, so it does the prefix work of both stages.
                    {
                        int4 (*TB)[PCOLS] = s_T2[(decltype(utag)::value + UN - 2 % UN) % UN][slot];
                        // (general planes in the two-job geometry: role C prefixes its own block one tick later instead, see there)
                        const bool da = k < nblk, db = !(NJ > 1 && KIND == 2) && k >= 2 && k < nblk + 2;
                        constexpr bool kPair = KIND != 2;
#ifndef LES_MARCH_PREFIX_PAIR
#define LES_MARCH_PREFIX_PAIR 1
#endif
                        if (LES_MARCH_PREFIX_PAIR && kPair && da && db) march_prefix_pair<BY, PCOLS>(s_T1[k & 1][slot], TB, ci0, lane);
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
        else march_a(std::integral_constant<int, 2>{});
    } else if (role == 1 && (LES_MARCH_ROLE_MASK & 2)) {
        // ================================================= role C =================================================
        const bool s1_col = col_in && ci >= R && ci < WGC - R;            // stage-1 column with a complete horizontal window
        const bool general_plane = !view.raw_off && !(plane.x == 0.0f && plane.y == 0.0f);  // (role A's KIND 2)
        // a, b are zero outside the clip and before the march is primed: the column part of that rule is folded into the lane's
        // normalisation factors, the row part into the row's 1/count_y (0 * finite = 0, and v_cvt_rpi(+-0) = 0)
        const float kap_x = s1_col ? view.kapS * (float)s_rtab[nx] : 0.0f;
        const float up_x = s1_col ? view.upS * (float)s_rtab[nx] : 0.0f;
        const uint32_t sx48 = (uint32_t)sx * 48u;
        mstat4 st[BY][3];
#pragma unroll
        for (int i = 0; i < BY; i++) st[i][0] = st[i][1] = st[i][2] = mstat4{0.0f, 0.0f, 0.0f, 0.0f};
        float rny[BY];                                                    // 1 / count_y of the rows of the block in flight, 0 for rows outside the clip or before the march is primed (wave-uniform: scalar registers)
        int nx_srow = 0;
        float nx_rny = 0.0f;
        auto prep = [&](int b) __attribute__((always_inline)) {
            const int t = b * BY + lane;
            const int gy1 = job.ty0 - 3 * R + t;                          // centre of the vertical window that ends at p-row t
            nx_rny = (float)s_rtab[window_count(gy1, R, job.cy0, job.cy1)];
            nx_srow = (int)(((uint32_t)min(max(gy1, job.cy0), cy1m) * (uint32_t)g.W) | ((gy1 >= job.cy0 && gy1 < job.cy1 && t >= 2 * R && t < Ttot) ? 0x80000000u : 0u));
        };
        auto issue_row = [&](auto itag) __attribute__((always_inline)) {      // rolling prefetch, see role A
            constexpr int i = decltype(itag)::value;
            const uint32_t srow = (uint32_t)readlane_i32(nx_srow, i);
            const float r = readlane_f32(nx_rny, i);
            rny[i] = (srow >> 31) ? r : 0.0f;
            const float4* sp = view.mstats + (size_t)(srow & 0x7fffffffu) * 3;          // scalar row base + the lane's column
            mstat4 &d0 = st[i][0], &d1 = st[i][1], &d2 = st[i][2];                    // (named here: operands of an asm statement alone do not capture)
            const uint32_t off = sx48;
            if (LES_MARCH_EXP & 1) return;
            LES_STATS_LOAD(d0, sp, off, 0); LES_STATS_LOAD(d1, sp, off, 16); LES_STATS_LOAD(d2, sp, off, 32);
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
                            if (LES_MARCH_EXP & 2) { pp[S & 1][j] = int4{lane, k, 2, 3}; pm[S & 1][j] = int4{3, 2, k, lane}; px[S & 1][j] = int4{0, 0, 0, 0}; }
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
                    {
                        mstat4 (&ra)[3] = st[LO];
                        mstat4 (&rb)[3] = st[LO + N - 1];
                        if constexpr (N == 2) march_stats_wait6<3 * (BY - 2)>(ra, rb);
                        else march_stats_wait3<3 * (BY - 1)>(ra);
                    }
                    static_for<N>([&](auto jtag) {
                        constexpr int j = decltype(jtag)::value;
                        constexpr int i = LO + j;
                        const int s = pp[CB][j].x - pm[CB][j].x + px[CB][j].x;                         // sum of pi over the window (exact)
                        const int t0c = pp[CB][j].y - pm[CB][j].y + px[CB][j].y, t1c = pp[CB][j].z - pm[CB][j].z + px[CB][j].z, t2c = pp[CB][j].w - pm[CB][j].w + px[CB][j].w;
                        const mstat4 q0 = st[i][0], q1 = st[i][1], q2 = st[i][2];
                        const int M0 = __float_as_int(q2.y), M1 = __float_as_int(q2.z), M2 = __float_as_int(q2.w);
                        // N cov_c in units of 2^SH (u8 * pi): t_c - mean_c * s, the product rounded at 2^-32 of its own scale
                        const float d0 = (float)(t0c - (int)(((long long)M0 * (long long)s + (1ll << 31)) >> 32));
                        const float d1 = (float)(t1c - (int)(((long long)M1 * (long long)s + (1ll << 31)) >> 32));
                        const float d2 = (float)(t2c - (int)(((long long)M2 * (long long)s + (1ll << 31)) >> 32));
                        // LES/GuidedFilter.h:204-221 with the scale of the integer stage 2 folded into the normalisation
                        const float ka = kap_x * rny[i];
                        const float a0 = fmaf(q0.z, d2, fmaf(q0.y, d1, q0.x * d0)) * ka;     // inv00 inv01 inv02
                        const float a1 = fmaf(q1.x, d2, fmaf(q0.w, d1, q0.y * d0)) * ka;     // inv01 inv11 inv12
                        const float a2 = fmaf(q1.y, d2, fmaf(q1.x, d1, q0.z * d0)) * ka;     // inv02 inv12 inv22
                        const float mp = (float)s * (up_x * rny[i]);
                        const float bb = fmaf(-a2, q2.x, fmaf(-a1, q1.w, fmaf(-a0, q1.z, mp)));
                        int4 o;
                        o.x = cvt_rpi_i32(a0);
                        o.y = cvt_rpi_i32(a1);
                        o.z = cvt_rpi_i32(a2);
                        o.w = cvt_rpi_i32(bb);
                        T2[i][pcS] = o;
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
    } else if (role == 2 && (LES_MARCH_ROLE_MASK & 4)) {
        // ================================================= role D =================================================
        const bool out_col = ci >= 2 * R && ci < 2 * R + job.tw && job.th > 0;
        const uint32_t oc4 = (uint32_t)max(ci - 2 * R, 0) * 4u;           // byte offset of the lane's output column in a row of the output tile
        // IsValiLabel (LES/StereoEnergy.h:560-610) without branches: ds = ((x a + y b) + 1 c) + 0 v and the four corner values
        // ds +- 5a +- 5b must all lie in [MIN, MAX]  <=>  min of the five >= MIN and max <= MAX (a NaN only arises next to an
        // infinity, which fails the range test; an all-NaN set fails the comparison itself)
        const float vl_xa = (float)gx * plane.x, vl_c = 1.0f * plane.z, vl_zv = 0.0f * plane.w, vl_a5 = plane.x * 5, vl_b5 = plane.y * 5;
        const double c_lane = view.qscale * s_rtab[nx];                    // 1 / (255 scale count_x)
        int ring2[4][KS];                // horizontal box sums of (a_0, a_1, a_2, b) of the last 2R+1 stage-1 rows
#pragma unroll
        for (int k = 0; k < KS; k++) ring2[0][k] = ring2[1][k] = ring2[2][k] = ring2[3][k] = 0;
        double S2[4] = {0.0, 0.0, 0.0, 0.0};
        uint32_t gq[BY];
        float rny2[BY];                                                   // 1 / count_y of the block's output rows
        uint32_t okbits = 0;
        int nx_grow = 0;
        float nx_rny = 0.0f;
        auto prep = [&](int b) __attribute__((always_inline)) {
            const int t = b * BY + lane;
            const int gy2 = job.ty0 - 4 * R + t;
            nx_rny = (float)s_rtab[window_count(gy2, R, job.cy0, job.cy1)];
            nx_grow = (int)(((uint32_t)min(max(gy2, job.cy0), cy1m) * (uint32_t)g.W) | ((t >= 4 * R && t < Ttot) ? 0x80000000u : 0u));
        };
        auto issue_row = [&](auto itag) __attribute__((always_inline)) {
            constexpr int i = decltype(itag)::value;
            const uint32_t grow = (uint32_t)readlane_i32(nx_grow, i);
            rny2[i] = readlane_f32(nx_rny, i);
            okbits = (okbits & ~(1u << i)) | ((grow >> 31) << i);
            const uint32_t* rg = view.ipk8 + (size_t)(grow & 0x7fffffffu);
            if (LES_MARCH_EXP & 128) { gq[i] = (uint32_t)lane; return; }
            gq[i] = ld_sbase(rg, sx4);
        };
        // two specialisations (label check on / off), selected once per job -- see role A
        auto march_d = [&](auto check_tag) __attribute__((always_inline)) {
        constexpr bool CHECK = decltype(check_tag)::value != 0;
        LES_TICK_BEGIN();
        constexpr int UN = KS / BY;
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
                                    if (LES_MARCH_EXP & 16) { pp[S & 1][j] = int4{lane, b, 2, 3}; pm[S & 1][j] = int4{3, 2, b, lane}; px[S & 1][j] = int4{0, 0, 0, 0}; }
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
                                constexpr int SLOT = BASE + i;
                                const int h0 = pp[CB][j].x - pm[CB][j].x + px[CB][j].x, h1 = pp[CB][j].y - pm[CB][j].y + px[CB][j].y;
                                const int h2 = pp[CB][j].z - pm[CB][j].z + px[CB][j].z, h3 = pp[CB][j].w - pm[CB][j].w + px[CB][j].w;
                                S2[0] += (double)(h0 - ring2[0][SLOT]); ring2[0][SLOT] = h0;
                                S2[1] += (double)(h1 - ring2[1][SLOT]); ring2[1][SLOT] = h1;
                                S2[2] += (double)(h2 - ring2[2][SLOT]); ring2[2][SLOT] = h2;
                                S2[3] += (double)(h3 - ring2[3][SLOT]); ring2[3][SLOT] = h3;
                                if (out_col && ((okbits >> i) & 1u)) {
                                    const int t = b * BY + i;
                                    const uint32_t gi = gq[i];
                                    const double i0 = (double)(((int)(gi << 24)) >> 24), i1 = (double)(((int)(gi << 16)) >> 24), i2 = (double)(((int)(gi << 8)) >> 24);
                                    // LES/GuidedFilter.h:243: (b + a . I) / N on the centred guide, in integers < 2^53
                                    const double acc = fma(S2[2], i2, fma(S2[1], i1, fma(S2[0], i0, S2[3] * 255.0)));
                                    // the 1/count_y factor and the offset are applied in fp32 (relative error 1e-7 of q - vmin)
                                    float q = fmaf((float)(acc * c_lane), rny2[i], view.vmin);
                                    if constexpr (CHECK) {
                                        const int gy2 = job.ty0 + t - 4 * R;
                                        const float ds = ((vl_xa + (float)gy2 * plane.y) + vl_c) + vl_zv;
                                        const float dp = ds + vl_a5, dm = ds - vl_a5;
                                        const float d1 = dp + vl_b5, d2 = dp - vl_b5, d3 = dm + vl_b5, d4 = dm - vl_b5;
                                        const float mn = fmin3(fmin3(ds, d1, d2), d3, d4), mx = fmax3(fmax3(ds, d1, d2), d3, d4);
                                        if (!(mn >= g.mind && mx <= g.maxd)) q = LES_COST_INVALID;
                                    }
                                    if (!((LES_MARCH_EXP & 128) && q != 12345.0f))
                                    st_sbase(out + (job.out_off + (long long)(t - 4 * R) * job.out_stride), oc4, q);
                                }
                            });
                            LES_MARCH_SCHED_FENCE();
                        });
                    }
                    if (k >= 2 && k <= nblk + 1) {                // guide rows of the block this role handles at the next tick (7 dwords per
                        prep(k - 2);                              // lane; this role is never the last to arrive at the barrier, and it has no
                        static_for<BY>([&](auto itag) { issue_row(itag); });   // registers to spare for a rolling issue)
                    }
                    LES_TICK_BARRIER();
                }
            });
        }
        LES_TICK_END(2);
        };
        if (check) march_d(std::integral_constant<int, 1>{});
        else march_d(std::integral_constant<int, 0>{});
    }
}

// ---------------------------------------------------------------------------------------------------
// One-time preparation for the march kernel.
// ---------------------------------------------------------------------------------------------------
// guide as signed bytes + statistics in the march format (from the same fp64 horizontal sums as les_stats_finish_kernel);
// *inv_diag_max (float bits, positive) collects max over pixels of the diagonal of the inverse covariance
__global__ void les_march_stats_kernel(const double* __restrict__ hs, const uint32_t* __restrict__ ipk, uint32_t* __restrict__ ipk8,
                                       float4* __restrict__ mstats, unsigned* __restrict__ inv_diag_max, int H, int W, int R, double eps)
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
    const double c0 = m0 * 255.0 - 128.0, c1 = m1 * 255.0 - 128.0, c2 = m2 * 255.0 - 128.0;    // centred means in u8 units
    const int M0 = (int)rint(c0 * 8388608.0), M1 = (int)rint(c1 * 8388608.0), M2 = (int)rint(c2 * 8388608.0);
    mstats[px * 3 + 0] = make_float4((float)irr, (float)irg, (float)irb, (float)igg);
    mstats[px * 3 + 1] = make_float4((float)igb, (float)ibb, (float)(c0 / 255.0), (float)(c1 / 255.0));
    mstats[px * 3 + 2] = make_float4((float)(c2 / 255.0), __int_as_float(M0), __int_as_float(M1), __int_as_float(M2));
    const uint32_t v = ipk[px];
    const uint32_t b0 = ((v & 0xffu) - 128u) & 0xffu, b1 = (((v >> 8) & 0xffu) - 128u) & 0xffu, b2 = (((v >> 16) & 0xffu) - 128u) & 0xffu;
    ipk8[px] = b0 | (b1 << 8) | (b2 << 16);
    const float dmax = fmaxf(fmaxf((float)irr, (float)igg), (float)ibb);
    if (dmax > 0.0f) atomicMax(inv_diag_max, __float_as_uint(dmax));
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
