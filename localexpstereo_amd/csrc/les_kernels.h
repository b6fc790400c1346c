// les_kernels.h -- HIP kernels of the matching-cost hot path (gfx950 / CDNA4, wave64).
//
// What is computed (reference: LES/CostVolumeEnergy.h:55-183, LES/GuidedFilter.h:142-266,301-326):
//   for a plane (a,b,c), a target rect and a clip rect (== the reference's filterRect):
//     p(s)   = min( lerp_d vol[a*x+b*y+c][y][x], th_col )              s in clip        (gather)
//     Sp, SIp_c = 21x21 zero-padded window sums of p, I_c*p over the clip               (stage 1)
//     a_c, b    = per-pixel 3x3 solve with the precomputed guide statistics
//     q(y)   = ( box(b) + sum_c box(a_c) I_c(y) ) / N(y)                                 (stage 2)
//   q is written for the target pixels only; pixels with an invalid label get 1e6.
//
// Kernel structure ("strip march"): one workgroup owns a strip of TW output columns of one job and
// marches down the rows in blocks of BY rows.  Per block:
//   G   gather p (2 volume taps, coalesced along x) for BY new rows x WP=TW+4R columns -> LDS
//   H1  horizontal 2R+1 sums of (I'_0 p, I'_1 p, I'_2 p, p) : threads own (row, quantity, segment)
//       and slide along x with an fp64 cumulative sum held in a register ring          -> LDS  T[row][x][k]
//   V   threads own (column, quantity) = lane quads; vertical sums by fp64 cumulative register
//       rings that persist across blocks; the 3x3 algebra is distributed over the quad with DPP
//       quad_perm broadcasts; the vertical sums of (a_0,a_1,a_2,b) follow immediately   -> LDS  T (in place)
//   H2  horizontal sums of the 4 stage-2 quantities, weighted by I'_c(y) and reduced over the quad
//   F   coalesced store of q (+ validity overwrite)
// All box sums are accumulated in fp64 (the reference's default "GF" filter is double); only
// 21-term partial sums are rounded to fp32 when they pass through LDS.  The guide is centred
// (I' = I - 1/2: cov and q are exactly invariant to a constant shift of I, for any clip rect), which
// keeps those roundings far below the 1e-4 parity bound (measured: see DESIGN.md "Numerics").
// The cost p itself is NOT centred: that would only be exact where the clipped window of the
// sub-region equals the whole-image window, and the operator must reproduce the reference for any
// (filterRect, targetRect), including targets closer than 2R to the filterRect border.
//
// This header is also compiled by tools/hipsim (CPU fiber simulator, test infrastructure only) with
// LES_SIM defined; nothing in the product build depends on that.
#pragma once

#include "les_simt.h"

#include <utility>

namespace les {

struct Geom {
    int H, W, D, D0;          // D0 = int(-MIN_DISPARITY), LES/CostVolumeEnergy.h:67
    float th_col, pad_;       // truncation threshold
    float maxd, mind;         // MAX_DISPARITY, MIN_DISPARITY
};

struct View {
    const float* vol;          // [D][H][W]   (null for the image-based "naive" matching cost)
    const float4* stats;       // [H*W][3] : {mean_I'_k, inv[k][0], inv[k][1], inv[k][2]}, k = 0..2
    const uint32_t* ipk;       // [H*W] guide pixel packed B | G<<8 | R<<16
    const uint32_t* ipk10;     // [H*W] the same pixel as three signed 10-bit fields 2u - 255 (bits 10k..10k+9): H1's operand format
    // NaiveStereoEnergy (LES/StereoEnergy.h:629-764): 4-channel feature images {(1-alpha) B, G, R, alpha Gx} of this
    // view and of the other one; the raw cost is computed from them instead of a volume when feat_self != null
    const float4* feat_self;
    const float4* feat_other;
    float sign;                // +1 for the left view, -1 for the right one (x_src = x - sign * d)
    float th_color, th_grad;   // (1-alpha) th_col, alpha th_grad   (LES/StereoEnergy.h:662-663)
};

struct Job {
    int tx0, ty0, tw, th;      // target strip (image coordinates), tw <= TW
    int cx0, cy0, cx1, cy1;    // clip rect [cx0,cx1) x [cy0,cy1) == reference filterRect
    long long out_off;         // float offset of output element (ty0, tx0)
    int out_stride;            // floats per output row
    int plane_idx;             // index into the planes array of this launch (LES/Plane.h labels)
};

#define LES_COST_INVALID 1000000.0f

// ---------------------------------------------------------------------------------------------------
// LES/CostVolumeEnergy.h:70-98 (interpolate == 1).  Compiled with -ffp-contract=off: d must be the
// un-fused float expression a*x + (b*y + c) so that int(d) and the validity thresholds match the CPU.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gather_cost(const Geom& g, const float* __restrict__ vol, float a, float b,
                                             float c, int gx, int gy)
{
    const size_t HW = (size_t)g.H * g.W;
    const size_t px = (size_t)gy * g.W + gx;
    float d_base = b * (float)gy + c;
    float d = a * (float)gx + d_base;
    float C;
    if (d < g.mind) C = vol[px];
    else if (d >= g.maxd) C = vol[(size_t)(g.D - 1) * HW + px];
    else if (d != d || fabsf(d) == INFINITY) C = LES_COST_INVALID;
    else {
        int d0 = (int)d + g.D0;
        int d1 = d0 + 1;
        float f1 = d - floorf(d);
        float f0 = 1.0f - f1;
        if (d1 >= g.D || d0 < 0) C = LES_COST_INVALID;
        else {
            float v0 = vol[(size_t)d0 * HW + px];
            float v1 = vol[(size_t)d1 * HW + px];
            C = f0 * v0 + f1 * v1;
        }
    }
    return (g.th_col < C) ? g.th_col : C;       // std::min(C, th_col), NaN-propagating like the reference
}

// The same computation split into an address phase and an arithmetic phase so that a thread can put
// the loads of several pixels in flight before it needs any of them (branch-free, bit-identical).
// Element offsets are 32-bit (the context refuses volumes of 2^32 or more floats).
struct GatherPrep {
    uint32_t i0, i1;    // element offsets of the two volume taps (i1 == i0 when only one tap is used)
    float f1;           // lerp weight of the second tap
    int mode;           // 0: C = f0*v0 + f1*v1   1: C = v0 (clamped)   2: C = COST_FOR_INVALID   3: outside clip -> p = 0
};
// ax = a * x and d_base = b * y + c are the reference's own sub-expressions (LES/CostVolumeEnergy.h:73,76)
__device__ __forceinline__ GatherPrep gather_prepare(const Geom& g, float ax, float d_base, uint32_t px, uint32_t HW, bool inside)
{
    GatherPrep r;
    const float d = ax + d_base;
    const bool lo = d < g.mind, hi = d >= g.maxd;
    const bool mid = !(lo || hi || d != d);
    // (int)d is only meaningful on the `mid` path; substitute 0 so the conversion is always defined
    const float dsafe = mid ? d : 0.0f;
    const int d0 = (int)dsafe + g.D0;
    const bool ok = mid && d0 >= 0 && d0 + 1 < g.D;
    r.f1 = dsafe - floorf(dsafe);
    const uint32_t s0 = lo ? 0u : (hi ? (uint32_t)(g.D - 1) : (ok ? (uint32_t)d0 : 0u));
    r.i0 = s0 * HW + px;
    r.i1 = ok ? r.i0 + HW : r.i0;
    r.mode = !inside ? 3 : ((lo || hi) ? 1 : (ok ? 0 : 2));
    return r;
}
__device__ __forceinline__ float gather_finish(const Geom& g, const GatherPrep& r, float v0, float v1)
{
    const float f0 = 1.0f - r.f1;
    float C = f0 * v0 + r.f1 * v1;
    C = r.mode == 1 ? v0 : C;
    C = r.mode == 2 ? LES_COST_INVALID : C;
    const float p = (g.th_col < C) ? g.th_col : C;
    return r.mode == 3 ? 0.0f : p;
}

// NaiveStereoEnergy raw cost, LES/StereoEnergy.h:702-742: the other view is sampled at x - sign * d on the same row with
// the bilinear interpolation of cv::warpAffine (source coordinate rounded to 1/32 pixel, replicated border).
struct NaivePrep {
    uint32_t ia, ib;    // element offsets of the two feature taps in the other view
    float w1;
};
__device__ __forceinline__ NaivePrep naive_prepare(const Geom& g, float sign, float a, float b, float c, int gx, int gy)
{
    NaivePrep r;
    const float z = (a * (float)gx + b * (float)gy) + c;
    const double sx = (double)gx - (double)sign * (double)z;
    const double q = floor(sx * 32.0 + 0.5) / 32.0;
    const double fl = floor(q);
    r.w1 = (float)(q - fl);
    const double flc = fmin(fmax(fl, -2.0), (double)g.W + 1.0);       // NaN -> -2: defined conversion, weights stay NaN
    const int x0 = (int)flc;
    const int xa = min(max(x0, 0), g.W - 1), xb = min(max(x0 + 1, 0), g.W - 1);
    r.ia = (uint32_t)gy * (uint32_t)g.W + (uint32_t)xa;
    r.ib = (uint32_t)gy * (uint32_t)g.W + (uint32_t)xb;
    return r;
}
__device__ __forceinline__ float naive_finish(const View& v, const NaivePrep& r, float4 own, float4 fa, float4 fb)
{
    const float w0 = 1.0f - r.w1;
    const float v0 = w0 * fa.x + r.w1 * fb.x, v1 = w0 * fa.y + r.w1 * fb.y, v2 = w0 * fa.z + r.w1 * fb.z, v3 = w0 * fa.w + r.w1 * fb.w;
    const float col = (fabsf(own.x - v0) + fabsf(own.y - v1)) + fabsf(own.z - v2);
    const float grad = fabsf(own.w - v3);
    return ((col < v.th_color) ? col : v.th_color) + ((grad < v.th_grad) ? grad : v.th_grad);      // std::min(th, x)
}

// LES/StereoEnergy.h:560-610
__device__ __forceinline__ bool label_valid(const Geom& g, float a, float b, float c, float v, int gx, int gy)
{
    float fx = (float)gx, fy = (float)gy;
    float ds = ((fx * a + fy * b) + 1.0f * c) + 0.0f * v;
    float a5 = a * 5, b5 = b * 5;
    float d;
    return (ds >= g.mind && ds <= g.maxd
            && ((d = ds + a5 + b5) >= g.mind) && d <= g.maxd
            && ((d = ds + a5 - b5) >= g.mind) && d <= g.maxd
            && ((d = ds - a5 + b5) >= g.mind) && d <= g.maxd
            && ((d = ds - a5 - b5) >= g.mind) && d <= g.maxd);
}

// I'_k = I_k - 1/2 with I_k = u8/255 (LES/GuidedFilter.h:62-65 scaling), k = 0..2; k = 3 -> 1
__device__ __forceinline__ float guide_centred_f32(uint32_t ipk, int k)
{
    int u = (int)((ipk >> (8 * k)) & 0xffu);
    float w = (float)(2 * u - 255) * (1.0f / 510.0f);
    return k == 3 ? 1.0f : w;
}
// the same value from the signed 10-bit field format (one bit-field extract instead of shift / mask / shift / add)
// H1 form without the k == 3 select: `off` = 10 * k for k < 3; the k == 3 lanes read a constant word whose field 0 is 510, and
// (float)510 * (1/510.f) == 1.0f exactly
__device__ __forceinline__ float guide10_field_f32(uint32_t w10, int off)
{
#if defined(LES_SIM)
    const int v = ((int)(w10 << (22 - off))) >> 22;
#else
    const int v = __builtin_amdgcn_sbfe((int)w10, (unsigned)off, 10u);
#endif
    return (float)v * (1.0f / 510.0f);
}
__device__ __forceinline__ float guide10_centred_f32(uint32_t w10, int k)
{
#if defined(LES_SIM)
    const int v = ((int)(w10 << (22 - 10 * (k < 3 ? k : 0)))) >> 22;
#else
    const int v = __builtin_amdgcn_sbfe((int)w10, (unsigned)(10 * k), 10u);
#endif
    const float w = (float)v * (1.0f / 510.0f);
    return k == 3 ? 1.0f : w;
}
__device__ __forceinline__ double guide_centred_f64(uint32_t ipk, int k)
{
    int u = (int)((ipk >> (8 * k)) & 0xffu);
    double w = fma((double)u, 1.0 / 255, -0.5);
    return k == 3 ? 1.0 : w;
}

__device__ __forceinline__ int window_count(int c, int R, int lo, int hi)
{   // number of integers in [c-R, c+R] intersected with [lo, hi)
    int a = c - R < lo ? lo : c - R;
    int b = c + R + 1 > hi ? hi : c + R + 1;
    return b > a ? b - a : 0;
}

// The H chains are fully unrolled; without a fence the scheduler hoists dozens of LDS loads to the top and
// the register footprint (and with it the occupancy) is set by that phase alone.
#if defined(LES_SIM)
#define LES_SCHED_FENCE(step) ((void)0)
#else
#define LES_SCHED_FENCE(step) do { if (((step) & 3) == 3) __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
// In phase V the rows are overlapped in groups of 3 (more would spill at the 168-VGPR occupancy target).
#if defined(LES_SIM)
#define LES_SCHED_FENCE_V(row) ((void)0)
#else
#define LES_SCHED_FENCE_V(row) do { if (((row) % 3) == 2) __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

#ifndef LES_H1_F32
typedef double H1ACC;
#else
typedef float H1ACC;      // experiment: fp32 running sums in the H1 chains (their output is rounded to fp32 anyway)
#endif

template <int V>
struct IntTag { static constexpr int value = V; };

// compile-time loop: f(IntTag<0>{}), ..., f(IntTag<N-1>{})
template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f)
{
    (f(IntTag<Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// ---------------------------------------------------------------------------------------------------
// Vertical phase state of one (column, quantity) lane.  The window sums are exact running sums in
// fp64 of fp32 inputs: S += in - out, with the last RS >= 2R+1 inputs held in a statically indexed
// register ring (slot = row mod RS, RS a multiple of the block height so slots are compile-time).
// ---------------------------------------------------------------------------------------------------
template <int R, int RS>
struct VLane {
    static constexpr int KS = 2 * R + 1;
    static_assert(RS >= KS, "ring shorter than the window");
    float ring1[RS] = {};
    float ring2[RS] = {};
    double S1 = 0.0, S2 = 0.0;
    int k;               // quantity: 0..2 -> I'_k p / a_k, 3 -> p / b

    // One row.  S = row index mod RS (compile time): the ring keeps the last RS inputs, the one that
    // leaves the 2R+1 window is the input of KS rows ago.
    //   in  : H1 sum of this lane's quantity for the incoming p-row
    //   rn1 : 1/N of the stage-1 pixel;  st: its statistics {mean_I'_k, inv[k][0..2]} (used when in_clip && k < 3)
    // returns the vertical stage-2 sum (centred R rows above the stage-1 row)
    template <int S>
    __device__ __forceinline__ float step(float in, bool keep, double rn1, float4 s)
    {
        constexpr int OLD = (S + RS - KS) % RS;
        S1 += (double)in - (double)ring1[OLD];             // sum over the last 2R+1 p-rows
        ring1[S] = in;
        // LES/GuidedFilter.h:204-221 on the centred guide.  Straight-line code: consecutive rows only meet in S1 / S2.
        // `s` = statistics of the stage-1 pixel for lanes k < 3, the all-zero entry for lane 3 (so lane 3 contributes
        // nothing to b's correction and its cov is the plain mean); pixels outside the clip use the statistics of the
        // clamped pixel -- finite numbers -- and their result is discarded by `keep`.
        const double m = S1 * rn1;                         // lane k<3: mean(I'_k p), lane 3: mean(p)
        const double mp = quad_bcast<3>(m);
        const float cov = (float)fma(-(double)s.x, mp, m); // cov_k = mean(I'_k p) - mean_I'_k * mean_p  (one rounding: explicit fma)
        const float ak = fmaf(s.w, quad_bcast<2>(cov), fmaf(s.z, quad_bcast<1>(cov), s.y * quad_bcast<0>(cov)));
        // b = mean_p - sum_k a_k mean_I'_k : lanes 0..2 contribute a_k * mean_I'_k, lane 3 contributes 0
        const float bb = (float)mp - quad_sum(ak * s.x);
        float val = (k < 3) ? ak : bb;
        if (!keep) val = 0.0f;                             // a, b are zero-padded outside the sub-region and before the march is primed
        S2 += (double)val - (double)ring2[OLD];            // sum over the last 2R+1 stage-1 rows
        ring2[S] = val;
        return (float)S2;
    }
};

// ---------------------------------------------------------------------------------------------------
// The fused strip kernel.   NT = 4*WA threads; TW = WA-2R output columns per strip; the H phases use
// 4*BY*SEG lanes (row, quantity, segment).
// ---------------------------------------------------------------------------------------------------
template <int R, int WA, int BY, int SEG>
struct StripCfg {
    static constexpr int TW = WA - 2 * R;
    static constexpr int WP = WA + 2 * R;
    static constexpr int NT = 4 * WA;
    static constexpr int HL = 4 * BY * SEG;                // lanes active in the H phases
    static constexpr int L1 = (WA + SEG - 1) / SEG;        // H1 outputs per segment
    static constexpr int L2 = (TW + SEG - 1) / SEG;        // H2 outputs per segment
    static constexpr int TPITCH = WA * 4 + 4;              // floats per T row (+4: bank spread across rows)
    static constexpr int KS = 2 * R + 1;
    static constexpr int RS = ((KS + BY - 1) / BY) * BY;   // V ring length: multiple of BY, so a block's rows hit static slots
    static_assert(HL <= NT, "more H lanes than threads");
    static_assert(TW > 0, "strip too narrow for this radius");
};

// -DLES_PHASE_TIMING (tools/phase_probe.py builds such a variant next to the product library): lane 0 of every workgroup
// accumulates the cycles between the barriers of the march; read back with les_hip_debug_phases().
#if defined(LES_PHASE_TIMING) && !defined(LES_SIM)
__device__ unsigned long long les_dbg[12];
#define LES_PHASE_BEGIN() unsigned long long ph_[5] = {0, 0, 0, 0, 0}; unsigned long long tl_ = clock64()
#define LES_PHASE_MARK(k) do { const unsigned long long now_ = clock64(); ph_[k] += now_ - tl_; tl_ = now_; } while (0)
#define LES_PHASE_END() do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 5; k_++) atomicAdd(&les_dbg[k_], ph_[k_]); atomicAdd(&les_dbg[5], 1ull); } } while (0)
#else
#define LES_PHASE_BEGIN() ((void)0)
#define LES_PHASE_MARK(k) ((void)0)
#define LES_PHASE_END() ((void)0)
#endif

template <int R, int WA, int BY, int SEG, int MW, int SRC = 0>       // SRC 0: cost volume, 1: image-based matching cost
__global__ void __launch_bounds__(4 * WA, MW)
les_strip_kernel(Geom g, View view, const Job* __restrict__ jobs, const float4* __restrict__ planes,
                 float* __restrict__ out, int njobs, int check)
{
    using Cfg = StripCfg<R, WA, BY, SEG>;
    constexpr int TW = Cfg::TW, WP = Cfg::WP, NT = Cfg::NT, HL = Cfg::HL, L1 = Cfg::L1, L2 = Cfg::L2;
    constexpr int TPITCH = Cfg::TPITCH, KS = Cfg::KS, RS = Cfg::RS;

    __shared__ float s_p[BY][WP];            // truncated cost p (0 outside the clip); dead after H1 ...
    __shared__ uint32_t s_ipk[BY][WP];       // guide pixel of the same p-rows (signed 10-bit field format)
    __shared__ float s_T[BY][TPITCH];        // H1 sums, then (in place) vertical stage-2 sums
    __shared__ uint32_t s_ipk2[BY][TW];      // packed guide pixel of the output rows of this block
    __shared__ double s_rtab[2 * R + 2];     // 1/n, n = 0..2R+1
    __shared__ uint32_t s_c510[WP];          // constant "guide" row of the k = 3 lanes in H1: field 0 = 510, i.e. weight 1
    // per-row scalars of phase V for the current block (written by BY lanes in phase G): without the table every lane
    // quad re-derives them on the scalar unit -- 27 SALU instructions per row that stall the three resident waves
    struct RowInfo { double rny; int flags; uint32_t soff; };      // 1/count_y; bit 0: row in clip, bit 1: t >= 2R; stats row offset of row t + PD
    __shared__ RowInfo s_row[BY];
    // per-row terms of phase G (row of the clip the p-row maps to): pixel offset of the row, d_base = b*y + c
    // (LES/CostVolumeEnergy.h:73), whether the row lies inside the clip and the march, the clamped row itself
    struct GRow { uint32_t rowpx; float d_base; };                  // bit 31 of rowpx: row inside the clip and the march
    __shared__ GRow s_grow[BY];
    __shared__ double s_rnx2[TW + 4];        // 1/count_x of the strip's output columns (constant for the whole job; +4: padding columns)
    __shared__ int s_gsy[BY];                                       // the clamped row itself (image-based matching cost only)
    float (*s_q)[WP] = s_p;                  // ... so the finished q of the block's output rows reuses it (H2 -> F)

    // XCD-aware job order (guide T1): consecutive jobs (same strip, consecutive planes / neighbouring
    // cells) run on the same XCD so that guide statistics and volume halos are shared in its L2.
    int job_id;
    {
        const int nwg = (int)gridDim.x, orig = (int)blockIdx.x;
        const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        job_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    }
    if (job_id >= njobs) return;
    const Job job = jobs[job_id];
    const float4 plane = planes[job.plane_idx];          // (a, b, c, v)
    const int tid = (int)threadIdx.x;
    const int Ttot = job.th + 4 * R;         // p-rows to march over

    if (tid < 2 * R + 2) s_rtab[tid] = tid > 0 ? 1.0 / (double)tid : 0.0;
    if (tid < WP) s_c510[tid] = 510u;

    // ---- V-phase identity: lane quad = 4 quantities of stage-1 column vx
    const int vx = tid >> 2, vk = tid & 3;
    VLane<R, RS> vl;
    vl.k = vk;
    const int gx1 = job.tx0 - R + vx;                                   // stage-1 column (image coords)
    const bool col_in_clip = gx1 >= job.cx0 && gx1 < job.cx1;
    const int nx1 = window_count(gx1, R, job.cx0, job.cx1);

    // ---- H-phase identity: (segment, row, quantity); lanes of a wave differ in row first
    const bool h_active = tid < HL;
    const int hk = tid & 3, hrow = (tid >> 2) % BY, hseg = (tid >> 2) / BY;
    const uint32_t* h1_guide = hk < 3 ? &s_ipk[hrow][0] : &s_c510[0];   // H1: guide words of this lane's row, or the constant row
    const int h1_off = hk < 3 ? 10 * hk : 0;

    __syncthreads();
    const double rnx1 = s_rtab[nx1];

    // ---- G-phase identity: a lane keeps one p column (and one output column) for the whole job
    constexpr int GRP = NT / WP, GPASS = (BY + GRP - 1) / GRP;          // rows per pass, passes per block
    constexpr int ORP = NT / TW, OPASS = (BY + ORP - 1) / ORP;
    const int g_ri = tid / WP, g_xi = tid - g_ri * WP;
    const bool g_lane = tid < GRP * WP;
    const int g_gx = job.tx0 - 2 * R + g_xi;
    const bool g_col_in = g_gx >= job.cx0 && g_gx < job.cx1;
    const int g_sx = min(max(g_gx, job.cx0), job.cx1 - 1);
    const float g_ax = plane.x * (float)g_sx;                           // a * x, LES/CostVolumeEnergy.h:76
    const int o_ri = tid / TW, o_xo = tid - o_ri * TW;
    const bool o_lane = tid < ORP * TW;
    const int o_sx = min(max(job.tx0 + o_xo, job.cx0), job.cx1 - 1);
    const uint32_t HWu = (uint32_t)g.H * (uint32_t)g.W;

    // Guide statistics of the stage-1 pixels are prefetched PD rows ahead into registers (the pipeline
    // runs across block boundaries), so their L2/HBM latency never sits on the V-phase critical path.
    constexpr int PD = (BY % 4 == 0) ? 4 : ((BY % 3 == 0) ? 3 : 1);      // must divide BY (static register names)
    const int sgx1 = min(max(gx1, job.cx0), job.cx1 - 1);
    // lanes k < 3 walk the statistics of their stage-1 column; lane 3 (the p / b quantity) always reads the all-zero entry
    // that follows the table
    const float4* st_col = vk < 3 ? view.stats + (size_t)sgx1 * 3 + vk : view.stats + (size_t)g.H * g.W * 3;
    const size_t st_stride = (size_t)g.W * 3;
    auto stats_row = [&](int t) -> float4 {
        const int gy1 = min(max(job.ty0 - 3 * R + t, job.cy0), job.cy1 - 1);
        return st_col[(vk < 3 ? (size_t)gy1 : (size_t)0) * st_stride];
    };
    float4 pre[PD];
#pragma unroll
    for (int j = 0; j < PD; j++) pre[j] = stats_row(j);

    // Row tables of the block that starts at p-row tb, written by BY lanes one block ahead (block 0: here, before the
    // barrier below; block n+1: in phase F of block n), so no phase re-derives per-row scalars lane by lane.
    auto fill_row_tables = [&](int tb) {
        const int t = tb + tid;
        const int gy1 = job.ty0 - 3 * R + t;                           // stage-1 row (phase V)
        RowInfo ri;
        ri.rny = s_rtab[window_count(gy1, R, job.cy0, job.cy1)];
        ri.flags = ((gy1 >= job.cy0 && gy1 < job.cy1) ? 1 : 0) | (t >= 2 * R ? 2 : 0);
        ri.soff = (uint32_t)min(max(gy1 + PD, job.cy0), job.cy1 - 1) * (uint32_t)(g.W * 3);
        s_row[tid] = ri;
        const int gy = job.ty0 - 2 * R + t;                            // p-row (phase G)
        GRow gr;
        const int sy = min(max(gy, job.cy0), job.cy1 - 1);
        gr.rowpx = ((uint32_t)sy * (uint32_t)g.W) | ((t < Ttot && gy >= job.cy0 && gy < job.cy1) ? 0x80000000u : 0u);
        gr.d_base = plane.y * (float)sy + plane.z;
        s_grow[tid] = gr;
        if constexpr (SRC == 1) s_gsy[tid] = sy;
    };
    if (tid < BY) fill_row_tables(0);
    if (tid < TW + 4) s_rnx2[tid] = s_rtab[window_count(job.tx0 + tid, R, job.cx0, job.cx1)];
    __syncthreads();

    LES_PHASE_BEGIN();
    for (int t0 = 0; t0 < Ttot; t0 += BY) {
        // ===================== G: gather =====================
        // A lane keeps its column for the whole job (column terms hoisted out of the march); the loads of
        // up to GB rows per lane are issued before any is consumed (memory-level parallelism, bounded so
        // that the register footprint stays small).
        {
            if constexpr (SRC == 1) {
                // image-based matching cost (one row per lane in flight keeps the register footprint below the volume path's)
#pragma unroll 1
                for (int jp = 0; jp < GPASS; jp++) {
                    const int i = jp * GRP + g_ri;
                    const GRow gr = s_grow[i < BY ? i : BY - 1];
                    const bool inside = g_lane && g_col_in && i < BY && (gr.rowpx >> 31);
                    const int sy = s_gsy[i < BY ? i : BY - 1];
                    const uint32_t px = (gr.rowpx & 0x7fffffffu) + (uint32_t)g_sx;
                    const NaivePrep np = naive_prepare(g, view.sign, plane.x, plane.y, plane.z, g_sx, sy);
                    const float4 own = view.feat_self[px], fa = view.feat_other[np.ia], fb = view.feat_other[np.ib];
                    const uint32_t ip = view.ipk10[px];
                    if (g_lane && i < BY) {
                        s_p[i][g_xi] = inside ? naive_finish(view, np, own, fa, fb) : 0.0f;
                        s_ipk[i][g_xi] = ip;
                    }
                }
            } else {
                constexpr int GB = 7;
                static_for<(GPASS + GB - 1) / GB>([&](auto btag) {
                    constexpr int J0 = decltype(btag)::value * GB;
                    constexpr int JN = (GPASS - J0) < GB ? (GPASS - J0) : GB;
                    GatherPrep gp[JN];
                    float v0[JN], v1[JN];
                    uint32_t ipa[JN];
#pragma unroll
                    for (int j = 0; j < JN; j++) {
                        const int i = (J0 + j) * GRP + g_ri;
                        const GRow gr = s_grow[i < BY ? i : BY - 1];
                        const bool inside = g_lane && g_col_in && i < BY && (gr.rowpx >> 31);
                        const uint32_t px = (gr.rowpx & 0x7fffffffu) + (uint32_t)g_sx;
                        gp[j] = gather_prepare(g, g_ax, gr.d_base, px, HWu, inside);
                        v0[j] = view.vol[gp[j].i0];
                        v1[j] = view.vol[gp[j].i1];
                        ipa[j] = view.ipk10[px];
                    }
#pragma unroll
                    for (int j = 0; j < JN; j++) {
                        const int i = (J0 + j) * GRP + g_ri;
                        if (g_lane && i < BY) {
                            s_p[i][g_xi] = gather_finish(g, gp[j], v0[j], v1[j]);
                            s_ipk[i][g_xi] = ipa[j];
                        }
                    }
                });
            }
            uint32_t ipo[OPASS];
#pragma unroll
            for (int j = 0; j < OPASS; j++) {
                const int i = j * ORP + o_ri;
                const int sy = min(max(job.ty0 + t0 + i - 4 * R, job.cy0), job.cy1 - 1);
                ipo[j] = view.ipk[(uint32_t)sy * (uint32_t)g.W + (uint32_t)o_sx];
            }
#pragma unroll
            for (int j = 0; j < OPASS; j++) {
                const int i = j * ORP + o_ri;
                if (o_lane && i < BY) s_ipk2[i][o_xo] = ipo[j];
            }
        }
        __syncthreads();
        LES_PHASE_MARK(0);

        // ===================== H1: horizontal sums of F_k = I'_k p =====================
        if (h_active) {
            float ring[KS];
#pragma unroll
            for (int i = 0; i < KS; i++) ring[i] = 0.0f;
            H1ACC S = 0;
            const int x0 = hseg * L1;
#pragma unroll
            for (int s = 0; s < L1 + 2 * R; s++) {
                const int xi = x0 + s;                                  // p column index
                const float f = xi < WP ? guide10_field_f32(h1_guide[xi], h1_off) * s_p[hrow][xi] : 0.0f;
                S += (H1ACC)f - (H1ACC)ring[s % KS];
                ring[s % KS] = f;
                if (s >= 2 * R && x0 + s - 2 * R < WA) s_T[hrow][(x0 + s - 2 * R) * 4 + hk] = (float)S;
                LES_SCHED_FENCE(s);
            }
        }
        __syncthreads();
        LES_PHASE_MARK(1);

        // ===================== V: vertical sums, algebra, vertical sums =====================
        // Rows of a block map to compile-time ring slots: slot = (t0 % RS) + i, and t0 % RS takes only
        // RS/BY distinct values, so the whole block is straight-line code selected by one uniform branch.
        {
            const float* trow = &s_T[0][vx * 4 + vk];
            auto vblock = [&](auto base_tag) {
                constexpr int BASE = decltype(base_tag)::value;
                static_for<BY>([&](auto itag) {
                    constexpr int i = decltype(itag)::value;
                    const RowInfo ri = s_row[i];                        // uniform address: one broadcast 16-byte LDS read
                    const double rn1 = rnx1 * ri.rny;
                    const float o = vl.template step<BASE + i>(trow[i * TPITCH], col_in_clip && ri.flags == 3, rn1, pre[i % PD]);
                    pre[i % PD] = st_col[vk < 3 ? ri.soff : 0u];
                    s_T[i][vx * 4 + vk] = o;
                    LES_SCHED_FENCE_V(i);
                });
            };
            const int base = t0 % RS;
            if constexpr (RS / BY == 1) vblock(IntTag<0>{});
            else if constexpr (RS / BY == 2) { if (base == 0) vblock(IntTag<0>{}); else vblock(IntTag<BY>{}); }
            else if constexpr (RS / BY == 3) { if (base == 0) vblock(IntTag<0>{}); else if (base == BY) vblock(IntTag<BY>{}); else vblock(IntTag<2 * BY>{}); }
            else {
                static_assert(RS / BY == 4, "unsupported ring/block ratio");
                if (base == 0) vblock(IntTag<0>{}); else if (base == BY) vblock(IntTag<BY>{});
                else if (base == 2 * BY) vblock(IntTag<2 * BY>{}); else vblock(IntTag<3 * BY>{});
            }
        }
        __syncthreads();
        LES_PHASE_MARK(2);

        // ===================== H2: horizontal sums + guide weighting + quad reduction =====================
        if (h_active) {
            const int t = t0 + hrow;
            const bool row_ok = t >= 4 * R && t < Ttot;                 // uniform per (row) quad
            const int gy2 = job.ty0 + t - 4 * R;
            const double rny2 = s_rtab[window_count(gy2, R, job.cy0, job.cy1)];
            float ring[KS];
#pragma unroll
            for (int i = 0; i < KS; i++) ring[i] = 0.0f;
            double S = 0.0;
            const int x0 = hseg * L2;
#pragma unroll
            for (int s = 0; s < L2 + 2 * R; s++) {
                const int xa = x0 + s;                                  // stage-1 column index
                const float f = xa < WA ? s_T[hrow][xa * 4 + hk] : 0.0f;
                S += (double)f - (double)ring[s % KS];
                ring[s % KS] = f;
                if (s >= 2 * R) {
                    const int xo = x0 + s - 2 * R;                      // output column of the strip
                    const uint32_t ip = xo < TW ? s_ipk2[hrow][xo] : 0u;
                    // LES/GuidedFilter.h:243: (b + a_r I_r + a_g I_g + a_b I_b) / N
                    const double qn = quad_sum(S * guide_centred_f64(ip, hk));
                    // branch-free: every lane stores; the lanes that do not own a result (k != 0, padding columns, rows outside
                    // the job) write into the unused tail of their q row instead of being masked by a divergent region per output
                    const double rn2 = rny2 * s_rnx2[xo < TW + 4 ? xo : TW + 3];
                    const int col = (hk == 0 && xo < TW && row_ok) ? xo : TW + hk;
                    s_q[hrow][col] = (float)(qn * rn2);
                }
                LES_SCHED_FENCE(s);
            }
        }
        __syncthreads();
        LES_PHASE_MARK(3);

        // ===================== F: store =====================
        if (tid < BY) fill_row_tables(t0 + BY);
        for (int idx = tid; idx < BY * TW; idx += NT) {
            const int i = idx / TW, xo = idx - i * TW;
            const int t = t0 + i;
            if (t >= 4 * R && t < Ttot && xo < job.tw) {
                const int gy2 = job.ty0 + t - 4 * R, gx2 = job.tx0 + xo;
                float q = s_q[i][xo];
                if (check && !label_valid(g, plane.x, plane.y, plane.z, plane.w, gx2, gy2)) q = LES_COST_INVALID;
                out[job.out_off + (long long)(t - 4 * R) * job.out_stride + xo] = q;
            }
        }
        __syncthreads();
        LES_PHASE_MARK(4);
    }
    LES_PHASE_END();
}

// ---------------------------------------------------------------------------------------------------
// Guide statistics (one-time per view): LES/GuidedFilter.h:58-102 in fp64, stored as fp32 AoS.
// Pass 1: horizontal 2R+1 sums of the 9 moments (I_c, I_a I_b); pass 2: vertical sums + inverse.
// ---------------------------------------------------------------------------------------------------
__global__ void les_pack_guide_kernel(const uint8_t* __restrict__ bgr, uint32_t* __restrict__ ipk, uint32_t* __restrict__ ipk10, int P)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= P) return;
    const uint32_t b = bgr[3 * (size_t)i], g = bgr[3 * (size_t)i + 1], r = bgr[3 * (size_t)i + 2];
    ipk[i] = b | (g << 8) | (r << 16);
    ipk10[i] = ((2u * b - 255u) & 1023u) | (((2u * g - 255u) & 1023u) << 10) | (((2u * r - 255u) & 1023u) << 20);
}

__global__ void les_stats_hsum_kernel(const uint32_t* __restrict__ ipk, double* __restrict__ hs, int H, int W, int R)
{
    int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), y = (int)blockIdx.y;
    if (x >= W) return;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int dx = -R; dx <= R; dx++) {
        int xx = x + dx;
        if (xx < 0 || xx >= W) continue;
        uint32_t v = ipk[(size_t)y * W + xx];
        double I0 = (double)(v & 0xff) * (1.0 / 255), I1 = (double)((v >> 8) & 0xff) * (1.0 / 255),
               I2 = (double)((v >> 16) & 0xff) * (1.0 / 255);
        s[0] += I0; s[1] += I1; s[2] += I2;
        s[3] += I0 * I0; s[4] += I0 * I1; s[5] += I0 * I2; s[6] += I1 * I1; s[7] += I1 * I2; s[8] += I2 * I2;
    }
    size_t P = (size_t)H * W;
    for (int k = 0; k < 9; k++) hs[k * P + (size_t)y * W + x] = s[k];
}

__global__ void les_stats_finish_kernel(const double* __restrict__ hs, float4* __restrict__ stats, int H, int W, int R, double eps)
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
    double N = (double)(window_count(x, R, 0, W) * window_count(y, R, 0, H));     // :69
    double m0 = s[0] / N, m1 = s[1] / N, m2 = s[2] / N;                             // :70-72
    double rr = s[3] / N - m0 * m0 + eps, rg = s[4] / N - m0 * m1, rb = s[5] / N - m0 * m2;   // :79-84
    double gg = s[6] / N - m1 * m1 + eps, gb = s[7] / N - m1 * m2, bb = s[8] / N - m2 * m2 + eps;
    double irr = gg * bb - gb * gb, irg = gb * rb - rg * bb, irb = rg * gb - gg * rb;         // :87-92
    double igg = rr * bb - rb * rb, igb = rb * rg - rr * gb, ibb = rr * gg - rg * rg;
    double det = irr * rr + irg * rg + irb * rb;                                              // :94
    irr /= det; irg /= det; irb /= det; igg /= det; igb /= det; ibb /= det;                   // :96-101
    size_t px = (size_t)y * W + x;
    stats[px * 3 + 0] = make_float4((float)(m0 - 0.5), (float)irr, (float)irg, (float)irb);
    stats[px * 3 + 1] = make_float4((float)(m1 - 0.5), (float)irg, (float)igg, (float)igb);
    stats[px * 3 + 2] = make_float4((float)(m2 - 0.5), (float)irb, (float)igb, (float)ibb);
}

// NaiveStereoEnergy constructor, LES/StereoEnergy.h:644-664: feature image {(1-alpha) B, (1-alpha) G, (1-alpha) R,
// alpha Gx}, Gx = 0.5 * (gray(x+1) - gray(x-1)) with replicated border, gray = 0.114 B + 0.587 G + 0.299 R.
__global__ void les_naive_features_kernel(const uint8_t* __restrict__ img, float4* __restrict__ feat, int H, int W, float alpha)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    auto gray = [&](int xx) {
        const uint8_t* p = img + ((size_t)y * W + xx) * 3;
        return ((float)p[0] * 0.114f + (float)p[1] * 0.587f) + (float)p[2] * 0.299f;
    };
    const float gx = 0.5f * (gray(min(x + 1, W - 1)) - gray(max(x - 1, 0)));
    const uint8_t* p = img + ((size_t)y * W + x) * 3;
    const double k = 1.0 - (double)alpha;
    float4 f;
    f.x = (float)((double)p[0] * k); f.y = (float)((double)p[1] * k); f.z = (float)((double)p[2] * k);
    f.w = gx * alpha;
    feat[(size_t)y * W + x] = f;
}

// Raw image-based matching cost of whole calls (LES/StereoEnergy.h:686-719, the loop that fills the cost patch before the
// guided filter): call i evaluates its plane on every pixel of its filterRect and stores the patch compactly at raw + off.
// The march kernel (les_march.h, role A KIND 3) then reads that patch like one slice of a cost volume, so the image-based
// energy runs on the same fixed-point filter as the volume-based one.  Every value lies in [0, th_color + th_grad] (a
// comparison with a NaN operand selects the threshold), which is the range the march kernel's fixed point is scaled for.
struct RawCall {
    int fx, fy, fw, fh;        // filterRect
    long long off;             // float offset of its patch
};
__global__ void les_naive_raw_kernel(Geom g, View view, const RawCall* __restrict__ calls, const float4* __restrict__ planes,
                                     float* __restrict__ raw)
{
    const RawCall rc = calls[blockIdx.x];
    const float4 plane = planes[blockIdx.x];
    const long long area = (long long)rc.fw * rc.fh;
    const long long per = (area + gridDim.y - 1) / gridDim.y;
    const long long p0 = per * blockIdx.y, p1 = p0 + per < area ? p0 + per : area;
    for (long long p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const int yy = (int)(p / rc.fw), xx = (int)(p - (long long)yy * rc.fw);
        const int gx = rc.fx + xx, gy = rc.fy + yy;
        const NaivePrep np = naive_prepare(g, view.sign, plane.x, plane.y, plane.z, gx, gy);
        const uint32_t px = (uint32_t)gy * (uint32_t)g.W + (uint32_t)gx;
        raw[rc.off + p] = naive_finish(view, np, view.feat_self[px], view.feat_other[np.ia], view.feat_other[np.ib]);
    }
}

// One dword per lane streaming copy: the calibration pattern for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (the strip
// kernel reads and writes dwords; MI355X_MICROARCH.md asks for a calibration on a known byte count in the same access width).
__global__ void les_calib_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// 16 bytes per lane, four chunks per thread: the streaming-copy ceiling of the chip for this process (bench.py quotes the rate
// it measures in the same run as roofline.peak_achievable)
__global__ void les_calib_copy_wide_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4)
{
    const size_t i = ((size_t)blockIdx.x * 4) * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t j = i + (size_t)k * blockDim.x;
        if (j < n4) dst[j] = src[j];
    }
}

// ---------------------------------------------------------------------------------------------------
// Volume preparation ("next" row N3): LES/main.cpp:146-176 fillOutOfView and :178-199 convertVolumeL2R with
// margin = 0 (interp_margin, LES/main.cpp:359).  grid = (ceil(W/256), H, D).
// ---------------------------------------------------------------------------------------------------
__global__ void les_fill_out_of_view_kernel(float* __restrict__ vol, int D, int H, int W, int mode)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), y = (int)blockIdx.y, d = (int)blockIdx.z;
    float* row = vol + ((size_t)d * H + y) * W;
    if (mode == 0) {                                    // left view: columns x < d see nothing -> first valid column
        const int q = d < W - 1 ? d : W - 1;
        if (x < q) row[x] = row[q];
    } else {                                            // right view: columns x >= W-d -> last valid column
        int p = W - d;
        if (p < 1) p = 1;
        if (x >= p && x < W) row[x] = row[p - 1];
    }
}

__global__ void les_convert_l2r_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H, int W)
{
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x), y = (int)blockIdx.y, d = (int)blockIdx.z;
    if (x >= W) return;
    const float* s0 = src + ((size_t)d * H + y) * W;
    float v;
    if (d >= W) v = s0[x];                               // (the reference assumes d < W)
    else v = x < W - 1 - d ? s0[x + d] : s0[W - 1];      // slice d shifted left by d, right edge replicated
    dst[((size_t)d * H + y) * W + x] = v;
}

// ---------------------------------------------------------------------------------------------------
// PatchMatch-style winner-take-all update (LES/FastGCStereo.h:56-60) for a batch of shared regions.
// ---------------------------------------------------------------------------------------------------
struct WtaJob { int x, y, w, h; };

__global__ void les_wta_kernel(const WtaJob* __restrict__ jobs, const float4* __restrict__ planes,
                               float* __restrict__ cur_cost, const float* __restrict__ prop_cost,
                               float4* __restrict__ labels, int W)
{
    const WtaJob j = jobs[blockIdx.x];
    const float4 pl = planes[blockIdx.x];
    // grid = (regions, chunks): large shared regions (layer 2: ~400 x 390 px) are split over several blocks
    for (int idx = (int)(blockIdx.y * blockDim.x + threadIdx.x); idx < j.w * j.h; idx += (int)(blockDim.x * gridDim.y)) {
        int yy = idx / j.w, xx = idx - yy * j.w;
        size_t k = (size_t)(j.y + yy) * W + j.x + xx;
        float pc = prop_cost[k];
        if (cur_cost[k] > pc) {                 // strict, as the reference
            cur_cost[k] = pc;
            labels[k] = pl;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Multi-GPU tile exchange: after the lock-steps of a disjoint set every rank publishes the label / cost tiles of its
// own cells.  A rank's slot of the all-gather buffer is [labels of its pixels: float4 x lmax][costs: float x lmax]; `off` is the
// pixel offset of a rect inside its rank's slot.  grid = (rects, row chunks).
// ---------------------------------------------------------------------------------------------------
struct XchgRect { int x, y, w, h; int off; int rank; };

__global__ void les_xchg_pack_kernel(const XchgRect* __restrict__ rects, int first, const float4* __restrict__ labels,
                                     const float* __restrict__ cost, float* __restrict__ slot, int lmax, int W)
{
    const XchgRect r = rects[first + blockIdx.x];
    float4* sl = reinterpret_cast<float4*>(slot);
    float* sc = slot + 4 * (size_t)lmax;
    for (int idx = (int)(blockIdx.y * blockDim.x + threadIdx.x); idx < r.w * r.h; idx += (int)(blockDim.x * gridDim.y)) {
        const int yy = idx / r.w, xx = idx - yy * r.w;
        const size_t k = (size_t)(r.y + yy) * W + r.x + xx;
        sl[r.off + idx] = labels[k];
        sc[r.off + idx] = cost[k];
    }
}

// every rect of every OTHER rank (own_rank's rects are skipped: this rank's maps already hold them)
__global__ void les_xchg_unpack_kernel(const XchgRect* __restrict__ rects, const float* __restrict__ recv, float4* __restrict__ labels,
                                       float* __restrict__ cost, int lmax, int W, int own_rank)
{
    const XchgRect r = rects[blockIdx.x];
    if (r.rank == own_rank) return;
    const float* slot = recv + 5 * (size_t)lmax * (size_t)r.rank;
    const float4* sl = reinterpret_cast<const float4*>(slot);
    const float* sc = slot + 4 * (size_t)lmax;
    for (int idx = (int)(blockIdx.y * blockDim.x + threadIdx.x); idx < r.w * r.h; idx += (int)(blockDim.x * gridDim.y)) {
        const int yy = idx / r.w, xx = idx - yy * r.w;
        const size_t k = (size_t)(r.y + yy) * W + r.x + xx;
        labels[k] = sl[r.off + idx];
        cost[k] = sc[r.off + idx];
    }
}

}  // namespace les
