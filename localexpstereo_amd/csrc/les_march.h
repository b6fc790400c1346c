// les_march.h -- the fixed-point "column march" kernel: gather + guided-filter aggregation with exact integer box sums
// (gfx950 / CDNA4, wave64).  It computes the same quantity as les_strip_kernel (les_kernels.h):
//
//     p(s)  = min( lerp_d vol[a*x+b*y+c][y][x], th_col )                       LES/CostVolumeEnergy.h:70-98
//     q     = guided filter of p with the colour guide, radius R, eps           LES/GuidedFilter.h:142-266, 301-326
//     q    -> 1e6 where the label is invalid                                    LES/CostVolumeEnergy.h:176-183
//
// but is organised around what the MI355X micro-benchmarks say is cheap (tools/ubench/valu_rates.hip): 32-bit integer /
// fp32 adds issue in 2 cycles per wave, every fp64 operation, conversion, DPP move or 64-bit integer MAD in 4, an LDS
// ds_write_b128 costs ~14 CU cycles and a ds_read_b128 4.  All four box-filter passes are therefore EXACT INTEGER sums:
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

#if defined(LES_SIM)
struct alignas(16) int4 { int x, y, z, w; };
__device__ inline int readfirstlane_i32(int v) { return v; }
__device__ inline int cvt_rpi_i32(float x)
{
    if (!(x == x)) return 0;
    const float f = floorf(x + 0.5f);
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int)0x80000000;
    return (int)f;
}
// exclusive prefix sum over the 16 lanes of a DPP row
__device__ inline int row16_excl_scan(int v)
{
    int o[16];
    hipsim::group16_allgather(v, o);
    const int l = hipsim::g_block->current & 15;
    int s = 0;
    for (int j = 0; j < l; j++) s = (int)((unsigned)s + (unsigned)o[j]);
    return s;
}
#else
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
__device__ __forceinline__ int row16_excl_scan(int v)
{
    int s = v;
    s += dpp_row_shr<1>(s);
    s += dpp_row_shr<2>(s);
    s += dpp_row_shr<4>(s);
    s += dpp_row_shr<8>(s);
    return s - v;
}
#endif

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
};

template <int R, int WGC, int NJ, int BY>
struct MarchCfg {
    static constexpr int KS = 2 * R + 1;
    static constexpr int TW = WGC - 4 * R;                 // output columns per job
    static constexpr int NT = WGC * NJ;
    static constexpr int SEGL = 8;                         // columns per prefix segment
    static constexpr int NSEG = WGC / SEGL;                // <= 16: one DPP row per (job, row)
    static constexpr int PCOLS = 1 + WGC + NSEG;           // physical columns: leading zero element (P(-1)) + one pad per segment
    static constexpr int NBL = NJ * BY * 16;               // lanes of the prefix phases
    static_assert(KS % BY == 0, "the block height must divide the ring length (compile-time ring slots)");
    static_assert(WGC % 64 == 0, "a job slot is a whole number of waves");
    static_assert(NSEG <= 16 && WGC % SEGL == 0, "one DPP row per prefix row");
    static_assert(NBL <= NT, "more prefix lanes than threads");
    static_assert(TW > 0, "job too narrow for this radius");
    __host__ __device__ static constexpr int pcol(int ci) { return 1 + ci + ci / SEGL; }
};

template <int R, int WGC, int NJ, int BY, int MW>
__global__ void __launch_bounds__(WGC * NJ, MW)
les_march_kernel(Geom g, MarchView view, const Job* __restrict__ jobs, const float4* __restrict__ planes,
                 float* __restrict__ out, int ngroups, int check)
{
    using Cfg = MarchCfg<R, WGC, NJ, BY>;
    constexpr int KS = Cfg::KS, NT = Cfg::NT, SEGL = Cfg::SEGL, NSEG = Cfg::NSEG, PCOLS = Cfg::PCOLS, NBL = Cfg::NBL;

    __shared__ int4 s_T1[NJ][BY][PCOLS];     // stage 1: vertical sums, then (in place) their prefix sums along x
    __shared__ int4 s_T2[NJ][BY][PCOLS];     // stage 2: quantised (a_0, a_1, a_2, b), then their prefix sums
    __shared__ double s_rtab[KS + 1];        // 1/n, n = 0..2R+1
    // per-row scalars, written one block ahead by NJ*BY lanes (double buffered by block parity)
    struct RowA { uint32_t rowpx; float d_base; };                   // p-row: pixel offset of the (clamped) row | bit 31: row inside clip and march
    struct RowC { float rny1; uint32_t srow; };                      // stage-1 row: 1/count_y, statistics row offset | bit 31: row inside clip and primed
    struct RowD { double rny2; uint32_t grow; uint32_t pad_; };      // output row: 1/count_y, guide row offset | bit 31: the row is an output row of the job
    __shared__ RowA s_rowA[2][NJ][BY];
    __shared__ RowC s_rowC[2][NJ][BY];
    __shared__ RowD s_rowD[2][NJ][BY];

    // XCD-aware group order (cf. les_strip_kernel): consecutive groups (same strip, consecutive planes) share an XCD's L2
    int grp;
    {
        const int nwg = (int)gridDim.x, orig = (int)blockIdx.x;
        const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
        grp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    }
    if (grp >= ngroups) return;
    const int tid = (int)threadIdx.x;
    const int slot = readfirstlane_i32(tid / WGC);       // a job slot is a whole number of waves
    const int ci = tid - slot * WGC;
    const Job job = jobs[grp * NJ + slot];
    int th_max = 0;
#pragma unroll
    for (int s = 0; s < NJ; s++) th_max = max(th_max, jobs[grp * NJ + s].th);
    const int TtotMax = th_max + 4 * R;
    const int Ttot = job.th > 0 ? job.th + 4 * R : 0;     // an empty slot (padding of the last group) never passes a row test
    const float4 plane = planes[job.plane_idx];

    if (tid <= KS) s_rtab[tid] = tid > 0 ? 1.0 / (double)tid : 0.0;
    // zero the LDS tiles once: element 0 of every row is P(-1) = 0 and stays untouched; the pads are never read
    for (int k = tid; k < NJ * BY * PCOLS; k += NT) {
        (&s_T1[0][0][0])[k] = int4{0, 0, 0, 0};
        (&s_T2[0][0][0])[k] = int4{0, 0, 0, 0};
    }

    // ---- per-thread column constants
    const int gx = job.tx0 - 2 * R + ci;                                  // image column of this lane (p, stage-1 and output column alike)
    const bool col_in = gx >= job.cx0 && gx < job.cx1 && job.th > 0;
    const int sx = min(max(gx, job.cx0), max(job.cx1 - 1, job.cx0));      // clamped: addresses stay inside the image
    const bool s1_col = col_in && ci >= R && ci < WGC - R;                // stage-1 column with a complete horizontal window
    const bool out_col = ci >= 2 * R && ci < 2 * R + job.tw && job.th > 0;
    const int nx = window_count(gx, R, job.cx0, job.cx1);                 // the same count serves stage 1 and stage 2 (same column)
    const uint32_t HWu = (uint32_t)g.H * (uint32_t)g.W;
    const float g_ax = plane.x * (float)sx;                               // a * x, LES/CostVolumeEnergy.h:76
    // fronto-parallel planes (a = b = 0): d = c for every pixel, so taps / weight / mode are per-job constants
    const bool fronto = plane.x == 0.0f && plane.y == 0.0f;
    GatherPrep gpc = gather_prepare(g, 0.0f, plane.z, 0u, HWu, true);
    if (gpc.f1 == 0.0f) gpc.i1 = gpc.i0;                                  // weight 0 on a finite volume: the second tap is never needed
    const int pcP = Cfg::pcol(min(ci + R, WGC - 1));                      // physical columns of P(x+R) and P(x-R-1)
    const int pcM = ci - R - 1 >= 0 ? Cfg::pcol(ci - R - 1) : 0;
    const int pcS = Cfg::pcol(ci);

    __syncthreads();
    const float rnx_f = (float)s_rtab[nx];
    const double rnx_d = s_rtab[nx];

    // ---- row tables of the block that starts at p-row tb (lanes tid < NJ*BY; slot / row of the TABLE entry, not of the lane's own job)
    auto fill_tables = [&](int tb, int par) {
        const int s = tid / BY, i = tid - s * BY;
        const Job js = jobs[grp * NJ + s];
        const float4 pl = planes[js.plane_idx];
        const int t = tb + i;
        const int tt = js.th > 0 ? js.th + 4 * R : 0;
        const int cy1m = max(js.cy1 - 1, js.cy0);
        {   // p-row (phase A)
            const int gy = js.ty0 - 2 * R + t;
            const int sy = min(max(gy, js.cy0), cy1m);
            RowA ra;
            ra.rowpx = ((uint32_t)sy * (uint32_t)g.W) | ((t < tt && gy >= js.cy0 && gy < js.cy1) ? 0x80000000u : 0u);
            ra.d_base = pl.y * (float)sy + pl.z;                        // b*y + c, LES/CostVolumeEnergy.h:73
            s_rowA[par][s][i] = ra;
        }
        {   // stage-1 row (phase C): centre of the vertical window that ends at p-row t
            const int gy1 = js.ty0 - 3 * R + t;
            RowC rc;
            rc.rny1 = (float)s_rtab[window_count(gy1, R, js.cy0, js.cy1)];
            rc.srow = ((uint32_t)min(max(gy1, js.cy0), cy1m) * (uint32_t)g.W) | ((gy1 >= js.cy0 && gy1 < js.cy1 && t >= 2 * R && t < tt) ? 0x80000000u : 0u);
            s_rowC[par][s][i] = rc;
        }
        {   // output row (phase D)
            const int gy2 = js.ty0 - 4 * R + t;
            RowD rd;
            rd.rny2 = s_rtab[window_count(gy2, R, js.cy0, js.cy1)];
            rd.grow = ((uint32_t)min(max(gy2, js.cy0), cy1m) * (uint32_t)g.W) | ((t >= 4 * R && t < tt) ? 0x80000000u : 0u);
            rd.pad_ = 0;
            s_rowD[par][s][i] = rd;
        }
    };
    if (tid < NJ * BY) fill_tables(0, 0);

    // ---- persistent state of the vertical passes
    int ringP[KS];                       // fixed-point cost of the last 2R+1 p-rows of this column
    uint32_t ringG[KS];                  // their guide pixels
    int ring2[4][KS];                    // horizontal box sums of (a_0, a_1, a_2, b) of the last 2R+1 stage-1 rows
#pragma unroll
    for (int k = 0; k < KS; k++) { ringP[k] = 0; ringG[k] = 0u; ring2[0][k] = ring2[1][k] = ring2[2][k] = ring2[3][k] = 0; }
    int Sp = 0;
    long long Sc[3] = {1ll << (kMarchSH - 1), 1ll << (kMarchSH - 1), 1ll << (kMarchSH - 1)};   // rounding bias of the >> SH folded in
    double S2[4] = {0.0, 0.0, 0.0, 0.0};

    __syncthreads();

    int par = 0;
    LES_PHASE_BEGIN();
    for (int t0 = 0; t0 < TtotMax; t0 += BY, par ^= 1) {
        const int base = t0 % KS;                            // ring slot of the block's first row: one of KS/BY values
        // ===================== A: gather, fixed point, vertical running sums =====================
        {
            GatherPrep gp[BY];
            float v0[BY], v1[BY];
            uint32_t gw[BY];
#pragma unroll
            for (int i = 0; i < BY; i++) {
                const RowA ra = s_rowA[par][slot][i];
                const bool inside = col_in && (ra.rowpx >> 31);
                const uint32_t px = (ra.rowpx & 0x7fffffffu) + (uint32_t)sx;
                if (fronto) {
                    gp[i] = gpc;
                    gp[i].i0 += px; gp[i].i1 += px;
                    gp[i].mode = inside ? gpc.mode : 3;
                } else {
                    gp[i] = gather_prepare(g, g_ax, ra.d_base, px, HWu, inside);
                    if (gp[i].f1 == 0.0f) gp[i].i1 = gp[i].i0;
                }
                v0[i] = view.vol[gp[i].i0];
                v1[i] = view.vol[gp[i].i1];
                gw[i] = view.ipk8[px];
            }
            auto blockA = [&](auto base_tag) {
                constexpr int BASE = decltype(base_tag)::value;
                static_for<BY>([&](auto itag) {
                    constexpr int i = decltype(itag)::value;
                    constexpr int SLOT = BASE + i;           // ring slot of p-row t0 + i; it holds the row that leaves the window (2R+1 rows ago)
                    const float p = gather_finish(g, gp[i], v0[i], v1[i]);
                    const int pi = gp[i].mode == 3 ? 0 : (int)fmaf(p - view.vmin, view.sp, 0.5f);
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
                    s_T1[slot][i][pcS] = int4{Sp, (int)(Sc[0] >> kMarchSH), (int)(Sc[1] >> kMarchSH), (int)(Sc[2] >> kMarchSH)};
                });
            };
            if constexpr (KS / BY == 1) blockA(IntTag<0>{});
            else if constexpr (KS / BY == 3) { if (base == 0) blockA(IntTag<0>{}); else if (base == BY) blockA(IntTag<BY>{}); else blockA(IntTag<2 * BY>{}); }
            else {
                static_assert(KS / BY == 1 || KS / BY == 3, "unsupported ring / block ratio");
            }
        }
        __syncthreads();
        LES_PHASE_MARK(0);

        // ===================== B1: prefix sums of T1 along x (in place, modulo 2^32) =====================
        auto prefix_phase = [&](int4 (*T)[BY][PCOLS]) {
            if (tid < NBL) {
                const int rh = tid >> 4, seg = tid & 15;
                const int s = rh / BY, i = rh - s * BY;
                int4 v[SEGL];
                int4* row = &T[s][i][1 + (seg < NSEG ? seg : 0) * (SEGL + 1)];
#pragma unroll
                for (int j = 0; j < SEGL; j++) v[j] = row[j];
#pragma unroll
                for (int j = 1; j < SEGL; j++) {
                    v[j].x += v[j - 1].x; v[j].y += v[j - 1].y; v[j].z += v[j - 1].z; v[j].w += v[j - 1].w;
                }
                const bool act = seg < NSEG;
                int4 off;
                off.x = row16_excl_scan(act ? v[SEGL - 1].x : 0);
                off.y = row16_excl_scan(act ? v[SEGL - 1].y : 0);
                off.z = row16_excl_scan(act ? v[SEGL - 1].z : 0);
                off.w = row16_excl_scan(act ? v[SEGL - 1].w : 0);
                if (act) {
#pragma unroll
                    for (int j = 0; j < SEGL; j++) {
                        v[j].x += off.x; v[j].y += off.y; v[j].z += off.z; v[j].w += off.w;
                        row[j] = v[j];
                    }
                }
            }
        };
        prefix_phase(s_T1);
        if (tid < NJ * BY) fill_tables(t0 + BY, par ^ 1);     // the next block's row tables (other parity: nobody reads them before the barriers below)
        __syncthreads();
        LES_PHASE_MARK(1);

        // ===================== C: box sums, covariance, 3x3 algebra, quantise =====================
        {
            auto rowsC = [&](auto lo_tag, auto n_tag) {
                constexpr int LO = decltype(lo_tag)::value, N = decltype(n_tag)::value;
                float4 st[N][3];
                int4 pp[N], pm[N];
                RowC rc[N];
#pragma unroll
                for (int k = 0; k < N; k++) {
                    rc[k] = s_rowC[par][slot][LO + k];
                    const float4* sp = view.mstats + (size_t)((rc[k].srow & 0x7fffffffu) + (uint32_t)sx) * 3;
                    st[k][0] = sp[0]; st[k][1] = sp[1]; st[k][2] = sp[2];
                    pp[k] = s_T1[slot][LO + k][pcP];
                    pm[k] = s_T1[slot][LO + k][pcM];
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const int s = pp[k].x - pm[k].x;                       // sum of pi over the window (exact)
                    const int t0c = pp[k].y - pm[k].y, t1c = pp[k].z - pm[k].z, t2c = pp[k].w - pm[k].w;
                    const float4 q0 = st[k][0], q1 = st[k][1], q2 = st[k][2];
                    const int M0 = __float_as_int(q2.y), M1 = __float_as_int(q2.z), M2 = __float_as_int(q2.w);
                    // N cov_c in units of 2^SH (u8 * pi): t_c - mean_c * s, the product rounded at 2^-32 of its own scale
                    const float d0 = (float)(t0c - (int)(((long long)M0 * (long long)s + (1ll << 31)) >> 32));
                    const float d1 = (float)(t1c - (int)(((long long)M1 * (long long)s + (1ll << 31)) >> 32));
                    const float d2 = (float)(t2c - (int)(((long long)M2 * (long long)s + (1ll << 31)) >> 32));
                    // LES/GuidedFilter.h:204-221 with the scale of the integer stage 2 folded into the normalisation
                    const float rn = rnx_f * rc[k].rny1;
                    const float ka = view.kapS * rn;
                    const float a0 = fmaf(q0.z, d2, fmaf(q0.y, d1, q0.x * d0)) * ka;     // inv00 inv01 inv02
                    const float a1 = fmaf(q1.x, d2, fmaf(q0.w, d1, q0.y * d0)) * ka;     // inv01 inv11 inv12
                    const float a2 = fmaf(q1.y, d2, fmaf(q1.x, d1, q0.z * d0)) * ka;     // inv02 inv12 inv22
                    const float mp = (float)s * (view.upS * rn);
                    const float bb = fmaf(-a2, q2.x, fmaf(-a1, q1.w, fmaf(-a0, q1.z, mp)));
                    const bool keep = s1_col && (rc[k].srow >> 31);
                    int4 o;
                    o.x = keep ? cvt_rpi_i32(a0) : 0;
                    o.y = keep ? cvt_rpi_i32(a1) : 0;
                    o.z = keep ? cvt_rpi_i32(a2) : 0;
                    o.w = keep ? cvt_rpi_i32(bb) : 0;
                    s_T2[slot][LO + k][pcS] = o;
                }
            };
            constexpr int GC = 2;                               // rows in flight (statistics + prefix reads: 24 registers per row)
            static_for<(BY + GC - 1) / GC>([&](auto gtag) {
                constexpr int LO = decltype(gtag)::value * GC;
                constexpr int N = (BY - LO) < GC ? (BY - LO) : GC;
                rowsC(IntTag<LO>{}, IntTag<N>{});
            });
        }
        __syncthreads();
        LES_PHASE_MARK(2);

        // ===================== B2: prefix sums of T2 =====================
        prefix_phase(s_T2);
        __syncthreads();
        LES_PHASE_MARK(3);

        // ===================== D: horizontal box, vertical running sums, output =====================
        {
            auto blockD = [&](auto base_tag) {
                constexpr int BASE = decltype(base_tag)::value;
                constexpr int GD = 4;                           // rows in flight
                static_for<(BY + GD - 1) / GD>([&](auto gtag) {
                    constexpr int LO = decltype(gtag)::value * GD;
                    constexpr int N = (BY - LO) < GD ? (BY - LO) : GD;
                    RowD rd[N];
                    uint32_t gq[N];
                    int4 pp[N], pm[N];
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        rd[k] = s_rowD[par][slot][LO + k];
                        gq[k] = view.ipk8[(rd[k].grow & 0x7fffffffu) + (uint32_t)sx];
                        pp[k] = s_T2[slot][LO + k][pcP];
                        pm[k] = s_T2[slot][LO + k][pcM];
                    }
                    static_for<N>([&](auto ktag) {
                        constexpr int k = decltype(ktag)::value;
                        constexpr int i = LO + k;
                        constexpr int SLOT = BASE + i;
                        const int h0 = pp[k].x - pm[k].x, h1 = pp[k].y - pm[k].y, h2 = pp[k].z - pm[k].z, h3 = pp[k].w - pm[k].w;
                        S2[0] += (double)(h0 - ring2[0][SLOT]); ring2[0][SLOT] = h0;
                        S2[1] += (double)(h1 - ring2[1][SLOT]); ring2[1][SLOT] = h1;
                        S2[2] += (double)(h2 - ring2[2][SLOT]); ring2[2][SLOT] = h2;
                        S2[3] += (double)(h3 - ring2[3][SLOT]); ring2[3][SLOT] = h3;
                        if (out_col && (rd[k].grow >> 31)) {
                            const int t = t0 + i;
                            const uint32_t gi = gq[k];
                            const double i0 = (double)(((int)(gi << 24)) >> 24), i1 = (double)(((int)(gi << 16)) >> 24), i2 = (double)(((int)(gi << 8)) >> 24);
                            // LES/GuidedFilter.h:243: (b + a . I) / N on the centred guide, in integers < 2^53
                            const double acc = fma(S2[2], i2, fma(S2[1], i1, fma(S2[0], i0, S2[3] * 255.0)));
                            float q = (float)fma(acc, view.qscale * (rnx_d * rd[k].rny2), (double)view.vmin);
                            const int gy2 = job.ty0 + t - 4 * R;
                            if (check && !label_valid(g, plane.x, plane.y, plane.z, plane.w, gx, gy2)) q = LES_COST_INVALID;
                            out[job.out_off + (long long)(t - 4 * R) * job.out_stride + (ci - 2 * R)] = q;
                        }
                    });
                });
            };
            if constexpr (KS / BY == 1) blockD(IntTag<0>{});
            else { if (base == 0) blockD(IntTag<0>{}); else if (base == BY) blockD(IntTag<BY>{}); else blockD(IntTag<2 * BY>{}); }
        }
        // no barrier here: phase A of the next block writes T1 (last read in phase C, two barriers ago) and reads the
        // other parity of the row tables (written before the barrier that ended B1)
        LES_PHASE_MARK(4);
    }
    LES_PHASE_END();
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
