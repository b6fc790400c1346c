// les_maxflow_cell.h -- the minimum cut of an expansion move for a cell that fits ONE workgroup (the finest layer: 42 x 42 ... 45 x 45 nodes), with
// the iteration of the tiled solver instead of the one of les_maxflow.h (round 6).
//
// Same job as les_maxflow_kernel (les_maxflow.h: `graph.maxflow()` + `graph.what_segment()` of LES/FastGCStereo.h:553-559 on the 5-float node payload
// of les_expansion_graph_kernel), same outputs, same segment rule.  What differs is the cost of an iteration.  les_maxflow.h keeps the residuals in
// LDS and pushes two grid directions per pass: ten barriers per push-relabel iteration, and a barrier step of sixteen waves costs about a
// microsecond whatever happens between two of them (measured on the tiled solver with clock stamps, tools/lab/mt_probe.py) -- the slowest of the 450
// cells of a lock-step needs 86 iterations and 317 relabel sweeps, and the launch lasts as long as that cell.  Here, as in les_maxflow_tiled.h:
//   * residuals, excess and own height of a node live in REGISTERS of its owner (two nodes per thread, 1024 threads);
//   * one push half for all eight directions (an exchange word per node and direction), one receive + relabel half: TWO barriers per iteration,
//     the heights double-buffered so that the relabel never writes what a neighbour may still be reading (the snapshot rule: deterministic);
//   * exact relabelling = chaotic relaxation over live LDS values with row flags (a sweep only looks at rows next to one in which a distance
//     fell), one barrier per sweep; the heights are RESET to the exact distances every `round_iters` iterations (les_maxflow.h: raised to them).
// LDS: two halo-pitched height arrays of uint16 (N + 2 <= 2050), 8 exchange words and a flag byte per node: 77 728 B, two workgroups per CU.
// Cells of up to 2048 nodes whose halo-pitched rectangle (w + 2) x (h + 2) has at most 2304 entries and at most 72 rows qualify
// (mc_fits); a lock-step with any other cell runs les_maxflow_kernel as before.  Masks are the unique sink-side set of a maximum preflow: equal to
// les_maxflow_kernel's up to ties that float rounding of the residuals moves; flow values agree to rounding.
#pragma once

#include <cstdint>

#include "les_maxflow.h"
#include "les_maxflow_tiled.h"

namespace les {

#if !defined(LES_MC_THREADS)
#define LES_MC_THREADS 1024              // (lab builds: tools/build_variant.sh mcT -DLES_MC_THREADS=512 -DLES_MC_NPT=4; tools/lab/ab_cell_shape.sh)
#define LES_MC_NPT 2
#endif
constexpr int kMcThreads = LES_MC_THREADS;
constexpr int kMcNpt = LES_MC_NPT;
constexpr int kMcMaxNodes = 2048;
constexpr int kMcMaxHalo = 2304;
constexpr int kMcMaxRows = 72;                              // h + 2
static_assert(kMcThreads * kMcNpt >= kMcMaxNodes, "every node needs an owner");
constexpr size_t kMcLdsBytes = (size_t)kMcMaxHalo * 2 * 2 + (size_t)kMcMaxNodes * (8 * 4 + 1) + 64 + 3 * kMcMaxRows * 4;
static_assert(2 * kMcLdsBytes <= 160 * 1024, "two workgroups per CU");
static_assert(kMcThreads * 8 <= kMcMaxNodes * 8 * 4, "the final reduction borrows one double per thread from the exchange words");

__host__ __device__ inline bool mc_fits(int w, int h)
{
    return w > 0 && h > 0 && (long long)w * h <= kMcMaxNodes && (long long)(w + 2) * (h + 2) <= kMcMaxHalo && h + 2 <= kMcMaxRows;
}

// grid = cells; block = kMcThreads; dynamic LDS = kMcLdsBytes.  Every cell of the launch must satisfy mc_fits (or be empty).
__global__ void __launch_bounds__(kMcThreads, kMcThreads / 128)          // two workgroups per CU (1024 threads: 8 waves per SIMD -> at most 64 VGPRs)
les_maxflow_cell_kernel(const GraphCellMf* __restrict__ cells, const long long* __restrict__ offsets, const float* __restrict__ payload,
                        int max_iter, int round_iters, uint8_t* __restrict__ masks, int* __restrict__ status, double* __restrict__ flows,
                        int* __restrict__ unsolved_total)
{
#if defined(LES_SIM)
    static thread_local int s_raw[kMcLdsBytes / 4 + 16];
    char* base = reinterpret_cast<char*>(s_raw);
#else
    extern __shared__ __attribute__((aligned(16))) char s_dyn_mc[];
    char* base = s_dyn_mc;
#endif
    uint16_t* hgA = reinterpret_cast<uint16_t*>(base);                      // heights / distances, halo-pitched: (h + 2) x (w + 2)
    uint16_t* hgB = hgA + kMcMaxHalo;
    float* sent = reinterpret_cast<float*>(hgB + kMcMaxHalo);               // sent[k * NP + v]: what node v pushed along direction k in this iteration
    constexpr int NP = kMcMaxNodes;
    uint8_t* flg = reinterpret_cast<uint8_t*>(sent + 8 * NP);               // "something was sent to this node"
    int* sflag = reinterpret_cast<int*>(flg + NP);                          // [0..2] rotating "changed" flags of relax, [4] any active node, [8..10] rotating "still active" flags of the iterations
    int* rowchg = sflag + 16;                                               // [3][72]: rows (halo rows included) in which a distance fell, per sweep

    const GraphCellMf c = cells[blockIdx.x];
    const int W = c.w, H = c.h, N = W * H;
    const int tid = (int)threadIdx.x;
    if (N <= 0) { if (tid == 0) { status[blockIdx.x] = 0; if (flows) flows[blockIdx.x] = 0.0; } return; }
    const float* p5 = payload + 5 * offsets[blockIdx.x];
    const int BIG = N + 2;                                                  // "cannot reach the sink" (fits uint16: N <= 2048)
    const int hp = W + 2;
    const float inv_w = 1.0f / (float)W;

    // ---- own nodes: v = tid + j * kMcThreads (row-major over the cell)
    int hi[kMcNpt];
    bool has[kMcNpt];
    unsigned edge[kMcNpt];             // bit k: the neighbour in direction k lies outside the cell
    float r[kMcNpt][8], e[kMcNpt];
    int hv[kMcNpt];
#pragma unroll
    for (int j = 0; j < kMcNpt; j++) {
        const int v = tid + j * kMcThreads;
        has[j] = v < N;
        const int vv = has[j] ? v : 0;
        const int ly = (int)(((float)vv + 0.5f) * inv_w), lx = vv - ly * W;      // (vv < 2048, W <= 766: the same argument as for yrow below; les_maxflow_tiled.h)
        hi[j] = (ly + 1) * hp + lx + 1;
        edge[j] = (lx == W - 1 ? kMtDirsE : 0u) | (lx == 0 ? kMtDirsW : 0u) | (ly == H - 1 ? kMtDirsS : 0u) | (ly == 0 ? kMtDirsN : 0u);
        e[j] = 0.0f; hv[j] = BIG;
#pragma unroll
        for (int k = 0; k < 8; k++) r[j][k] = 0.0f;
        if (!has[j]) { edge[j] = 0xffu; continue; }
        const float* q = p5 + 5 * (size_t)v;
        e[j] = q[0];                                                   // source arcs are saturated at the start; a sink arc is the negative part
        r[j][0] = (edge[j] >> 0 & 1u) ? 0.0f : q[1];                   // arcs that would leave the region carry no capacity
        r[j][2] = (edge[j] >> 2 & 1u) ? 0.0f : q[2];
        r[j][4] = (edge[j] >> 4 & 1u) ? 0.0f : q[3];
        r[j][6] = (edge[j] >> 6 & 1u) ? 0.0f : q[4];
#pragma unroll
        for (int k = 0; k < 8; k++) sent[k * NP + v] = 0.0f;
        flg[v] = 0;
    }
    auto hoff = [&](int k) { return mf_dy(k) * hp + mf_dx(k); };            // neighbour k in the halo-pitched arrays
    auto loff = [&](int k) { return mf_dy(k) * W + mf_dx(k); };             // neighbour k in the node-indexed arrays

    if (tid < 16) sflag[tid] = 0;
    {   // the halo ring of both height arrays: BIG for good
        const int ring = 2 * (W + 2) + 2 * H;
        for (int i = tid; i < ring; i += kMcThreads) {
            int hx, hy;
            if (i < W + 2) { hx = i; hy = 0; }
            else if (i < 2 * (W + 2)) { hx = i - (W + 2); hy = H + 1; }
            else if (i < 2 * (W + 2) + H) { hx = 0; hy = i - 2 * (W + 2) + 1; }
            else { hx = W + 1; hy = i - 2 * (W + 2) - H + 1; }
            hgA[hy * hp + hx] = (uint16_t)BIG;
            hgB[hy * hp + hx] = (uint16_t)BIG;
        }
    }

    // Residual distances to the sink over the live values of hgA, one barrier per sweep (les_maxflow_tiled.h: relax)
    const float inv_hp = 1.0f / (float)hp;
    auto relax = [&](const unsigned (&rm)[kMcNpt]) {
        int yrow[kMcNpt];                                                    // halo-pitched row of the own nodes (hi < 2304, hp <= 768: (hi + 0.5) / hp keeps 0.5 / hp >= 6.5e-4 from every integer, the rounding stays below 2e-5)
#pragma unroll
        for (int j = 0; j < kMcNpt; j++) yrow[j] = (int)(((float)hi[j] + 0.5f) * inv_hp);
        for (int s = 0;; s++) {
            const int cur = s % 3, nxt = (s + 1) % 3, nn2 = (s + 2) % 3;
            if (tid == 0) sflag[nxt] = 0;
            if (tid < kMcMaxRows) rowchg[nn2 * kMcMaxRows + tid] = 0;        // (read in sweep s + 2; last read in sweep s - 1)
            bool changed = false;
#pragma unroll
            for (int j = 0; j < kMcNpt; j++) {
                const int y = yrow[j];
                const bool look = has[j] && (rowchg[cur * kMcMaxRows + y - 1] | rowchg[cur * kMcMaxRows + y] | rowchg[cur * kMcMaxRows + y + 1]) != 0;
                if (!mt_wave_any(look)) continue;
                const int d = hgA[hi[j]];
                int dn[8];
#pragma unroll
                for (int k = 0; k < 8; k++) dn[k] = hgA[hi[j] + hoff(k)];
                int best = d;
#pragma unroll
                for (int k = 0; k < 8; k++) best = ((rm[j] >> k & 1u) && dn[k] + 1 < best) ? dn[k] + 1 : best;
                if (look && best < d) { hgA[hi[j]] = (uint16_t)best; rowchg[nxt * kMcMaxRows + y] = 1; changed = true; }
            }
            if (changed) sflag[cur] = 1;
            __syncthreads();
            if (!sflag[cur]) break;
        }
    };

    int git = 0;                       // iterations so far (the rotating flags and the height buffers go on across the rounds)
    bool converged = false;
    for (;;) {
        // ---- exact relabelling: distances to the sink in the residual graph, from scratch
        unsigned rm[kMcNpt];
#pragma unroll
        for (int j = 0; j < kMcNpt; j++) {
            rm[j] = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) rm[j] |= (r[j][k] > 0.0f ? 1u : 0u) << k;
            if (has[j]) hgA[hi[j]] = (uint16_t)(e[j] < 0.0f ? 1 : BIG);
        }
        if (tid < 3) sflag[tid] = 0;
        if (tid == 0) sflag[4] = 0;
        for (int i = tid; i < 3 * kMcMaxRows; i += kMcThreads) rowchg[i] = i < kMcMaxRows ? 1 : 0;     // sweep 0 looks at every row
        __syncthreads();
        relax(rm);
        bool mine = false;
#pragma unroll
        for (int j = 0; j < kMcNpt; j++) {
            if (!has[j]) continue;
            hv[j] = hgA[hi[j]];
            hgB[hi[j]] = (uint16_t)hv[j];
            if (e[j] > 0.0f && hv[j] < BIG) mine = true;
        }
        if (mine) sflag[4] = 1;
        __syncthreads();
        if (!sflag[4]) { converged = true; break; }                          // no excess can reach the sink: a maximum preflow
        if (git >= max_iter) break;                                           // gives up: status 1, the caller cuts the cell on the host
        const int until = git + (round_iters < max_iter - git ? round_iters : max_iter - git);

        // ---- synchronous push-relabel iterations: pushes | barrier | receive + relabel | barrier (les_maxflow_tiled.h, DISCHARGE)
        bool stale[kMcNpt];
#pragma unroll
        for (int j = 0; j < kMcNpt; j++) stale[j] = false;
        for (; git < until; git++) {
            uint16_t* hc = (git & 1) ? hgB : hgA;
            uint16_t* hn = (git & 1) ? hgA : hgB;
            const int fl = 8 + git % 3;
            if (tid == 0) sflag[8 + (git + 1) % 3] = 0;
            bool act = false;
#pragma unroll
            for (int j = 0; j < kMcNpt; j++) {
                LES_MARCH_SCHED_FENCE();
                if (stale[j]) hn[hi[j]] = (uint16_t)hv[j];                    // (the buffer this node did not write when it was raised: nobody reads it now)
                const bool on = has[j] && e[j] > 0.0f && hv[j] < BIG;
                if (!mt_wave_any(on)) continue;
                int hw[8];
                int hij = hi[j];
                MT_OPAQUE(hij);
#pragma unroll
                for (int k = 0; k < 8; k++) hw[k] = hc[hij + hoff(k)];
                if (!on) continue;
                int v = tid + j * kMcThreads;
                MT_OPAQUE(v);
                float ee = e[j];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float rk = r[j][k];
                    if (rk > 0.0f && ee > 0.0f && hv[j] > hw[k]) {             // (rk > 0 implies the neighbour exists)
                        const float d = ee < rk ? ee : rk;
                        r[j][k] = rk - d;
                        ee -= d;
                        sent[k * NP + v] = d;
                        flg[v + loff(k)] = 1;
                    }
                }
                e[j] = ee;
            }
            __syncthreads();
            {
                bool f[kMcNpt];
                uint8_t fb[kMcNpt];
#pragma unroll
                for (int j = 0; j < kMcNpt; j++) {
                    const int v = has[j] ? tid + j * kMcThreads : 0;
                    fb[j] = flg[v];                                           // (unconditional: a guarded read is a round trip of its own)
                }
#pragma unroll
                for (int j = 0; j < kMcNpt; j++) f[j] = has[j] && fb[j] != 0;
#pragma unroll
                for (int j = 0; j < kMcNpt; j++) {
                    LES_MARCH_SCHED_FENCE();
                    stale[j] = false;
                    int v = has[j] ? tid + j * kMcThreads : 0;
                    MT_OPAQUE(v);
                    if (mt_wave_any(f[j])) {
                        unsigned om = edge[j];
                        MT_OPAQUE(om);
                        float g[8];
#pragma unroll
                        for (int k = 0; k < 8; k++)                          // (the sender along k would lie outside the cell: read the own word, ignored)
                            g[k] = sent[k * NP + ((om >> (k ^ 1) & 1u) ? v : v - loff(k))];
                        if (f[j]) {
                            flg[v] = 0;
                            float add = 0.0f;
#pragma unroll
                            for (int k = 0; k < 8; k++) {
                                if ((om >> (k ^ 1) & 1u) || !(g[k] > 0.0f)) continue;
                                sent[k * NP + v - loff(k)] = 0.0f;
                                r[j][k ^ 1] += g[k];
                                add += g[k];
                            }
                            e[j] += add;                                     // (a sink arc absorbs what it can right here)
                        }
                    }
                    const bool on = has[j] && e[j] > 0.0f && hv[j] < BIG;
                    if (!mt_wave_any(on)) continue;
                    // relabel when no residual arc leads downhill -- from a SNAPSHOT (the buffer every wave reads in this iteration)
                    int hw[8];
                    int hij = hi[j];
                    MT_OPAQUE(hij);
#pragma unroll
                    for (int k = 0; k < 8; k++) hw[k] = hc[hij + hoff(k)];
                    if (!on) continue;
                    int best = BIG;
#pragma unroll
                    for (int k = 0; k < 8; k++) best = (r[j][k] > 0.0f && hw[k] + 1 < best) ? hw[k] + 1 : best;
                    if (best > hv[j]) { hv[j] = best; hn[hij] = (uint16_t)best; stale[j] = true; }
                    if (best < BIG) act = true;
                }
            }
            if (act) sflag[fl] = 1;
            __syncthreads();
            if (!sflag[fl]) { git++; break; }                                 // nothing can move any more under these heights: relabel exactly
        }
    }

    // ---- the cut: nodes that can still reach the sink keep the current label (SINK), the others take the proposal (SOURCE)
    uint8_t* m = masks + offsets[blockIdx.x];
    double t_used = 0.0;               // sink capacity of the own nodes at load time (read again: two registers less through the whole kernel) minus what is left of it
#pragma unroll
    for (int j = 0; j < kMcNpt; j++) {
        if (!has[j]) continue;
        const int v = tid + j * kMcThreads;
        m[v] = hv[j] >= BIG ? 255 : 0;
        const float tr = p5[5 * (size_t)v];
        if (tr < 0.0f) t_used += (double)(-tr);
        if (e[j] < 0.0f) t_used -= (double)(-e[j]);
    }
    // flow into the sink = sink capacity used; block reduction in the (no longer needed) exchange words
    __syncthreads();
    double* red = reinterpret_cast<double*>(sent);
    red[tid] = t_used;
    __syncthreads();
    for (int s2 = kMcThreads / 2; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
    }
    if (tid == 0) {
        status[blockIdx.x] = converged ? 0 : 1;
        if (flows) flows[blockIdx.x] = red[0];
        if (!converged && unsolved_total) atomicAdd(unsolved_total, 1);
    }
}

}  // namespace les
