// les_maxflow_tiled.h -- the minimum cut of an expansion move on the device for cells of ANY size (round 5): the cell's graph lives
// in global memory, a workgroup owns a TILE of it, and a lock-step is a sequence of launches in which every cell advances through
// its own small state machine.
//
// Replaces, for the cells les_maxflow.h cannot hold in one workgroup's LDS (the coarse layers: 129 x 129 ... 404 x 387 nodes at the
// Adirondack shape, 840 x 920 at 3000 x 2000), `graph.maxflow()` and `graph.what_segment()` of FastGCStereo::expansionMoveBK
// (LES/FastGCStereo.h:553-559) on the 5-float node payload of les_expansion_graph_kernel (les_pairwise.h), i.e. the graph of
// LES/FastGCStereo.h:411-551.
//
// Algorithm: push-relabel (Goldberg-Tarjan, first phase), region-parallel.  A cell is cut into tiles of at most 1920 nodes; a
// launch lets every tile of every unfinished cell do ONE step of the cell's phase:
//   RELABEL0  load the tile (launch 0: from the payload; later: residuals + excess, applying what the neighbouring tiles pushed
//             across the border in the previous launch), distances to the sink inside the tile alone;
//   RELABEL   relax the distances against the neighbouring tiles' (a one-node halo) until nothing changes in the tile; the
//             phase repeats until no tile of the cell changed anything: the distances are then the exact residual distances;
//   DISCHARGE up to K synchronous push / relabel iterations in LDS with the halo's heights frozen; what leaves the tile goes to an
//             outbox (one slot per node and direction) that the owner of the receiving node applies in its next launch;
//             after S sweeps, or when no excess is left that could move, back to RELABEL0;
//   FINAL     after a RELABEL phase that found no excess with a finite distance: SINK side = the nodes that can still reach the
//             sink in the residual graph (the segment rule of the reference's solver, `what_segment` with SOURCE as the default),
//             masks written, cell DONE.
// The last workgroup of a cell to finish a launch (an atomic counter per cell) decides the cell's next phase, so the host only
// enqueues launches and looks at "cells done" every few of them; tiles of finished cells return at once.
// Correctness does not rest on the labelling staying valid across tile borders (heights of the halo are one launch old): every
// push keeps a feasible preflow whatever the heights are, and the loop only ends on an EXACT relabelling that finds no excess able
// to reach the sink -- a maximum preflow, whose sink-side set is unique.  Pushes go downhill (height(v) > height(w)), which is the
// usual rule under a valid labelling and does not stall under a momentarily invalid one.  Everything is deterministic: a node is
// the only writer of its residuals, what it receives it adds in the fixed order of the eight directions, heights are relabelled from a
// snapshot (stored after a barrier), and the flow value is a sum of 64-bit integers (no float atomics).
// tools/tiled_pr_probe.py is the numpy model this was designed with (cuts identical to the host solver on dumped lock-steps).
#pragma once

#include <cstdint>

#include "les_maxflow.h"
#include "les_simt.h"

namespace les {

constexpr int kMtThreads = 1024;
constexpr int kMtNpt = 2;                                  // nodes per thread
constexpr int kMtMaxTileNodes = 1920;                      // tw * th
constexpr int kMtMaxSide = 64;                             // tw, th <= 64
constexpr int kMtMaxHalo = kMtMaxTileNodes + 2 * (kMtMaxSide + kMtMaxTileNodes / kMtMaxSide) + 4;    // (tw + 2) * (th + 2) <= 2112
static_assert(kMtThreads * kMtNpt >= kMtMaxTileNodes, "every node needs an owner");
// LDS: heights incl. halo TWICE (DISCHARGE alternates between the two: the relabel of an iteration writes the other buffer, so an iteration has two barriers
// instead of three), 8 exchange words and a flag byte per node, 16 control words, 3 x 72 row flags -- 81 184 B, two workgroups per CU.  The excess of a
// node is only ever looked at by its owner: a register.
constexpr size_t kMtLdsBytes = (size_t)kMtMaxHalo * 8 + (size_t)kMtMaxTileNodes * (8 * 4 + 1) + 64 + 3 * 72 * 4;
static_assert(2 * kMtLdsBytes <= 160 * 1024, "two workgroups per CU");

// the directions k (les_maxflow.h: E W S N SW NE SE NW) whose step has dx = +1 / dx = -1 / dy = +1 / dy = -1, as bit sets
constexpr unsigned kMtDirsE = 1u << 0 | 1u << 5 | 1u << 6, kMtDirsW = 1u << 1 | 1u << 4 | 1u << 7, kMtDirsS = 1u << 2 | 1u << 4 | 1u << 6, kMtDirsN = 1u << 3 | 1u << 5 | 1u << 7;
constexpr int mt_cdx(int k) { return (int)((0x02201102u >> (4 * k)) & 0xfu) - 1; }      // (mf_dx / mf_dy as constant expressions, for the check below)
constexpr int mt_cdy(int k) { return (int)((0x02020211u >> (4 * k)) & 0xfu) - 1; }
constexpr unsigned mt_dirs_where(int dx, int dy) { unsigned m = 0; for (int k = 0; k < 8; k++) if ((dx && mt_cdx(k) == dx) || (dy && mt_cdy(k) == dy)) m |= 1u << k; return m; }
static_assert(mt_dirs_where(1, 0) == kMtDirsE && mt_dirs_where(-1, 0) == kMtDirsW && mt_dirs_where(0, 1) == kMtDirsS && mt_dirs_where(0, -1) == kMtDirsN, "direction sets");

enum MtPhase : int { kMtRelabel0 = 0, kMtRelabel = 1, kMtDischarge = 2, kMtFinal = 3, kMtDone = 4, kMtHandover = 5 };      // (>= kMtDone: the launches leave the cell alone)
constexpr int kMtFlowShift = 22;                           // flow values are accumulated as 64-bit integers in units of 2^-22 (integer additions commute: the sum over the tiles is the same in any order)

struct MtTile { int cell, x0, y0, tw, th, W, H, pad; long long off, pad2; };   // cell-local rectangle + the cell's size and node offset; 48 B
struct MtCtl {                                                                // per cell; 64 B
    int phase, launches;
    long long votes;                  // this launch's arrivals: bits 0-20 tiles arrived, 21-41 of them "changed", 42-62 "active" (ONE relaxed atomic per tile)
    int parity, sweeps, rounds, ntiles;
    int hand, pad0;                   // hand: the cell's slot in the hand-over list (kMtHandover)
    long long hoff;                   // ... and its node offset in the hand-over staging arrays
    long long flow_fix;               // sink capacity at load time minus what is left of it, in units of 2^-kMtFlowShift
    int pad[2];
};
static_assert(sizeof(MtCtl) == 64, "one control record per 64 bytes");
struct MtHandCell { int cell, pad; long long hoff; };                         // one entry of the hand-over list (host-mapped)
struct MtHeader { int cells_done, cells_failed, pad[14]; };                   // 64 B at the start of the workspace (device-side copy of the two counters)

// workspace carve-up for `nodes` graph nodes and `cells` cells (all regions 256-byte aligned)
struct MtLayout {
    size_t ctl, r, ex, hgt, rmask, outbox, total;
};
__host__ __device__ inline MtLayout mt_layout(long long nodes, int cells)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    MtLayout L;
    size_t o = up(sizeof(MtHeader));
    L.ctl = o; o = up(o + (size_t)cells * sizeof(MtCtl));
    L.r = o; o = up(o + (size_t)nodes * 32);
    L.ex = o; o = up(o + (size_t)nodes * 4);
    L.hgt = o; o = up(o + (size_t)nodes * 8);                                 // [2 parities][node]: a launch reads the heights of the previous one
    L.rmask = o; o = up(o + (size_t)nodes);
    L.outbox = o; o = up(o + (size_t)nodes * 64);                             // [2 parities][node][8 directions]
    L.total = o;
    return L;
}

#if defined(LES_SIM)
#define MT_OPAQUE(x) ((void)0)
__device__ inline bool mt_wave_any(bool) { return true; }                    // (skipping is an optimisation only: the skipped code is a no-op for the wave)
__device__ inline int mt_atomic_add(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
__device__ inline int mt_atomic_or(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
__device__ inline int mt_load(const int* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
__device__ inline void mt_store(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
__device__ inline void mt_fence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
__device__ inline void mt_host_add(int* p, int v) { __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
__device__ inline void mt_atomic_add_i64(long long* p, long long v) { __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
__device__ inline long long mt_vote(long long* p, long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
__device__ inline long long mt_load_i64(const long long* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
#else
// Keeps the LDS addresses derived from x inside the loop they are used in: hoisted out of the iteration loop (32 of them are loop
// invariant) they do not fit the 128 registers and come back from scratch memory on every iteration; recomputing one is one v_add.
#define MT_OPAQUE(x) asm volatile("" : "+v"(x))
__device__ __forceinline__ bool mt_wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
__device__ __forceinline__ int mt_atomic_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }     // (counters: read after the launch)
__device__ __forceinline__ int mt_atomic_or(int* p, int v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int mt_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mt_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mt_fence() { __threadfence(); }
__device__ __forceinline__ void mt_host_add(int* p, int v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // (fine-grained host memory)
__device__ __forceinline__ void mt_atomic_add_i64(long long* p, long long v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the arrival of a tile: RELAXED -- nothing another tile wrote in this launch is read before the next launch (the kernel boundary orders it), and a
// release here is a write-back of the XCD's whole L2 (measured with the clock stamps of tools/lab/mt_probe.py: 4.5 us median, 30 us worst, per tile and launch)
__device__ __forceinline__ long long mt_vote(long long* p, long long v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long mt_load_i64(const long long* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// Lab build (-DLES_MT_PROBE, tools/lab/mt_probe.py): thread 0 of every tile stamps the 100-MHz clock at eight points of a launch into
// g_mt_probe[launch of the cell][tile][16] -- where the time of a launch goes.  Empty in the product.
#if defined(LES_MT_PROBE) && !defined(LES_SIM)
__device__ unsigned long long* g_mt_probe = nullptr;
__device__ int g_mt_probe_launches = 0;
#define MT_STAMP0() const unsigned long long mt_t0 = __builtin_amdgcn_s_memrealtime()
#define MT_STAMP(i) do { if (threadIdx.x == 0 && g_mt_probe && launch < g_mt_probe_launches) { unsigned long long* q_ = g_mt_probe + ((size_t)launch * gridDim.x + blockIdx.x) * 16; \
                         if ((i) == 1) { q_[0] = mt_t0; q_[2] = q_[3] = q_[4] = q_[5] = q_[6] = q_[7] = 0; q_[8] = (unsigned long long)phase; q_[9] = gridDim.x; } \
                         q_[(i)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define MT_STAMP0() ((void)0)
#define MT_STAMP(i) ((void)0)
#endif

struct MtArgs {
    const GraphCellMf* cells;        // target rects of the lock-step (x, y unused here; w, h = the cell's graph)
    const long long* offsets;        // node offset of every cell in the payload / masks / workspace arrays
    const float* payload;            // 5 floats per node (les_pairwise.h)
    const MtTile* tiles;
    char* ws;                        // workspace (mt_layout)
    long long nodes;                 // total node count of the lock-step
    int ncells;
    int K, S;                        // inner iterations per discharge sweep; sweeps between exact relabellings -- in the first round (dense: every wave has work)
    int K2, S2;                      // ... and in the later rounds (a few active nodes per tile: an iteration is cheap, a launch is not)
    int max_launches;                // a cell that needs more reports status 1 (cut on the host)
    uint8_t* masks;
    int* status;
    double* flows;                   // optional
    int* host_flags;                 // host-mapped (pinned) words the host reads after a stream synchronisation, no copy: [0] cells finished, [1] of them: gave up
};

// grid = tiles of the lock-step; block = kMtThreads; dynamic LDS = kMtLdsBytes
__global__ void __launch_bounds__(kMtThreads, 8)          // two 16-wave workgroups per CU (eight waves per SIMD) -> at most 64 VGPRs
les_maxflow_tiled_kernel(MtArgs a)
{
#if defined(LES_SIM)
    static thread_local int s_raw[kMtLdsBytes / 4 + 16];
    char* base = reinterpret_cast<char*>(s_raw);
#else
    extern __shared__ __attribute__((aligned(16))) char s_dyn_mt[];
    char* base = s_dyn_mt;
#endif
    MT_STAMP0();
    const MtTile t = a.tiles[blockIdx.x];
    const MtLayout L = mt_layout(a.nodes, a.ncells);
    MtCtl* ctl = reinterpret_cast<MtCtl*>(a.ws + L.ctl) + t.cell;
    MtHeader* hdr = reinterpret_cast<MtHeader*>(a.ws);
    const int tid = (int)threadIdx.x;
    // (written by the last tile of the previous launch: plain loads see them across the kernel boundary)
    const int phase = ctl->phase, launch = ctl->launches, parity = ctl->parity;
    const bool first_round = ctl->rounds <= 1;
    const int Kit = first_round ? a.K : a.K2, Ssw = first_round ? a.S : a.S2;
    if (phase >= kMtDone) return;
    MT_STAMP(1);

    int* hg = reinterpret_cast<int*>(base);                                 // heights / distances, halo-pitched: (th + 2) x (tw + 2)
    int* hg2 = hg + kMtMaxHalo;                                             // the second height buffer of DISCHARGE
    float* sent = reinterpret_cast<float*>(hg2 + kMtMaxHalo);               // sent[k * NP + v]
    uint8_t* flg = reinterpret_cast<uint8_t*>(sent + 8 * kMtMaxTileNodes);  // "something was sent to this node"
    int* sflag = reinterpret_cast<int*>(flg + kMtMaxTileNodes);             // [0..2] rotating "changed" flags of relax, [3] tile changed, [4] busy, [7] tile active, [8..10] rotating "still active" flags of the iterations
    int* rowchg = sflag + 16;                                               // [3][72]: rows of the tile (halo rows included) in which a distance fell, per sweep
    constexpr int NP = kMtMaxTileNodes;

    const int W = t.W, H = t.H;
    const long long off = t.off;
    const int tw = t.tw, th = t.th, n = tw * th, hp = tw + 2;
    const long long Ncell = (long long)W * H;
    const int BIG = (int)(Ncell + 2 < 0x7ffffff0ll ? Ncell + 2 : 0x7ffffff0ll);

    float* g_r = reinterpret_cast<float*>(a.ws + L.r);
    float* g_ex = reinterpret_cast<float*>(a.ws + L.ex);
    // heights are double-buffered like the outboxes: a launch that moves flow reads what the previous one wrote and writes the other
    // buffer, so what a tile sees of its neighbours never depends on which of them ran first (bit-reproducible flows)
    const int* g_h_rd = reinterpret_cast<const int*>(a.ws + L.hgt) + (size_t)(parity ^ 1) * (size_t)a.nodes;
    int* g_h_wr = reinterpret_cast<int*>(a.ws + L.hgt) + (size_t)parity * (size_t)a.nodes;
    int* g_h_cur = reinterpret_cast<int*>(a.ws + L.hgt) + (size_t)(parity ^ 1) * (size_t)a.nodes;      // RELABEL works in place on the latest buffer
    uint8_t* g_rm = reinterpret_cast<uint8_t*>(a.ws + L.rmask);
    float* g_out = reinterpret_cast<float*>(a.ws + L.outbox);               // [parity][node][8]
    const size_t out_par = (size_t)a.nodes * 8;

    // ---- own nodes: v = tid + j * kMtThreads (row-major over the tile)
    const float inv_tw = 1.0f / (float)tw;
    int hi[kMtNpt];                    // index of the node in the halo-pitched height array (a harmless own index for a missing node)
    int ly_[kMtNpt];
    bool has[kMtNpt];
    unsigned outm[kMtNpt];             // bit k: the neighbour in direction k lies outside the TILE
    unsigned incell[kMtNpt];           // bit k: the neighbour in direction k lies inside the CELL
    int gl[kMtNpt];                    // node index inside the cell (gy * W + gx); + off = global node index
#pragma unroll
    for (int j = 0; j < kMtNpt; j++) {
        const int v = tid + j * kMtThreads;
        has[j] = v < n;
        const int vv = has[j] ? v : 0;
        // (every launch of every tile pays for this set-up, 32 waves per CU at once: the quotient through one reciprocal -- vv < 2048, tw <= 64, so
        // (vv + 0.5) / tw keeps 1/128 of distance from every integer, three orders of magnitude above the rounding -- and the eight border tests as
        // four comparisons against the direction sets)
        const int ly = (int)(((float)vv + 0.5f) * inv_tw), lx = vv - ly * tw;
        ly_[j] = ly;
        hi[j] = (ly + 1) * hp + lx + 1;
        const int gx = t.x0 + lx, gy = t.y0 + ly;
        gl[j] = gy * W + gx;
        outm[j] = (lx == tw - 1 ? kMtDirsE : 0u) | (lx == 0 ? kMtDirsW : 0u) | (ly == th - 1 ? kMtDirsS : 0u) | (ly == 0 ? kMtDirsN : 0u);
        incell[j] = 0xffu & ~((gx == W - 1 ? kMtDirsE : 0u) | (gx == 0 ? kMtDirsW : 0u) | (gy == H - 1 ? kMtDirsS : 0u) | (gy == 0 ? kMtDirsN : 0u));
        if (!has[j]) { outm[j] = 0; incell[j] = 0; }
    }
    auto gi = [&](int j) -> size_t { return (size_t)(off + gl[j]); };
    auto hoff = [&](int k) { return mf_dy(k) * hp + mf_dx(k); };            // neighbour k in the halo-pitched array
    auto loff = [&](int k) { return mf_dy(k) * tw + mf_dx(k); };            // neighbour k in the tile-local arrays
    auto goff = [&](int k) { return (long long)mf_dy(k) * W + mf_dx(k); };  // neighbour k in the global arrays

    // No barrier behind this initialisation on the device: every use of these words lies behind the phase's own first barrier (the one that waits for
    // the tile's loads) -- measured with the clock stamps: a barrier here costs 2.9 us per launch, the waves of a workgroup do not start together.
    // DISCHARGE keeps its "is there anything to do in this tile" vote in rowchg[0 .. 15] (one word per wave, written unconditionally) and does not use the row flags.
    if (tid < 16) sflag[tid] = 0;
    if (phase != kMtDischarge)
        for (int i = tid; i < 3 * 72; i += kMtThreads) rowchg[i] = i < 72 ? 1 : 0;     // sweep 0 looks at every row
#if defined(LES_SIM)
    __syncthreads();
#endif
    MT_STAMP(2);

    // halo of the height array from the global heights (nodes outside the cell: BIG); own nodes are loaded by the phases
    auto load_halo = [&](bool from_global, bool both = false) {
        const int ring = 2 * (tw + 2) + 2 * th;
        for (int i = tid; i < ring; i += kMtThreads) {
            int hx, hy;
            if (i < tw + 2) { hx = i; hy = 0; }
            else if (i < 2 * (tw + 2)) { hx = i - (tw + 2); hy = th + 1; }
            else if (i < 2 * (tw + 2) + th) { hx = 0; hy = i - 2 * (tw + 2) + 1; }
            else { hx = tw + 1; hy = i - 2 * (tw + 2) - th + 1; }
            const int gx = t.x0 + hx - 1, gy = t.y0 + hy - 1;
            int val = BIG;
            if (from_global && gx >= 0 && gx < W && gy >= 0 && gy < H) val = g_h_rd[off + (long long)gy * W + gx];
            hg[hy * hp + hx] = val;
            if (both) hg2[hy * hp + hx] = val;
        }
    };

    // what the neighbouring tiles pushed towards node j in the previous outbox launch (added in direction order)
    auto apply_inbox = [&](int j, float (&r)[8], float& e) -> bool {        // -> something arrived
        const float* ob = g_out + (size_t)(parity ^ 1) * out_par;
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; k++)             // the sender along k is the neighbour in direction k ^ 1
            g[k] = ((outm[j] >> (k ^ 1) & 1u) && (incell[j] >> (k ^ 1) & 1u)) ? ob[(size_t)(gi(j) - goff(k)) * 8 + k] : 0.0f;
        bool any = false;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (g[k] > 0.0f) { r[k ^ 1] += g[k]; e += g[k]; any = true; }
        return any;
    };
    auto store_r = [&](int j, const float (&r)[8]) {
        float4* p = reinterpret_cast<float4*>(g_r + (size_t)gi(j) * 8);
        p[0] = make_float4(r[0], r[1], r[2], r[3]);
        p[1] = make_float4(r[4], r[5], r[6], r[7]);
    };
    auto load_r = [&](int j, float (&r)[8]) {
        const float4* p = reinterpret_cast<const float4*>(g_r + (size_t)gi(j) * 8);
        const float4 lo = p[0], hi4 = p[1];
        r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi4.x; r[5] = hi4.y; r[6] = hi4.z; r[7] = hi4.w;
    };
    auto store_outbox = [&](int j, const float (&o)[8]) {
        float4* p = reinterpret_cast<float4*>(g_out + (size_t)parity * out_par + (size_t)gi(j) * 8);
        p[0] = make_float4(o[0], o[1], o[2], o[3]);
        p[1] = make_float4(o[4], o[5], o[6], o[7]);
    };

    // Residual distances to the sink inside the tile, against the (fixed) halo: chaotic relaxation over the live values, one barrier
    // per sweep; the "changed" flag rotates through three words (les_maxflow.h).  rm[j]: bit k = arc k of node j is residual.  A sweep
    // only looks at nodes next to a row in which a distance fell in the previous sweep (row flags, rotating like the "changed" word):
    // the frontier of such a relaxation is a band of the tile, and a wave whose rows are quiet skips its LDS reads altogether.
    auto relax = [&](const unsigned (&rm)[kMtNpt]) {
        for (int s = 0;; s++) {
            const int cur = s % 3, nxt = (s + 1) % 3, nn2 = (s + 2) % 3;
            if (tid == 0) sflag[nxt] = 0;
            if (tid < 72) rowchg[nn2 * 72 + tid] = 0;                        // (read in sweep s + 2; last read in sweep s - 1)
            bool changed = false;
#pragma unroll
            for (int j = 0; j < kMtNpt; j++) {
                const int y = ly_[j] + 1;                                    // halo-pitched row
                const bool look = has[j] && (rowchg[cur * 72 + y - 1] | rowchg[cur * 72 + y] | rowchg[cur * 72 + y + 1]) != 0;
                if (!mt_wave_any(look)) continue;
                const int d = hg[hi[j]];
                int dn[8];
#pragma unroll
                for (int k = 0; k < 8; k++) dn[k] = hg[hi[j] + hoff(k)];
                int best = d;
#pragma unroll
                for (int k = 0; k < 8; k++) best = ((rm[j] >> k & 1u) && dn[k] + 1 < best) ? dn[k] + 1 : best;
                if (look && best < d) { hg[hi[j]] = best; rowchg[nxt * 72 + y] = 1; changed = true; }
            }
            if (changed) sflag[cur] = 1;
            __syncthreads();
            if (!sflag[cur]) break;
        }
    };

    bool tile_changed = false, tile_active = false;

    if (phase == kMtRelabel0) {
        // ---- (re)load the tile's residuals, fold in the inbox, distances inside the tile alone
        load_halo(false);
        unsigned rm[kMtNpt];
        float ex0[kMtNpt];
        double t_in = 0.0;
#pragma unroll
        for (int j = 0; j < kMtNpt; j++) {
            rm[j] = 0; ex0[j] = 0.0f;
            if (!has[j]) continue;
            float r[8], e;
            if (launch == 0) {
                const float* p5 = a.payload + 5 * (size_t)gi(j);
                e = p5[0];
                r[0] = (incell[j] >> 0 & 1u) ? p5[1] : 0.0f;          // arcs that would leave the region carry no capacity
                r[2] = (incell[j] >> 2 & 1u) ? p5[2] : 0.0f;
                r[4] = (incell[j] >> 4 & 1u) ? p5[3] : 0.0f;
                r[6] = (incell[j] >> 6 & 1u) ? p5[4] : 0.0f;
                r[1] = r[3] = r[5] = r[7] = 0.0f;
                if (e < 0.0f) t_in += (double)(-e);
            } else {
                load_r(j, r);
                e = g_ex[gi(j)];
                if (outm[j]) apply_inbox(j, r, e);
            }
            store_r(j, r);
            g_ex[gi(j)] = e;
            ex0[j] = e;
#pragma unroll
            for (int k = 0; k < 8; k++) rm[j] |= (r[k] > 0.0f ? 1u : 0u) << k;
            g_rm[gi(j)] = (uint8_t)rm[j];
            hg[hi[j]] = e < 0.0f ? 1 : BIG;
            if (outm[j]) {                                            // nothing in flight after this launch
                const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                store_outbox(j, z);
                if (launch == 0) {
                    float4* p = reinterpret_cast<float4*>(g_out + (size_t)(parity ^ 1) * out_par + (size_t)gi(j) * 8);
                    p[0] = make_float4(0, 0, 0, 0); p[1] = make_float4(0, 0, 0, 0);
                }
            }
        }
        if (launch == 0 && a.flows) {
            // sink capacity of the cell at load time (the flow value is what of it has been used at the end)
            double* red = reinterpret_cast<double*>(sent);
            red[tid] = t_in;
            __syncthreads();
            for (int s2 = kMtThreads / 2; s2 > 0; s2 >>= 1) {
                if (tid < s2) red[tid] += red[tid + s2];
                __syncthreads();
            }
            if (tid == 0 && red[0] != 0.0) mt_atomic_add_i64(&ctl->flow_fix, (long long)rint(red[0] * (double)(1ll << kMtFlowShift)));
        }
        __syncthreads();
        MT_STAMP(3);
        relax(rm);
#pragma unroll
        for (int j = 0; j < kMtNpt; j++) {
            if (!has[j]) continue;
            const int d = hg[hi[j]];
            g_h_wr[gi(j)] = d;
            if (ex0[j] > 0.0f && d < BIG) tile_active = true;
        }
        tile_changed = true;                                          // (the neighbours have not seen these distances yet)
    } else if (phase == kMtRelabel) {
        load_halo(true);
        unsigned rm[kMtNpt];
        int d0[kMtNpt];
#pragma unroll
        for (int j = 0; j < kMtNpt; j++) {
            rm[j] = 0; d0[j] = 0;
            if (!has[j]) continue;
            rm[j] = g_rm[gi(j)];
            d0[j] = g_h_cur[gi(j)];
            hg[hi[j]] = d0[j];
        }
        __syncthreads();
        MT_STAMP(3);
        relax(rm);
#pragma unroll
        for (int j = 0; j < kMtNpt; j++) {
            if (!has[j]) continue;
            const int d = hg[hi[j]];
            if (d != d0[j]) { g_h_cur[gi(j)] = d; tile_changed = true; }
            if (d < BIG && g_ex[gi(j)] > 0.0f) tile_active = true;
        }
    } else if (phase == kMtDischarge) {
        load_halo(true, true);
        float r[kMtNpt][8];
        float e[kMtNpt];                   // excess of the own nodes (after what the neighbouring tiles pushed across the border)
        int hv[kMtNpt];                    // heights of the own nodes (the neighbours read them from LDS)
        bool mine = false;
#pragma unroll
        for (int j = 0; j < kMtNpt; j++) {
            e[j] = 0.0f; hv[j] = BIG;
#pragma unroll
            for (int k = 0; k < 8; k++) r[j][k] = 0.0f;
            if (!has[j]) continue;
            const int v = tid + j * kMtThreads;
            hv[j] = g_h_rd[gi(j)];
            hg[hi[j]] = hv[j];
            hg2[hi[j]] = hv[j];
            const float e_in = g_ex[gi(j)];
            // the residuals are loaded together with the rest of the tile's state (one trip to memory instead of two: an idle tile reads them for nothing)
            load_r(j, r[j]);
            e[j] = e_in;
            const bool got = outm[j] ? apply_inbox(j, r[j], e[j]) : false;
            flg[v] = 0;
            if ((e_in > 0.0f && hv[j] < BIG) || got) mine = true;
        }
#if defined(LES_SIM)
        if (mine) sflag[4] = 1;
        __syncthreads();
        const bool busy = sflag[4] != 0;
#else
        {
            const bool any = mt_wave_any(mine);
            if ((tid & 63) == 0) rowchg[tid >> 6] = any ? 1 : 0;
        }
        __syncthreads();
        bool busy = false;
#pragma unroll
        for (int w = 0; w < kMtThreads / 64; w++) busy = busy || rowchg[w] != 0;
#endif
        MT_STAMP(3);
        if (busy) {
#pragma unroll
            for (int j = 0; j < kMtNpt; j++) {
                if (!has[j]) continue;
                const int v = tid + j * kMtThreads;
#pragma unroll
                for (int k = 0; k < 8; k++) sent[k * NP + v] = 0.0f;
            }
            __syncthreads();
            MT_STAMP(4);
            // ---- K synchronous iterations: pushes | barrier | receive + relabel | barrier.  All LDS reads of a step are issued together and
            // unconditionally (a chain of conditional reads costs a round trip each); a wave none of whose lanes has work skips the step.
            // The heights live in two buffers: an iteration reads `hc`, its relabel writes the raised heights into `hn`, which the next iteration
            // reads -- the new height of a node is never stored where a neighbour may still be reading the old one (the snapshot rule that makes
            // the flows bit-reproducible) without a third barrier.  A raised node brings the buffer it did NOT write up to date in the push half
            // of the next iteration, when nobody reads it.
            bool stale[kMtNpt];
#pragma unroll
            for (int j = 0; j < kMtNpt; j++) stale[j] = false;
            for (int it = 0; it < Kit; it++) {
                int* hc = (it & 1) ? hg2 : hg;
                int* hn = (it & 1) ? hg : hg2;
                const int fl = 8 + it % 3;
                if (tid == 0) sflag[8 + (it + 1) % 3] = 0;
                bool act = false;
#pragma unroll
                for (int j = 0; j < kMtNpt; j++) {
                    LES_MARCH_SCHED_FENCE();                                  // (keeps the eight reads of one node together instead of hoisting all thirty-two)
                    if (stale[j]) hn[hi[j]] = hv[j];
                    const bool on = has[j] && e[j] > 0.0f && hv[j] < BIG;
                    if (!mt_wave_any(on)) continue;
                    int hw[8];
                    int hij = hi[j];
                    MT_OPAQUE(hij);
#pragma unroll
                    for (int k = 0; k < 8; k++) hw[k] = hc[hij + hoff(k)];
                    if (!on) continue;
                    int v = tid + j * kMtThreads;
                    MT_OPAQUE(v);
                    unsigned om = outm[j];
                    MT_OPAQUE(om);                                            // (tested bit by bit here: 32 precomputed lane masks would live in spilled SGPRs)
                    float ee = e[j];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const float rk = r[j][k];
                        if (rk > 0.0f && ee > 0.0f && hv[j] > hw[k]) {
                            const float d = ee < rk ? ee : rk;
                            r[j][k] = rk - d;
                            ee -= d;
                            if (om >> k & 1u) sent[k * NP + v] += d;             // leaves the tile: accumulates for the outbox
                            else { sent[k * NP + v] = d; flg[v + loff(k)] = 1; }
                        }
                    }
                    e[j] = ee;
                }
                __syncthreads();
                {
                    bool f[kMtNpt];
                    uint8_t fb[kMtNpt];
#pragma unroll
                    for (int j = 0; j < kMtNpt; j++) {
                        const int v = has[j] ? tid + j * kMtThreads : 0;
                        fb[j] = flg[v];                                           // (unconditional: a guarded read is a round trip of its own)
                    }
#pragma unroll
                    for (int j = 0; j < kMtNpt; j++) f[j] = has[j] && fb[j] != 0;
#pragma unroll
                    for (int j = 0; j < kMtNpt; j++) {
                        LES_MARCH_SCHED_FENCE();
                        stale[j] = false;
                        int v = has[j] ? tid + j * kMtThreads : 0;
                        MT_OPAQUE(v);
                        if (mt_wave_any(f[j])) {
                            unsigned om = outm[j];
                            MT_OPAQUE(om);
                            float g[8];
#pragma unroll
                            for (int k = 0; k < 8; k++)                          // (the sender along k lies outside the tile: read the own word, ignored)
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
                        if (best > hv[j]) { hv[j] = best; hn[hij] = best; stale[j] = true; }
                        if (best < BIG) act = true;
                    }
                }
                if (act) sflag[fl] = 1;
                __syncthreads();
                if (!sflag[fl]) break;
            }
            MT_STAMP(5);
            // ---- write the tile back
#pragma unroll
            for (int j = 0; j < kMtNpt; j++) {
                if (!has[j]) continue;
                const int v = tid + j * kMtThreads;
                store_r(j, r[j]);
                g_ex[gi(j)] = e[j];
                g_h_wr[gi(j)] = hv[j];
                if (e[j] > 0.0f && hv[j] < BIG) tile_active = true;
                if (outm[j]) {
                    float o[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        o[k] = (outm[j] >> k & 1u) ? sent[k * NP + v] : 0.0f;
                        if (o[k] > 0.0f) tile_active = true;                        // in flight: the receiver may become active
                    }
                    store_outbox(j, o);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kMtNpt; j++) {
                if (!has[j]) continue;
                g_h_wr[gi(j)] = hv[j];                                  // (an idle tile only carries its heights over)
                if (!outm[j]) continue;
                const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                store_outbox(j, z);
            }
        }
    } else {        // kMtFinal
        uint8_t* m = a.masks;
        double t_out = 0.0;
#pragma unroll
        for (int j = 0; j < kMtNpt; j++) {
            if (!has[j]) continue;
            m[gi(j)] = g_h_rd[gi(j)] >= BIG ? 255 : 0;
            const float e = g_ex[gi(j)];
            if (e < 0.0f) t_out += (double)(-e);
        }
        if (a.flows) {
            double* red = reinterpret_cast<double*>(sent);
            red[tid] = t_out;
            __syncthreads();
            for (int s2 = kMtThreads / 2; s2 > 0; s2 >>= 1) {
                if (tid < s2) red[tid] += red[tid + s2];
                __syncthreads();
            }
            if (tid == 0 && red[0] != 0.0) mt_atomic_add_i64(&ctl->flow_fix, -(long long)rint(red[0] * (double)(1ll << kMtFlowShift)));
        }
    }

    // ---- the cell's verdict on this launch: every tile reports, the last one to arrive decides the next phase
    if (tile_changed) sflag[3] = 1;
    if (tile_active) sflag[7] = 1;
    __syncthreads();
    MT_STAMP(6);
    if (tid == 0) {
        // (the flow value is read by the last tile of the FINAL launch: there, and only there, a tile's share must be visible before its arrival)
        const bool ordered = a.flows != nullptr && phase == kMtFinal;
        if (ordered) mt_fence();
        const long long vote = 1ll + (sflag[3] ? 1ll << 21 : 0ll) + (sflag[7] ? 1ll << 42 : 0ll);
        const long long all = mt_vote(&ctl->votes, vote) + vote;
        if ((int)(all & 0x1fffff) == ctl->ntiles) {
            if (ordered) mt_fence();
            const int changed = (int)((all >> 21) & 0x1fffff), active = (int)((all >> 42) & 0x1fffff);
            int next = phase, par = parity, sweeps = ctl->sweeps, rounds = ctl->rounds;
            if (phase == kMtRelabel0) { next = ctl->ntiles > 1 ? kMtRelabel : (active ? kMtDischarge : kMtFinal); par ^= 1; sweeps = 0; rounds++; }
            else if (phase == kMtRelabel) next = changed ? kMtRelabel : (active ? kMtDischarge : kMtFinal);
            else if (phase == kMtDischarge) { par ^= 1; sweeps++; next = (!active || sweeps >= Ssw) ? kMtRelabel0 : kMtDischarge; }
            else next = kMtDone;
            if (next == kMtDone) {
                a.status[t.cell] = 0;
                if (a.flows) a.flows[t.cell] = (double)mt_load_i64(&ctl->flow_fix) * (1.0 / (double)(1ll << kMtFlowShift));      // (every tile's share arrived before its count did)
                mt_atomic_add(&hdr->cells_done, 1);
                mt_host_add(a.host_flags, 1);
            } else if (launch + 1 >= a.max_launches) {
                next = kMtDone;                                          // gives up: status stays 1, the caller cuts the cell on the host
                mt_atomic_add(&hdr->cells_done, 1);
                mt_atomic_add(&hdr->cells_failed, 1);
                mt_host_add(a.host_flags + 1, 1);
                mt_host_add(a.host_flags, 1);
            }
            ctl->parity = par; ctl->sweeps = sweeps; ctl->rounds = rounds;
            ctl->votes = 0;
            ctl->launches = launch + 1;
            ctl->phase = next;
        }
    }
    MT_STAMP(7);
}

// sets up the per-cell control words of a lock-step: grid = ceil(cells / 256), block = 256
__global__ void les_maxflow_tiled_init_kernel(char* ws, long long nodes, int ncells, const int* __restrict__ tiles_per_cell, int* __restrict__ status,
                                              double* __restrict__ flows, int* __restrict__ host_flags)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const MtLayout L = mt_layout(nodes, ncells);
    if (i == 0) {
        MtHeader* h = reinterpret_cast<MtHeader*>(ws);
        h->cells_done = 0; h->cells_failed = 0;
    }
    if (i >= ncells) return;
    MtCtl* c = reinterpret_cast<MtCtl*>(ws + L.ctl) + i;
    c->phase = kMtRelabel0; c->votes = 0;
    c->parity = 0; c->sweeps = 0; c->launches = 0; c->rounds = 0;
    c->ntiles = tiles_per_cell[i];
    c->hand = -1; c->pad0 = 0; c->hoff = 0; c->flow_fix = 0;
    status[i] = 1;
    if (flows) flows[i] = 0.0;
    if (tiles_per_cell[i] == 0) {                     // an empty cell has nothing to cut
        c->phase = kMtDone;
        status[i] = 0;
        mt_atomic_add(&reinterpret_cast<MtHeader*>(ws)->cells_done, 1);
        mt_host_add(host_flags, 1);
    }
}

// ---- hand-over of straggler cells to the host cores (round 6; host/ResidualCut.h) -------------------------------------------------
// A lock-step lasts as long as its slowest cell, and the scheme above is at its worst on the tail of a hard cell (a few hundred small
// excesses, hundreds of launches, one cell's tiles on a 256-CU chip).  Once few cells are still open the host stops enqueueing launches:
//   collect  (one thread) lists the open cells if they are few enough -- all of them or none --, gives every one a slot and a node offset in
//            the staging arrays and parks it in kMtHandover (launches leave it alone);
//   pack     (grid = tiles) writes the residual graph of the parked cells -- 8 residual capacities and the excess of every node, with what
//            the neighbouring tiles still had in flight folded in exactly as the next launch would have -- into host-mapped memory, and
//            takes the sink capacity that is left out of the cell's flow value;
//   the host finishes each cell with a search from the remaining excess nodes (same cut: the residual graph of a feasible preflow has the
//            minimum cuts of the graph it came from) and writes the masks and the flow it routed into host-mapped memory;
//   unpack   (grid = tiles) copies the masks into place and completes the flow value and the status word.
struct MtHandArgs {
    const MtTile* tiles;
    char* ws;
    long long nodes;
    int ncells;
    const GraphCellMf* cells;
    int max_cells;                   // policy: hand over only when at most this many cells ...
    long long max_nodes;             // ... of at most this many nodes in total are still open
    long long cap_nodes;             // capacity of the staging arrays (never exceeded, whatever the policy says)
    long long max_cell_nodes;        // policy: ... and none of them is larger than this
    MtHandCell* list;                // host-mapped, [max_cells]
    float* rc8;                      // host-mapped staging: [max_nodes][8]
    float* ex;                       // [max_nodes]
    const uint8_t* hmasks;           // [max_nodes], written by the host
    const double* hflows;            // [max_cells], written by the host: the flow it routed
    uint8_t* masks;
    int* status;
    double* flows;                   // optional
    int* host_flags;                 // [2] cells handed over, [3] their nodes
};

__global__ void les_maxflow_tiled_collect_kernel(MtHandArgs a)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const MtLayout L = mt_layout(a.nodes, a.ncells);
    MtCtl* ctl = reinterpret_cast<MtCtl*>(a.ws + L.ctl);
    int open = 0;
    long long nodes = 0;
    long long largest = 0;
    for (int i = 0; i < a.ncells; i++)
        if (ctl[i].phase < kMtDone) {
            const long long cn = (long long)a.cells[i].w * a.cells[i].h;
            open++; nodes += cn; largest = cn > largest ? cn : largest;
        }
    if (open == 0 || open > a.max_cells || nodes > a.max_nodes || nodes > a.cap_nodes || largest > a.max_cell_nodes) { mt_store(a.host_flags + 2, 0); return; }
    int slot = 0;
    long long hoff = 0;
    for (int i = 0; i < a.ncells; i++) {
        if (ctl[i].phase >= kMtDone) continue;
        ctl[i].hand = slot; ctl[i].hoff = hoff;
        ctl[i].phase = kMtHandover;
        a.list[slot].cell = i; a.list[slot].pad = 0; a.list[slot].hoff = hoff;
        slot++;
        hoff += (long long)a.cells[i].w * a.cells[i].h;
    }
    mt_fence();
    mt_store(a.host_flags + 3, (int)hoff);
    mt_store(a.host_flags + 2, slot);
}

// grid = tiles of the lock-step; block = kMtThreads
__global__ void __launch_bounds__(kMtThreads)
les_maxflow_tiled_pack_kernel(MtHandArgs a)
{
#if defined(LES_SIM)
    static thread_local double red[kMtThreads];
#else
    __shared__ double red[kMtThreads];
#endif
    const MtTile t = a.tiles[blockIdx.x];
    const MtLayout L = mt_layout(a.nodes, a.ncells);
    MtCtl* ctl = reinterpret_cast<MtCtl*>(a.ws + L.ctl) + t.cell;
    if (ctl->phase != kMtHandover) return;
    const int tid = (int)threadIdx.x;
    const int parity = ctl->parity;
    const long long hoff = ctl->hoff;
    const int W = t.W, H = t.H, tw = t.tw, th = t.th, n = tw * th;
    const float* g_r = reinterpret_cast<const float*>(a.ws + L.r);
    const float* g_ex = reinterpret_cast<const float*>(a.ws + L.ex);
    const float* ob = reinterpret_cast<const float*>(a.ws + L.outbox) + (size_t)(parity ^ 1) * (size_t)a.nodes * 8;     // what is still in flight (zeros outside a discharge)
    double t_out = 0.0;
    for (int v = tid; v < n; v += kMtThreads) {
        const int ly = v / tw, lx = v - ly * tw;
        const int gx = t.x0 + lx, gy = t.y0 + ly;
        const long long gl = (long long)gy * W + gx;
        const size_t gi = (size_t)(t.off + gl);
        float r[8];
        const float4* p = reinterpret_cast<const float4*>(g_r + gi * 8);
        const float4 lo = p[0], hi4 = p[1];
        r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi4.x; r[5] = hi4.y; r[6] = hi4.z; r[7] = hi4.w;
        float e = g_ex[gi];
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {            // the sender along k is the neighbour in direction k ^ 1: counted when it lies outside the tile and inside the cell
            const int sx = lx + mf_dx(k ^ 1), sy = ly + mf_dy(k ^ 1);
            const int cx = gx + mf_dx(k ^ 1), cy = gy + mf_dy(k ^ 1);
            const bool out_tile = sx < 0 || sx >= tw || sy < 0 || sy >= th;
            const bool in_cell = cx >= 0 && cx < W && cy >= 0 && cy < H;
            g[k] = (out_tile && in_cell) ? ob[(size_t)((long long)gi - ((long long)mf_dy(k) * W + mf_dx(k))) * 8 + k] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (g[k] > 0.0f) { r[k ^ 1] += g[k]; e += g[k]; }
        float4* q = reinterpret_cast<float4*>(a.rc8 + (size_t)(hoff + gl) * 8);
        q[0] = make_float4(r[0], r[1], r[2], r[3]);
        q[1] = make_float4(r[4], r[5], r[6], r[7]);
        a.ex[hoff + gl] = e;
        if (e < 0.0f) t_out += (double)(-e);
    }
    if (a.flows) {
        red[tid] = t_out;
        __syncthreads();
        for (int s2 = kMtThreads / 2; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] += red[tid + s2];
            __syncthreads();
        }
        if (tid == 0 && red[0] != 0.0) mt_atomic_add_i64(&ctl->flow_fix, -(long long)rint(red[0] * (double)(1ll << kMtFlowShift)));
    }
}

// grid = tiles of the lock-step; block = 256
__global__ void les_maxflow_tiled_unpack_kernel(MtHandArgs a)
{
    const MtTile t = a.tiles[blockIdx.x];
    const MtLayout L = mt_layout(a.nodes, a.ncells);
    MtCtl* ctl = reinterpret_cast<MtCtl*>(a.ws + L.ctl) + t.cell;
    if (ctl->phase != kMtHandover) return;
    const long long hoff = ctl->hoff;
    const int n = t.tw * t.th;
    for (int v = (int)threadIdx.x; v < n; v += (int)blockDim.x) {
        const int ly = v / t.tw, lx = v - ly * t.tw;
        const long long gl = (long long)(t.y0 + ly) * t.W + t.x0 + lx;
        a.masks[t.off + gl] = a.hmasks[hoff + gl];
    }
    if (threadIdx.x == 0 && t.x0 == 0 && t.y0 == 0) {
        a.status[t.cell] = 0;
        if (a.flows) a.flows[t.cell] = (double)mt_load_i64(&ctl->flow_fix) * (1.0 / (double)(1ll << kMtFlowShift)) + a.hflows[ctl->hand];
    }
}

}  // namespace les
