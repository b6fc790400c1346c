// les_maxflow.h -- the minimum cut of an expansion move on the device, one workgroup per cell, the whole graph in LDS.
//
// Replaces, for cells that fit (layer-0 cells: 42 x 42 .. 45 x 45 nodes), the host side of `expansionMoveBK` after the graph
// construction -- `graph.maxflow()` and `graph.what_segment()` (LES/FastGCStereo.h:553-559) -- on the 5-float node payload of
// les_expansion_graph_kernel (les_pairwise.h): terminal residual + capacities of the arcs to the E, S, SW, SE neighbours.
//
// Algorithm: synchronous push-relabel (Goldberg-Tarjan, first phase only) in the data-parallel form that needs no atomics:
//   per iteration   1. every active node pushes to the sink if its height is 1;
//                   2. for each of the 8 grid directions in turn: every active node pushes along its arc of that direction when
//                      the arc is admissible (height(v) == height(w) + 1); a node receives from at most one sender per direction,
//                      so the receiving side is a second pass behind a barrier (`sent`), no two lanes ever write the same word;
//                   3. active nodes without an admissible arc are relabelled from a snapshot of the heights;
//   every G iterations and at the end: global relabelling = residual distance to the sink by Jacobi sweeps until nothing changes.
// Termination: no node with excess can reach the sink.  The cut is then read off the final distances: SINK side = the nodes that
// can still reach the sink in the residual graph, which is the segment rule of the reference's solver (`what_segment` with
// SOURCE as the default) and does not depend on which maximum preflow was found.  Capacities float as in Graph<float,float,double>.
// tools/pushrelabel_probe.py is the numpy model of exactly this scheme (median 16 iterations on 42 x 42 crops of real graphs,
// cuts identical to the host solver).  A cell that does not converge within LES_MF_MAX_ITER reports status 1 and is cut on the host.
#pragma once

#include <cstdint>

#include "les_simt.h"

namespace les {

// 512 threads = 8 waves per workgroup; a 42 x 42 cell needs 81 208 B of LDS, so two workgroups share a CU (4 waves per SIMD): the
// kernel is a chain of dependent LDS accesses between barriers and lives on latency hiding (measured: 256 threads and one
// workgroup per CU took 8.6 ms for a lock-step of 450 cells that the host team cuts in 6.1 ms)
constexpr int kMfThreads = 512;
constexpr int kMfNodesPerThread = 5;                      // up to 2560 >= 2304 nodes (48 x 48) per cell
constexpr int kMfMaxNodes = 2304;
static_assert(kMfThreads * kMfNodesPerThread >= kMfMaxNodes, "every node needs an owner");
#ifndef LES_MF_G
#define LES_MF_G 16
#endif
constexpr int kMfGlobalRelabelEvery = LES_MF_G;
#ifndef LES_MF_MAX_ITER
#define LES_MF_MAX_ITER 6000
#endif

// LDS bytes for a launch whose largest cell has `nodes` nodes: r[8] + e + tcap + sent (float each) + height (uint16), + flags
__host__ __device__ inline size_t mf_lds_bytes(int nodes)
{
    const size_t n = (size_t)((nodes + 7) / 8) * 8;
    const size_t bytes = n * (8 * 4 + 4 + 4 + 4 + 2) + 64;
    return bytes < 4160 ? 4160 : bytes;                    // (the final reduction borrows kMfThreads doubles)
}

struct GraphCellMf { int x, y, w, h; };                   // same layout as GraphCell (les_pairwise.h)

// arc directions E W S N SW NE SE NW (sister(k) == k ^ 1) as offsets; packed in nibbles so that a run-time direction index needs
// no indexed register array
__host__ __device__ inline int mf_dx(int k) { return (int)((0x02201102u >> (4 * k)) & 0xfu) - 1; }      // +1 -1  0  0 -1 +1 +1 -1
__host__ __device__ inline int mf_dy(int k) { return (int)((0x02020211u >> (4 * k)) & 0xfu) - 1; }      //  0  0 +1 -1 +1 -1 +1 -1

// grid = cells; block = kMfThreads; dynamic LDS = mf_lds_bytes(max nodes of the launch)
__global__ void __launch_bounds__(kMfThreads)
les_maxflow_kernel(const GraphCellMf* __restrict__ cells, const long long* __restrict__ offsets, const float* __restrict__ payload,
                   int nmax_padded, uint8_t* __restrict__ masks, int* __restrict__ status, double* __restrict__ flows)
{
#if defined(LES_SIM)
    static thread_local float s_raw[(kMfMaxNodes * 46 + 4160) / 4 + 16];
    char* base = reinterpret_cast<char*>(s_raw);
#else
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    char* base = s_dyn;
#endif
    const int NP = nmax_padded;                            // array pitch (multiple of 8)
    float* r = reinterpret_cast<float*>(base);             // r[k * NP + v], k = E W S N SW NE SE NW (sister = k ^ 1)
    float* e = r + 8 * NP;
    float* tcap = e + NP;
    float* sent = tcap + NP;
    uint16_t* hgt = reinterpret_cast<uint16_t*>(sent + NP);
    int* flag = reinterpret_cast<int*>(hgt + NP);          // [0] any active, [1] relabel sweep changed something

    const GraphCellMf c = cells[blockIdx.x];
    const int W = c.w, H = c.h, N = W * H;
    const int tid = (int)threadIdx.x;
    const float* p5 = payload + 5 * offsets[blockIdx.x];
    if (N <= 0) { if (tid == 0) { status[blockIdx.x] = 0; if (flows) flows[blockIdx.x] = 0.0; } return; }
    const int BIG = N + 2;                                 // "cannot reach the sink" (fits uint16: N <= 2304)

    // ---- own nodes: v = tid + j * 256
    int vx[kMfNodesPerThread], vy[kMfNodesPerThread];
#pragma unroll
    for (int j = 0; j < kMfNodesPerThread; j++) {
        const int v = tid + j * kMfThreads;
        vy[j] = v < N ? v / W : -1;
        vx[j] = v < N ? v - vy[j] * W : 0;
    }
    double t_in = 0.0;                                     // sink capacity of the own nodes at load time (for the flow value)
#pragma unroll
    for (int j = 0; j < kMfNodesPerThread; j++) {
        const int v = tid + j * kMfThreads;
        if (v >= N) continue;
        const float tr = p5[5 * v];
        const int x = vx[j], y = vy[j];
        // arcs that would leave the region carry no capacity
        r[0 * NP + v] = (x + 1 < W) ? p5[5 * v + 1] : 0.0f;
        r[2 * NP + v] = (y + 1 < H) ? p5[5 * v + 2] : 0.0f;
        r[4 * NP + v] = (y + 1 < H && x > 0) ? p5[5 * v + 3] : 0.0f;
        r[6 * NP + v] = (y + 1 < H && x + 1 < W) ? p5[5 * v + 4] : 0.0f;
        r[1 * NP + v] = 0.0f; r[3 * NP + v] = 0.0f; r[5 * NP + v] = 0.0f; r[7 * NP + v] = 0.0f;
        e[v] = tr > 0.0f ? tr : 0.0f;
        tcap[v] = tr < 0.0f ? -tr : 0.0f;
        t_in += (double)tcap[v];
        hgt[v] = 0;
    }
    __syncthreads();

    // residual distance to the sink (Jacobi sweeps); on return hgt = max(hgt, distance) when raise_only, else = distance
    int dbg_sweeps = 0;                                    // (measurement builds report it through `flows`)
    auto global_relabel = [&](bool raise_only) {
        int d[kMfNodesPerThread];
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * kMfThreads;
            d[j] = (v < N && tcap[v] > 0.0f) ? 1 : BIG;
        }
        // the distances live in `sent` (as integers) during the sweeps: the heights stay readable for raise_only
        int* dist = reinterpret_cast<int*>(sent);
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * kMfThreads;
            if (v < N) dist[v] = d[j];
        }
        __syncthreads();
        for (;;) {
            if (tid == 0) flag[1] = 0;
            __syncthreads();
            dbg_sweeps++;
            bool changed = false;
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * kMfThreads;
                if (v >= N) continue;
                int best = d[j];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (!(r[k * NP + v] > 0.0f)) continue;                   // (a positive residual implies the neighbour exists)
                    const int w = (vy[j] + mf_dy(k)) * W + vx[j] + mf_dx(k);
                    const int dw = dist[w] + 1;
                    best = dw < best ? dw : best;
                }
                if (best < d[j]) { d[j] = best; changed = true; }
            }
            __syncthreads();                                                 // every lane has read the old distances
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * kMfThreads;
                if (v < N) dist[v] = d[j];
            }
            if (changed) flag[1] = 1;
            __syncthreads();
            if (!flag[1]) break;
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * kMfThreads;
            if (v >= N) continue;
            const int dd = d[j] > BIG ? BIG : d[j];
            const int hh = (int)hgt[v];
            hgt[v] = (uint16_t)((raise_only && hh > dd) ? hh : dd);
        }
        __syncthreads();
    };

    global_relabel(false);
    int it = 0;
    bool converged = false;
    for (; it < LES_MF_MAX_ITER; it++) {
        // ---- any active node?
        if (tid == 0) flag[0] = 0;
        __syncthreads();
        bool act = false;
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * kMfThreads;
            if (v < N && e[v] > 0.0f && (int)hgt[v] < BIG) act = true;
        }
        if (act) flag[0] = 1;
        __syncthreads();
        if (!flag[0]) { converged = true; break; }
        __syncthreads();
        // ---- 1. sink pushes (own data only)
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * kMfThreads;
            if (v >= N) continue;
            const float ev = e[v], tc = tcap[v];
            if (ev > 0.0f && tc > 0.0f && hgt[v] == 1) {
                const float d = ev < tc ? ev : tc;
                e[v] = ev - d;
                tcap[v] = tc - d;
            }
        }
        // ---- 2. one direction at a time: push, barrier, receive, barrier.  All loads of a pass are issued before the first store
        // (the passes are chains of LDS round trips; the own-node loop is unrolled so that the loads of all nodes overlap)
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            const int dx = mf_dx(k), dy = mf_dy(k);
            const int woff = dy * W + dx;
            float ev[kMfNodesPerThread], rv[kMfNodesPerThread];
            int hv[kMfNodesPerThread], hw[kMfNodesPerThread];
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * kMfThreads;
                const bool in = v < N;
                const int vs = in ? v : 0;
                ev[j] = e[vs]; rv[j] = r[k * NP + vs]; hv[j] = (int)hgt[vs];
                // the neighbour exists whenever the arc has capacity; otherwise read a harmless in-range word
                const int nx = vx[j] + dx, ny = vy[j] + dy;
                const int w = (in && nx >= 0 && nx < W && ny >= 0 && ny < H) ? vs + woff : vs;
                hw[j] = (int)hgt[w];
            }
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * kMfThreads;
                if (v >= N) continue;
                float d = 0.0f;
                if (ev[j] > 0.0f && rv[j] > 0.0f && hv[j] < BIG && hv[j] == hw[j] + 1) {
                    d = ev[j] < rv[j] ? ev[j] : rv[j];
                    e[v] = ev[j] - d;
                    r[k * NP + v] = rv[j] - d;
                }
                sent[v] = d;
            }
            __syncthreads();
            float got[kMfNodesPerThread];
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * kMfThreads;
                const int ux = vx[j] - dx, uy = vy[j] - dy;                  // the node that pushes towards v in direction k
                const bool has = v < N && ux >= 0 && ux < W && uy >= 0 && uy < H;
                got[j] = has ? sent[v - woff] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * kMfThreads;
                if (got[j] > 0.0f) {
                    e[v] += got[j];
                    r[(k ^ 1) * NP + v] += got[j];
                }
            }
            __syncthreads();
        }
        // ---- 3. relabel (from a snapshot of the heights)
        int hn[kMfNodesPerThread];
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * kMfThreads;
            hn[j] = -1;
            if (v >= N) continue;
            const int hv = (int)hgt[v];
            if (!(e[v] > 0.0f) || hv >= BIG) continue;
            int best = tcap[v] > 0.0f ? 1 : BIG;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (!(r[k * NP + v] > 0.0f)) continue;
                const int w = (vy[j] + mf_dy(k)) * W + vx[j] + mf_dx(k);
                const int hw = (int)hgt[w] + 1;
                best = hw < best ? hw : best;
            }
            if (best > BIG) best = BIG;
            if (best > hv) hn[j] = best;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * kMfThreads;
            if (v < N && hn[j] >= 0) hgt[v] = (uint16_t)hn[j];
        }
        __syncthreads();
        if ((it + 1) % kMfGlobalRelabelEvery == 0) global_relabel(true);
    }
    // ---- the cut: nodes that can still reach the sink keep the current label (SINK), the others take the proposal (SOURCE)
    global_relabel(false);
    uint8_t* m = masks + offsets[blockIdx.x];
    double t_out = 0.0;
#pragma unroll
    for (int j = 0; j < kMfNodesPerThread; j++) {
        const int v = tid + j * kMfThreads;
        if (v >= N) continue;
        m[v] = (int)hgt[v] >= BIG ? 255 : 0;
        t_out += (double)tcap[v];
    }
    // flow into the sink = sink capacity used (block reduction through `sent`, 256 doubles fit: N >= 128 is not required, the
    // array pitch is at least 64 floats ... use the r array, which is no longer needed)
    __syncthreads();
    double* red = reinterpret_cast<double*>(r);
    red[tid] = t_in - t_out;
    __syncthreads();
    for (int s = kMfThreads / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
#if defined(LES_MF_DEBUG_ITERS)
        status[blockIdx.x] = converged ? -it : 1;             // measurement builds: the iteration count, negated
#else
        status[blockIdx.x] = converged ? 0 : 1;
#endif
#if defined(LES_MF_DEBUG_ITERS)
        if (flows) flows[blockIdx.x] = (double)dbg_sweeps;
#else
        if (flows) flows[blockIdx.x] = red[0];
#endif
    }
}

}  // namespace les
