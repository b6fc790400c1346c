// les_maxflow.h -- the minimum cut of an expansion move on the device, one workgroup per cell, the whole graph in LDS.
//
// Replaces, for cells that fit (layer-0 cells: 42 x 42 .. 45 x 45 nodes), the host side of `expansionMoveBK` after the graph
// construction -- `graph.maxflow()` and `graph.what_segment()` (LES/FastGCStereo.h:553-559) -- on the 5-float node payload of
// les_expansion_graph_kernel (les_pairwise.h): terminal residual + capacities of the arcs to the E, S, SW, SE neighbours.
//
// Algorithm: synchronous push-relabel (Goldberg-Tarjan, first phase only) in the data-parallel form that needs no atomics:
//   per iteration   1. four passes of two grid directions each: every active node pushes along its arcs of those directions when
//                      they are admissible (height(v) == height(w) + 1); a node receives from at most one sender per direction,
//                      so the receiving side is a second half behind a barrier (`sentA`, `sentB`) and no two lanes ever write
//                      the same word.  Sink arcs are the negative part of the excess array: arriving excess is absorbed by the
//                      addition itself (a node with sink capacity sits at height 1, the push to the sink is always admissible);
//                   2. active nodes without an admissible arc are relabelled from a snapshot of the heights;
//   every G iterations and at the end: global relabelling = residual distance to the sink by Jacobi sweeps until nothing changes.
// Termination: no node with excess can reach the sink.  The cut is then read off the final distances: SINK side = the nodes that
// can still reach the sink in the residual graph, which is the segment rule of the reference's solver (`what_segment` with
// SOURCE as the default) and does not depend on which maximum preflow was found.  Capacities float as in Graph<float,float,double>.
// tools/pushrelabel_probe.py is the numpy model of exactly this scheme (median 16 iterations on 42 x 42 crops of real graphs,
// cuts identical to the host solver).  A cell that does not converge within the iteration limit reports status 1 and is cut on the host.
#pragma once

#include <cstdint>

#include "les_simt.h"

namespace les {

// 512 threads = 8 waves per workgroup; a 42 x 42 cell needs 81 208 B of LDS, so two workgroups share a CU (4 waves per SIMD): the
// kernel is a chain of dependent LDS accesses between barriers and lives on latency hiding (measured: 256 threads and one
// workgroup per CU took 8.6 ms for a lock-step of 450 cells that the host team cuts in 6.1 ms)
constexpr int kMfThreads = 512;
// nodes per thread: a template parameter (4 for cells of up to 2048 nodes, 5 up to the limit; neither spills at 128 VGPRs)
constexpr int kMfMaxNodes = 2304;
static_assert(kMfThreads * 5 >= kMfMaxNodes, "every node needs an owner");
#ifndef LES_MF_G
#define LES_MF_G 8
#endif
constexpr int kMfGlobalRelabelEvery = LES_MF_G;
constexpr int kMfMaxIter = 6000;                          // default iteration limit (LES_HIP_MAXFLOW_MAX_ITER overrides it: tests of the host fall-back)

// LDS bytes for a launch whose largest cell has `nodes` nodes: r[8] + excess + two exchange words (float each) + height (uint16), + flags
__host__ __device__ inline size_t mf_lds_bytes(int nodes)
{
    const size_t n = (size_t)((nodes + 7) / 8) * 8;
    const size_t bytes = n * (8 * 4 + 4 + 4 + 4 + 2) + 64;
    return bytes < 8256 ? 8256 : bytes;                    // (the final reduction borrows one double per thread: up to 1024)
}

struct GraphCellMf { int x, y, w, h; };                   // same layout as GraphCell (les_pairwise.h)

// arc directions E W S N SW NE SE NW (sister(k) == k ^ 1) as offsets; packed in nibbles so that a run-time direction index needs
// no indexed register array
__host__ __device__ inline int mf_dx(int k) { return (int)((0x02201102u >> (4 * k)) & 0xfu) - 1; }      // +1 -1  0  0 -1 +1 +1 -1
__host__ __device__ inline int mf_dy(int k) { return (int)((0x02020211u >> (4 * k)) & 0xfu) - 1; }      //  0  0 +1 -1 +1 -1 +1 -1

// grid = cells; block = THREADS; dynamic LDS = mf_lds_bytes(max nodes of the launch); NPT * THREADS >= nodes of every cell.
// Two shapes (round 5): <2, 1024> for cells of up to 2048 nodes -- sixteen waves with two node slots each: what an iteration costs is the instruction
// stream of its slowest wave (one wave issues an instruction per ~4.2 cycles whatever the others do), and two slots are half the stream of four; 64 VGPRs,
// two workgroups per CU as before -- and <5, 512> (128 VGPRs) for the cells between 2049 and 2304 nodes, where three slots of 1024 threads spill.
template <int kMfNodesPerThread, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 128)
les_maxflow_kernel(const GraphCellMf* __restrict__ cells, const long long* __restrict__ offsets, const float* __restrict__ payload,
                   int nmax_padded, int max_iter, uint8_t* __restrict__ masks, int* __restrict__ status, double* __restrict__ flows,
                   int* __restrict__ unsolved_total)         // optional: += 1 per cell that hits the iteration limit (callers that check once per several launches)
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
    float* ex = r + 8 * NP;                                // > 0: excess of the node; < 0: its remaining capacity to the sink
    float* sentA = ex + NP;                                // what a node pushed along the first / second direction of the pass
    float* sentB = sentA + NP;
    uint16_t* hgt = reinterpret_cast<uint16_t*>(sentB + NP);
    int* flag = reinterpret_cast<int*>(hgt + NP);          // [0] any active, [1..3] relabel sweep changed something (rotating)

    const GraphCellMf c = cells[blockIdx.x];
    const int W = c.w, H = c.h, N = W * H;
    const int tid = (int)threadIdx.x;
    const float* p5 = payload + 5 * offsets[blockIdx.x];
    if (N <= 0) { if (tid == 0) { status[blockIdx.x] = 0; if (flows) flows[blockIdx.x] = 0.0; } return; }
    const int BIG = N + 2;                                 // "cannot reach the sink" (fits uint16: N <= 2304)

    // ---- own nodes: v = tid + j * THREADS
    int vx[kMfNodesPerThread], vy[kMfNodesPerThread];
#pragma unroll
    for (int j = 0; j < kMfNodesPerThread; j++) {
        const int v = tid + j * THREADS;
        vy[j] = v < N ? v / W : -1;
        vx[j] = v < N ? v - vy[j] * W : 0;
    }
    double t_in = 0.0;                                     // sink capacity of the own nodes at load time (for the flow value)
#pragma unroll
    for (int j = 0; j < kMfNodesPerThread; j++) {
        const int v = tid + j * THREADS;
        if (v >= N) continue;
        const float tr = p5[5 * v];
        const int x = vx[j], y = vy[j];
        // arcs that would leave the region carry no capacity
        r[0 * NP + v] = (x + 1 < W) ? p5[5 * v + 1] : 0.0f;
        r[2 * NP + v] = (y + 1 < H) ? p5[5 * v + 2] : 0.0f;
        r[4 * NP + v] = (y + 1 < H && x > 0) ? p5[5 * v + 3] : 0.0f;
        r[6 * NP + v] = (y + 1 < H && x + 1 < W) ? p5[5 * v + 4] : 0.0f;
        r[1 * NP + v] = 0.0f; r[3 * NP + v] = 0.0f; r[5 * NP + v] = 0.0f; r[7 * NP + v] = 0.0f;
        // source arcs are saturated at the start (excess = their capacity); a sink arc is the negative part: excess that arrives
        // at such a node is absorbed by the addition itself (the node sits at height 1, the push to the sink is always admissible)
        ex[v] = tr;
        if (tr < 0.0f) t_in += (double)(-tr);
        hgt[v] = 0;
    }
    __syncthreads();

    // residual distance to the sink (Jacobi sweeps); on return hgt = max(hgt, distance) when raise_only, else = distance
    int dbg_sweeps = 0;                                    // (measurement builds report it through `flows`)
    auto global_relabel = [&](bool raise_only) {
        int d[kMfNodesPerThread];
        // the distances live in `sentA` (as integers) during the sweeps: the heights stay readable for raise_only
        int* dist = reinterpret_cast<int*>(sentA);
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * THREADS;
            d[j] = (v < N && ex[v] < 0.0f) ? 1 : BIG;
            if (v < N) dist[v] = d[j];
        }
        if (tid < 3) flag[1 + tid] = 0;
        __syncthreads();
        // Chaotic relaxation: a node takes min(own, neighbour + 1) over the LIVE distances (another lane may have lowered a neighbour
        // in this very sweep: the values only fall and the fixed point -- the breadth-first distances -- is the same whatever the order),
        // so an update can travel several hops per sweep, and a sweep costs one barrier: "did anything change" rotates through three
        // flags (the one for sweep s + 1 is cleared during sweep s, two barriers after its last reader).
        for (int s = 0;; s++) {
            const int cur = 1 + s % 3, nxt = 1 + (s + 1) % 3;
            if (tid == 0) flag[nxt] = 0;
            dbg_sweeps++;
            bool changed = false;
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * THREADS;
                if (v >= N) continue;
                float rk[8];
#pragma unroll
                for (int k = 0; k < 8; k++) rk[k] = r[k * NP + v];
                int best = d[j];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // (a positive residual implies the neighbour exists; otherwise read the node's own word)
                    const int w = rk[k] > 0.0f ? (vy[j] + mf_dy(k)) * W + vx[j] + mf_dx(k) : v;
                    const int dw = dist[w] + 1;
                    best = (rk[k] > 0.0f && dw < best) ? dw : best;
                }
                if (best < d[j]) { d[j] = best; dist[v] = best; changed = true; }
            }
            if (changed) flag[cur] = 1;
            __syncthreads();
            if (!flag[cur]) break;
        }
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * THREADS;
            if (v >= N) continue;
            const int dd = d[j] > BIG ? BIG : d[j];
            const int hh = (int)hgt[v];
            hgt[v] = (uint16_t)((raise_only && hh > dd) ? hh : dd);
        }
        __syncthreads();
    };

    global_relabel(false);
    // "is any node active?" (excess and a height below BIG) is evaluated where the heights are written -- here and at the end of
    // every iteration -- so the loop head only reads the flag.  A periodic global relabelling in between can only deactivate
    // nodes: the flag may then be stale by one (harmless) iteration.
    auto note_active = [&]() {
        bool act = false;
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * THREADS;
            if (v < N && ex[v] > 0.0f && (int)hgt[v] < BIG) act = true;
        }
        if (act) flag[0] = 1;
    };
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    note_active();
    __syncthreads();
    int it = 0;
    bool converged = false;
    for (; it < max_iter; it++) {
        if (!flag[0]) { converged = true; break; }
        // ---- pushes, two directions per pass: a pass is "push, barrier, receive, barrier"; every word has one writer per half
        // (a node has at most one sender per direction, and of the two arcs between a pair of nodes only one can be admissible).
        // All loads of a half are issued before its first store: the halves are chains of LDS round trips.
#pragma unroll 1
        for (int kp = 0; kp < 8; kp += 2) {
            const int dxa = mf_dx(kp), dya = mf_dy(kp), dxb = mf_dx(kp + 1), dyb = mf_dy(kp + 1);
            const int offa = dya * W + dxa, offb = dyb * W + dxb;
            // (a lane only writes words of its own nodes in this half and reads no word another lane writes in it, so the own nodes are
            // handled in batches of JB: with five nodes per thread all six operands of all of them do not fit 128 VGPRs)
            constexpr int JB = kMfNodesPerThread > 4 ? 3 : kMfNodesPerThread;
#pragma unroll
            for (int j0 = 0; j0 < kMfNodesPerThread; j0 += JB) {
                float ev[JB], ra[JB], rb[JB];
                int hv[JB], ha[JB], hb[JB];
#pragma unroll
                for (int jj = 0; jj < JB; jj++) {
                    const int j = j0 + jj;
                    if (j >= kMfNodesPerThread) continue;
                    const int v = tid + j * THREADS;
                    const bool in = v < N;
                    const int vs = in ? v : 0;
                    ev[jj] = ex[vs]; ra[jj] = r[kp * NP + vs]; rb[jj] = r[(kp + 1) * NP + vs]; hv[jj] = (int)hgt[vs];
                    // the neighbour exists whenever the arc has capacity; otherwise read a harmless in-range word
                    const int ax = vx[j] + dxa, ay = vy[j] + dya, bx = vx[j] + dxb, by = vy[j] + dyb;
                    ha[jj] = (int)hgt[(in && ax >= 0 && ax < W && ay >= 0 && ay < H) ? vs + offa : vs];
                    hb[jj] = (int)hgt[(in && bx >= 0 && bx < W && by >= 0 && by < H) ? vs + offb : vs];
                }
#pragma unroll
                for (int jj = 0; jj < JB; jj++) {
                    const int j = j0 + jj;
                    if (j >= kMfNodesPerThread) continue;
                    const int v = tid + j * THREADS;
                    if (v >= N) continue;
                    float e0 = ev[jj], da = 0.0f, db = 0.0f;
                    if (e0 > 0.0f && hv[jj] < BIG) {
                        if (ra[jj] > 0.0f && hv[jj] == ha[jj] + 1) { da = e0 < ra[jj] ? e0 : ra[jj]; e0 -= da; r[kp * NP + v] = ra[jj] - da; }
                        if (e0 > 0.0f && rb[jj] > 0.0f && hv[jj] == hb[jj] + 1) { db = e0 < rb[jj] ? e0 : rb[jj]; e0 -= db; r[(kp + 1) * NP + v] = rb[jj] - db; }
                        if (da > 0.0f || db > 0.0f) ex[v] = e0;
                    }
                    sentA[v] = da;
                    sentB[v] = db;
                }
            }
            __syncthreads();
            if (kp == 0 && tid == 0) flag[0] = 0;                            // (every lane has read the flag at the loop head)
            float ga[kMfNodesPerThread], gb[kMfNodesPerThread];
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * THREADS;
                const int uax = vx[j] - dxa, uay = vy[j] - dya, ubx = vx[j] - dxb, uby = vy[j] - dyb;   // the nodes that push towards v
                ga[j] = (v < N && uax >= 0 && uax < W && uay >= 0 && uay < H) ? sentA[v - offa] : 0.0f;
                gb[j] = (v < N && ubx >= 0 && ubx < W && uby >= 0 && uby < H) ? sentB[v - offb] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < kMfNodesPerThread; j++) {
                const int v = tid + j * THREADS;
                if (ga[j] > 0.0f) r[(kp ^ 1) * NP + v] += ga[j];
                if (gb[j] > 0.0f) r[((kp + 1) ^ 1) * NP + v] += gb[j];
                if (ga[j] > 0.0f || gb[j] > 0.0f) ex[v] += ga[j] + gb[j];     // (a sink arc absorbs what it can right here)
            }
            __syncthreads();
        }
        // ---- relabel (from a snapshot of the heights)
        int hn[kMfNodesPerThread];
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * THREADS;
            hn[j] = -1;
            if (v >= N) continue;
            const int hv = (int)hgt[v];
            if (!(ex[v] > 0.0f) || hv >= BIG) continue;
            int best = BIG;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (!(r[k * NP + v] > 0.0f)) continue;
                const int w = (vy[j] + mf_dy(k)) * W + vx[j] + mf_dx(k);
                const int hw = (int)hgt[w] + 1;
                best = hw < best ? hw : best;
            }
            if (best > hv) hn[j] = best;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kMfNodesPerThread; j++) {
            const int v = tid + j * THREADS;
            if (v < N && hn[j] >= 0) hgt[v] = (uint16_t)hn[j];
        }
        note_active();                                                       // (own nodes only: their excess and new heights)
        __syncthreads();
        if ((it + 1) % kMfGlobalRelabelEvery == 0) global_relabel(true);
    }
    // ---- the cut: nodes that can still reach the sink keep the current label (SINK), the others take the proposal (SOURCE)
    global_relabel(false);
    uint8_t* m = masks + offsets[blockIdx.x];
    double t_out = 0.0;
#pragma unroll
    for (int j = 0; j < kMfNodesPerThread; j++) {
        const int v = tid + j * THREADS;
        if (v >= N) continue;
        m[v] = (int)hgt[v] >= BIG ? 255 : 0;
        const float xv = ex[v];
        if (xv < 0.0f) t_out += (double)(-xv);
    }
    // flow into the sink = sink capacity used; block reduction in the (no longer needed) residual array
    __syncthreads();
    double* red = reinterpret_cast<double*>(r);
    red[tid] = t_in - t_out;
    __syncthreads();
    for (int s2 = THREADS / 2; s2 > 0; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
    }
    if (tid == 0) {
#if defined(LES_MF_DEBUG_ITERS)
        status[blockIdx.x] = converged ? -it : 1;             // measurement builds: the iteration count, negated
        if (flows) flows[blockIdx.x] = (double)dbg_sweeps;
#else
        status[blockIdx.x] = converged ? 0 : 1;
        if (flows) flows[blockIdx.x] = red[0];
        if (!converged && unsolved_total) atomicAdd(unsolved_total, 1);
#endif
    }
}

}  // namespace les
