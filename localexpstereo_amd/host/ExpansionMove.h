// ExpansionMove.h -- one local alpha-expansion (graph-cut fusion of the current labelling with one proposal) over a
// shared region: FastGCStereo::expansionMoveBK, LES/FastGCStereo.h:411-597 ("next" row N2), on top of GridMaxFlow.h
// (the grid-specialised form of MaxFlow.h: same cut, ~5x faster on these graphs).
//
// Nodes = pixels of `region`.  t-links: source capacity = current unary cost, sink capacity = proposal unary cost
// (:433), plus for region-border pixels the pairwise terms towards their fixed neighbours outside the region
// (:455-475).  n-links for the four forward neighbour directions with B = cost10, C = cost01, D = cost00:
// add_edge(i, j, max(0, B + C - D), 0), add_tweights(i, C, 0), add_tweights(j, D - C, 0) (:485-551).
// A pixel takes the proposal iff it ends on the SOURCE side (:555-559).
#pragma once

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "BandPool.h"
#include "GridMaxFlow.h"
#include "GridPushRelabel.h"
#include "StereoEnergy.h"

namespace les_host {

// updateMask: region.height x region.width (255 = take the proposal).  proposalCost / currentCost: H x W maps.
// Returns the flow (= the energy of the fused labelling restricted to the terms that touch the region).
// cost maps as raw row-major H x W arrays (row stride = image width)
struct CostView {
    const float* p; int stride;
    CostView(const float* p_, int stride_) : p(p_), stride(stride_) {}
    CostView(const CostMap& m) : p(m.data.data()), stride(m.cols) {}
    float at(int y, int x) const { return p[(size_t)y * stride + x]; }
};

// graph construction of expansionMoveBK (LES/FastGCStereo.h:425-551)
inline void buildExpansionGraph(GridMaxFlow& graph, const StereoEnergy& E, const LabelMap& currentLabeling, CostView currentCost,
                                CostView proposalCost, const Plane& label1, const Rect& region, int mode = 0)
{
    const int w = region.width, h = region.height;
    const int W = E.getWidth(), H = E.getHeight();
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const Point ps{region.x + x, region.y + y};
            graph.add_tweights(x, y, currentCost.at(ps.y, ps.x), proposalCost.at(ps.y, ps.x));
            if (x == 0 || x == w - 1 || y == 0 || y == h - 1) {
                for (int k = 0; k < 8; k++) {
                    const Point pt{ps.x + E.neighbors[k].x, ps.y + E.neighbors[k].y};
                    const bool in_region = pt.x >= region.x && pt.x < region.x + w && pt.y >= region.y && pt.y < region.y + h;
                    if (in_region || pt.x < 0 || pt.x >= W || pt.y < 0 || pt.y >= H) continue;
                    // pt keeps its current label
                    const float c00 = E.computeSmoothnessTerm(currentLabeling.at(ps.y, ps.x), currentLabeling.at(pt.y, pt.x), ps, k, mode);
                    const float c10 = E.computeSmoothnessTerm(label1, currentLabeling.at(pt.y, pt.x), ps, k, mode);
                    graph.add_tweights(x, y, c00, c10);
                }
            }
        }
    auto link = [&](int k, int dir, int x0, int x1, int y1, int dx, int dy) {
        for (int y = 0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                float B, C, D;                                             // cost10, cost01, cost00 (LES/StereoEnergy.h:398-453)
                E.smoothnessTermsExpansionAt(currentLabeling, label1, region.x + x, region.y + y, k, D, C, B, mode);
                graph.add_edge(x, y, dir, std::max(0.f, B + C - D), 0);     // B+C-D can be slightly negative numerically
                graph.add_tweights(x, y, C, 0);
                graph.add_tweights(x + dx, y + dy, D - C, 0);
            }
    };
    link(StereoEnergy::NB_GE, GridMaxFlow::E, 0, w - 1, h, +1, 0);           // ee <-> ge
    link(StereoEnergy::NB_EG, GridMaxFlow::S, 0, w, h - 1, 0, +1);           // ee <-> eg
    link(StereoEnergy::NB_LG, GridMaxFlow::SW, 1, w, h - 1, -1, +1);         // ee <-> lg
    link(StereoEnergy::NB_GG, GridMaxFlow::SE, 0, w - 1, h - 1, +1, +1);     // ee <-> gg
}

// max-flow + segment readout (LES/FastGCStereo.h:553-559)
inline double solveExpansionGraph(GridMaxFlow& graph, const Rect& region, std::vector<uint8_t>& updateMask, int bands = 1)
{
    const int w = region.width, h = region.height;
    const double flow = graph.maxflow(bands);
    updateMask.resize((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) updateMask[(size_t)y * w + x] = graph.what_segment(x, y) == GridMaxFlow::SOURCE ? 255 : 0;
    return flow;
}

inline double expansionMove(const StereoEnergy& E, const LabelMap& currentLabeling, CostView currentCost,
                            CostView proposalCost, const Plane& label1, const Rect& region, std::vector<uint8_t>& updateMask,
                            int mode = 0)
{
    GridMaxFlow graph(region.width, region.height);
    buildExpansionGraph(graph, E, currentLabeling, currentCost, proposalCost, label1, region, mode);
    return solveExpansionGraph(graph, region, updateMask);
}

// The same move on a graph whose capacities were computed on the device (5 floats per node, row-major over the region;
// include/localexp_hip.h: les_hip_batch_expansion_graph).
// mask: region.width * region.height bytes (255 = take the proposal).  The solver object is reused per thread.
// bands > 1: parallel first phase of the max-flow on that many row bands (large regions when few cells share a lock-step)
// Solver policy for device-built graphs.  Boykov-Kolmogorov (GridMaxFlow.h, band-parallel first phase for large cells) is the faster
// solver on the common, easy moves; on the moves where a large part of a coarse cell switches it degrades (tools/cpp/prbench.cpp: 250 ms
// against 90 ms for push-relabel on a 387 x 387 cell).  So cells of at least pushRelabelMinNodes() nodes get a work budget of
// kBkOpsPerNode search operations per node; when it runs out, the flow routed so far is kept and FIFO push-relabel (GridPushRelabel.h)
// finishes on the residual graph.  Both return the canonical cut; budget and threshold are functions of the region size only, so every
// rank and every host cuts a given cell the same way.  LES_GC_PUSH_RELABEL_MIN_NODES overrides the threshold (0: never switch),
// LES_GC_BK_OPS_PER_NODE the budget (A/B measurements).
// The budget: since the graphs are pre-pushed while they are loaded (GridMaxFlow::load_rows_prepushed: the search starts from the 0.2 .. 1 %
// of the nodes that still have source excess) an ordinary move needs 0.05 .. 1.6 operations per node, the moves on which the trees are
// rebuilt over and over 5 .. 70 (sampled lock-steps of a two-view run, tools/cut_replay.py / DESIGN 6.3); 3 separates the two and costs a
// hard 129 x 129 cell about a millisecond before push-relabel takes over.  (It was 12 when every node started in the queue.)
inline long long pushRelabelMinNodes()
{
    static const long long v = [] {
        const char* e = getenv("LES_GC_PUSH_RELABEL_MIN_NODES");
        if (!e) return 10000ll;
        const long long x = atoll(e);
        return x <= 0 ? (1ll << 62) : x;
    }();
    return v;
}
inline double bkOpsPerNode()
{
    static const double v = [] { const char* e = getenv("LES_GC_BK_OPS_PER_NODE"); return e ? atof(e) : 3.0; }();
    return v;
}

// tooling (LES_GC_PROFILE=1): where the time of the prebuilt cuts goes, summed over all threads, printed when the process ends
struct CutProfile {
    std::atomic<long long> ns[5], cells[3], nodes[3];          // phases: load + pre-push, search, hand-over, push-relabel, read-out; classes: small, large, large handed over
    static CutProfile& get() { static CutProfile p; return p; }
    static bool on() { static const bool v = [] { const char* e = getenv("LES_GC_PROFILE"); return e && atoi(e) != 0; }(); return v; }
    CutProfile()
    {
        for (auto& x : ns) x = 0;
        for (auto& x : cells) x = 0;
        for (auto& x : nodes) x = 0;
        if (on()) atexit([] {
            CutProfile& p = get();
            fprintf(stderr, "host cuts (all threads): load+prepush %.3f s, search %.3f s, hand-over %.3f s, push-relabel %.3f s, read-out %.3f s | cells: %lld searched only (%lld nodes), %lld large searched only (%lld), %lld handed to push-relabel (%lld)\n",
                    p.ns[0] * 1e-9, p.ns[1] * 1e-9, p.ns[2] * 1e-9, p.ns[3] * 1e-9, p.ns[4] * 1e-9, (long long)p.cells[0], (long long)p.nodes[0], (long long)p.cells[1], (long long)p.nodes[1],
                    (long long)p.cells[2], (long long)p.nodes[2]);
        });
    }
};

inline double expansionMovePrebuilt(const float* payload, double base_flow, const Rect& region, uint8_t* mask, int bands = 1)
{
    const bool prof = CutProfile::on();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](int phase, std::chrono::steady_clock::time_point& t) {
        if (!prof) return;
        const auto t1 = now();
        CutProfile::get().ns[phase] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t).count();
        t = t1;
    };
    auto tp = prof ? now() : std::chrono::steady_clock::time_point();
    static thread_local GridMaxFlow graph_tls;
    GridMaxFlow& graph = graph_tls;                 // (a reference: the helper threads below must use THIS thread's solver, not their own)
    const int w = region.width, h = region.height;
    graph.reset_for_load(w, h);
    // large regions: the node load and the segment read-out are split over the same number of threads as the first max-flow phase
    auto rows_parallel = [&](auto&& body) {
        if (bands <= 1) { body(0, 0, h); return; }
        BandPool::mine().run(bands, [&](int b) { body(b, (int)((long long)h * b / bands), (int)((long long)h * (b + 1) / bands)); });
    };
    static const bool prepush = [] { const char* e = getenv("LES_GC_PREPUSH"); return !e || atoi(e) != 0; }();
    std::vector<double> routed((size_t)std::max(1, bands), 0.0);
    rows_parallel([&](int b, int y0, int y1) {
        if (prepush) { routed[(size_t)b] = graph.load_rows_prepushed(payload, y0, y1); return; }
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < w; x++) graph.load_node(x, y, payload + 5 * ((size_t)y * w + x));
    });
    lap(0, tp);
    graph.set_base_flow(base_flow);
    for (double f : routed) graph.add_base_flow(f);                         // (in band order: the sum does not depend on the threads' timing)
    const bool budgeted = (long long)w * h >= pushRelabelMinNodes();
    // (the band searches of the coarsest layer's cells run in parallel while the push-relabel that would take over does not: they get four
    // times the allowance -- two-view run at the Adirondack shape: 6.2 s with 12 per band node, 6.5 with 3 or 40, 6.8 with 200)
    static const double band_ops = [] { const char* e = getenv("LES_GC_BK_BAND_OPS_PER_NODE"); return e ? atof(e) : 4.0 * bkOpsPerNode(); }();
    const double flow = graph.maxflow(bands, budgeted ? bkOpsPerNode() : 0.0, band_ops);
    lap(1, tp);
    if (prof) { const int cls = graph.exhausted() ? 2 : budgeted ? 1 : 0; CutProfile::get().cells[cls]++; CutProfile::get().nodes[cls] += (long long)w * h; }
    if (graph.exhausted()) {
        // a hard move: push-relabel continues from the feasible flow found so far
        static thread_local GridPushRelabel pr_tls;
        GridPushRelabel& pr = pr_tls;
        pr.reset_for_load(w, h);
        rows_parallel([&](int, int y0, int y1) {
            float rc8[8], tr;
            for (int y = y0; y < y1; y++)
                for (int x = 0; x < w; x++) { graph.residual(x, y, rc8, &tr); pr.load_residual(x, y, rc8, tr); }
        });
        pr.set_base_flow(flow);
        lap(2, tp);
        const double total = pr.maxflow(bands);                 // (the coarsest layer's cells: the same row bands as the search)
        lap(3, tp);
        rows_parallel([&](int, int y0, int y1) {
            for (int y = y0; y < y1; y++)
                for (int x = 0; x < w; x++) mask[(size_t)y * w + x] = pr.what_segment(x, y) == GridPushRelabel::SOURCE ? 255 : 0;
        });
        lap(4, tp);
        return total;
    }
    rows_parallel([&](int, int y0, int y1) {
        for (int y = y0; y < y1; y++) graph.segment_row(y, mask + (size_t)y * w);
    });
    lap(4, tp);
    return flow;
}
// Row bands of the parallel first max-flow phase: only for large regions (the coarsest layer: 4-6 cells of ~400 x 400 nodes per
// lock-step, which leave most of the host idle), up to 8 bands.  Cells x bands may exceed the CPUs the process is granted (a cgroup
// quota of 16 on the MI355X boxes): the helpers of such a lock-step then sleep instead of spinning between phases (tuneBandSpin).
// The band count is a function of the region size ONLY (not of the machine or of how many cells a rank happens to cut in the
// lock-step): the search order inside a cut -- and with float capacities possibly a tie between equal-energy cuts -- must not
// depend on the host or on the world size.
inline int bandsFor(const Rect& region, int /*cells_in_lockstep*/ = 0)
{
    const long long nodes = (long long)region.width * region.height;
    if (nodes < 40000) return 1;
    return (int)std::max<long long>(2, std::min<long long>(8, nodes / 20000));
}

// idle-helper policy of a lock-step of n cells (BandPool::spinLimit): spin only while every thread that could run has a CPU
template <class RegionAt>
inline void tuneBandSpin(int n, int team, RegionAt&& region_at)
{
    long long want = 0;
    int big = 0;
    for (int i = 0; i < n; i++) {
        const int b = bandsFor(region_at(i), n);
        if (b > 1) { want += b; big++; }
    }
    // at most `team` cells are cut at once
    if (big > team && big > 0) want = want * team / big;
    BandPool::spinLimit().store(want > cpuBudget() ? 64 : 20000, std::memory_order_relaxed);
}

inline double expansionMovePrebuilt(const float* payload, double base_flow, const Rect& region, std::vector<uint8_t>& updateMask)
{
    updateMask.resize((size_t)region.width * region.height);
    return expansionMovePrebuilt(payload, base_flow, region, updateMask.data());
}

// The reference's (disabled) self-check of the graph construction, LES/FastGCStereo.h:561-594: the flow equals the
// unary cost of the fused labelling over the region plus every forward pairwise term with an endpoint in the region.
inline double fusedEnergy(const StereoEnergy& E, const LabelMap& currentLabeling, CostView currentCost, CostView proposalCost,
                          const Plane& label1, const Rect& region, const std::vector<uint8_t>& updateMask, int mode = 0)
{
    const int W = E.getWidth(), H = E.getHeight();
    auto in_region = [&](int x, int y) { return x >= region.x && x < region.x + region.width && y >= region.y && y < region.y + region.height; };
    auto label_at = [&](int x, int y) -> Plane {
        if (in_region(x, y) && updateMask[(size_t)(y - region.y) * region.width + (x - region.x)]) return label1;
        return currentLabeling.at(y, x);
    };
    double e = 0;
    for (int y = region.y; y < region.y + region.height; y++)
        for (int x = region.x; x < region.x + region.width; x++)
            e += updateMask[(size_t)(y - region.y) * region.width + (x - region.x)] ? proposalCost.at(y, x) : currentCost.at(y, x);
    const Rect m = Rect(region.x - 1, region.y - 1, region.width + 2, region.height + 2) & Rect(0, 0, W, H);
    for (int y = m.y; y < m.y + m.height; y++)
        for (int x = m.x; x < m.x + m.width; x++)
            for (int k : {(int)StereoEnergy::NB_GE, (int)StereoEnergy::NB_EG, (int)StereoEnergy::NB_LG, (int)StereoEnergy::NB_GG}) {
                const int xn = x + E.neighbors[k].x, yn = y + E.neighbors[k].y;
                if (xn < 0 || xn >= W || yn < 0 || yn >= H) continue;
                if (!in_region(x, y) && !in_region(xn, yn)) continue;
                e += E.computeSmoothnessTerm(label_at(x, y), label_at(xn, yn), Point{x, y}, k, mode);
            }
    return e;
}

}  // namespace les_host
