// ResidualCut.h -- finishes the minimum cut of an expansion move from a RESIDUAL graph: the state in which the tiled device max-flow
// (csrc/les_maxflow_tiled.h) hands a straggler cell over to the host cores (round 6).
//
// A lock-step of the coarse layers lasts as long as its slowest cell, and the region-parallel push-relabel of the device is worst at the
// tail of a hard cell: a few hundred small excesses behind arcs of capacity 1e-4 ... 1e-2 that need hundreds of launches while 250 of the
// 256 CUs idle (per-cell launch counts of dumped lock-steps, tools/tiled_cell_stats.py: 42 of 48 cells done after <= 26 launches, the other
// six after 77 ... 236).  A search from the few remaining excess nodes is what the host solvers are good at, and the host cores idle during
// device cuts.  The residual graph of any feasible preflow is a valid max-flow problem with the same minimum cuts: whatever is routed
// first, every s-t cut loses the same amount, so SINK = "can still reach the sink in the residual graph of a maximum (pre)flow" -- the
// reference solver's what_segment with SOURCE as the default (LES/FastGCStereo.h:553-559) -- is the same set (up to float rounding of the
// residual capacities, as between any two of the solvers here).
//
// Input per node (row-major w x h): rc8 = residual capacities towards E W S N SW NE SE NW, ex > 0: excess (source residual), < 0: remaining
// capacity to the sink.  Output: mask 255 = SOURCE segment (the proposal is taken), 0 = SINK; returns the flow routed here.
#pragma once

#include <cstdint>
#include <cstdlib>

#include "BandPool.h"
#include "GridMaxFlow.h"
#include "GridPushRelabel.h"

namespace les_host {

// solver: 0 = search trees from the excess nodes (GridMaxFlow, lazy mode) with the push-relabel continuation when its work budget runs
// out (the same allowance per node as expansionMovePrebuilt); 1 = push-relabel only
inline double residualBkOpsPerNode()
{
    const char* e = getenv("LES_GC_RESIDUAL_BK_OPS_PER_NODE");      // (read per call: A/B measurements; 0 = no budget)
    return e ? atof(e) : 12.0;
}

inline double finishResidualCut(const float* rc8, const float* ex, int w, int h, uint8_t* mask, int bands = 1, int solver = 0)
{
    const double bk_ops_per_node = residualBkOpsPerNode();
    auto rows_parallel = [&](auto&& body) {
        if (bands <= 1) { body(0, 0, h); return; }
        BandPool::mine().run(bands, [&](int b) { body(b, (int)((long long)h * b / bands), (int)((long long)h * (b + 1) / bands)); });
    };
    static thread_local GridPushRelabel pr_tls;
    GridPushRelabel& pr = pr_tls;                   // (references: the helper threads below must use THIS thread's solvers, not their own)
    if (solver == 0) {
        static thread_local GridMaxFlow graph_tls;
        GridMaxFlow& graph = graph_tls;
        graph.reset_for_load(w, h);
        rows_parallel([&](int, int y0, int y1) {
            for (int y = y0; y < y1; y++)
                for (int x = 0; x < w; x++) { const size_t i = (size_t)y * w + x; graph.load_residual(x, y, rc8 + 8 * i, ex[i]); }
        });
        const double flow = graph.maxflow(bands, bk_ops_per_node, 4.0 * bk_ops_per_node);
        if (!graph.exhausted()) {
            rows_parallel([&](int, int y0, int y1) {
                for (int y = y0; y < y1; y++) graph.segment_row(y, mask + (size_t)y * w);
            });
            return flow;
        }
        pr.reset_for_load(w, h);
        rows_parallel([&](int, int y0, int y1) {
            float r8[8], tr;
            for (int y = y0; y < y1; y++)
                for (int x = 0; x < w; x++) { graph.residual(x, y, r8, &tr); pr.load_residual(x, y, r8, tr); }
        });
        pr.set_base_flow(flow);
    } else {
        pr.reset_for_load(w, h);
        rows_parallel([&](int, int y0, int y1) {
            for (int y = y0; y < y1; y++)
                for (int x = 0; x < w; x++) { const size_t i = (size_t)y * w + x; pr.load_residual(x, y, rc8 + 8 * i, ex[i]); }
        });
        pr.set_base_flow(0.0);
    }
    const double total = pr.maxflow(bands);
    rows_parallel([&](int, int y0, int y1) {
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < w; x++) mask[(size_t)y * w + x] = pr.what_segment(x, y) == GridPushRelabel::SOURCE ? 255 : 0;
    });
    return total;
}

// Row bands of a residual cut: ONE.  The parallel first phase of the host solvers pays on whole problems (ExpansionMove.h: bandsFor); on what the device
// hands over -- a few hundred excess nodes, most of the flow routed -- the band phase finds nothing the whole-graph run does not find as fast, and its
// threads compete with the other cells' (and the other view's) finishers: measured on 387 x 387 residuals, 7 bands / 1 band: 35 / 31, 26 / 15, 19 / 7.4,
// 29 / 18 ms (tools/residual_probe.py).  LES_GC_RESIDUAL_BAND_NODES=n (tests, A/B): n nodes per band, at most 8 bands.
inline int residualBands(int w, int h)
{
    const char* e = getenv("LES_GC_RESIDUAL_BAND_NODES");
    if (!e || atoll(e) <= 0) return 1;
    const long long nodes = (long long)w * h, per_band = atoll(e);
    if (nodes < 2 * per_band) return 1;
    return (int)std::max<long long>(2, std::min<long long>(8, nodes / per_band));
}

}  // namespace les_host
