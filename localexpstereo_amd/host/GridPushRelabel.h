// GridPushRelabel.h -- s/t minimum cut on the 8-connected w x h pixel grid of an expansion move (LES/FastGCStereo.h:485-559) by
// FIFO push-relabel, first phase only, with exact global relabelling and the gap heuristic.
//
// Why a second host solver next to GridMaxFlow.h (Boykov-Kolmogorov): on the moves where a large part of a coarse cell changes
// its label -- the ones that decide how long a lock-step of the coarsest layer takes -- the search trees of BK are rebuilt over and over
// (a 387 x 387 cell with 22 % of its pixels switching: 163 000 augmentations, 400 000 adoptions, 6.4 M steps of origin walks, 250 ms), while
// push-relabel moves the same flow with 1.5 M pushes and 0.75 M relabels in 95 ms; on easy cells the two are within 15 % of each other
// (tools/cpp/prbench.cpp on dumped lock-steps).  The cut is the same object: SINK side = the nodes that can still reach the
// sink in the residual graph of a maximum (pre)flow, SOURCE otherwise -- the reference solver's what_segment with SOURCE as the default
// (LES/FastGCStereo.h:557) -- which does not depend on the algorithm that found the flow.  Capacities float like Graph<float,float,double>;
// the flow value is accumulated in double.
//
// Sink arcs are the negative part of the excess array, as in the device kernel (csrc/les_maxflow.h): excess that arrives at a node with
// remaining sink capacity is absorbed by the addition itself, and such a node sits at height 1.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "BandPool.h"

namespace les_host {

class GridPushRelabel {
public:
    enum termtype { SOURCE = 0, SINK = 1 };
    // arc directions as in GridMaxFlow: E W S N SW NE SE NW, sister(k) == k ^ 1

    // (re)initialise for a w x h grid whose every interior node is then set with load_node; the storage is reused
    void reset_for_load(int w, int h)
    {
        w_ = w; h_ = h; pw_ = w + 2;
        const size_t n = (size_t)(w + 2) * (h + 2);
        if (ex_.size() < n) { rc_.resize(n * 8); ex_.resize(n); d_.resize(n); queued_.resize(n); }
        big_ = w * h + 2;
        // the padding ring: no capacity, no excess, unreachable
        auto blank = [&](size_t i) { for (int k = 0; k < 8; k++) rc_[i * 8 + k] = 0.f; ex_[i] = 0.f; d_[i] = big_; queued_[i] = 0; };
        for (int x = 0; x < pw_; x++) { blank((size_t)x); blank((size_t)(h + 1) * pw_ + x); }
        for (int y = 1; y <= h; y++) { blank((size_t)y * pw_); blank((size_t)y * pw_ + w + 1); }
        const int o[8] = {+1, -1, +pw_, -pw_, pw_ - 1, -pw_ + 1, pw_ + 1, -pw_ - 1};
        for (int k = 0; k < 8; k++) off_[k] = o[k];
        flow_ = 0;
    }
    int id(int x, int y) const { return (y + 1) * pw_ + (x + 1); }
    // the 5-float node payload {terminal residual, caps E, S, SW, SE} of les_hip_batch_expansion_graph
    void load_node(int x, int y, const float* p5)
    {
        const size_t i = (size_t)id(x, y);
        float* r = &rc_[i * 8];
        r[0] = p5[1]; r[1] = 0; r[2] = p5[2]; r[3] = 0; r[4] = p5[3]; r[5] = 0; r[6] = p5[4]; r[7] = 0;
        ex_[i] = p5[0]; d_[i] = big_; queued_[i] = 0;
    }
    // a node of a residual graph (all 8 residual capacities + terminal residual): continuation of another solver's feasible flow
    void load_residual(int x, int y, const float* rc8, float tr)
    {
        const size_t i = (size_t)id(x, y);
        for (int k = 0; k < 8; k++) rc_[i * 8 + k] = rc8[k];
        ex_[i] = tr; d_[i] = big_; queued_[i] = 0;
    }
    void set_base_flow(double f) { flow_ = f; }

    // bands > 1: parallel first phase.  The rows are cut into `bands` bands; every band runs the algorithm on its own thread as if the
    // arcs into the other bands did not exist (what it routes is a feasible flow of the whole graph; excess that cannot reach a sink
    // inside its band stays where it is), then one run on the whole graph continues from that state.  The result is a maximum
    // preflow of the whole graph, so the cut read-out is the same as for bands == 1 (up to float rounding of the capacities: the
    // band count must therefore be a function of the problem, never of the machine).  Measured on the hard 387 x 387 moves of a
    // two-view run (8 bands, one core per band): 106 -> 14 + 50 ms, 39 -> 7 + 22 ms, 28 -> 5 + 15 ms -- the bands route 95 % of the
    // flow, the whole-graph run needs half as many pushes again for the rest; on 129 x 129 cells there is nothing to gain (4.4 -> 1.5 + 2.8).
    double maxflow(int bands = 1)
    {
        if (bands > h_ / 8) bands = h_ / 8;                       // at least 8 rows per band
        if (bands > 64) bands = 64;
        double absorbed = 0;
        if (bands > 1) {
            if (band_.size() < ex_.size()) band_.resize(ex_.size());
            std::fill(band_.begin(), band_.begin() + (size_t)pw_ * (h_ + 2), (uint8_t)255);
            std::vector<int> row0((size_t)bands + 1);
            for (int b = 0; b <= bands; b++) row0[b] = (int)((long long)h_ * b / bands);
            for (int b = 0; b < bands; b++)
                for (int y = row0[b]; y < row0[b + 1]; y++) std::fill(band_.begin() + (size_t)(y + 1) * pw_ + 1, band_.begin() + (size_t)(y + 1) * pw_ + 1 + w_, (uint8_t)b);
            if ((int)ctx_.size() < bands) ctx_.resize((size_t)bands);
            BandPool::mine().run(bands, [&](int b) { ctx_[b].absorbed = run(ctx_[b], row0[b], row0[b + 1], b, false); });
            for (int b = 0; b < bands; b++) absorbed += ctx_[b].absorbed;          // (in band order: the sum does not depend on the threads' timing)
        }
        absorbed += run(main_, 0, h_, -1, true);
        return flow_ + absorbed;
    }
    termtype what_segment(int x, int y) const { return d_[(size_t)id(x, y)] >= big_ ? SOURCE : SINK; }

private:
    // structure of arrays: the heights (4 bytes per node, three image rows = a few KB) are what every push and every sweep of a
    // relabelling reads; packed 64-byte nodes measured 30 % slower
    int w_ = 0, h_ = 0, pw_ = 2, big_ = 2;
    int off_[8];
    double flow_ = 0;
    std::vector<float> rc_;                        // [node][8] residual capacity towards the 8 neighbours
    std::vector<float> ex_;                        // > 0: excess; < 0: remaining capacity to the sink
    std::vector<int> d_;                           // height; >= big_: cannot reach the sink
    std::vector<uint8_t> queued_;
    std::vector<uint8_t> band_;                    // row band of the node during the parallel first phase (padding: 255)
    // state of one run: the whole graph, or one band of rows whose arcs into other bands are ignored
    struct alignas(128) Ctx {
        std::vector<int> queue, bfs, cnt;
        double absorbed = 0;
    };
    Ctx main_;
    std::vector<Ctx> ctx_;

    bool allowed(int band, int u) const { return band < 0 || band_[(size_t)u] == band; }

    // FIFO push-relabel on the rows [y0, y1) (band >= 0: arcs leaving the band are ignored); final: finish with an exact relabelling
    // (the distances the cut is read from).  Returns the flow absorbed by the sink arcs.
    double run(Ctx& c, int y0, int y1, int band, bool final)
    {
        const int n_int = w_ * (y1 - y0);
        const int lim = n_int + 2;                                   // heights of nodes that can reach a sink inside the rows are < lim
        if ((int)c.cnt.size() < lim + 2) c.cnt.resize((size_t)lim + 2);
        if (c.queue.size() < (size_t)n_int + 1) c.queue.resize((size_t)n_int + 1);
        global_relabel(c, y0, y1, band, lim);
        size_t qh = 0, qt = 0, qn = 0;
        const size_t qcap = (size_t)n_int + 1;
        auto enqueue = [&](int i) { if (!queued_[i]) { queued_[i] = 1; c.queue[qt] = i; qt = qt + 1 == qcap ? 0 : qt + 1; qn++; } };
        for (int y = y0 + 1; y <= y1; y++)
            for (int x = 1; x <= w_; x++) { const int i = y * pw_ + x; if (ex_[i] > 0 && d_[i] < big_) enqueue(i); }
        long long since = 0;
        const long long period = (long long)n_int / 4 + 1;           // relabels between two global relabellings (hard moves of the two-view runs: n/4 4.4 / 6.6 / 28.9 ms, n/2 5.6 / 6.9 / 33.7, n/8 5.0 / 7.7 / 35.1; a period that starts short and doubles: no fewer pushes)
        double absorbed = 0;
        while (qn) {
            const int v = c.queue[qh]; qh = qh + 1 == qcap ? 0 : qh + 1; qn--;
            queued_[v] = 0;
            if (d_[v] >= big_) continue;
            float e = ex_[v];
            float* r = &rc_[(size_t)v * 8];
            while (e > 0) {
                const int dv = d_[v];
                for (int k = 0; k < 8 && e > 0; k++) {
                    if (!(r[k] > 0)) continue;
                    const int u = v + off_[k];
                    if (!allowed(band, u) || dv != d_[u] + 1) continue;      // (band test first: another band's heights are being written by its thread)
                    const float f = e < r[k] ? e : r[k];
                    r[k] -= f; rc_[(size_t)u * 8 + (k ^ 1)] += f; e -= f;
                    const float eu = ex_[u];
                    if (eu < 0) absorbed += (double)(f < -eu ? f : -eu);      // the part the sink arc of u takes
                    ex_[u] = eu + f;
                    if (eu + f > 0 && d_[u] < big_) enqueue(u);
                }
                if (!(e > 0)) break;
                int best = big_;
                for (int k = 0; k < 8; k++)
                    if (r[k] > 0) { const int u = v + off_[k]; if (!allowed(band, u)) continue; const int du = d_[u] + 1; if (du < best) best = du; }
                since++;
                c.cnt[dv]--;
                if (best >= lim) d_[v] = big_;
                else { d_[v] = best; c.cnt[best]++; }
                if (c.cnt[dv] == 0) raise_above(c, y0, y1, dv);        // gap: nobody left at height dv, everything above it is cut off
                if (d_[v] >= big_) break;
                if (since >= period) { ex_[v] = e; since = 0; global_relabel(c, y0, y1, band, lim); if (d_[v] >= big_) break; }
            }
            ex_[v] = e;
        }
        if (final) global_relabel(c, y0, y1, band, lim);             // the cut is read off exact distances
        return absorbed;
    }

    // exact residual distances to the sink: breadth-first search over reversed residual arcs
    void global_relabel(Ctx& c, int y0, int y1, int band, int lim)
    {
        c.bfs.clear();
        for (int y = y0 + 1; y <= y1; y++)
            for (int x = 1; x <= w_; x++) {
                const int i = y * pw_ + x;
                if (ex_[i] < 0) { d_[i] = 1; c.bfs.push_back(i); } else d_[i] = big_;
            }
        for (size_t head = 0; head < c.bfs.size(); head++) {
            const int v = c.bfs[head];
            const int dv = d_[v] + 1;
            for (int k = 0; k < 8; k++) {
                const int u = v + off_[k];                            // the arc u -> v is u's direction k ^ 1 (padding nodes have no capacity)
                if (!allowed(band, u)) continue;                      // (first: a band run must not read another band's state)
                if (d_[u] != big_ || !(rc_[(size_t)u * 8 + (k ^ 1)] > 0)) continue;
                d_[u] = dv;
                c.bfs.push_back(u);
            }
        }
        std::fill(c.cnt.begin(), c.cnt.begin() + lim + 1, 0);
        for (int v : c.bfs) c.cnt[d_[v]]++;
    }
    void raise_above(Ctx& c, int y0, int y1, int level)
    {
        for (int y = y0 + 1; y <= y1; y++)
            for (int x = 1; x <= w_; x++) {
                const int i = y * pw_ + x;
                if (d_[i] > level && d_[i] < big_) { c.cnt[d_[i]]--; d_[i] = big_; }
            }
    }
};

}  // namespace les_host
