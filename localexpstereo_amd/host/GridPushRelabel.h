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

    double maxflow()
    {
        const int n_int = w_ * h_;
        if ((int)cnt_.size() < big_ + 2) cnt_.resize((size_t)big_ + 2);
        queue_.resize((size_t)n_int + 1);
        global_relabel();
        size_t qh = 0, qt = 0, qn = 0;
        const size_t qcap = queue_.size();
        auto enqueue = [&](int i) { if (!queued_[i]) { queued_[i] = 1; queue_[qt] = i; qt = qt + 1 == qcap ? 0 : qt + 1; qn++; } };
        for (int y = 1; y <= h_; y++)
            for (int x = 1; x <= w_; x++) { const int i = y * pw_ + x; if (ex_[i] > 0 && d_[i] < big_) enqueue(i); }
        long long since = 0;
        const long long period = (long long)n_int / 4 + 1;           // relabels between two global relabellings (hard moves of the two-view runs: n/4 4.4 / 6.6 / 28.9 ms, n/2 5.6 / 6.9 / 33.7, n/8 5.0 / 7.7 / 35.1)
        double absorbed = 0;
        while (qn) {
            const int v = queue_[qh]; qh = qh + 1 == qcap ? 0 : qh + 1; qn--;
            queued_[v] = 0;
            if (d_[v] >= big_) continue;
            float e = ex_[v];
            float* r = &rc_[(size_t)v * 8];
            while (e > 0) {
                const int dv = d_[v];
                for (int k = 0; k < 8 && e > 0; k++) {
                    if (!(r[k] > 0)) continue;
                    const int u = v + off_[k];
                    if (dv != d_[u] + 1) continue;
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
                    if (r[k] > 0) { const int du = d_[v + off_[k]] + 1; if (du < best) best = du; }
                since++;
                cnt_[dv]--;
                if (best >= big_) d_[v] = big_;
                else { d_[v] = best; cnt_[best]++; }
                if (cnt_[dv] == 0) raise_above(dv);                   // gap: nobody left at height dv, everything above it is cut off
                if (d_[v] >= big_) break;
                if (since >= period) { ex_[v] = e; since = 0; global_relabel(); if (d_[v] >= big_) break; }
            }
            ex_[v] = e;
        }
        global_relabel();                                             // the cut is read off exact distances
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
    std::vector<int> queue_, bfs_, cnt_;

    // exact residual distances to the sink: breadth-first search over reversed residual arcs
    void global_relabel()
    {
        bfs_.clear();
        for (int y = 1; y <= h_; y++)
            for (int x = 1; x <= w_; x++) {
                const int i = y * pw_ + x;
                if (ex_[i] < 0) { d_[i] = 1; bfs_.push_back(i); } else d_[i] = big_;
            }
        for (size_t head = 0; head < bfs_.size(); head++) {
            const int v = bfs_[head];
            const int dv = d_[v] + 1;
            for (int k = 0; k < 8; k++) {
                const int u = v + off_[k];                            // the arc u -> v is u's direction k ^ 1 (padding nodes have no capacity)
                if (d_[u] != big_ || !(rc_[(size_t)u * 8 + (k ^ 1)] > 0)) continue;
                d_[u] = dv;
                bfs_.push_back(u);
            }
        }
        std::fill(cnt_.begin(), cnt_.begin() + big_ + 1, 0);
        for (int v : bfs_) cnt_[d_[v]]++;
    }
    void raise_above(int level)
    {
        for (int y = 1; y <= h_; y++)
            for (int x = 1; x <= w_; x++) {
                const int i = y * pw_ + x;
                if (d_[i] > level && d_[i] < big_) { cnt_[d_[i]]--; d_[i] = big_; }
            }
    }
};

}  // namespace les_host
