// GridMaxFlow.h -- s/t minimum cut on an 8-connected w x h pixel grid, the only graph shape the expansion moves build
// (LES/FastGCStereo.h:485-551: forward neighbours E, S, SW, SE of every pixel of the region).
//
// Same algorithm and segment rule as MaxFlow.h (Boykov-Kolmogorov search trees, FIFO orphan adoption, time-stamp /
// distance heuristic; SINK iff the node can still reach the sink in the residual graph, SOURCE otherwise), but the
// arcs are implicit: every node keeps its 8 residual capacities inline (one cache line per node together with its
// tree state) and neighbours are index offsets on a grid padded by one ring of capacity-less nodes, so the inner
// loops have no arc records, no `next` chains and no bounds checks.  The minimum cut -- hence every label decision --
// is a property of the graph, not of the augmentation order (up to float rounding of the residual capacities); it is
// tested for equality against MaxFlow.h on random instances.  Capacities float, flow double, like the reference's
// Graph<float,float,double>.
#pragma once

#include <algorithm>
#include <cstdint>
#include <limits>
#include <vector>

#include "BandPool.h"

namespace les_host {

class GridMaxFlow {
public:
    enum termtype { SOURCE = 0, SINK = 1 };
    // arc directions; sister(k) == k ^ 1
    enum { E = 0, W = 1, S = 2, N = 3, SW = 4, NE = 5, SE = 6, NW = 7 };

    GridMaxFlow() : w_(0), h_(0), pw_(2), flow_(0) {}
    GridMaxFlow(int w, int h) : GridMaxFlow() { reset(w, h); }
    // (re)initialise for a w x h grid; the node storage is reused, so a thread that solves many small graphs (one per
    // cell and proposal) does not go through the allocator for each of them
    void reset(int w, int h)
    {
        w_ = w; h_ = h; pw_ = w + 2;
        const size_t n = (size_t)(w + 2) * (h + 2);
        if (nodes_.size() < n) nodes_.resize(n);
        std::fill(nodes_.begin(), nodes_.begin() + n, Node());
        const int o[8] = {+1, -1, +pw_, -pw_, pw_ - 1, -pw_ + 1, pw_ + 1, -pw_ - 1};
        for (int k = 0; k < 8; k++) off_[k] = o[k];
        flow_ = 0;
        main_ = Ctx();
        have_sign_ = false; lazy_ = false;
    }
    int id(int x, int y) const { return (y + 1) * pw_ + (x + 1); }

    void add_tweights(int x, int y, float cap_source, float cap_sink)
    {
        Node& n = nodes_[id(x, y)];
        const float delta = n.tr;
        if (delta > 0) cap_source += delta;
        else cap_sink -= delta;
        flow_ += (cap_source < cap_sink) ? cap_source : cap_sink;
        n.tr = cap_source - cap_sink;
    }
    // arc (x, y) -> neighbour in direction k with capacity cap, reverse capacity rev_cap (capacities accumulate)
    void add_edge(int x, int y, int k, float cap, float rev_cap)
    {
        const int i = id(x, y);
        nodes_[i].rc[k] += cap;
        nodes_[i + off_[k]].rc[k ^ 1] += rev_cap;
    }

    // reset for a graph whose every interior node is then initialised with load_node: only the padding ring is cleared
    void reset_for_load(int w, int h)
    {
        w_ = w; h_ = h; pw_ = w + 2;
        const size_t n = (size_t)(w + 2) * (h + 2);
        if (nodes_.size() < n) nodes_.resize(n);
        const Node blank;
        // the byte maps (see load_node): every interior entry is written by load_node, the ring is "sink side, nothing to do"
        if (sign_.size() < n) { sign_.resize(n); dirty_.resize(n); }
        auto ring = [&](size_t i) { nodes_[i] = blank; sign_[i] = 0; dirty_[i] = 0; };
        for (int x = 0; x < pw_; x++) { ring((size_t)x); ring((size_t)(h + 1) * pw_ + x); }
        for (int y = 1; y <= h; y++) { ring((size_t)y * pw_); ring((size_t)y * pw_ + w + 1); }
        have_sign_ = true; lazy_ = false;
        const int o[8] = {+1, -1, +pw_, -pw_, pw_ - 1, -pw_ + 1, pw_ + 1, -pw_ - 1};
        for (int k = 0; k < 8; k++) off_[k] = o[k];
        flow_ = 0;
        main_ = Ctx();
    }
    // direct initialisation from / export to the 5-float node payload {terminal residual, caps E, S, SW, SE} produced by
    // the device (include/localexp_hip.h: les_hip_batch_expansion_graph); base_flow = flow already routed by the t-links
    void load_node(int x, int y, const float* p5)          // sets every field of the node (see reset_for_load)
    {
        Node& n = nodes_[id(x, y)];
        n.rc[E] = p5[1]; n.rc[W] = 0; n.rc[S] = p5[2]; n.rc[N] = 0; n.rc[SW] = p5[3]; n.rc[NE] = 0; n.rc[SE] = p5[4]; n.rc[NW] = 0;
        const float tr = p5[0];
        const bool src = tr > 0, snk = tr < 0;
        n.tr = tr; n.next_active = NOT_QUEUED; n.ts = 0;
        // the initial trees (what init_trees did in a pass of its own): terminal-connected nodes are roots at distance 1
        n.dist = (src || snk) ? 1 : 0; n.parent = (src || snk) ? P_TERMINAL : P_NONE; n.is_sink = snk ? 1 : 0;
        // Byte maps next to the 64-byte nodes.  Most moves of the later iterations are easy: nearly every node prefers its current label
        // (tr < 0: a root of the sink tree) and so do its eight neighbours.  Such a node has nothing to do when it is taken from the
        // active queue -- no free neighbour to claim, no source-tree neighbour to meet, equal distances so no re-parenting -- and unless
        // an augmentation orphans it, it is on the sink side at the end.  sign_ lets init_trees leave those nodes out of the queue
        // (the queue order of the others and every comparison of time stamps is unchanged: the clock ticks per node taken from the
        // queue, and the events of two different ticks stay on different ticks); dirty_ lets the read-out skip them.
        const size_t i = (size_t)id(x, y);
        sign_[i] = src ? 2 : snk ? 0 : 1;
        dirty_[i] = snk ? 0 : 1;
    }
    // A node of a RESIDUAL graph (after reset_for_load): all 8 residual capacities + the terminal residual, i.e. the continuation of a
    // feasible (pre)flow another solver routed -- the tiled device max-flow hands its straggler cells over in this form
    // (csrc/les_maxflow_tiled.h, ResidualCut.h).  Lazy mode as after prepush_rows: only the nodes that still have source excess are
    // queued, the sink trees are not grown, classify() reads the segments off the residual graph.
    void load_residual(int x, int y, const float* rc8, float tr)
    {
        lazy_ = true;
        const size_t i = (size_t)id(x, y);
        Node& n = nodes_[i];
        for (int k = 0; k < 8; k++) n.rc[k] = rc8[k];
        const bool src = tr > 0, snk = tr < 0;
        n.tr = tr; n.next_active = NOT_QUEUED; n.ts = 0;
        n.dist = (src || snk) ? 1 : 0; n.parent = (src || snk) ? P_TERMINAL : P_NONE; n.is_sink = snk ? 1 : 0;
        sign_[i] = src ? 2 : snk ? 0 : 1;
        dirty_[i] = snk ? 1 : 0;                 // lazy mode: the map of the nodes with sink capacity (kept by augment)
    }
    // mask row for the caller: 255 = SOURCE segment (LES/FastGCStereo.h:557: the proposal is taken), 0 = SINK
    void segment_row(int y, uint8_t* out) const
    {
        const size_t r = (size_t)(y + 1) * pw_ + 1;
        if (lazy_) { for (int x = 0; x < w_; x++) out[x] = seg_[r + x] ? 0 : 255; return; }
        if (!have_sign_) { for (int x = 0; x < w_; x++) out[x] = what_segment(x, y) == SOURCE ? 255 : 0; return; }
        const uint8_t* d = &dirty_[r];
        for (int x = 0; x < w_; x++) {
            if (!d[x]) { out[x] = 0; continue; }
            const Node& n = nodes_[r + x];
            out[x] = (n.parent != P_NONE && n.is_sink) ? 0 : 255;
        }
    }
    // Local pre-push for a graph loaded with load_node: rows [y0, y1), to be called before maxflow (once per row band, the bands
    // may run on different threads; a band pushes only inside itself).  Returns the flow routed (add it with add_base_flow).
    //
    // Why: the expansion-move construction (LES/FastGCStereo.h:485-551) writes every pairwise term as +c at one endpoint, -c at
    // the other and an arc between them that can carry c.  On the moves of an almost converged labelling 44 % of the nodes of a
    // coarse cell come out with source excess -- and one sweep in which every such node pushes along its four forward arcs into
    // neighbours with sink capacity cancels 99.5 % of it (a 387 x 387 cell: 28 896 units of flow, 36 left on 0.2 % of the nodes).
    // The search trees then start from those few nodes instead of from every second one.  Any feasible flow may be routed first:
    // every s-t cut loses the same amount, so the minimum cuts -- and the one the segment rule picks -- are unchanged.
    //
    // The sweep leaves the former excess nodes with tr == 0 ("free").  Growing the sink tree over them is what would cost the
    // time now, so this mode does without: only source roots are queued (a search from the source side alone finds every
    // augmenting path; sink roots stay roots, orphans are adopted as usual), and the segments are read off the residual graph
    // afterwards by classify(): SINK = the nodes that can still reach a node with sink capacity, which is the same rule.
    double prepush_rows(int y0, int y1)
    {
        lazy_ = true;
        double routed = 0;
        for (int y = y0; y < y1; y++) routed += prepush_row(y, y + 1 >= y1);
        return routed;
    }
    // load_node for the rows [y0, y1) of a w x h payload (5 floats per node, row-major) with the pre-push of a row done as soon as
    // the row below it is in place, while both are still in the cache
    // sources (optional): += the number of nodes of these rows that still have source excess afterwards
    double load_rows_prepushed(const float* payload, int y0, int y1, long long* sources = nullptr)
    {
        lazy_ = true;
        double routed = 0;
        for (int y = y0; y < y1; y++) {
            const float* p = payload + 5 * (size_t)y * w_;
            for (int x = 0; x < w_; x++) load_node(x, y, p + 5 * x);
            if (y > y0) routed += prepush_row(y - 1, false);
        }
        if (y1 > y0) routed += prepush_row(y1 - 1, true);
        if (sources) {
            long long n = 0;
            for (int y = y0; y < y1; y++) {
                const uint8_t* sg = &sign_[(size_t)(y + 1) * pw_ + 1];
                for (int x = 0; x < w_; x++) n += sg[x] == 2;
            }
            *sources += n;
        }
        return routed;
    }
    void add_base_flow(double f) { flow_ += f; }

private:
    double prepush_row(int y, bool last)
    {
        double routed = 0;
        {
            const size_t r = (size_t)(y + 1) * pw_ + 1;
            for (int x = 0; x < w_; x++) {
                Node& n = nodes_[r + x];
                float e = n.tr;
                if (e > 0) {
                    static constexpr int kForward[4] = {E, S, SW, SE};      // the arcs that have capacity at load time
                    for (int q = 0; q < 4 && e > 0; q++) {
                        const int k = kForward[q];
                        if (k != E && last) break;
                        const float c = n.rc[k];
                        if (!(c > 0)) continue;
                        Node& m = nodes_[r + x + off_[k]];
                        const float d = m.tr;
                        if (!(d < 0)) continue;
                        float f = e < c ? e : c;
                        if (-d < f) f = -d;
                        n.rc[k] = c - f; m.rc[k ^ 1] += f;
                        e -= f; m.tr = d + f;
                        routed += (double)f;
                    }
                    n.tr = e;
                }
            }
            // row y is final now (rows above have pushed into it, it has pushed east and down): the initial trees and the maps
            for (int x = 0; x < w_; x++) {
                Node& n = nodes_[r + x];
                const bool src = n.tr > 0, snk = n.tr < 0;
                n.dist = (src || snk) ? 1 : 0; n.parent = (src || snk) ? P_TERMINAL : P_NONE; n.is_sink = snk ? 1 : 0;
                sign_[r + x] = src ? 2 : snk ? 0 : 1;
                dirty_[r + x] = snk ? 1 : 0;             // lazy mode: dirty_ is the map of the nodes with sink capacity (kept by augment)
            }
        }
        return routed;
    }

public:

    void store_node(int x, int y, float* p5) const
    {
        const Node& n = nodes_[id(x, y)];
        p5[0] = n.tr; p5[1] = n.rc[E]; p5[2] = n.rc[S]; p5[3] = n.rc[SW]; p5[4] = n.rc[SE];
    }
    void set_base_flow(double f) { flow_ = f; }
    double base_flow() const { return flow_; }

    // bands > 1: parallel first phase.  The rows are cut into `bands` bands; every band runs the search on its own thread as
    // if the arcs into the other bands did not exist (the flows found are feasible for the whole graph and the trees are
    // valid), then one search continues on the whole graph from the union of the trees with the nodes next to the band
    // borders re-activated (the standard "capacities increased, reuse the trees" continuation).  The final flow is a
    // maximum flow of the whole graph, so the cut read-out is the same as for bands == 1.
    // ops_per_node > 0: give up after that much search work per node (hard instances: the caller continues with push-relabel on the
    // residual graph, see exhausted() / residual()); the band searches of the parallel first phase get the same allowance per band node
    double maxflow(int bands = 1, double ops_per_node = 0.0, double band_ops_per_node = -1.0)
    {
        if (band_ops_per_node < 0) band_ops_per_node = ops_per_node;
        exhausted_ = false;
        if (bands > h_ / 8) bands = h_ / 8;                      // at least 8 rows per band
        if (bands > 64) bands = 64;
        const long long budget = ops_per_node > 0 ? (long long)(ops_per_node * w_ * h_) + 1 : 0;
        if (bands <= 1) {
            main_.budget = budget;
            init_trees(main_, 0, h_);
            exhausted_ = !search(main_);
            if (lazy_ && !exhausted_) classify(1);
            return flow_ + main_.flow;
        }
        std::vector<int> row0(bands + 1);
        for (int b = 0; b <= bands; b++) row0[b] = (int)((long long)h_ * b / bands);
        for (int b = 0; b < bands; b++)
            for (int y = row0[b]; y < row0[b + 1]; y++)
                for (int x = 0; x < w_; x++) nodes_[id(x, y)].band = (uint8_t)b;
        markPadding();
        std::vector<Ctx> ctx(bands);
        std::vector<char> band_done(bands, 1);
        BandPool::mine().run(bands, [&](int b) {
            ctx[b].band = b;
            ctx[b].budget = budget > 0 ? (long long)(band_ops_per_node * w_ * (row0[b + 1] - row0[b])) + 1 : 0;
            init_trees(ctx[b], row0[b], row0[b + 1]);
            band_done[b] = search(ctx[b]) ? 1 : 0;
        });
        // continuation on the whole graph
        main_ = Ctx();
        for (int b = 0; b < bands; b++) { main_.flow += ctx[b].flow; main_.time = std::max(main_.time, ctx[b].time); }
        // A band that ran out of budget still has active nodes inside, which the continuation below (it re-activates the band borders only)
        // would never visit: the instance is a hard one anyway -- hand the feasible flow found so far to the caller's other solver.
        for (int b = 0; b < bands; b++)
            if (!band_done[b]) { exhausted_ = true; return flow_ + main_.flow; }
        main_.time += 1;
        main_.epoch = main_.time;
        for (int b = 1; b < bands; b++)
            for (int dy = -1; dy <= 0; dy++) {                   // last row of band b-1 and first row of band b
                const int y = row0[b] + dy;
                for (int x = 0; x < w_; x++) {
                    const int i = id(x, y);
                    if (nodes_[i].parent != P_NONE) set_active(main_, i);      // (lazy mode: set_active keeps the source-tree nodes only)
                }
            }
        main_.budget = budget;
        exhausted_ = !search(main_);
        if (lazy_ && !exhausted_) classify(bands);
        return flow_ + main_.flow;
    }
    // after maxflow(..., ops_per_node): true = the budget ran out; the return value is then the flow routed so far and residual(x, y)
    // the remaining problem (8 residual capacities E W S N SW NE SE NW + the terminal residual)
    bool exhausted() const { return exhausted_; }
    void residual(int x, int y, float* rc8, float* tr) const
    {
        const Node& n = nodes_[id(x, y)];
        for (int k = 0; k < 8; k++) rc8[k] = n.rc[k];
        *tr = n.tr;
    }

    termtype what_segment(int x, int y) const
    {
        if (lazy_) return seg_[(size_t)id(x, y)] ? SINK : SOURCE;
        const Node& n = nodes_[id(x, y)];
        return (n.parent != P_NONE && n.is_sink) ? SINK : SOURCE;
    }

private:
    static constexpr int NONE_NODE = -1, NOT_QUEUED = -2;
    static constexpr int8_t P_TERMINAL = 8, P_ORPHAN = 9, P_NONE = 10;

    struct alignas(64) Node {
        float rc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // residual capacity towards the 8 neighbours
        float tr = 0;                              // > 0: source -> node residual, < 0: node -> sink
        int next_active = NOT_QUEUED;
        int ts = 0, dist = 0;
        int8_t parent = P_NONE;                    // direction towards the tree parent, or P_TERMINAL / P_ORPHAN / P_NONE (free)
        uint8_t is_sink = 0;
        uint8_t band = 0;                          // row band of the node (parallel first phase); padding nodes: 255
    };

    // state of one search: the whole graph, or -- in the parallel first phase -- one band of rows whose arcs into other bands
    // are ignored
    struct alignas(128) Ctx {                      // one cache-line pair per search: the band searches run on different cores
        int queue_first[2] = {NONE_NODE, NONE_NODE}, queue_last[2] = {NONE_NODE, NONE_NODE};
        std::vector<int> orphans;
        size_t orphan_head = 0;
        int time = 0;
        int epoch = 0;                             // marks older than this come from another clock: not used by the re-parenting heuristic
        double flow = 0;
        int band = -1;                             // >= 0: restricted to this band
        long long ops = 0, budget = 0;             // work done (grow steps, adoptions, steps of origin walks) / allowed (0: unlimited)
    };

    int w_, h_, pw_;
    int off_[8];
    double flow_;                                  // flow routed by the t-links while the graph was built
    bool exhausted_ = false;
    bool have_sign_ = false;                       // the graph was loaded with load_node: sign_ / dirty_ are valid
    std::vector<Node> nodes_;
    std::vector<uint8_t> sign_;                    // per node at load time: 0 tr < 0, 1 tr == 0 (free), 2 tr > 0; read-only during the searches
    bool lazy_ = false;                            // prepush_rows was used: sink trees are not grown, classify() reads the segments
    std::vector<uint8_t> seg_;                     // lazy mode, after maxflow: 1 = the node can reach the sink in the residual graph
    std::vector<int> bfs_;
    std::vector<std::vector<int>> bandq_;          // per band: queue, nodes whose propagation crosses the band border
    std::vector<uint8_t> dirty_;                   // 1: the node was not a sink root at load time or has been freed since (may be SOURCE at the end)
    Ctx main_;

    // lazy mode: residual reachability of the sink.  Start: the nodes with sink capacity left; then every node with a residual arc
    // into the set joins it: one raster pass that pulls, then a queue for what the raster order missed.  bands > 1: every row band
    // does that inside itself on its own thread (it reads the neighbouring bands' bytes, which only ever change from 0 to 1, and
    // never writes them); what would cross a band border is finished by one thread afterwards.  The result is the least fixed
    // point whatever the interleaving.
    static uint8_t ldb(const uint8_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
    static void stb(uint8_t* p) { __atomic_store_n(p, (uint8_t)1, __ATOMIC_RELAXED); }
    void classify_rows(int y0, int y1, std::vector<int>& q, std::vector<int>* spill)
    {
        uint8_t* seg = seg_.data();
        q.clear();
        for (int y = y0; y < y1; y++) {
            const size_t r = (size_t)(y + 1) * pw_ + 1;
            const bool edge = spill && (y == y0 || y + 1 == y1);
            for (int x = 0; x < w_; x++) {
                const size_t i = r + x;
                if (ldb(seg + i)) continue;
                const Node& nd = nodes_[i];
                for (int k = 0; k < 8; k++)
                    if (nd.rc[k] > 0 && ldb(seg + i + off_[k])) {
                        stb(seg + i);
                        // the nodes later in raster order pull from this one themselves; only an undecided earlier neighbour needs the
                        // queue (and, in a band, the rows next to another band: its thread may have passed already)
                        if (edge || !(ldb(seg + i - 1) & ldb(seg + i - pw_ - 1) & ldb(seg + i - pw_) & ldb(seg + i - pw_ + 1))) q.push_back((int)i);
                        break;
                    }
            }
        }
        const int lo = (y0 + 1) * pw_, hi = (y1 + 1) * pw_;          // node indices of the band's rows
        for (size_t head = 0; head < q.size(); head++) {
            const int u = q[head];
            bool spilled = false;
            for (int k = 0; k < 8; k++) {
                const int v = u + off_[k];
                if (ldb(seg + v) || !(nodes_[v].rc[k ^ 1] > 0)) continue;      // residual arc v -> u
                if (spill && (v < lo || v >= hi)) { if (!spilled) { spill->push_back(u); spilled = true; } continue; }
                stb(seg + v);
                q.push_back(v);
            }
        }
    }
    void classify(int bands)
    {
        const size_t n = (size_t)pw_ * (h_ + 2);
        if (seg_.size() < n) seg_.resize(n);
        std::copy(dirty_.begin(), dirty_.begin() + n, seg_.begin());
        if (bands <= 1) { classify_rows(0, h_, bfs_, nullptr); return; }
        if ((int)bandq_.size() < 2 * bands) bandq_.resize((size_t)2 * bands);
        BandPool::mine().run(bands, [&](int b) {
            bandq_[2 * b + 1].clear();
            classify_rows((int)((long long)h_ * b / bands), (int)((long long)h_ * (b + 1) / bands), bandq_[2 * b], &bandq_[2 * b + 1]);
        });
        // across the borders: the spilled nodes once more, now without a fence
        bfs_.clear();
        for (int b = 0; b < bands; b++) bfs_.insert(bfs_.end(), bandq_[2 * b + 1].begin(), bandq_[2 * b + 1].end());
        uint8_t* seg = seg_.data();
        for (size_t head = 0; head < bfs_.size(); head++) {
            const int u = bfs_[head];
            for (int k = 0; k < 8; k++) {
                const int v = u + off_[k];
                if (seg[v] || !(nodes_[v].rc[k ^ 1] > 0)) continue;
                seg[v] = 1;
                bfs_.push_back(v);
            }
        }
    }

    void markPadding()
    {
        for (int x = 0; x < pw_; x++) { nodes_[x].band = 255; nodes_[(size_t)(h_ + 1) * pw_ + x].band = 255; }
        for (int y = 1; y <= h_; y++) { nodes_[(size_t)y * pw_].band = 255; nodes_[(size_t)y * pw_ + w_ + 1].band = 255; }
    }
    // may the search `c` look at node j?  (whole graph: always; band search: only nodes of its band -- padding has band 255)
    bool allowed(const Ctx& c, const Node& nj) const { return c.band < 0 || nj.band == c.band; }

    void set_active(Ctx& c, int i)
    {
        if (nodes_[i].next_active != NOT_QUEUED) return;
        if (lazy_ && nodes_[i].is_sink) return;                 // lazy mode: the search runs from the source side only (see prepush_rows)
        nodes_[i].next_active = i;
        if (c.queue_last[1] != NONE_NODE) nodes_[c.queue_last[1]].next_active = i;
        else c.queue_first[1] = i;
        c.queue_last[1] = i;
    }
    int next_active(Ctx& c)
    {
        for (;;) {
            int i = c.queue_first[0];
            if (i == NONE_NODE) {
                c.queue_first[0] = i = c.queue_first[1];
                c.queue_last[0] = c.queue_last[1];
                c.queue_first[1] = c.queue_last[1] = NONE_NODE;
                if (i == NONE_NODE) return NONE_NODE;
            }
            if (nodes_[i].next_active == i) c.queue_first[0] = c.queue_last[0] = NONE_NODE;
            else c.queue_first[0] = nodes_[i].next_active;
            nodes_[i].next_active = NOT_QUEUED;
            if (nodes_[i].parent != P_NONE) return i;
        }
    }
    void init_trees(Ctx& c, int y0, int y1)
    {
        if (have_sign_) {
            // the trees were set up by load_node; queue the source roots and the sink roots with a neighbour that is not a sink root
            std::vector<uint8_t> f((size_t)pw_);
            for (int y = y0; y < y1; y++) {
                const size_t r = (size_t)(y + 1) * pw_;
                const uint8_t *a = &sign_[r - pw_], *b = &sign_[r], *d = &sign_[r + pw_];
                if (lazy_) {
                    for (int x = 1; x <= w_; x++) if (b[x] == 2) set_active(c, (int)(r + x));
                    continue;
                }
                for (int x = 1; x <= w_; x++) f[x] = (uint8_t)(a[x - 1] | a[x] | a[x + 1] | b[x - 1] | b[x + 1] | d[x - 1] | d[x] | d[x + 1]);
                for (int x = 1; x <= w_; x++)
                    if (b[x] == 2 || (b[x] == 0 && f[x])) set_active(c, (int)(r + x));
            }
            return;
        }
        for (int y = y0; y < y1; y++)
            for (int x = 0; x < w_; x++) {
                const int i = id(x, y);
                Node& n = nodes_[i];
                if (n.tr > 0) { n.is_sink = 0; n.parent = P_TERMINAL; n.dist = 1; set_active(c, i); }
                else if (n.tr < 0) { n.is_sink = 1; n.parent = P_TERMINAL; n.dist = 1; set_active(c, i); }
            }
    }
    void make_orphan(Ctx& c, int i) { nodes_[i].parent = P_ORPHAN; c.orphans.push_back(i); }

    // -> false when the work budget ran out (the flow routed so far is a feasible flow: the residual graph is a valid max-flow problem)
    bool search(Ctx& c)
    {
        int current = NONE_NODE;
        for (;;) {
            if (c.budget > 0 && c.ops > c.budget) {
                if (current != NONE_NODE) nodes_[current].next_active = NOT_QUEUED;
                return false;
            }
            c.ops++;
            int i = current;
            if (i != NONE_NODE) {
                nodes_[i].next_active = NOT_QUEUED;
                if (nodes_[i].parent == P_NONE) i = NONE_NODE;
            }
            if (i == NONE_NODE) {
                i = next_active(c);
                if (i == NONE_NODE) break;
            }
            Node& ni = nodes_[i];
            int mid_s = NONE_NODE, mid_k = 0;                   // connecting arc: S-tree node mid_s, direction mid_k
            if (!ni.is_sink) {
                for (int k = 0; k < 8; k++) {
                    if (!(ni.rc[k] > 0)) continue;
                    const int j = i + off_[k];
                    Node& nj = nodes_[j];
                    if (!allowed(c, nj)) continue;
                    if (nj.parent == P_NONE) {
                        nj.is_sink = 0; nj.parent = (int8_t)(k ^ 1); nj.ts = ni.ts; nj.dist = ni.dist + 1;
                        set_active(c, j);
                    } else if (nj.is_sink) { mid_s = i; mid_k = k; break; }
                    else if (nj.ts <= ni.ts && nj.dist > ni.dist && nj.ts >= c.epoch) { nj.parent = (int8_t)(k ^ 1); nj.ts = ni.ts; nj.dist = ni.dist + 1; }
                }
            } else {
                for (int k = 0; k < 8; k++) {
                    const int j = i + off_[k];
                    Node& nj = nodes_[j];
                    if (!allowed(c, nj)) continue;              // (first: a band search must not even read another band's capacities)
                    if (!(nj.rc[k ^ 1] > 0)) continue;          // residual capacity j -> i
                    if (nj.parent == P_NONE) {
                        nj.is_sink = 1; nj.parent = (int8_t)(k ^ 1); nj.ts = ni.ts; nj.dist = ni.dist + 1;
                        set_active(c, j);
                    } else if (!nj.is_sink) { mid_s = j; mid_k = k ^ 1; break; }
                    else if (nj.ts <= ni.ts && nj.dist > ni.dist && nj.ts >= c.epoch) { nj.parent = (int8_t)(k ^ 1); nj.ts = ni.ts; nj.dist = ni.dist + 1; }
                }
            }
            c.time++;
            if (mid_s != NONE_NODE) {
                ni.next_active = i;                              // stays active (not queued): more paths may start here
                current = i;
                augment(c, mid_s, mid_k);
                while (c.orphan_head < c.orphans.size()) {
                    const int o = c.orphans[c.orphan_head++];
                    if (nodes_[o].is_sink) adopt<true>(c, o);
                    else adopt<false>(c, o);
                }
                c.orphans.clear();
                c.orphan_head = 0;
            } else current = NONE_NODE;
        }
        return true;
    }

    void augment(Ctx& c, int s, int k)
    {
        const int t = s + off_[k];
        float bottleneck = nodes_[s].rc[k];
        for (int i = s;;) {                                      // up the S tree
            const int p = nodes_[i].parent;
            if (p == P_TERMINAL) { if (bottleneck > nodes_[i].tr) bottleneck = nodes_[i].tr; break; }
            const int m = i + off_[p];
            const float cc = nodes_[m].rc[p ^ 1];                // parent -> i
            if (bottleneck > cc) bottleneck = cc;
            i = m;
        }
        for (int i = t;;) {                                      // down to the sink
            const int p = nodes_[i].parent;
            if (p == P_TERMINAL) { if (bottleneck > -nodes_[i].tr) bottleneck = -nodes_[i].tr; break; }
            const float cc = nodes_[i].rc[p];
            if (bottleneck > cc) bottleneck = cc;
            i += off_[p];
        }
        nodes_[s].rc[k] -= bottleneck;
        nodes_[t].rc[k ^ 1] += bottleneck;
        for (int i = s;;) {
            const int p = nodes_[i].parent;
            if (p == P_TERMINAL) {
                nodes_[i].tr -= bottleneck;
                if (!(nodes_[i].tr > 0)) make_orphan(c, i);
                break;
            }
            const int m = i + off_[p];
            nodes_[i].rc[p] += bottleneck;
            nodes_[m].rc[p ^ 1] -= bottleneck;
            if (!(nodes_[m].rc[p ^ 1] > 0)) make_orphan(c, i);
            i = m;
        }
        for (int i = t;;) {
            const int p = nodes_[i].parent;
            if (p == P_TERMINAL) {
                nodes_[i].tr += bottleneck;
                if (!(nodes_[i].tr < 0)) { make_orphan(c, i); if (lazy_) dirty_[i] = 0; }
                break;
            }
            const int m = i + off_[p];
            nodes_[m].rc[p ^ 1] += bottleneck;
            nodes_[i].rc[p] -= bottleneck;
            if (!(nodes_[i].rc[p] > 0)) make_orphan(c, i);
            i = m;
        }
        c.flow += bottleneck;
    }

    int origin_distance(Ctx& c, int j)
    {
        int d = 0, k = j;
        for (;;) {
            c.ops++;
            if (nodes_[k].ts == c.time) { d += nodes_[k].dist; break; }
            const int p = nodes_[k].parent;
            d++;
            if (p == P_TERMINAL) { nodes_[k].ts = c.time; nodes_[k].dist = 1; break; }
            if (p == P_ORPHAN || p == P_NONE) return -1;
            k += off_[p];
        }
        int dd = d;
        for (k = j; nodes_[k].ts != c.time; k += off_[nodes_[k].parent]) {
            nodes_[k].ts = c.time;
            nodes_[k].dist = dd--;
        }
        return d;
    }

    template <bool SINKTREE>
    void adopt(Ctx& c, int i)
    {
        c.ops++;
        Node& ni = nodes_[i];
        int best = -1, best_d = std::numeric_limits<int>::max();
        for (int k = 0; k < 8; k++) {
            const int j = i + off_[k];
            const Node& nj = nodes_[j];
            if (!allowed(c, nj)) continue;
            // source tree: need residual j -> i; sink tree: need residual i -> j
            if (SINKTREE ? !(ni.rc[k] > 0) : !(nj.rc[k ^ 1] > 0)) continue;
            if ((bool)nj.is_sink != SINKTREE || nj.parent == P_NONE) continue;
            const int d = origin_distance(c, j);
            if (d >= 0 && d < best_d) { best = k; best_d = d; }
        }
        if (best >= 0) {
            ni.parent = (int8_t)best; ni.ts = c.time; ni.dist = best_d + 1;
            return;
        }
        ni.ts = 0;
        for (int k = 0; k < 8; k++) {
            const int j = i + off_[k];
            Node& nj = nodes_[j];
            if (!allowed(c, nj)) continue;
            const int pa = nj.parent;
            if ((bool)nj.is_sink == SINKTREE && pa != P_NONE) {
                if (SINKTREE ? (ni.rc[k] > 0) : (nj.rc[k ^ 1] > 0)) set_active(c, j);
                if (pa == (k ^ 1)) make_orphan(c, j);            // j's parent is i
            }
        }
        ni.parent = P_NONE;
        if (have_sign_ && !lazy_) dirty_[i] = 1;
    }
};

}  // namespace les_host
