// MaxFlow.h -- s/t minimum cut for the expansion moves ("next" row N2 of SURVEY.md section 8(f)).
//
// The reference links the Boykov-Kolmogorov max-flow library v3.01/3.04 as Graph<float,float,double>
// (maxflow/README.TXT; call sites LES/FastGCStereo.h:425-428,433,474,492-494,553,557), which is NOT in the
// reference tree.  This is an independent implementation of the published algorithm (Boykov & Kolmogorov,
// "An Experimental Comparison of Min-Cut/Max-Flow Algorithms for Energy Minimization in Vision", PAMI 2004:
// two search trees grown from the terminals, augmentation on contact, adoption of orphans, with the time-stamp /
// distance heuristic of section 3.2) behind the same small interface:
//     add_node(n), add_tweights(i, cap_source, cap_sink), add_edge(i, j, cap, rev_cap), maxflow(), what_segment(i).
// Segment convention (decides label assignments on ties, LES/FastGCStereo.h:557): a node belongs to SINK iff it can
// still reach the sink in the residual graph, every other node -- including nodes cut off from both terminals --
// is SOURCE (the library's default_segm).  Capacities float, flow accumulated in double, like the reference's
// instantiation.
#pragma once

#include <cstdint>
#include <limits>
#include <vector>

namespace les_host {

class MaxFlowGraph {
public:
    enum termtype { SOURCE = 0, SINK = 1 };
    typedef int node_id;

    MaxFlowGraph(int node_num_max, int edge_num_max) : flow_(0)
    {
        nodes_.reserve(node_num_max);
        arcs_.reserve(2 * (size_t)edge_num_max);
    }

    node_id add_node(int num = 1)
    {
        const node_id first = (node_id)nodes_.size();
        nodes_.resize(nodes_.size() + num);
        return first;
    }

    // terminal capacities accumulate; only their difference matters for the cut, the common part is flow
    void add_tweights(node_id i, float cap_source, float cap_sink)
    {
        const float delta = nodes_[i].tr_cap;
        if (delta > 0) cap_source += delta;
        else cap_sink -= delta;
        flow_ += (cap_source < cap_sink) ? cap_source : cap_sink;
        nodes_[i].tr_cap = cap_source - cap_sink;
    }

    void add_edge(node_id i, node_id j, float cap, float rev_cap)
    {
        const int a = (int)arcs_.size();
        arcs_.push_back(Arc{j, nodes_[i].first, a + 1, cap});
        arcs_.push_back(Arc{i, nodes_[j].first, a, rev_cap});
        nodes_[i].first = a;
        nodes_[j].first = a + 1;
    }

    double maxflow()
    {
        init_trees();
        std::vector<int> orphans;
        int current = NONE;
        for (;;) {
            int i = current;
            if (i != NONE) {
                nodes_[i].next_active = NONE_ACTIVE;            // remove from the active list (it is re-examined below)
                if (nodes_[i].parent == NONE) i = NONE;
            }
            if (i == NONE) {
                i = next_active();
                if (i == NONE) break;
            }
            // ---- growth: look for a residual arc that connects the two trees
            int path_arc = NONE;                                // arc from the S tree into the T tree
            if (!nodes_[i].is_sink) {
                for (int a = nodes_[i].first; a != NONE; a = arcs_[a].next) {
                    if (arcs_[a].r_cap <= 0) continue;
                    const int j = arcs_[a].head;
                    Node& nj = nodes_[j];
                    if (nj.parent == NONE) {
                        nj.is_sink = false; nj.parent = arcs_[a].sister; nj.ts = nodes_[i].ts; nj.dist = nodes_[i].dist + 1;
                        set_active(j);
                    } else if (nj.is_sink) { path_arc = a; break; }
                    else if (nj.ts <= nodes_[i].ts && nj.dist > nodes_[i].dist) {   // heuristic: shorter route to the source
                        nj.parent = arcs_[a].sister; nj.ts = nodes_[i].ts; nj.dist = nodes_[i].dist + 1;
                    }
                }
            } else {
                for (int a = nodes_[i].first; a != NONE; a = arcs_[a].next) {
                    const int as = arcs_[a].sister;              // arc j -> i
                    if (arcs_[as].r_cap <= 0) continue;
                    const int j = arcs_[a].head;
                    Node& nj = nodes_[j];
                    if (nj.parent == NONE) {
                        nj.is_sink = true; nj.parent = as; nj.ts = nodes_[i].ts; nj.dist = nodes_[i].dist + 1;
                        set_active(j);
                    } else if (!nj.is_sink) { path_arc = as; break; }
                    else if (nj.ts <= nodes_[i].ts && nj.dist > nodes_[i].dist) {
                        nj.parent = as; nj.ts = nodes_[i].ts; nj.dist = nodes_[i].dist + 1;
                    }
                }
            }
            time_++;
            if (path_arc != NONE) {
                nodes_[i].next_active = i;                       // keep the active flag (not queued): more paths may start here
                current = i;
                augment(path_arc, orphans);
                while (!orphans.empty()) {
                    const int o = orphans.back();
                    orphans.pop_back();
                    if (nodes_[o].is_sink) adopt_sink(o, orphans);
                    else adopt_source(o, orphans);
                }
            } else current = NONE;
        }
        return flow_;
    }

    termtype what_segment(node_id i, termtype default_segm = SOURCE) const
    {
        if (nodes_[i].parent != NONE) return nodes_[i].is_sink ? SINK : SOURCE;
        return default_segm;
    }

private:
    static constexpr int NONE = -1;            // no arc / no node
    static constexpr int TERMINAL = -2;        // parent "arc" of a node attached directly to its terminal
    static constexpr int ORPHAN = -3;          // parent marker of an orphan
    static constexpr int NONE_ACTIVE = -4;     // next_active marker: not in the active list

    struct Node {
        int first = NONE;                      // first outgoing arc
        int parent = NONE;                     // arc from this node towards its tree parent (or TERMINAL / ORPHAN / NONE = free)
        int next_active = NONE_ACTIVE;
        int ts = 0, dist = 0;                  // time stamp and distance to the terminal (valid when ts == time_)
        bool is_sink = false;
        float tr_cap = 0;                      // > 0: residual capacity source -> node, < 0: node -> sink
    };
    struct Arc {
        int head, next, sister;
        float r_cap;
    };

    std::vector<Node> nodes_;
    std::vector<Arc> arcs_;
    double flow_;
    int time_ = 0;
    int queue_first_[2] = {NONE, NONE}, queue_last_[2] = {NONE, NONE};

    void set_active(int i)
    {
        if (nodes_[i].next_active != NONE_ACTIVE) return;       // already queued
        nodes_[i].next_active = i;                               // self-loop marks the tail
        if (queue_last_[1] != NONE) nodes_[queue_last_[1]].next_active = i;
        else queue_first_[1] = i;
        queue_last_[1] = i;
    }
    int next_active()
    {
        for (;;) {
            int i = queue_first_[0];
            if (i == NONE) {
                queue_first_[0] = i = queue_first_[1];
                queue_last_[0] = queue_last_[1];
                queue_first_[1] = queue_last_[1] = NONE;
                if (i == NONE) return NONE;
            }
            if (nodes_[i].next_active == i) queue_first_[0] = queue_last_[0] = NONE;
            else queue_first_[0] = nodes_[i].next_active;
            nodes_[i].next_active = NONE_ACTIVE;
            if (nodes_[i].parent != NONE) return i;              // only nodes that are still in a tree are active
        }
    }

    void init_trees()
    {
        queue_first_[0] = queue_last_[0] = queue_first_[1] = queue_last_[1] = NONE;
        time_ = 0;
        for (int i = 0; i < (int)nodes_.size(); i++) {
            Node& n = nodes_[i];
            n.next_active = NONE_ACTIVE;
            n.ts = time_;
            if (n.tr_cap > 0) { n.is_sink = false; n.parent = TERMINAL; n.dist = 1; set_active(i); }
            else if (n.tr_cap < 0) { n.is_sink = true; n.parent = TERMINAL; n.dist = 1; set_active(i); }
            else n.parent = NONE;
        }
    }

    void make_orphan(int i, std::vector<int>& orphans)
    {
        nodes_[i].parent = ORPHAN;
        orphans.push_back(i);
    }

    // mid: arc from a node of the S tree to a node of the T tree
    void augment(int mid, std::vector<int>& orphans)
    {
        // bottleneck
        float bottleneck = arcs_[mid].r_cap;
        for (int i = arcs_[arcs_[mid].sister].head;;) {          // walk up the S tree
            const int a = nodes_[i].parent;
            if (a == TERMINAL) { if (bottleneck > nodes_[i].tr_cap) bottleneck = nodes_[i].tr_cap; break; }
            const float c = arcs_[arcs_[a].sister].r_cap;        // capacity parent -> i
            if (bottleneck > c) bottleneck = c;
            i = arcs_[a].head;
        }
        for (int i = arcs_[mid].head;;) {                        // walk down to the sink
            const int a = nodes_[i].parent;
            if (a == TERMINAL) { if (bottleneck > -nodes_[i].tr_cap) bottleneck = -nodes_[i].tr_cap; break; }
            if (bottleneck > arcs_[a].r_cap) bottleneck = arcs_[a].r_cap;
            i = arcs_[a].head;
        }
        // push
        arcs_[arcs_[mid].sister].r_cap += bottleneck;
        arcs_[mid].r_cap -= bottleneck;
        for (int i = arcs_[arcs_[mid].sister].head;;) {
            const int a = nodes_[i].parent;
            if (a == TERMINAL) {
                nodes_[i].tr_cap -= bottleneck;
                if (!(nodes_[i].tr_cap > 0)) make_orphan(i, orphans);
                break;
            }
            arcs_[a].r_cap += bottleneck;
            arcs_[arcs_[a].sister].r_cap -= bottleneck;
            const int up = arcs_[a].head;
            if (!(arcs_[arcs_[a].sister].r_cap > 0)) make_orphan(i, orphans);
            i = up;
        }
        for (int i = arcs_[mid].head;;) {
            const int a = nodes_[i].parent;
            if (a == TERMINAL) {
                nodes_[i].tr_cap += bottleneck;
                if (!(nodes_[i].tr_cap < 0)) make_orphan(i, orphans);
                break;
            }
            arcs_[arcs_[a].sister].r_cap += bottleneck;
            arcs_[a].r_cap -= bottleneck;
            const int up = arcs_[a].head;
            if (!(arcs_[a].r_cap > 0)) make_orphan(i, orphans);
            i = up;
        }
        flow_ += bottleneck;
    }

    // distance of j to its terminal through valid tree arcs, or -1 if its tree route is broken; caches time stamps
    int origin_distance(int j)
    {
        int d = 0;
        int k = j;
        for (;;) {
            if (nodes_[k].ts == time_) { d += nodes_[k].dist; break; }
            const int a = nodes_[k].parent;
            d++;
            if (a == TERMINAL) { nodes_[k].ts = time_; nodes_[k].dist = 1; break; }
            if (a == ORPHAN || a == NONE) return -1;
            k = arcs_[a].head;
        }
        // stamp the walked part of the route
        int dd = d;
        for (k = j; nodes_[k].ts != time_; k = arcs_[nodes_[k].parent].head) {
            nodes_[k].ts = time_;
            nodes_[k].dist = dd--;
        }
        return d;
    }

    void adopt_source(int i, std::vector<int>& orphans)
    {
        int best = NONE, best_d = std::numeric_limits<int>::max();
        for (int a0 = nodes_[i].first; a0 != NONE; a0 = arcs_[a0].next) {
            if (!(arcs_[arcs_[a0].sister].r_cap > 0)) continue;  // need residual capacity j -> i
            const int j = arcs_[a0].head;
            if (nodes_[j].is_sink || nodes_[j].parent == NONE) continue;
            const int d = origin_distance(j);
            if (d >= 0 && d < best_d) { best = a0; best_d = d; }
        }
        if (best != NONE) {
            nodes_[i].parent = best; nodes_[i].ts = time_; nodes_[i].dist = best_d + 1;
            return;
        }
        // no parent: i becomes free; neighbours that could reach it become active, its children become orphans
        nodes_[i].ts = 0;
        for (int a0 = nodes_[i].first; a0 != NONE; a0 = arcs_[a0].next) {
            const int j = arcs_[a0].head;
            const int pa = nodes_[j].parent;
            if (!nodes_[j].is_sink && pa != NONE) {
                if (arcs_[arcs_[a0].sister].r_cap > 0) set_active(j);
                if (pa != TERMINAL && pa != ORPHAN && arcs_[pa].head == i) make_orphan(j, orphans);
            }
        }
        nodes_[i].parent = NONE;
    }

    void adopt_sink(int i, std::vector<int>& orphans)
    {
        int best = NONE, best_d = std::numeric_limits<int>::max();
        for (int a0 = nodes_[i].first; a0 != NONE; a0 = arcs_[a0].next) {
            if (!(arcs_[a0].r_cap > 0)) continue;                // need residual capacity i -> j
            const int j = arcs_[a0].head;
            if (!nodes_[j].is_sink || nodes_[j].parent == NONE) continue;
            const int d = origin_distance(j);
            if (d >= 0 && d < best_d) { best = a0; best_d = d; }
        }
        if (best != NONE) {
            nodes_[i].parent = best; nodes_[i].ts = time_; nodes_[i].dist = best_d + 1;
            return;
        }
        nodes_[i].ts = 0;
        for (int a0 = nodes_[i].first; a0 != NONE; a0 = arcs_[a0].next) {
            const int j = arcs_[a0].head;
            const int pa = nodes_[j].parent;
            if (nodes_[j].is_sink && pa != NONE) {
                if (arcs_[a0].r_cap > 0) set_active(j);
                if (pa != TERMINAL && pa != ORPHAN && arcs_[pa].head == i) make_orphan(j, orphans);
            }
        }
        nodes_[i].parent = NONE;
    }
};

}  // namespace les_host
