// prbench.cpp -- experiment: a sequential FIFO push-relabel (first phase, global relabelling + gap heuristic) on the 8-connected grid
// graphs of the expansion moves, next to the Boykov-Kolmogorov solver of host/GridMaxFlow.h, on dumped lock-steps.
//   g++ -O2 -std=c++17 -I localexpstereo_amd/host -I include tools/cpp/prbench.cpp -o tools/cpp/prbench -lpthread && tools/cpp/prbench cell.bin
// Same cut rule as the product solvers: SINK side = the nodes that can still reach the sink in the residual graph.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ExpansionMove.h"
using namespace les_host;
using clk = std::chrono::steady_clock;

struct GridPR {
    int w, h, pw, n;                 // padded by one ring
    int off[8];
    std::vector<float> rc;           // [node][8] residual capacities E W S N SW NE SE NW (sister = k ^ 1)
    std::vector<float> ex;           // > 0 excess, < 0 remaining capacity to the sink
    std::vector<int> d;              // height; >= BIG: cannot reach the sink
    std::vector<int> q;              // FIFO of active nodes (ring buffer)
    std::vector<uint8_t> inq;
    std::vector<int> cnt;            // nodes per height (gap heuristic)
    int BIG;
    double flow = 0;
    long long relabels = 0, pushes = 0, globals = 0, gaps = 0;

    void load(const float* pay, int W, int H)
    {
        w = W; h = H; pw = W + 2; n = (W + 2) * (H + 2);
        const int o[8] = {+1, -1, +pw, -pw, pw - 1, -pw + 1, pw + 1, -pw - 1};
        for (int k = 0; k < 8; k++) off[k] = o[k];
        rc.assign((size_t)n * 8, 0.f); ex.assign(n, 0.f); d.assign(n, 0); inq.assign(n, 0);
        BIG = W * H + 2;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float* p = pay + 5 * ((size_t)y * W + x);
                const int i = (y + 1) * pw + x + 1;
                ex[i] = p[0];
                if (x + 1 < W) rc[(size_t)i * 8 + 0] = p[1];
                if (y + 1 < H) rc[(size_t)i * 8 + 2] = p[2];
                if (y + 1 < H && x > 0) rc[(size_t)i * 8 + 4] = p[3];
                if (y + 1 < H && x + 1 < W) rc[(size_t)i * 8 + 6] = p[4];
            }
        for (int i = 0; i < n; i++) d[i] = BIG;      // padding stays unreachable
    }
    bool inside(int i) const { const int x = i % pw, y = i / pw; return x >= 1 && x <= w && y >= 1 && y <= h; }

    // exact residual distances to the sink (BFS over reversed residual arcs)
    double t_glob = 0;
    void global_relabel()
    {
        globals++;
        auto tg0 = clk::now();
        std::vector<int>& bq = bfs;
        bq.clear();
        for (int y = 1; y <= h; y++)
            for (int x = 1; x <= w; x++) {
                const int i = y * pw + x;
                if (ex[i] < 0) { d[i] = 1; bq.push_back(i); } else d[i] = BIG;
            }
        for (size_t head = 0; head < bq.size(); head++) {
            const int v = bq[head];
            const int dv = d[v] + 1;
            for (int k = 0; k < 8; k++) {
                const int u = v + off[k];                       // arc u -> v is u's direction k ^ 1
                if (d[u] != BIG || !(rc[(size_t)u * 8 + (k ^ 1)] > 0) || !inside(u)) continue;
                d[u] = dv;
                bq.push_back(u);
            }
        }
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int y = 1; y <= h; y++)
            for (int x = 1; x <= w; x++) { const int dd = d[y * pw + x]; if (dd < BIG) cnt[dd]++; }
        t_glob += std::chrono::duration<double>(clk::now() - tg0).count();
    }
    std::vector<int> bfs;

    double run()
    {
        cnt.assign((size_t)BIG + 2, 0);
        double t_in = 0;
        for (int i = 0; i < n; i++) if (ex[i] < 0) t_in += (double)-ex[i];
        global_relabel();
        q.assign(n, 0);
        size_t qh = 0, qt = 0, qn = 0;
        auto push_q = [&](int i) { if (!inq[i]) { inq[i] = 1; q[qt] = i; qt = (qt + 1) % n; qn++; } };
        for (int y = 1; y <= h; y++)
            for (int x = 1; x <= w; x++) { const int i = y * pw + x; if (ex[i] > 0 && d[i] < BIG) push_q(i); }
        long long since = 0;
        const long long period = (long long)((double)w * h * (getenv("PR_PERIOD") ? atof(getenv("PR_PERIOD")) : 0.5)) + 1;       // global relabelling after this many relabels
        const bool use_gap = !getenv("PR_NOGAP");
        while (qn) {
            const int v = q[qh]; qh = (qh + 1) % n; qn--; inq[v] = 0;
            if (d[v] >= BIG) continue;
            float e = ex[v];
            float* r = &rc[(size_t)v * 8];
            while (e > 0) {
                int best = BIG;
                for (int k = 0; k < 8 && e > 0; k++) {
                    if (!(r[k] > 0)) continue;
                    const int u = v + off[k];
                    if (d[v] == d[u] + 1) {
                        const float f = e < r[k] ? e : r[k];
                        r[k] -= f; rc[(size_t)u * 8 + (k ^ 1)] += f; e -= f;
                        const float eu = ex[u];
                        ex[u] = eu + f;                           // a sink arc absorbs what it can
                        if (eu + f > 0 && d[u] < BIG) push_q(u);
                        pushes++;
                    } else if (d[u] + 1 < best) best = d[u] + 1;
                }
                if (!(e > 0)) break;
                // relabel (admissible arcs are exhausted; `best` may miss arcs that were admissible and saturated: recompute)
                best = BIG;
                for (int k = 0; k < 8; k++) if (r[k] > 0) { const int u = v + off[k]; if (d[u] + 1 < best) best = d[u] + 1; }
                const int old = d[v];
                relabels++; since++;
                if (best >= BIG) { d[v] = BIG; cnt[old]--; }
                else { d[v] = best; cnt[old]--; cnt[best]++; }
                if (use_gap && cnt[old] == 0 && old < BIG) {
                    gaps++;
                    // gap: nobody at height `old` -> every node above it is cut off from the sink
                    for (int y = 1; y <= h; y++)
                        for (int x = 1; x <= w; x++) { const int i = y * pw + x; if (d[i] > old && d[i] < BIG) { cnt[d[i]]--; d[i] = BIG; } }
                }
                if (d[v] >= BIG) break;
                if (since >= period) { ex[v] = e; since = 0; global_relabel(); if (d[v] >= BIG) break; }
            }
            ex[v] = e;
        }
        global_relabel();                                         // the cut: nodes with a distance reach the sink
        double t_out = 0;
        for (int i = 0; i < n; i++) if (ex[i] < 0) t_out += (double)-ex[i];
        flow = t_in - t_out;
        return flow;
    }
    // highest-label selection: active nodes in per-height stacks (next-pointer lists), always discharge a node of the largest height
    std::vector<int> head, nxt;
    int highest = 0, dmax = 0;
    void hl_push(int i) { if (!inq[i]) { inq[i] = 1; nxt[i] = head[d[i]]; head[d[i]] = i; if (d[i] > highest) highest = d[i]; } }
    void hl_rebuild()
    {
        std::fill(head.begin(), head.begin() + dmax + 2, -1);
        highest = 0;
        int mx = 0;
        for (int y = 1; y <= h; y++)
            for (int x = 1; x <= w; x++) {
                const int i = y * pw + x;
                inq[i] = 0;
                if (d[i] < BIG && d[i] > mx) mx = d[i];
            }
        dmax = mx;
        for (int y = 1; y <= h; y++)
            for (int x = 1; x <= w; x++) { const int i = y * pw + x; if (ex[i] > 0 && d[i] < BIG) hl_push(i); }
    }
    double run_hl()
    {
        cnt.assign((size_t)BIG + 2, 0);
        head.assign((size_t)BIG + 2, -1); nxt.assign(n, -1);
        double t_in = 0;
        for (int i = 0; i < n; i++) if (ex[i] < 0) t_in += (double)-ex[i];
        global_relabel();
        dmax = BIG; hl_rebuild();
        long long since = 0;
        const long long period = (long long)((double)w * h * (getenv("PR_PERIOD") ? atof(getenv("PR_PERIOD")) : 0.5)) + 1;
        for (;;) {
            while (highest > 0 && head[highest] < 0) highest--;
            if (highest <= 0) break;
            const int v = head[highest];
            head[highest] = nxt[v]; inq[v] = 0;
            if (d[v] != highest || d[v] >= BIG) continue;       // stale entry
            float e = ex[v];
            float* r = &rc[(size_t)v * 8];
            bool again = false;
            while (e > 0) {
                for (int k = 0; k < 8 && e > 0; k++) {
                    if (!(r[k] > 0)) continue;
                    const int u = v + off[k];
                    if (d[v] == d[u] + 1) {
                        const float f = e < r[k] ? e : r[k];
                        r[k] -= f; rc[(size_t)u * 8 + (k ^ 1)] += f; e -= f;
                        const float eu = ex[u];
                        ex[u] = eu + f;
                        if (eu + f > 0 && d[u] < BIG) hl_push(u);
                        pushes++;
                    }
                }
                if (!(e > 0)) break;
                int best = BIG;
                for (int k = 0; k < 8; k++) if (r[k] > 0) { const int u = v + off[k]; if (d[u] + 1 < best) best = d[u] + 1; }
                const int old = d[v];
                relabels++; since++;
                cnt[old]--;
                if (best >= BIG) d[v] = BIG; else { d[v] = best; cnt[best]++; if (best > dmax) dmax = best; }
                if (cnt[old] == 0 && old < BIG) {
                    gaps++;
                    for (int y = 1; y <= h; y++)
                        for (int x = 1; x <= w; x++) { const int i = y * pw + x; if (d[i] > old && d[i] < BIG) { cnt[d[i]]--; d[i] = BIG; } }
                }
                if (d[v] >= BIG) break;
                if (since >= period) { ex[v] = e; since = 0; global_relabel(); hl_rebuild(); again = true; break; }
            }
            if (!again) ex[v] = e;
        }
        global_relabel();
        double t_out = 0;
        for (int i = 0; i < n; i++) if (ex[i] < 0) t_out += (double)-ex[i];
        flow = t_in - t_out;
        return flow;
    }
    bool is_source(int x, int y) const { return d[(y + 1) * pw + x + 1] >= BIG; }
};

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: prbench cell.bin [reps]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    int wh[2];
    if (fread(wh, 4, 2, f) != 2) return 1;
    const int w = wh[0], h = wh[1];
    std::vector<float> pay((size_t)5 * w * h);
    if (fread(pay.data(), 4, pay.size(), f) != pay.size()) return 1;
    fclose(f);
    const int reps = argc > 2 ? atoi(argv[2]) : 2;
    std::vector<uint8_t> mask((size_t)w * h), mask2((size_t)w * h);
    double bk = 1e30, pr = 1e30, flow_bk = 0, flow_pr = 0;
    long long rel = 0, pu = 0, gl = 0;
    for (int r = 0; r < reps; r++) {
        auto t0 = clk::now();
        flow_bk = expansionMovePrebuilt(pay.data(), 0.0, Rect(0, 0, w, h), mask.data(), 1);
        bk = std::min(bk, std::chrono::duration<double>(clk::now() - t0).count());
        GridPR g;
        t0 = clk::now();
        g.load(pay.data(), w, h);
        flow_pr = getenv("PR_HL") ? g.run_hl() : g.run();
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) mask2[(size_t)y * w + x] = g.is_source(x, y) ? 255 : 0;
        pr = std::min(pr, std::chrono::duration<double>(clk::now() - t0).count());
        rel = g.relabels; pu = g.pushes; gl = g.globals; printf("global relabel time %.2f ms ", g.t_glob * 1e3); printf("gaps %lld ", g.gaps);
    }
    size_t diff = 0, ch = 0;
    for (size_t i = 0; i < mask.size(); i++) { diff += (mask[i] != 0) != (mask2[i] != 0); ch += mask[i] != 0; }
    printf("%s %dx%d: BK %.2f ms  push-relabel %.2f ms (%lld pushes, %lld relabels, %lld global relabellings)  flows %.6f / %.6f  changed %.4f  mask differences %zu\n",
           argv[1], w, h, bk * 1e3, pr * 1e3, pu, rel, gl, flow_bk, flow_pr, (double)ch / mask.size(), diff);
    return 0;
}
