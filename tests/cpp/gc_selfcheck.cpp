// gc_selfcheck.cpp -- TEST INFRASTRUCTURE: exercises the host graph-cut side (ExpansionMove.h / MaxFlow.h / the doGC
// loop of PMStereo.h) without a GPU.  The unary operator used here is NOT the product path and not a fallback for it:
// it is a deliberately different, un-aggregated per-pixel cost (truncated volume lookup) that only exists so that the
// local expansion moves have something to fuse on a CPU-only box.
//
// Checks (exit code != 0 on failure):
//   1. the reference's own disabled self-check (LES/FastGCStereo.h:561-594): for every move, max-flow value ==
//      energy of the fused labelling over the terms touching the region, within 1e-5 relative
//   2. every expansion move is optimal against brute force on tiny regions (all 2^N masks)
//   3. the total energy (data + smoothness) never increases over graph-cut iterations and the scene converges
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <thread>

#include "PMStereo.h"
#include "DemoScene.h"
#include "MaxFlow.h"

using namespace les_host;

class PointwiseTestEnergy : public StereoEnergy {
public:
    PointwiseTestEnergy(const Scene& s, Parameters p, float maxd) : StereoEnergy(s.W, s.H, std::move(p), maxd, 0), s_(s)
    {
        setImages(s.im.data(), s.im.data());
    }
    void ComputeUnaryPotentialWithoutCheck(const Rect& fr, const Rect& tr, float* costs, int row_stride, const Plane& plane, Reusable&,
                                           int = 0) const override
    {
        for (int y = tr.y; y < tr.y + tr.height; y++)
            for (int x = tr.x; x < tr.x + tr.width; x++) {
                float d = plane.GetZ((float)x, (float)y);
                d = std::min(std::max(d, 0.0f), (float)(s_.D - 1));
                const int d0 = std::min((int)d, s_.D - 2);
                const float f = d - (float)d0;
                const float c = s_.vol[((size_t)d0 * s_.H + y) * s_.W + x] * (1 - f) + s_.vol[((size_t)(d0 + 1) * s_.H + y) * s_.W + x] * f;
                costs[(size_t)(y - fr.y) * row_stride + (x - fr.x)] = std::min(c, params.th_col);
            }
    }
    void ComputeUnaryPotential(const Rect& fr, const Rect& tr, float* costs, int row_stride, const Plane& plane, Reusable& r, int mode = 0) const override
    {
        ComputeUnaryPotentialWithoutCheck(fr, tr, costs, row_stride, plane, r, mode);
        for (int y = tr.y; y < tr.y + tr.height; y++)
            for (int x = tr.x; x < tr.x + tr.width; x++)
                if (!IsValiLabel(plane, Point{x, y})) costs[(size_t)(y - fr.y) * row_stride + (x - fr.x)] = (float)COST_FOR_INVALID;
    }

private:
    const Scene& s_;
};

static int brute_force(const Scene& s, const Parameters& param, float maxd)
{
    PointwiseTestEnergy E(s, param, maxd);
    RNG rng(99);
    LabelMap lab(s.H, s.W);
    CostMap cur(s.H, s.W, 0.f), prop(s.H, s.W, 0.f);
    StereoEnergy::Reusable tmp;
    int fail = 0;
    for (int trial = 0; trial < 60; trial++) {
        // random piecewise labelling and costs, random small region (<= 12 pixels), random proposal
        for (int y = 0; y < s.H; y++)
            for (int x = 0; x < s.W; x++) {
                if ((x % 3 == 0 && y % 2 == 0) || (x == 0 && y == 0)) lab.at(y, x) = E.createRandomLabel(Point{x, y}, rng);
                else lab.at(y, x) = x % 3 ? lab.at(y, x - 1) : lab.at(y - 1, x);
                cur.at(y, x) = rng.uniform(0.0f, 1.0f);
                prop.at(y, x) = rng.uniform(0.0f, 1.0f);
            }
        const int w = rng.uniform(1, 5), h = rng.uniform(1, 4);
        const Rect region(rng.uniform(0, s.W - w + 1), rng.uniform(0, s.H - h + 1), w, h);
        const Plane label = E.createRandomLabel(Point{region.x, region.y}, rng);
        std::vector<uint8_t> mask;
        const double flow = expansionMove(E, lab, cur, prop, label, region, mask);
        const double e_cut = fusedEnergy(E, lab, cur, prop, label, region, mask);
        double best = 1e300;
        const int N = w * h;
        std::vector<uint8_t> m(N);
        for (int bits = 0; bits < (1 << N); bits++) {
            for (int i = 0; i < N; i++) m[i] = (bits >> i) & 1 ? 255 : 0;
            best = std::min(best, fusedEnergy(E, lab, cur, prop, label, region, m));
        }
        const double tol = 1e-5 * std::max(1.0, std::fabs(best));
        if (std::fabs(flow - e_cut) > tol || e_cut > best + tol) {
            printf("FAIL brute force trial %d: region %dx%d flow=%.7f cut energy=%.7f optimum=%.7f\n", trial, w, h, flow, e_cut, best);
            fail = 1;
        }
    }
    printf("brute force: 60 random moves %s\n", fail ? "FAILED" : "optimal");
    return fail;
}

// GridMaxFlow (implicit 8-connected grid arcs) against the generic MaxFlowGraph on random instances: same flow value, same
// segments (the minimum cut with the largest source side is unique).  Integer-valued capacities keep the arithmetic exact.
static int grid_vs_generic()
{
    RNG rng(2024);
    int fail = 0;
    for (int trial = 0; trial < 40; trial++) {
        const int w = rng.uniform(1, 24), h = rng.uniform(1, 20);
        GridMaxFlow g(w, h);
        MaxFlowGraph m(w * h, 4 * w * h);
        m.add_node(w * h);
        const int range = trial % 2 ? 4 : 50;                       // small range: many ties / saturated arcs
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                for (int rep = 0; rep < 2; rep++) {                 // accumulating t-links like the expansion move does
                    const float a = (float)rng.uniform(0, range), b = (float)rng.uniform(0, range);
                    g.add_tweights(x, y, a, b); m.add_tweights(y * w + x, a, b);
                }
                const int dx[4] = {1, 0, -1, 1}, dy[4] = {0, 1, 1, 1}, dir[4] = {GridMaxFlow::E, GridMaxFlow::S, GridMaxFlow::SW, GridMaxFlow::SE};
                for (int k = 0; k < 4; k++) {
                    const int xx = x + dx[k], yy = y + dy[k];
                    if (xx < 0 || xx >= w || yy >= h) continue;
                    const float c = (float)rng.uniform(0, range), r = (float)rng.uniform(0, trial % 3 ? 1 : range);
                    g.add_edge(x, y, dir[k], c, r); m.add_edge(y * w + x, yy * w + xx, c, r);
                }
            }
        const double fg = g.maxflow(), fm = m.maxflow();
        int diff = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) diff += (int)g.what_segment(x, y) != (int)m.what_segment(y * w + x);
        if (fg != fm || diff) { printf("FAIL grid max-flow trial %d (%dx%d): flow %.1f vs %.1f, %d segment differences\n", trial, w, h, fg, fm, diff); fail = 1; }
    }
    printf("grid max-flow vs generic: 40 random grids %s\n", fail ? "FAILED" : "identical");
    return fail;
}

// GridPushRelabel (the solver of the coarsest layer's cells) against GridMaxFlow on random graphs in the device payload format (terminal
// residual + caps E, S, SW, SE): integer-valued capacities keep the arithmetic exact, so flow and segments must be IDENTICAL -- the cut with
// the largest source side is unique -- for tiny grids, single rows / columns, grids with no arcs, with only one kind of terminal, with huge
// terminals next to small capacities, and for grids of a few thousand nodes where relabelling periods and gaps occur.
static int push_relabel_vs_bk()
{
    RNG rng(4711);
    int fail = 0, trials = 0, hybrids = 0, exhausted_none = 0, prepushed = 0, banded_pr = 0;
    const int shapes[][2] = {{1, 1}, {2, 1}, {1, 7}, {9, 1}, {2, 2}, {5, 3}, {16, 12}, {33, 27}, {64, 48}, {97, 61}, {120, 90}};
    for (const auto& sh : shapes)
        for (int variant = 0; variant < 6; variant++, trials++) {
            const int w = sh[0], h = sh[1];
            std::vector<float> pay((size_t)5 * w * h, 0.f);
            const int range = variant == 1 ? 3 : 40;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    float* p = &pay[5 * ((size_t)y * w + x)];
                    p[0] = (float)(rng.uniform(0, 2 * range + 1) - range);
                    if (variant == 2) p[0] = std::fabs(p[0]);                       // only sources
                    if (variant == 3) p[0] = -std::fabs(p[0]);                      // only sinks
                    if (variant == 4 && rng.uniform(0, 4) == 0) p[0] = rng.uniform(0, 2) ? 1048576.f : -1048576.f;
                    const int dx[4] = {1, 0, -1, 1}, dy[4] = {0, 1, 1, 1};
                    for (int k = 0; k < 4; k++) {
                        const int xx = x + dx[k], yy = y + dy[k];
                        if (xx < 0 || xx >= w || yy >= h) continue;
                        p[1 + k] = variant == 5 ? 0.f : (float)rng.uniform(0, rng.uniform(0, 3) ? range / 2 + 1 : 1);
                    }
                }
            GridMaxFlow bk;
            GridPushRelabel pr;
            bk.reset_for_load(w, h); pr.reset_for_load(w, h);
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) { bk.load_node(x, y, &pay[5 * ((size_t)y * w + x)]); pr.load_node(x, y, &pay[5 * ((size_t)y * w + x)]); }
            const double fb = bk.maxflow(), fp = pr.maxflow();
            int diff = 0;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) diff += (bk.what_segment(x, y) == GridMaxFlow::SOURCE) != (pr.what_segment(x, y) == GridPushRelabel::SOURCE);
            if (fb != fp || diff) { printf("FAIL push-relabel vs BK %dx%d variant %d: flow %.1f vs %.1f, %d segment differences\n", w, h, variant, fp, fb, diff); fail = 1; }
            // push-relabel with the band-parallel first phase
            for (int nb : {2, 4, 7}) {
                if (h < 8 * nb) continue;
                GridPushRelabel pb;
                pb.reset_for_load(w, h);
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) pb.load_node(x, y, &pay[5 * ((size_t)y * w + x)]);
                const double fpb = pb.maxflow(nb);
                int db = 0;
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) db += (bk.what_segment(x, y) == GridMaxFlow::SOURCE) != (pb.what_segment(x, y) == GridPushRelabel::SOURCE);
                banded_pr++;
                if (fpb != fb || db) { printf("FAIL banded push-relabel (%d bands) %dx%d variant %d: flow %.1f vs %.1f, %d segment differences\n", nb, w, h, variant, fpb, fb, db); fail = 1; }
            }
            // the path the drivers take (expansionMovePrebuilt): local pre-push while loading, search from the source side only, segments by
            // residual reachability of the sink -- one band, and row bands loaded / pre-pushed / classified separately
            for (int nb : {1, 3, 5}) {
                if (nb > 1 && h < 8 * nb) continue;
                GridMaxFlow lz;
                lz.reset_for_load(w, h);
                double routed = 0;
                for (int b2 = 0; b2 < nb; b2++) routed += lz.load_rows_prepushed(pay.data(), (int)((long long)h * b2 / nb), (int)((long long)h * (b2 + 1) / nb));
                lz.add_base_flow(routed);
                const double fl = lz.maxflow(nb);
                int dl = 0;
                std::vector<uint8_t> row((size_t)w);
                for (int y = 0; y < h; y++) {
                    lz.segment_row(y, row.data());
                    for (int x = 0; x < w; x++) {
                        dl += (bk.what_segment(x, y) == GridMaxFlow::SOURCE) != (lz.what_segment(x, y) == GridMaxFlow::SOURCE);
                        dl += (row[x] != 0) != (bk.what_segment(x, y) == GridMaxFlow::SOURCE);
                    }
                }
                prepushed++;
                if (fl != fb || dl) { printf("FAIL pre-push path (%d bands) %dx%d variant %d: flow %.1f vs %.1f, %d segment differences\n", nb, w, h, variant, fl, fb, dl); fail = 1; }
            }
            // the hybrid of expansionMovePrebuilt: BK until a (here: tiny) work budget runs out, push-relabel on the residual graph
            for (double budget : {0.25, 1.5, 4.0}) {
                GridMaxFlow part;
                part.reset_for_load(w, h);
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) part.load_node(x, y, &pay[5 * ((size_t)y * w + x)]);
                // (larger grids: with the band-parallel first phase, whose bands have their own share of the budget)
                const double f1 = part.maxflow(h >= 48 ? 4 : 1, budget);
                if (!part.exhausted()) { exhausted_none++; if (f1 != fb) { printf("FAIL budgeted BK finished with another flow\n"); fail = 1; } continue; }
                GridPushRelabel rest;
                rest.reset_for_load(w, h);
                float rc8[8], tr;
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) { part.residual(x, y, rc8, &tr); rest.load_residual(x, y, rc8, tr); }
                rest.set_base_flow(f1);
                const double f2 = rest.maxflow();
                int d2 = 0;
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) d2 += (bk.what_segment(x, y) == GridMaxFlow::SOURCE) != (rest.what_segment(x, y) == GridPushRelabel::SOURCE);
                hybrids++;
                if (f2 != fb || d2) { printf("FAIL BK(%.2f ops/node) + push-relabel %dx%d variant %d: flow %.1f vs %.1f, %d segment differences\n", budget, w, h, variant, f2, fb, d2); fail = 1; }
            }
        }
    // through the dispatcher the drivers call: forced to either solver, same mask
    {
        const int w = 70, h = 50;
        std::vector<float> pay((size_t)5 * w * h, 0.f);
        for (size_t i = 0; i < pay.size(); i++) pay[i] = (float)rng.uniform(0, 9) - ((i % 5 == 0) ? 4.f : 0.f);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { float* p = &pay[5 * ((size_t)y * w + x)]; if (x + 1 >= w) p[1] = p[4] = 0; if (y + 1 >= h) p[2] = p[3] = p[4] = 0; if (x == 0) p[3] = 0; }
        GridPushRelabel pr; pr.reset_for_load(w, h);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) pr.load_node(x, y, &pay[5 * ((size_t)y * w + x)]);
        const double fp = pr.maxflow();
        std::vector<uint8_t> mask;
        const double fb = expansionMovePrebuilt(pay.data(), 0.0, Rect(0, 0, w, h), mask);
        int diff = 0;
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) diff += (mask[(size_t)y * w + x] != 0) != (pr.what_segment(x, y) == GridPushRelabel::SOURCE);
        if (fb != fp || diff) { printf("FAIL push-relabel vs expansionMovePrebuilt: flow %.1f vs %.1f, %d differences\n", fp, fb, diff); fail = 1; }
    }
    printf("push-relabel vs Boykov-Kolmogorov: %d random grids %s; %d of them also cut as BK-with-a-budget + push-relabel on the residual graph, %d runs of the pre-push path, %d of push-relabel with a band-parallel first phase\n", trials, fail ? "FAILED" : "identical", hybrids, prepushed, banded_pr);
    return fail;
}

// band-parallel first phase vs the plain search on larger random grids: identical flow and identical segments
static int banded_vs_plain()
{
    RNG rng(77);
    int fail = 0;
    for (int trial = 0; trial < 12; trial++) {
        const int w = rng.uniform(20, 90), h = rng.uniform(64, 200), bands = rng.uniform(2, 9);
        GridMaxFlow a(w, h), b(w, h);
        const int range = trial % 2 ? 6 : 100;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                // smooth-ish terminals so that flow has to travel (and cross bands), plus noise
                const float bias = (float)((x * 7 + y * 13) % 23) - 11.0f;
                const float s = (float)rng.uniform(0, range) + (bias > 0 ? bias : 0), t = (float)rng.uniform(0, range) + (bias < 0 ? -bias : 0);
                a.add_tweights(x, y, s, t); b.add_tweights(x, y, s, t);
                const int dx[4] = {1, 0, -1, 1}, dy[4] = {0, 1, 1, 1}, dir[4] = {GridMaxFlow::E, GridMaxFlow::S, GridMaxFlow::SW, GridMaxFlow::SE};
                for (int k = 0; k < 4; k++) {
                    const int xx = x + dx[k], yy = y + dy[k];
                    if (xx < 0 || xx >= w || yy >= h) continue;
                    const float c = (float)rng.uniform(0, range), r = (float)rng.uniform(0, trial % 3 ? 1 : range);
                    a.add_edge(x, y, dir[k], c, r); b.add_edge(x, y, dir[k], c, r);
                }
            }
        const double fa = a.maxflow(1), fb = b.maxflow(bands);
        int diff = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) diff += (int)a.what_segment(x, y) != (int)b.what_segment(x, y);
        if (fa != fb || diff) { printf("FAIL banded max-flow trial %d (%dx%d, %d bands): flow %.1f vs %.1f, %d segment differences\n", trial, w, h, bands, fa, fb, diff); fail = 1; }
    }
    printf("band-parallel max-flow vs plain: 12 random grids %s\n", fail ? "FAILED" : "identical");
    // Float (non-integer) capacities as the expansion moves produce them: roundings make the flow VALUE depend on the augmentation
    // order in the last digits, and near-ties of the cut are possible in principle.  The minimum cut must agree up to such ties:
    // flows within 1e-6 relative and the two labellings have the same cut cost; they are reported (not required) to be identical.
    int ident = 0, trials = 0;
    for (int trial = 0; trial < 10; trial++) {
        const int w = rng.uniform(30, 80), h = rng.uniform(64, 160), bands = rng.uniform(2, 9);
        GridMaxFlow a(w, h), b(w, h);
        std::vector<float> ts((size_t)w * h), tt((size_t)w * h), cap((size_t)w * h * 4, 0.f);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const float s = rng.uniform(0.f, 0.5f), t = rng.uniform(0.f, 0.5f);
                ts[(size_t)y * w + x] = s; tt[(size_t)y * w + x] = t;
                a.add_tweights(x, y, s, t); b.add_tweights(x, y, s, t);
                const int dx[4] = {1, 0, -1, 1}, dy[4] = {0, 1, 1, 1}, dir[4] = {GridMaxFlow::E, GridMaxFlow::S, GridMaxFlow::SW, GridMaxFlow::SE};
                for (int k = 0; k < 4; k++) {
                    const int xx = x + dx[k], yy = y + dy[k];
                    if (xx < 0 || xx >= w || yy >= h) continue;
                    const float c = rng.uniform(0.f, 0.3f);
                    cap[((size_t)y * w + x) * 4 + k] = c;
                    a.add_edge(x, y, dir[k], c, 0.f); b.add_edge(x, y, dir[k], c, 0.f);
                }
            }
        const double fa = a.maxflow(1), fb = b.maxflow(bands);
        auto cut_cost = [&](GridMaxFlow& g) {
            double c = 0;
            const int dx[4] = {1, 0, -1, 1}, dy[4] = {0, 1, 1, 1};
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    const bool src = g.what_segment(x, y) == GridMaxFlow::SOURCE;
                    c += src ? tt[(size_t)y * w + x] : ts[(size_t)y * w + x];
                    for (int k = 0; k < 4; k++) {
                        const int xx = x + dx[k], yy = y + dy[k];
                        if (xx < 0 || xx >= w || yy >= h) continue;
                        if (src && g.what_segment(xx, yy) != GridMaxFlow::SOURCE) c += cap[((size_t)y * w + x) * 4 + k];
                    }
                }
            return c;
        };
        const double ca = cut_cost(a), cb = cut_cost(b);
        int diff = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) diff += (int)a.what_segment(x, y) != (int)b.what_segment(x, y);
        trials++; ident += diff == 0;
        if (std::fabs(fa - fb) > 1e-6 * fa || std::fabs(ca - cb) > 1e-6 * ca || std::fabs(ca - fa) > 1e-5 * fa) {
            printf("FAIL banded max-flow (float capacities) trial %d: flows %.9g %.9g, cut costs %.9g %.9g\n", trial, fa, fb, ca, cb);
            fail = 1;
        }
    }
    printf("band-parallel max-flow vs plain, float capacities: %d grids, equal minimum cut cost, identical labelling in %d\n", trials, ident);
    return fail;
}

// BandPool: every task of every run executes exactly once, also when several caller threads (each with its own persistent
// team) run teams at the same time and when the team size changes between runs
static int band_pool_check()
{
    int fail = 0;
    auto caller = [&](int seed) {
        RNG rng(seed);
        for (int rep = 0; rep < 400; rep++) {
            const int n = 1 + (int)(rng.uniform(0.f, 1.f) * 12);
            std::vector<std::atomic<int>> hits(n);
            for (auto& h : hits) h.store(0);
            std::atomic<int> concurrent{0}, peak{0};
            BandPool::mine().run(n, [&](int b) {
                const int c = concurrent.fetch_add(1) + 1;
                int p = peak.load();
                while (c > p && !peak.compare_exchange_weak(p, c)) {}
                volatile int spin = 0;
                for (int k = 0; k < 200; k++) spin = spin + k;
                hits[b].fetch_add(1);
                concurrent.fetch_sub(1);
            });
            for (int b = 0; b < n; b++) if (hits[b].load() != 1) fail = 1;
            if (peak.load() > n) fail = 1;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < 3; t++) th.emplace_back(caller, 100 + t);
    caller(99);
    for (auto& t : th) t.join();
    printf("band pool: 4 callers x 400 runs %s\n", fail ? "FAILED" : "every task exactly once");
    return fail;
}

int main(int argc, char** argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 120, H = argc > 2 ? atoi(argv[2]) : 80, D = argc > 3 ? atoi(argv[3]) : 24;
    const int iters = argc > 4 ? atoi(argv[4]) : 2;
    const bool check = argc > 5 ? atoi(argv[5]) != 0 : true;          // 0: skip the per-move energy recomputation (timing runs)
    Scene s = make_scene(W, H, D);
    Parameters param(1.0f, 20, "GF", 1e-4f);
    param.th_col = 0.5f;
    const float maxd = (float)D - 1;
    int fail = 0;
    {
        Scene tiny = make_scene(16, 12, 8);
        fail |= brute_force(tiny, param, 7.0f);
        fail |= grid_vs_generic();
        fail |= banded_vs_plain();
        fail |= push_relabel_vs_bk();
        fail |= band_pool_check();
    }
    PMStereo st(W, H, param, maxd);
    st.setSeed(11);
    st.setStereoEnergy(std::make_unique<PointwiseTestEnergy>(s, param, maxd));
    st.addLayer(std::max(2, int(W * 0.04)), {{LES_HIP_PROPOSE_EXPANSION, 1}, {LES_HIP_PROPOSE_RANDOM, 7}});
    st.addLayer(std::max(4, int(W * 0.12)), {{LES_HIP_PROPOSE_EXPANSION, 2}});
    st.checkFlowEnergy = check;
    st.initCurrentFast(0);
    double e_prev = st.totalEnergy(0);
    printf("init      E=%.2f  bad1.0=%.2f%%\n", e_prev, bad_pixels(st.computeDisparities(0), s, 1.0f));
    for (int it = 0; it < iters; it++) {
        for (size_t li = 0; li < st.layers().layers.size(); li++) st.localExpansionMovesForLayer((int)li, 0, it, true);
        const double e = st.totalEnergy(0);
        printf("gc iter %d E=%.2f  bad1.0=%.2f%%  moves=%ld  max |flow-E|/E = %.2e\n", it + 1, e, bad_pixels(st.computeDisparities(0), s, 1.0f),
               st.numMoves, st.maxFlowEnergyGap);
        if (e > e_prev * (1 + 1e-6)) { printf("FAIL: energy increased\n"); fail = 1; }
        e_prev = e;
    }
    if (st.maxFlowEnergyGap > 1e-5) { printf("FAIL: flow != energy\n"); fail = 1; }
    if (bad_pixels(st.computeDisparities(0), s, 1.0f) > 20.0) { printf("FAIL: did not converge\n"); fail = 1; }
    printf(fail ? "gc_selfcheck: FAILED\n" : "gc_selfcheck: OK\n");
    return fail;
}
