// les_gc.cpp -- C ABI (include/localexp_host.h) over the host graph-cut fusion (ExpansionMove.h / MaxFlow.h).
#include "../../include/localexp_host.h"

#include <algorithm>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <thread>
#include <functional>
#include <omp.h>

#include <cstdarg>
#include <cstdio>
#include <memory>
#include <string>

#include "ExpansionMove.h"
#include "ResidualCut.h"

using namespace les_host;

namespace {
thread_local std::string g_err;
int fail(const char* fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

// Team size of a lock-step.  The cuts of one lock-step are small (tens of microseconds for a layer-0 cell), so a team of
// every hardware thread costs more in fork/join, wake-ups and cache traffic than it buys: measured on a 2 x 64-core
// EPYC 9575F (1436 x 992): with the default (spinning) OpenMP wait policy 16 threads 0.7 s / iteration, 64 threads 1.2 s,
// 256 threads 1.8 s; with OMP_WAIT_POLICY=passive 24 threads 0.54 s and flat beyond.
int defaultThreads(int requested, int n)
{
    if (requested <= 0) requested = std::min(24, omp_get_max_threads());
    return std::max(1, std::min(requested, n));
}

// the pairwise half of StereoEnergy; the unary operator lives on the GPU and is never called through this object
class PairwiseEnergy : public StereoEnergy {
public:
    PairwiseEnergy(int W, int H, Parameters p) : StereoEnergy(W, H, std::move(p), 0.f, 0.f) {}
    void ComputeUnaryPotentialWithoutCheck(const Rect&, const Rect&, float*, int, const Plane&, Reusable&, int) const override {}
    void ComputeUnaryPotential(const Rect&, const Rect&, float*, int, const Plane&, Reusable&, int) const override {}
};
}  // namespace

struct les_gc_ctx {
    int H, W;
    std::unique_ptr<PairwiseEnergy> E;
    LabelMap labels[2];
    CostMap costs[2];
};

extern "C" {

const char* les_gc_last_error(void) { return g_err.c_str(); }

int les_gc_create(les_gc_ctx** out, int H, int W, const uint8_t* imL, const uint8_t* imR, float lambda, float th_smooth, float omega, float epsilon)
{
    if (!out || H <= 0 || W <= 0 || (!imL && !imR)) return fail("les_gc_create: bad argument");
    Parameters p;
    p.lambda = lambda; p.th_smooth = th_smooth; p.omega = omega; p.epsilon = epsilon;
    les_gc_ctx* c = new les_gc_ctx();
    c->H = H; c->W = W;
    c->E = std::make_unique<PairwiseEnergy>(W, H, p);
    c->E->setImages(imL, imR);
    for (int m = 0; m < 2; m++) {
        c->labels[m] = LabelMap(H, W);
        c->costs[m] = CostMap(H, W, 0.f);
    }
    *out = c;
    return 0;
}

void les_gc_destroy(les_gc_ctx* c) { delete c; }

float* les_gc_labels(les_gc_ctx* c, int mode) { return (c && mode >= 0 && mode < 2) ? reinterpret_cast<float*>(c->labels[mode].data.data()) : nullptr; }
float* les_gc_costs(les_gc_ctx* c, int mode) { return (c && mode >= 0 && mode < 2) ? c->costs[mode].data.data() : nullptr; }

int les_gc_expansion_moves(les_gc_ctx* c, int mode, int n, const les_hip_rect* regions, const les_hip_plane* planes, const float* proposal_cost,
                           int nthreads, int check, double* max_gap)
{
    if (!c || mode < 0 || mode > 1 || n < 0 || (n > 0 && (!regions || !planes || !proposal_cost))) return fail("les_gc_expansion_moves: bad argument");
    if (!c->E->hasImages(mode)) return fail("les_gc_expansion_moves: view %d has no image", mode);
    for (int i = 0; i < n; i++) {
        const les_hip_rect& r = regions[i];
        if (r.w < 0 || r.h < 0 || r.x < 0 || r.y < 0 || r.x + r.w > c->W || r.y + r.h > c->H) return fail("les_gc_expansion_moves: region %d outside the image", i);
    }
    nthreads = defaultThreads(nthreads, n);
    LabelMap& lab = c->labels[mode];
    CostMap& cur = c->costs[mode];
    const CostView prop(proposal_cost, c->W);
    double gap = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(max : gap)
    for (int i = 0; i < n; i++) {
        const Rect region(regions[i].x, regions[i].y, regions[i].w, regions[i].h);
        if (region.width == 0 || region.height == 0) continue;
        const Plane label(planes[i].a, planes[i].b, planes[i].c, planes[i].v);
        std::vector<uint8_t> mask;
        const double flow = expansionMove(*c->E, lab, cur, prop, label, region, mask, mode);
        if (check) {
            const double e = fusedEnergy(*c->E, lab, cur, prop, label, region, mask, mode);
            gap = std::max(gap, std::fabs(flow - e) / std::max(1.0, std::fabs(e)));
        }
        for (int y = 0; y < region.height; y++)                               // LES/FastGCStereo.h:61-62
            for (int x = 0; x < region.width; x++)
                if (mask[(size_t)y * region.width + x]) {
                    cur.at(region.y + y, region.x + x) = prop.at(region.y + y, region.x + x);
                    lab.at(region.y + y, region.x + x) = label;
                }
    }
    if (max_gap) *max_gap = gap;
    return 0;
}

int les_gc_expansion_moves_prebuilt(les_gc_ctx* c, int mode, int n, const les_hip_rect* regions, const les_hip_plane* planes, const float* proposal_cost,
                                    const float* payload, const long long* offsets, const double* flow0, int nthreads, double* flows)
{
    if (!c || mode < 0 || mode > 1 || n < 0 || (n > 0 && (!regions || !planes || !proposal_cost || !payload || !offsets)))
        return fail("les_gc_expansion_moves_prebuilt: bad argument");
    for (int i = 0; i < n; i++) {
        const les_hip_rect& r = regions[i];
        if (r.w < 0 || r.h < 0 || r.x < 0 || r.y < 0 || r.x + r.w > c->W || r.y + r.h > c->H) return fail("les_gc_expansion_moves_prebuilt: region %d outside the image", i);
    }
    nthreads = defaultThreads(nthreads, n);
    LabelMap& lab = c->labels[mode];
    CostMap& cur = c->costs[mode];
    const CostView prop(proposal_cost, c->W);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int i = 0; i < n; i++) {
        const Rect region(regions[i].x, regions[i].y, regions[i].w, regions[i].h);
        if (region.width == 0 || region.height == 0) continue;
        const Plane label(planes[i].a, planes[i].b, planes[i].c, planes[i].v);
        std::vector<uint8_t> mask;
        const double flow = expansionMovePrebuilt(payload + 5 * offsets[i], flow0 ? flow0[i] : 0.0, region, mask);
        if (flows) flows[i] = flow;
        for (int y = 0; y < region.height; y++)
            for (int x = 0; x < region.width; x++)
                if (mask[(size_t)y * region.width + x]) {
                    cur.at(region.y + y, region.x + x) = prop.at(region.y + y, region.x + x);
                    lab.at(region.y + y, region.x + x) = label;
                }
    }
    return 0;
}

int les_gc_solve_prebuilt(int n, const les_hip_rect* regions, const float* payload, const long long* offsets, int nthreads, unsigned char* masks,
                          double* flows)
{
    if (n < 0 || (n > 0 && (!regions || !payload || !offsets || !masks))) return fail("les_gc_solve_prebuilt: bad argument");
    // the payload / mask buffers are indexed through offsets[]: they must describe disjoint node ranges in call order
    for (int i = 0; i < n; i++) {
        if (regions[i].w < 0 || regions[i].h < 0 || offsets[i] < 0) return fail("les_gc_solve_prebuilt: negative region size or offset (call %d)", i);
        if (i > 0 && offsets[i] < offsets[i - 1] + (long long)regions[i - 1].w * regions[i - 1].h)
            return fail("les_gc_solve_prebuilt: offsets[%d] overlaps the nodes of call %d", i, i - 1);
    }
    nthreads = defaultThreads(nthreads, n);
    tuneBandSpin(n, nthreads, [&](int i) { return Rect(0, 0, regions[i].w, regions[i].h); });
    // tooling (LES_GC_TRACE=file): wall-clock of every call as seen from inside, one line per call
    static FILE* trace = [] { const char* p = getenv("LES_GC_TRACE"); return p ? fopen(p, "a") : (FILE*)nullptr; }();
    const auto t0 = std::chrono::steady_clock::now();
    // Longest cell first.  A lock-step lasts as long as its slowest cell, and one hard cell (10 .. 50 x the median) that the
    // dynamic schedule hands out last adds its whole length to the lock-step.  The drivers call with the same payload buffer for the same
    // cells of a disjoint set, proposal after proposal, and a cell that was hard is usually hard again: the time each cell took is
    // remembered under (staging buffer, cell rect) and the next call starts the cells in decreasing order of it.  (Only the order in
    // which independent cells are started changes, never a result.)
    static std::mutex hist_mu;
    static std::unordered_map<unsigned long long, float> hist;
    // (the drivers stage every lock-step of a view in one buffer: the cell is identified by buffer, position and size)
    auto key = [&](int i) {
        unsigned long long k = (unsigned long long)(uintptr_t)payload;
        for (unsigned long long v : {(unsigned long long)(unsigned)regions[i].x, (unsigned long long)(unsigned)regions[i].y, (unsigned long long)(unsigned)regions[i].w, (unsigned long long)(unsigned)regions[i].h})
            k = (k ^ v) * 0x9E3779B97F4A7C15ull + (k >> 29);
        return k;
    };
    std::vector<int> order((size_t)n);
    std::vector<float> cost((size_t)n, 0.0f);
    {
        std::lock_guard<std::mutex> lk(hist_mu);
        for (int i = 0; i < n; i++) {
            order[(size_t)i] = i;
            auto it = hist.find(key(i));
            cost[(size_t)i] = it != hist.end() ? it->second : 1e-9f * (float)regions[i].w * (float)regions[i].h;
        }
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int q = 0; q < n; q++) {
        const int i = order[(size_t)q];
        const Rect region(0, 0, regions[i].w, regions[i].h);
        if (region.width <= 0 || region.height <= 0) continue;
        const auto c0 = std::chrono::steady_clock::now();
        const double flow = expansionMovePrebuilt(payload + 5 * offsets[i], 0.0, region, masks + offsets[i], bandsFor(region, n));
        cost[(size_t)i] = std::chrono::duration<float>(std::chrono::steady_clock::now() - c0).count();
        if (flows) flows[i] = flow;
    }
    {
        std::lock_guard<std::mutex> lk(hist_mu);
        if (hist.size() > (1u << 20)) hist.clear();                 // (buffers come and go with the runs of a long-lived process)
        for (int i = 0; i < n; i++) hist[key(i)] = cost[(size_t)i];
    }
    if (trace) {
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fprintf(trace, "%zx %d %d %d %.6f\n", std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xffff, n, n > 0 ? regions[0].w : 0, nthreads, dt);
        fflush(trace);
    }
    return 0;
}

int les_gc_solve_residual(int n, const les_hip_rect* regions, const float* rc8, const float* ex, const long long* offsets, int nthreads, int solver,
                          unsigned char* masks, double* flows)
{
    if (n < 0 || (n > 0 && (!regions || !rc8 || !ex || !offsets || !masks))) return fail("les_gc_solve_residual: bad argument");
    for (int i = 0; i < n; i++)
        if (regions[i].w < 0 || regions[i].h < 0 || offsets[i] < 0) return fail("les_gc_solve_residual: negative region size or offset (call %d)", i);
    nthreads = defaultThreads(nthreads, n);
    tuneBandSpin(n, nthreads, [&](int i) { return Rect(0, 0, regions[i].w, regions[i].h); });
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int i = 0; i < n; i++) {
        const int w = regions[i].w, h = regions[i].h;
        if (w <= 0 || h <= 0) continue;
        const double flow = finishResidualCut(rc8 + 8 * offsets[i], ex + offsets[i], w, h, masks + offsets[i], residualBands(w, h), solver);
        if (flows) flows[i] = flow;
    }
    return 0;
}

int les_gc_build_graphs(les_gc_ctx* c, int mode, int n, const les_hip_rect* regions, const les_hip_plane* planes, const float* proposal_cost,
                        const long long* offsets, float* payload, double* flow0)
{
    if (!c || mode < 0 || mode > 1 || n < 0 || (n > 0 && (!regions || !planes || !proposal_cost || !payload || !offsets))) return fail("les_gc_build_graphs: bad argument");
    if (!c->E->hasImages(mode)) return fail("les_gc_build_graphs: view %d has no image", mode);
    for (int i = 0; i < n; i++) {
        const les_hip_rect& r = regions[i];
        if (r.w < 0 || r.h < 0 || (r.w > 0 && r.h > 0 && (r.x < 0 || r.y < 0 || r.x + r.w > c->W || r.y + r.h > c->H)))
            return fail("les_gc_build_graphs: region %d outside the image", i);
        if (offsets[i] < 0 || (i > 0 && offsets[i] < offsets[i - 1] + (long long)regions[i - 1].w * regions[i - 1].h))
            return fail("les_gc_build_graphs: offsets[%d] overlaps the nodes of call %d", i, i - 1);
    }
    const CostView prop(proposal_cost, c->W);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        const Rect region(regions[i].x, regions[i].y, regions[i].w, regions[i].h);
        if (region.width <= 0 || region.height <= 0) continue;
        GridMaxFlow graph(region.width, region.height);
        buildExpansionGraph(graph, *c->E, c->labels[mode], c->costs[mode], prop, Plane(planes[i].a, planes[i].b, planes[i].c, planes[i].v), region, mode);
        for (int y = 0; y < region.height; y++)
            for (int x = 0; x < region.width; x++) graph.store_node(x, y, payload + 5 * (offsets[i] + (long long)y * region.width + x));
        if (flow0) flow0[i] = graph.base_flow();
    }
    return 0;
}

double les_gc_smoothness_cost(les_gc_ctx* c, int mode) { return (c && mode >= 0 && mode < 2 && c->E->hasImages(mode)) ? c->E->computeSmoothnessCost(c->labels[mode], mode) : 0.0; }

double les_gc_data_cost(les_gc_ctx* c, int mode)
{
    double s = 0;
    if (c && mode >= 0 && mode < 2)
        for (float v : c->costs[mode].data) s += v;
    return s;
}

}  // extern "C"
