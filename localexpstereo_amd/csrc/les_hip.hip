// les_hip.hip -- C ABI (include/localexp_hip.h) + host-side launch logic of the MI355X matching-cost path.
//
// Build (see __graft_entry__.build / localexpstereo_amd/build.py):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared les_hip.hip -o liblocalexp_hip.so
// There is no CPU fallback in this library: every entry point needs a HIP device.
// (tools/hipsim compiles this same file against a CPU fiber simulator for logic tests only.)
#include "../../include/localexp_hip.h"
#include "les_kernels.h"
#include "les_march.h"
#include "les_propose.h"
#include "les_post.h"
#include "les_pairwise.h"
#include "les_maxflow.h"
#include "les_maxflow_tiled.h"

#include "../host/ResidualCut.h"      // the host cores' finisher of the tiled max-flow (plain C++: search trees / push-relabel on a residual graph)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <mutex>
#if !defined(LES_SIM)
#include <dlfcn.h>
#endif
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// One line on stderr, once per context and reason, when work that the march kernel could have served goes to the 2.2x slower strip
// kernel (a perf cliff nobody would otherwise see; les_hip_batch_kernel_kind reports the same fact per batch).  LES_HIP_QUIET=1 silences it.
enum FallbackReason { FB_RADIUS = 0, FB_NONFINITE, FB_RANGE, FB_THRESHOLD, FB_IMAGE_SIZE, FB_GEOMETRY, FB_PATCHES, FB_GUIDE, FB_COUNT };
void note_fallback(std::atomic<unsigned>& seen, FallbackReason r, const char* fmt, ...)
{
    // (contexts are shared by concurrent host threads -- the re-entrant per-call operator, the two views: fetch_or decides who reports)
    if (seen.fetch_or(1u << r, std::memory_order_relaxed) & (1u << r)) return;
    static const bool quiet = [] { const char* e = getenv("LES_HIP_QUIET"); return e && atoi(e) != 0; }();
    if (quiet) return;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "localexp_hip: strip kernel instead of the march kernel: %s (LES_HIP_QUIET=1 silences this note)\n", buf);
}

#define HIPCHECK(expr)                                                                               \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(LES_HIP_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- kernel configurations compiled into this build: guided-filter radius -> strip geometry
typedef void (*StripKernel)(les::Geom, les::View, const les::Job*, const float4*, float*, int, int);
struct StripEntry { int R; int variant; int TW; int NT; StripKernel fn; };

// (radius, stage-1 columns WA, rows per block BY, H-phase segments SEG, min waves/SIMD MW); variant 0 is
// the default of a radius, the others are selectable with LES_HIP_VARIANT for A/B measurements.
#define LES_STRIP_ENTRY(R_, V_, WA_, BY_, SEG_, MW_) \
    { R_, V_, les::StripCfg<R_, WA_, BY_, SEG_>::TW, les::StripCfg<R_, WA_, BY_, SEG_>::NT, les::les_strip_kernel<R_, WA_, BY_, SEG_, MW_> }
const StripEntry kStrips[] = {
    // Defaults per radius.  Measured on MI355X (R = 10, 1500x1000x256): occupancy without register spills wins --
    // (WA 64, BY 21 = ring length, SEG 3, 3 waves/SIMD, 0 B scratch) 7.0 ms vs 9.2 ms for (128,16,8,2 waves) and
    // 12+ ms for anything that spills.
    LES_STRIP_ENTRY(1, 0, 64, 16, 4, 2), LES_STRIP_ENTRY(2, 0, 64, 16, 4, 2), LES_STRIP_ENTRY(3, 0, 64, 16, 4, 2),
    LES_STRIP_ENTRY(4, 0, 64, 16, 4, 2), LES_STRIP_ENTRY(5, 0, 64, 16, 4, 2), LES_STRIP_ENTRY(6, 0, 64, 16, 4, 2),
    LES_STRIP_ENTRY(7, 0, 64, 16, 4, 2), LES_STRIP_ENTRY(8, 0, 64, 16, 4, 2), LES_STRIP_ENTRY(9, 0, 64, 16, 4, 2),
    LES_STRIP_ENTRY(10, 0, 64, 21, 3, 3), LES_STRIP_ENTRY(12, 0, 64, 16, 4, 2), LES_STRIP_ENTRY(15, 0, 96, 16, 6, 2),
    // A/B variants for radius 10 (LES_HIP_VARIANT=n)
    // Kept: the wide strip and the default geometry at 2 waves/SIMD.  Measured earlier in the round (ms per 1500x1000x256 pass,
    // default then 7.14): (64,21,3) at 2 waves/SIMD 9.57 | (128,21,SEG 3/4/6) at 2 waves 10.2 / 11.2 / 8.8 | (128,16,8) 9.6 |
    // (96,21,SEG 3/4) at 2 waves (6-wave workgroups) 15.1 / 12.4 | (80,21,3) 14.4 | anything that spills 11-15:
    // occupancy beats the lower instruction count of wide strips
    LES_STRIP_ENTRY(10, 8, 128, 21, 6, 2), LES_STRIP_ENTRY(10, 11, 64, 21, 3, 2),
};
// image-based matching cost (les_hip_create_naive): one conservative configuration per radius
#define LES_NAIVE_ENTRY(R_, WA_, BY_, SEG_, MW_) \
    { R_, 0, les::StripCfg<R_, WA_, BY_, SEG_>::TW, les::StripCfg<R_, WA_, BY_, SEG_>::NT, les::les_strip_kernel<R_, WA_, BY_, SEG_, MW_, 1> }
const StripEntry kNaiveStrips[] = {
    LES_NAIVE_ENTRY(1, 64, 16, 4, 2), LES_NAIVE_ENTRY(2, 64, 16, 4, 2), LES_NAIVE_ENTRY(3, 64, 16, 4, 2), LES_NAIVE_ENTRY(4, 64, 16, 4, 2),
    LES_NAIVE_ENTRY(5, 64, 16, 4, 2), LES_NAIVE_ENTRY(6, 64, 16, 4, 2), LES_NAIVE_ENTRY(7, 64, 16, 4, 2), LES_NAIVE_ENTRY(8, 64, 16, 4, 2),
    LES_NAIVE_ENTRY(9, 64, 16, 4, 2), LES_NAIVE_ENTRY(10, 64, 16, 4, 2), LES_NAIVE_ENTRY(12, 64, 16, 4, 2), LES_NAIVE_ENTRY(15, 96, 16, 6, 2),
};
const StripEntry* find_naive_strip(int R)
{
    for (const auto& e : kNaiveStrips)
        if (e.R == R) return &e;
    return nullptr;
}
const StripEntry* find_strip(int R)
{
    int variant = 0;
    if (const char* v = getenv("LES_HIP_VARIANT")) variant = atoi(v);
    const StripEntry* def = nullptr;
    for (const auto& e : kStrips) {
        if (e.R != R) continue;
        if (e.variant == variant) return &e;
        if (e.variant == 0) def = &e;
    }
    return def;
}

// ---- the fixed-point march kernel (les_march.h): (radius, columns per job slot, job slots per workgroup, rows per block)
typedef void (*MarchKernel)(les::Geom, les::MarchView, const les::Job*, const float4*, float*, int, int);
struct MarchEntry { int R; int TW; int NJ; int NT; int BY; MarchKernel fn; };
#define LES_MARCH_ENTRY(R_, WGC_, NJ_, BY_) \
    { R_, les::MarchCfg<R_, WGC_, NJ_, BY_>::TW, NJ_, les::MarchCfg<R_, WGC_, NJ_, BY_>::NT, BY_, les::les_march_kernel<R_, WGC_, NJ_, BY_> }
// two geometries per radius: wide jobs (WGC - 4R output columns: whole-image hypothesis slabs, layer-1/2 cells) and two narrow jobs
// per workgroup (128 - 4R output columns each: layer-0 cells); both run 12 waves per workgroup.
// Radii 4 .. 10 (windR 8 .. 21; the reference's option -filterRadious, LES/main.cpp:48,285,349; its default 20 -> radius 10).  The rings
// hold RS = 3 x BY >= 2R + 1 rows (three ticks per unrolled loop iteration, three stage-2 buffers; the row that leaves a window is read
// from slot (s + RS - (2R + 1)) mod RS), with BY <= 8 rows per prefix pass: the smallest such block height per radius.  Radius 11 would
// need BY = 8 and 176 KB of LDS, radius 12 and beyond more than 24 ring rows: they, and the radii below 4 (windows of at most 7 x 7, not worth
// two more instantiations each), stay on the strip kernel.
const MarchEntry kMarch[] = {
    LES_MARCH_ENTRY(10, 256, 1, 7),
#if defined(LES_MARCH_LAB) && defined(LES_MARCH_NARROW_NJ1)
    LES_MARCH_ENTRY(10, 128, 1, 7),          // experiment: one narrow job per workgroup, two workgroups per CU (independent tick barriers)
#else
    LES_MARCH_ENTRY(10, 128, 2, 7),
#endif
    LES_MARCH_ENTRY(7, 256, 1, 5),
    LES_MARCH_ENTRY(7, 128, 2, 5),
#if !defined(LES_MARCH_FEW_RADII)
    LES_MARCH_ENTRY(4, 256, 1, 3),  LES_MARCH_ENTRY(4, 128, 2, 3),
    LES_MARCH_ENTRY(5, 256, 1, 4),  LES_MARCH_ENTRY(5, 128, 2, 4),
    LES_MARCH_ENTRY(6, 256, 1, 5),  LES_MARCH_ENTRY(6, 128, 2, 5),
    LES_MARCH_ENTRY(8, 256, 1, 6),  LES_MARCH_ENTRY(8, 128, 2, 6),
    LES_MARCH_ENTRY(9, 256, 1, 7),  LES_MARCH_ENTRY(9, 128, 2, 7),
#endif
};
// wide != 0: the entry with the widest jobs, else the one with the narrowest
const MarchEntry* find_march(int R, int wide = 1)
{
    if (const char* k = getenv("LES_HIP_KERNEL")) if (!strcmp(k, "strip")) return nullptr;     // A/B measurements: force the fp64 strip kernel
    const MarchEntry* best = nullptr;
    for (const auto& e : kMarch)
        if (e.R == R && (!best || (wide ? e.TW > best->TW : e.TW < best->TW))) best = &e;
    return best;
}

constexpr long long kRawPatchCapFloats = 1ll << 30;   // 4 GB of raw-cost patches per batch and view (image-based energy on the march kernel)
constexpr int kRansacMaxSam = 500;   // RansacProposer default MAX_SAM, LES/Proposer.h:265
constexpr int kRansacChunks = 5;
constexpr int kRansacChunkEnds[kRansacChunks] = {16, 64, 128, 256, kRansacMaxSam};   // the candidates are evaluated in chunks that end here (les_propose.h)

struct ViewData {
    float* vol = nullptr;
    bool own_vol = false;
    float4* stats = nullptr;
    uint32_t* ipk = nullptr;
    uint32_t* ipk10 = nullptr;           // the guide pixel as signed 10-bit fields (H1 operand format of the strip kernel)
    float4* feat = nullptr;              // NaiveStereoEnergy feature image (image-based matching cost)
    // march kernel (les_march.h): guide as signed bytes, statistics in its format, cost range of the volume
    uint32_t* ipk8 = nullptr;
    float* mstats = nullptr;
    float* vol_t = nullptr;              // the volume once more, tiled [H][ceil(W/8)][D][8]: the taps of steep planes (les_march.h, role A's KIND 5); null when not built
    bool march_ok = false;               // volume finite, range condition met, tables built
    unsigned dmax_bits = 0;              // largest diagonal entry of the guide's inverse covariance (float bits): fixes the scale of a, b
    les::MarchView mv = {};
};

}  // namespace

struct MtHost;
namespace { void mt_host_free(MtHost* m); }

struct les_hip_ctx {
    les_hip_params p;
    int R;
    const StripEntry* strip;
    const MarchEntry* march = nullptr;   // null: radius not instantiated (or LES_HIP_KERNEL=strip)
    int ncu = 256;                       // compute units of the device (job cutting of the march kernel)
    std::atomic<unsigned> fallback_seen{0};   // reasons already reported by note_fallback
    hipStream_t stream;
    les::Geom geom;
    ViewData v[2];
    float naive_alpha = 0;
    int naive = 0;                       // 1: raw cost from the feature images (les_hip_create_naive), no volume
    float th_color = 0, th_grad = 0;
    // scratch reused by the non-prepared entry points and by batch_run
    float4* d_planes = nullptr; size_t planes_cap = 0;
    float* d_map = nullptr;                         // H*W floats
    les::WtaJob* d_wta = nullptr; size_t wta_cap = 0;
    float4* d_wta_planes = nullptr; size_t wta_planes_cap = 0;
    // smoothness-coefficient table of the pairwise terms, cached per (omega, epsilon)
    float* d_pw_tab = nullptr; float pw_omega = -1.f, pw_epsilon = -1.f;
    std::mutex mu;                       // guards the lazily built tables when two host threads (the two views) share the context
    unsigned long long gen = 0;          // unique id of this context: thread-local bindings compare it, not the address (an address can be reused)
    std::vector<les_hip_scratch*> idle_scratch;  // hidden scratches whose owning thread has exited, ready for the next new thread
    bool maxflow_lds_ready = false;      // the per-device dynamic-LDS opt-in of les_maxflow_kernel has been made on this context's device
    bool maxflow_tiled_lds_ready = false;   // ... and of les_maxflow_tiled_kernel
    std::vector<les_hip_scratch*> own_scratch;   // scratch objects created behind les_hip_unary_one (one per calling thread), freed with the context
    std::vector<MtHost*> mt_idle; // host-mapped flag words + hand-over staging of the tiled max-flow: one per CONCURRENT caller, reused, freed with the context
};

struct les_hip_batch {
    int n = 0, njobs = 0, out_slabs = 0, R = 0;
    les::Job* d_jobs = nullptr;
    // the same calls cut for the march kernel (groups of NJ jobs); march_ok: every target keeps 2R distance from clip borders
    // that are not image borders, so the kernel's bound on |a| holds (les_march.h)
    les::Job* d_mjobs = nullptr;
    const MarchEntry* mentry = nullptr;  // the geometry the table was cut for
    int nmgroups = 0;
    bool march_ok = false;
    // image-based energy on the march kernel: the calls' filterRects with the offsets of their raw-cost patches, and one patch
    // buffer per view (allocated on the view's first run; two host threads may drive the two views of one batch)
    les::RawCall* d_rawcalls = nullptr;
    long long* d_raw_off = nullptr;
    long long raw_floats = 0;
    int raw_chunks = 1;
    mutable float* d_raw[2] = {nullptr, nullptr};
    std::vector<les_hip_rect> targets;
    int device = 0;
    // cell geometry for the proposers / WTA
    les::Rect4* d_units = nullptr;
    les::WtaJob* d_targets = nullptr;
    les::RansacScratch rs = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr};   // RANSAC proposer scratch
    int wta_chunks = 1;                  // blocks per target rect in the WTA kernel
    // expansion-graph payload layout (les_hip_batch_expansion_graph): node offset of every target, total node count
    std::vector<long long> graph_off;
    long long graph_nodes = 0;
    long long* d_graph_off = nullptr;
    double* d_flow0 = nullptr;           // n * wta_chunks partial sums
    // tiled device max-flow (les_maxflow_tiled.h): the cells cut into tiles, built on first use (two host threads -- the two views -- may share a batch)
    mutable std::mutex mt_mu;
    mutable les::MtTile* d_mt_tiles = nullptr;
    mutable int* d_mt_tiles_per_cell = nullptr;
    mutable int mt_ntiles = -1;          // -1: not built yet
};

// Caller-owned scratch of the one-call operator (the reference's `Reusable`, LES/StereoEnergy.h:616-623): its own stream, a
// compact device tile for the target rect, pinned host staging, and the job tables of the (filterRect, targetRect) pairs it has
// seen -- a cell visit calls the operator ~10 times with the same rects (LES/FastGCStereo.h:40-49).  Distinct scratch objects
// may be used concurrently from distinct host threads on one context; nothing is allocated once a rect pair is known.
struct les_hip_scratch {
    les_hip_ctx* c = nullptr;
    hipStream_t stream = nullptr;
    float* d_tile = nullptr; float* h_tile = nullptr; size_t tile_cap = 0;        // floats
    float4* d_plane = nullptr; float4* h_plane = nullptr;
    struct Entry { les_hip_rect f, t; int want_march; const void* march; int njobs, ngroups; les::Job* d_jobs; unsigned long long stamp; };
    // image-based energy on the march kernel: raw-cost patch of the call's filterRect and its one-entry call table
    float* d_raw = nullptr; size_t raw_cap = 0;
    les::RawCall* d_rawcall = nullptr; long long* d_raw_off = nullptr;
    les_hip_rect raw_f = {-1, -1, -1, -1};
    std::vector<Entry> cache;
    unsigned long long clock = 0;
};

namespace {

// The stream the calling thread's launches go to: the context's stream, unless this host thread has bound its own for this
// context (les_hip_set_thread_stream: two views advanced by two host threads on one context, each on its own stream).
std::atomic<unsigned long long> g_ctx_gen{0};
// live contexts by generation id (the thread-exit hook of les_hip_unary_one returns a hidden scratch to its context only if that
// very context -- not a later one at the same address -- still exists)
std::mutex g_live_mu;
std::vector<std::pair<unsigned long long, les_hip_ctx*>> g_live;
void release_hidden_scratch(unsigned long long gen, les_hip_scratch* s)
{
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (auto& e : g_live)
        if (e.first == gen) {
            std::lock_guard<std::mutex> lk2(e.second->mu);
            e.second->idle_scratch.push_back(s);         // still owned (and eventually freed) by the context
            return;
        }
    // the context is gone: it has already destroyed the scratch
}
thread_local unsigned long long tl_stream_gen = 0;       // generation id of the context the stream below is bound to (0: none)
thread_local hipStream_t tl_stream = nullptr;
inline hipStream_t cur_stream(const les_hip_ctx* c) { return (tl_stream_gen != 0 && tl_stream_gen == c->gen) ? tl_stream : c->stream; }

int check_rects(const les_hip_ctx* c, const les_hip_rect& f, const les_hip_rect& t)
{
    if (f.w < 0 || f.h < 0 || t.w < 0 || t.h < 0) return fail(LES_HIP_ERR_ARG, "negative rect size");
    if (f.x < 0 || f.y < 0 || f.x + f.w > c->p.W || f.y + f.h > c->p.H) return fail(LES_HIP_ERR_ARG, "filterRect outside the image");
    if (t.w > 0 && t.h > 0 && (t.x < f.x || t.y < f.y || t.x + t.w > f.x + f.w || t.y + t.h > f.y + f.h))
        return fail(LES_HIP_ERR_ARG, "targetRect not inside filterRect");
    return LES_HIP_OK;
}

// Split every call's target rect into strips of TW columns (and row chunks when there are few calls),
// ordered by position so that neighbouring workgroups share volume halos / guide statistics in L2.
int build_jobs(const les_hip_ctx* c, int n, const les_hip_rect* frs, const les_hip_rect* trs, int out_slabs,
               std::vector<les::Job>& jobs)
{
    const int TW = c->strip->TW, R = c->R;
    const long long P = (long long)c->p.H * c->p.W;
    long long strips = 0;
    for (int i = 0; i < n; i++) {
        int rc = check_rects(c, frs[i], trs[i]);
        if (rc) return rc;
        if (trs[i].w > 0 && trs[i].h > 0) strips += (trs[i].w + TW - 1) / TW;
    }
    // row chunking: aim for >= ~2048 workgroups, never below 8*R rows per chunk (4R rows are halo work)
    int max_rows = 1 << 30;
    if (strips > 0 && strips < 2048) {
        long long want = (2048 + strips - 1) / strips;
        int tallest = 0;
        for (int i = 0; i < n; i++) tallest = std::max(tallest, trs[i].h);
        max_rows = std::max<long long>(std::max(8 * R, 64), (tallest + want - 1) / want);
    }
    jobs.clear();
    for (int i = 0; i < n; i++) {
        const les_hip_rect &f = frs[i], &t = trs[i];
        if (t.w <= 0 || t.h <= 0) continue;
        for (int sy = 0; sy < t.h; sy += max_rows)
            for (int sx = 0; sx < t.w; sx += TW) {
                les::Job j;
                j.tx0 = t.x + sx; j.ty0 = t.y + sy;
                j.tw = std::min(TW, t.w - sx); j.th = std::min(max_rows, t.h - sy);
                j.cx0 = f.x; j.cy0 = f.y; j.cx1 = f.x + f.w; j.cy1 = f.y + f.h;
                j.out_off = (out_slabs ? (long long)(i / out_slabs) * P : 0) + (long long)j.ty0 * c->p.W + j.tx0;
                j.out_stride = c->p.W;
                j.plane_idx = i;
                jobs.push_back(j);
            }
    }
    std::stable_sort(jobs.begin(), jobs.end(), [](const les::Job& a, const les::Job& b) {
        if (a.tx0 != b.tx0) return a.tx0 < b.tx0;
        if (a.ty0 != b.ty0) return a.ty0 < b.ty0;
        return a.plane_idx < b.plane_idx;
    });
    return LES_HIP_OK;
}

// The same calls cut for the march kernel: balanced strips of at most TW columns (a 45-column target becomes 23 + 22, never
// 44 + 1), groups of NJ consecutive jobs per workgroup (padded with empty jobs), and the geometric precondition of the kernel.
bool build_march_jobs(les_hip_ctx* c, int n, const les_hip_rect* frs, const les_hip_rect* trs, int out_slabs,
                      std::vector<les::Job>& jobs, bool& ok, const MarchEntry*& entry)
{
    jobs.clear();
    ok = false;
    entry = nullptr;
    if (!c->march) return true;
    const int R = c->R, W = c->p.W, H = c->p.H;
    const long long P = (long long)H * W;
    ok = true;
    for (int i = 0; i < n; i++) {
        const les_hip_rect &f = frs[i], &t = trs[i];
        if (t.w <= 0 || t.h <= 0) continue;
        if ((f.x > 0 && t.x - f.x < 2 * R) || (f.x + f.w < W && (f.x + f.w) - (t.x + t.w) < 2 * R) ||
            (f.y > 0 && t.y - f.y < 2 * R) || (f.y + f.h < H && (f.y + f.h) - (t.y + t.h) < 2 * R)) ok = false;
    }
    if (!ok) {
        note_fallback(c->fallback_seen, FB_GEOMETRY, "a target rectangle lies closer than 2 x radius = %d pixels to a filterRect border that is not an image border", 2 * R);
        return true;
    }
    // Cut: geometry (wide jobs, one per workgroup / narrow jobs, two per workgroup) and rows per job.  A workgroup fills a CU
    // (12 waves, ~155 KB LDS) and runs one block of BY rows (3 .. 7 by radius; 7 at radius 10) per ~2.3-2.7 us tick with a 2-tick pipeline fill and 4R halo rows per job, so
    // the launch time is about rounds(workgroups / CUs) x ticks(rows per job): pick the cut that minimises it.  (Layer-1/2 sets
    // have only 5..50 cells: whole cells would leave most CUs idle -- measured 13 and 8 G evaluations/s against 55 at layer 0.)
    const int ncu = c->ncu;
    const MarchEntry* cands[2] = {c->march, find_march(c->R, 0)};
    const MarchEntry* m = c->march;
    int max_rows = 1 << 30;
    {
        double best = 1e300;
        const int row_opts[] = {1 << 30, 1024, 512, 384, 256, 192, 128, 96, 64, 48, 32};
        for (int g = 0; g < 2; g++) {
            const MarchEntry* e = cands[g];
            if (!e || (g == 1 && e == cands[0])) continue;
            if (const char* w = getenv("LES_HIP_MARCH_WIDE")) if ((atoi(w) != 0) != (g == 0)) continue;
            for (int ro : row_opts) {
                long long njobs = 0;
                int rows = 0, useful = 0;
                for (int i = 0; i < n; i++) {
                    const les_hip_rect& t = trs[i];
                    if (t.w <= 0 || t.h <= 0) continue;
                    const int ns = (t.w + e->TW - 1) / e->TW;
                    const int nr = (t.h + ro - 1) / ro, sh = (t.h + nr - 1) / nr;
                    njobs += (long long)ns * nr;
                    rows = std::max(rows, sh);
                    useful = std::max(useful, (t.w + ns - 1) / ns);
                }
                if (njobs == 0) continue;
                const long long wgs = (njobs + e->NJ - 1) / e->NJ;
                const double ticks = (double)((rows + 4 * R + e->BY - 1) / e->BY + 3);
                // a partially filled last round costs as much as a full one; narrow jobs that leave most lanes idle cost the same tick
                const double cost = (double)((wgs + ncu - 1) / ncu) * ticks * (1.0 + 1e-3 * (double)wgs / ncu) + (ro == (1 << 30) ? 0.0 : 1e-6);
                if (cost < best) { best = cost; m = e; max_rows = ro; }
            }
        }
    }
    if (const char* e = getenv("LES_HIP_MARCH_ROWS")) max_rows = std::max(1, atoi(e));
    entry = m;
    const int TW = m->TW, NJ = m->NJ;
    for (int i = 0; i < n; i++) {
        const les_hip_rect &f = frs[i], &t = trs[i];
        if (t.w <= 0 || t.h <= 0) continue;
        const int ns = (t.w + TW - 1) / TW, sw = (t.w + ns - 1) / ns;
        const int nr = (t.h + max_rows - 1) / max_rows, sh = (t.h + nr - 1) / nr;
        for (int sy = 0; sy < t.h; sy += sh)
            for (int sx = 0; sx < t.w; sx += sw) {
                les::Job j;
                j.tx0 = t.x + sx; j.ty0 = t.y + sy;
                j.tw = std::min(sw, t.w - sx); j.th = std::min(sh, t.h - sy);
                j.cx0 = f.x; j.cy0 = f.y; j.cx1 = f.x + f.w; j.cy1 = f.y + f.h;
                j.out_off = (out_slabs ? (long long)(i / out_slabs) * P : 0) + (long long)j.ty0 * W + j.tx0;
                j.out_stride = W;
                j.plane_idx = i;
                jobs.push_back(j);
            }
    }
    std::stable_sort(jobs.begin(), jobs.end(), [](const les::Job& a, const les::Job& b) {
        if (a.tx0 != b.tx0) return a.tx0 < b.tx0;
        if (a.ty0 != b.ty0) return a.ty0 < b.ty0;
        return a.plane_idx < b.plane_idx;
    });
    while (!jobs.empty() && jobs.size() % NJ) {
        les::Job pad = jobs.back();
        pad.tw = 0; pad.th = 0;
        jobs.push_back(pad);
    }
    return true;
}

int ensure_planes(les_hip_ctx* c, size_t n)
{
    if (n <= c->planes_cap) return LES_HIP_OK;
    if (c->d_planes) HIPCHECK(hipFree(c->d_planes));
    c->d_planes = nullptr; c->planes_cap = 0;
    size_t cap = std::max<size_t>(n, 1024);
    HIPCHECK(hipMalloc((void**)&c->d_planes, cap * sizeof(float4)));
    c->planes_cap = cap;
    return LES_HIP_OK;
}

// The raw-cost patches of an image-based context (null for a cost-volume context): call table, patch offsets, patch buffer
struct RawPatches { const les::RawCall* calls; const long long* off; float* raw; int n, chunks; };

les::View strip_view(const les_hip_ctx* c, int mode)
{
    les::View view{c->v[mode].vol, c->v[mode].stats, c->v[mode].ipk, c->v[mode].ipk10, nullptr, nullptr, mode ? -1.0f : 1.0f, c->th_color, c->th_grad};
    if (c->naive) { view.feat_self = c->v[mode].feat; view.feat_other = c->v[1 - mode].feat; }
    return view;
}

int launch_march(les_hip_ctx* c, const MarchEntry* m, int mode, const les::Job* d_mjobs, int ngroups, const float4* d_planes, float* d_out, int check,
                 hipStream_t stream, const RawPatches* rp = nullptr)
{
    if (ngroups <= 0) return LES_HIP_OK;
    les::MarchView mv = c->v[mode].mv;
    if (c->naive) {
        if (!rp || !rp->raw || !c->v[1 - mode].feat) return fail(LES_HIP_ERR_ARG, "view %d was not supplied at creation", mode);
        hipLaunchKernelGGL(les::les_naive_raw_kernel, dim3(rp->n, rp->chunks), dim3(256), 0, stream, c->geom, strip_view(c, mode), rp->calls, d_planes, rp->raw);
        mv.vol = rp->raw; mv.raw_off = rp->off;
    }
    hipLaunchKernelGGL(m->fn, dim3(ngroups), dim3(m->NT), 0, stream, c->geom, mv, d_mjobs, d_planes, d_out, ngroups, check);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int launch_strips(les_hip_ctx* c, int mode, const les::Job* d_jobs, int njobs, const float4* d_planes, float* d_out, int check, hipStream_t stream)
{
    if (njobs <= 0) return LES_HIP_OK;
    if (mode < 0 || mode > 1 || !c->v[mode].stats || (c->naive ? !c->v[1 - mode].feat : !c->v[mode].vol))
        return fail(LES_HIP_ERR_ARG, "view %d was not supplied at creation", mode);
    const les::View view = strip_view(c, mode);
    hipLaunchKernelGGL(c->strip->fn, dim3(njobs), dim3(c->strip->NT), 0, stream, c->geom, view, d_jobs, d_planes, d_out, njobs, check);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

float naive_alpha(const les_hip_ctx* c);

// Tables and constants of the march kernel for view m (les_march.h): statistics in its format, the guide as signed bytes,
// the cost range of the volume and the stage-2 scale from the bound |a_c| <= sqrt(inv_cc) * sd(p) <= sqrt(max inv_cc) * range / 2.
int build_march_view(les_hip_ctx* c, int m, const double* d_hs)
{
    ViewData& v = c->v[m];
    const size_t P = (size_t)c->p.H * c->p.W;
    const int W = c->p.W, H = c->p.H;
    v.march_ok = false;
    // the kernel addresses image rows and statistics rows by 32-bit byte offsets (raw buffer access): images of 2^26 pixels or more stay on the strip kernel
    if ((unsigned long long)P * (4ull * les::kMarchStatWords) >= (1ull << 31)) {
        note_fallback(c->fallback_seen, FB_IMAGE_SIZE, "image of %d x %d pixels (32-bit row offsets reach 2^26 pixels)", W, H);
        return LES_HIP_OK;
    }
    // image-based energy: the raw cost min(|dcolor|, th_color) + min(|dgrad|, th_grad) lies in [0, th_color + th_grad] by construction
    const float th = c->naive ? c->th_color + c->th_grad : c->p.th_col;
    if (!(th > 0.0f) || !(th < INFINITY)) { note_fallback(c->fallback_seen, FB_THRESHOLD, "truncation threshold %g is not positive and finite", (double)th); return LES_HIP_OK; }
    // cost range
    const int nb = 2048;
    std::vector<float> hmin(nb, 0.0f); std::vector<int> hbad(nb, 0);
    if (!c->naive) {
        // one allocation for both partial arrays, released on every path (a failing call must not leak device memory)
        struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) (void)hipFree(p); } } part;
        HIPCHECK(hipMalloc(&part.p, nb * (sizeof(float) + sizeof(int))));
        float* d_min = static_cast<float*>(part.p);
        int* d_bad = reinterpret_cast<int*>(d_min + nb);
        hipLaunchKernelGGL(les::les_range_kernel, dim3(nb), dim3(256), 0, cur_stream(c), v.vol, P * (size_t)c->p.D, d_min, d_bad);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(hmin.data(), d_min, nb * sizeof(float), hipMemcpyDeviceToHost, cur_stream(c)));
        HIPCHECK(hipMemcpyAsync(hbad.data(), d_bad, nb * sizeof(int), hipMemcpyDeviceToHost, cur_stream(c)));
        HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    }
    float vmin = INFINITY; int bad = 0;
    for (int i = 0; i < nb; i++) { vmin = std::min(vmin, hmin[i]); bad |= hbad[i]; }
    if (bad || !(vmin < INFINITY)) {                               // NaN / inf costs: the fp64 strip kernel reproduces the reference's propagation
        note_fallback(c->fallback_seen, FB_NONFINITE, "the cost volume of view %d holds NaN or infinite entries", m);
        return LES_HIP_OK;
    }
    vmin = std::min(vmin, 0.5f * th);                                // a volume entirely above th_col: p == th_col everywhere
    const double range = (double)th - (double)vmin;
    if (!(range <= 8.0 * (double)th)) {                              // the 20-bit fixed point would resolve th_col too coarsely
        note_fallback(c->fallback_seen, FB_RANGE, "view %d: costs reach %g below the truncation threshold %g (more than 8 x the threshold)", m, range, (double)th);
        return LES_HIP_OK;
    }
    // tables (d_hs == nullptr: les_hip_refresh_volume -- the guide has not changed, its tables and the bound on the inverse covariance are kept)
    unsigned dbits = v.dmax_bits;
    if (d_hs) {
        struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) (void)hipFree(p); } } dm;
        HIPCHECK(hipMalloc(&dm.p, sizeof(unsigned)));
        unsigned* d_dmax = static_cast<unsigned*>(dm.p);
        HIPCHECK(hipMemsetAsync(d_dmax, 0, sizeof(unsigned), cur_stream(c)));
        HIPCHECK(hipMalloc((void**)&v.ipk8, P * sizeof(uint32_t)));
        HIPCHECK(hipMalloc((void**)&v.mstats, P * les::kMarchStatWords * sizeof(float)));
        hipLaunchKernelGGL(les::les_march_stats_kernel, dim3((W + 255) / 256, H), dim3(256), 0, cur_stream(c), d_hs, v.ipk, v.ipk8, v.mstats, d_dmax, H, W, c->R, c->p.eps);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(&dbits, d_dmax, sizeof(unsigned), hipMemcpyDeviceToHost, cur_stream(c)));
        HIPCHECK(hipStreamSynchronize(cur_stream(c)));
        v.dmax_bits = dbits;
    }
    float dmax;
    memcpy(&dmax, &dbits, sizeof dmax);
    if (!(dmax > 0.0f) || !(dmax < INFINITY)) { note_fallback(c->fallback_seen, FB_GUIDE, "view %d: the inverse covariance of the guide is not positive and finite (largest diagonal entry %g)", m, (double)dmax); return LES_HIP_OK; }
    const int K = 2 * c->R + 1;
    const double Ba = 0.5 * range * std::sqrt((double)dmax), Bb = range + 1.5 * Ba;
    const double scale = 1073741824.0 / ((double)K * Bb * 1.25);    // horizontal box sums of the quantised a, b stay below 2^30
    // centred fixed-point cost (les_march.h): count = rint(p sp) + c0, c0 an integer, so that [vmin, th] maps onto [-2^(PB-1), 2^(PB-1) - 1]
    const int PB = les::march_pb(c->R);
    const float spf = (float)((double)((1 << PB) - 1) / range);
    const double c0 = std::rint(-(double)vmin * (double)spf) - (double)(1 << (PB - 1));
    const double up = 1.0 / (double)spf;
    les::MarchView mv;
    mv.vol = v.vol; mv.ipk8 = v.ipk8; mv.mstats = v.mstats;
    mv.sp = spf;
    mv.pmagic = (float)(12582912.0 + c0);                           // 1.5 * 2^23 + c0: an integer below 2^24, exact
    mv.poff = (float)(-c0 * up);
    mv.kapS = (float)((double)(1 << les::kMarchSH) * up / 255.0 * scale);
    mv.upS = (float)(up * scale);
    mv.qscale = (float)((double)(1 << les::kMarchS2) / (255.0 * scale));
    mv.kmu = (float)(1.0 / ((double)(1ll << les::kMarchMB) * 255.0));
    mv.raw_off = nullptr;
    mv.vol_t = nullptr;
    // The tiled copy for the taps of planes that are steep along x (a second resident copy of the volume: 1.5 GB more at 1500 x 1000 x 256, of
    // 288 GB).  Optional: LES_HIP_TILED=0 turns it off, a failing allocation or too little free memory (below) leaves it out
    // -- such planes then gather from [D][H][W] as every other plane does (same values, more HBM traffic).
    if (!c->naive && v.vol) {
        const char* e = getenv("LES_HIP_TILED");
        const unsigned long long nt = (unsigned long long)H * (unsigned long long)((W + 7) / 8) * 8ull * (unsigned long long)c->p.D;
        // (any size since round 5: the kernel's descriptor starts at the first row a job gathers; one image row of tiles must stay below 2^31 bytes)
        if (!(e && atoi(e) == 0) && (unsigned long long)((W + 7) / 8) * (unsigned long long)c->p.D * 32ull < (1ull << 31)) {
            if (v.vol_t) { (void)hipFree(v.vol_t); v.vol_t = nullptr; }
            // the copy must not be what later makes a scratch, batch or graph allocation fail: it is only taken when it leaves at least as much
            // memory free again as it uses, and 4 GB on top
            size_t mem_free = 0, mem_total = 0;
            const bool room = hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && mem_free >= 2 * nt * sizeof(float) + (4ull << 30);
            if (room && hipMalloc((void**)&v.vol_t, nt * sizeof(float)) == hipSuccess) {
                hipLaunchKernelGGL(les::les_tile_volume_kernel, dim3((unsigned)(((W + 7) / 8 * 8 + 63) / 64), (unsigned)H), dim3(256), 0, cur_stream(c), v.vol, v.vol_t, c->p.D, H, W);
                HIPCHECK(hipGetLastError());
                mv.vol_t = v.vol_t;
            } else {
                (void)hipGetLastError();
                v.vol_t = nullptr;
            }
        }
    }
    v.mv = mv;
    v.march_ok = true;
    return LES_HIP_OK;
}

int build_view(les_hip_ctx* c, int m, const uint8_t* im, const float* vol)
{
    const size_t P = (size_t)c->p.H * c->p.W;
    ViewData& v = c->v[m];
    if (vol) {
        if (c->p.volumes_on_device) { v.vol = const_cast<float*>(vol); v.own_vol = false; }
        else {
            HIPCHECK(hipMalloc((void**)&v.vol, P * c->p.D * sizeof(float)));
            v.own_vol = true;
            HIPCHECK(hipMemcpy(v.vol, vol, P * c->p.D * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (!im) return LES_HIP_OK;
    uint8_t* d_img = nullptr;
    double* d_hs = nullptr;
    HIPCHECK(hipMalloc((void**)&d_img, P * 3));
    HIPCHECK(hipMemcpy(d_img, im, P * 3, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc((void**)&v.ipk, P * sizeof(uint32_t)));
    HIPCHECK(hipMalloc((void**)&v.ipk10, P * sizeof(uint32_t)));
    HIPCHECK(hipMalloc((void**)&v.stats, (P * 3 + 1) * sizeof(float4)));           // + one all-zero entry (read by the k = 3 lanes of phase V)
    HIPCHECK(hipMemsetAsync(v.stats + P * 3, 0, sizeof(float4), cur_stream(c)));
    HIPCHECK(hipMalloc((void**)&d_hs, P * 9 * sizeof(double)));
    const int W = c->p.W, H = c->p.H;
    if (c->naive) {
        HIPCHECK(hipMalloc((void**)&v.feat, P * sizeof(float4)));
        hipLaunchKernelGGL(les::les_naive_features_kernel, dim3((c->p.W + 255) / 256, c->p.H), dim3(256), 0, cur_stream(c), d_img, v.feat, c->p.H, c->p.W, naive_alpha(c));
    }
    hipLaunchKernelGGL(les::les_pack_guide_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, cur_stream(c), d_img, v.ipk, v.ipk10, (int)P);
    hipLaunchKernelGGL(les::les_stats_hsum_kernel, dim3((W + 255) / 256, H), dim3(256), 0, cur_stream(c), v.ipk, d_hs, H, W, c->R);
    hipLaunchKernelGGL(les::les_stats_finish_kernel, dim3((W + 255) / 256, H), dim3(256), 0, cur_stream(c), d_hs, v.stats, H, W, c->R, c->p.eps);
    HIPCHECK(hipGetLastError());
    if (c->march && (v.vol || c->naive)) {
        int rc = build_march_view(c, m, d_hs);
        if (rc) { (void)hipFree(d_img); (void)hipFree(d_hs); return rc; }
    }
    HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    HIPCHECK(hipFree(d_img));
    HIPCHECK(hipFree(d_hs));
    return LES_HIP_OK;
}

float naive_alpha(const les_hip_ctx* c) { return c->naive_alpha; }

}  // namespace

// Host side of one tiled solve in flight: the host-mapped words the kernels report through (pinned, fine-grained: the kernel adds to them with
// system-scope atomics, the host reads them after synchronising its stream -- the progress check of a lock-step costs no copy) and the pinned
// staging of the hand-over (les_maxflow_tiled.h: residual graphs out, masks and flow values back).  A context keeps a pool of them: a call
// takes one, returns it at the end; they are freed with the context (round 5 kept two words per host THREAD for ever).
struct MtHost {
    int* h_flags = nullptr; int* d_flags = nullptr;                    // [0] cells finished, [1] of them: gave up, [2] cells handed over, [3] their nodes
    char* h_stage = nullptr; char* d_stage = nullptr;                  // rc8 [cap_nodes][8] floats | ex [cap_nodes] floats | masks [cap_nodes] bytes | list [cap_cells] | flows [cap_cells]
    long long cap_nodes = 0;
    int cap_cells = 0;
    size_t off_ex() const { return (size_t)cap_nodes * 32; }
    size_t off_masks() const { return (size_t)cap_nodes * 36; }
    size_t off_list() const { return ((size_t)cap_nodes * 37 + 255) & ~(size_t)255; }
    size_t off_flows() const { return off_list() + (size_t)cap_cells * sizeof(les::MtHandCell); }
    size_t bytes() const { return off_flows() + (size_t)cap_cells * sizeof(double); }
};
namespace {
void mt_host_free(MtHost* m)
{
    if (!m) return;
    if (m->h_flags) (void)hipHostFree(m->h_flags);
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    delete m;
}
int mt_host_map(void** h, void** d, size_t bytes)
{
#if defined(LES_SIM)
    HIPCHECK(hipHostMalloc(h, bytes, 0));
    *d = *h;
#else
    HIPCHECK(hipHostMalloc(h, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHECK(hipHostGetDevicePointer(d, *h, 0));
#endif
    return LES_HIP_OK;
}
int mt_host_acquire(les_hip_ctx* c, MtHost** out)
{
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->mt_idle.empty()) { *out = c->mt_idle.back(); c->mt_idle.pop_back(); return LES_HIP_OK; }
    }
    MtHost* m = new MtHost();
    const int rc = mt_host_map((void**)&m->h_flags, (void**)&m->d_flags, 64);
    if (rc) { delete m; return rc; }
    *out = m;
    return LES_HIP_OK;
}
void mt_host_release(les_hip_ctx* c, MtHost* m)
{
    std::lock_guard<std::mutex> lk(c->mu);
    c->mt_idle.push_back(m);
}
// staging for `nodes` graph nodes (37 bytes each) of `cells` cells, grown on demand
int mt_host_stage(MtHost* m, long long nodes, int cells)
{
    if (m->cap_nodes >= nodes && m->cap_cells >= cells) return LES_HIP_OK;
    if (m->h_stage) { (void)hipHostFree(m->h_stage); m->h_stage = nullptr; }
    m->cap_nodes = std::max(m->cap_nodes, (nodes + 4095) & ~4095ll);
    m->cap_cells = std::max(m->cap_cells, (cells + 15) & ~15);
    const int rc = mt_host_map((void**)&m->h_stage, (void**)&m->d_stage, m->bytes());
    if (rc) { m->cap_nodes = 0; m->cap_cells = 0; return rc; }
    return LES_HIP_OK;
}
struct MtHostLease {                       // returns the MtHost to the pool on every exit path
    les_hip_ctx* c; MtHost* m;
    ~MtHostLease() { if (m) mt_host_release(c, m); }
};
}  // namespace

extern "C" {

const char* les_hip_last_error(void) { return g_err.c_str(); }

#if defined(LES_PHASE_TIMING)
int les_hip_debug_phases(unsigned long long* out)          // experiment builds only (tools/phase_probe.py); not part of the ABI header
{
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(les::les_dbg), 12 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(les::les_dbg), z, sizeof z);
    return 0;
}
#endif

int les_hip_strip_width(int R)
{
    const StripEntry* e = find_strip(R);
    return e ? e->TW : 0;
}

static int create_common(les_hip_ctx** out, const les_hip_params* params, const uint8_t* imL, const uint8_t* imR,
                         const float* volL, const float* volR, int naive, float alpha, float th_grad)
{
    if (!out || !params) return fail(LES_HIP_ERR_ARG, "null argument");
    *out = nullptr;
    les_hip_params p = *params;
    if (naive) { p.D = 1; p.volumes_on_device = 0; }
    if (p.H <= 0 || p.W <= 0 || p.D <= 0 || p.windR < 2) return fail(LES_HIP_ERR_ARG, "bad dimensions");
    if (naive && (!imL || !imR)) return fail(LES_HIP_ERR_ARG, "the image-based matching cost needs both views");
    if ((unsigned long long)p.H * p.W * p.D >= (1ull << 32)) return fail(LES_HIP_ERR_UNSUPPORTED, "volumes of 2^32 or more floats are not supported (32-bit element offsets)");
    const StripEntry* strip = naive ? find_naive_strip(p.windR / 2) : find_strip(p.windR / 2);
    if (!strip) return fail(LES_HIP_ERR_UNSUPPORTED, "no kernel instantiated for guided-filter radius %d (windR %d)", p.windR / 2, p.windR);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(LES_HIP_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (p.device < 0 || p.device >= ndev) return fail(LES_HIP_ERR_ARG, "device %d out of range (%d devices)", p.device, ndev);
    HIPCHECK(hipSetDevice(p.device));
    les_hip_ctx* c = new les_hip_ctx();
    c->gen = ++g_ctx_gen;
    { std::lock_guard<std::mutex> lk(g_live_mu); g_live.emplace_back(c->gen, c); }
    c->p = p;
    c->R = p.windR / 2;
    c->strip = strip;
    c->march = find_march(p.windR / 2);
    if (!c->march && !(getenv("LES_HIP_KERNEL") && !strcmp(getenv("LES_HIP_KERNEL"), "strip")))
        note_fallback(c->fallback_seen, FB_RADIUS, "no march kernel for guided-filter radius %d (windR %d; instantiated: 4 .. 10)", p.windR / 2, p.windR);
#if !defined(LES_SIM)
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, p.device) == hipSuccess && v > 0) c->ncu = v;
    }
#endif
    c->stream = nullptr;
    c->geom.H = p.H; c->geom.W = p.W; c->geom.D = p.D;
    c->geom.D0 = (int)(-p.min_disparity);                       // LES/CostVolumeEnergy.h:67
    c->geom.th_col = p.th_col; c->geom.pad_ = 0.0f;
    c->geom.maxd = p.max_disparity; c->geom.mind = p.min_disparity;
    c->naive = naive;
    if (naive) {
        c->th_color = p.th_col * (1.0f - alpha);                 // LES/StereoEnergy.h:662
        c->th_grad = th_grad * alpha;                            // LES/StereoEnergy.h:663
        c->naive_alpha = alpha;
    }
    const uint8_t* ims[2] = {imL, imR};
    const float* vols[2] = {volL, volR};
    for (int m = 0; m < 2; m++) {
        int rc = build_view(c, m, ims[m], vols[m]);
        if (rc) { les_hip_destroy(c); return rc; }
    }
    if (hipMalloc((void**)&c->d_map, (size_t)p.H * p.W * sizeof(float)) != hipSuccess) {
        les_hip_destroy(c);
        return fail(LES_HIP_ERR_DEVICE, "hipMalloc of the scratch cost map failed");
    }
    *out = c;
    return LES_HIP_OK;
}

int les_hip_create(les_hip_ctx** out, const les_hip_params* params, const uint8_t* imL, const uint8_t* imR,
                   const float* volL, const float* volR)
{
    return create_common(out, params, imL, imR, volL, volR, 0, 0.0f, 0.0f);
}

int les_hip_create_naive(les_hip_ctx** out, const les_hip_params* params, const uint8_t* imL, const uint8_t* imR, float alpha, float th_grad)
{
    return create_common(out, params, imL, imR, nullptr, nullptr, 1, alpha, th_grad);
}

void les_hip_destroy(les_hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (tl_stream_gen == c->gen) { tl_stream_gen = 0; tl_stream = nullptr; }     // the destroying thread's own binding (other threads' bindings die with the id)
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.erase(std::remove_if(g_live.begin(), g_live.end(), [c](const std::pair<unsigned long long, les_hip_ctx*>& e) { return e.second == c; }), g_live.end());
    }
    for (int m = 0; m < 2; m++) {
        if (c->v[m].own_vol && c->v[m].vol) (void)hipFree(c->v[m].vol);
        if (c->v[m].stats) (void)hipFree(c->v[m].stats);
        if (c->v[m].ipk) (void)hipFree(c->v[m].ipk);
        if (c->v[m].ipk10) (void)hipFree(c->v[m].ipk10);
        if (c->v[m].feat) (void)hipFree(c->v[m].feat);
        if (c->v[m].ipk8) (void)hipFree(c->v[m].ipk8);
        if (c->v[m].mstats) (void)hipFree(c->v[m].mstats);
        if (c->v[m].vol_t) (void)hipFree(c->v[m].vol_t);
    }
    for (les_hip_scratch* sc : c->own_scratch) les_hip_scratch_destroy(sc);
    c->own_scratch.clear();
    for (MtHost* m : c->mt_idle) mt_host_free(m);
    c->mt_idle.clear();
    if (c->d_planes) (void)hipFree(c->d_planes);
    if (c->d_map) (void)hipFree(c->d_map);
    if (c->d_wta) (void)hipFree(c->d_wta);
    if (c->d_wta_planes) (void)hipFree(c->d_wta_planes);
    if (c->d_pw_tab) (void)hipFree(c->d_pw_tab);
    delete c;
}

int les_hip_set_stream(les_hip_ctx* c, void* s)
{
    if (!c) return fail(LES_HIP_ERR_ARG, "null context");
    c->stream = (hipStream_t)s;
    return LES_HIP_OK;
}

int les_hip_set_thread_stream(les_hip_ctx* c, void* s, int bind)
{
    if (!c) return fail(LES_HIP_ERR_ARG, "null context");
    if (bind) { tl_stream_gen = c->gen; tl_stream = (hipStream_t)s; }
    else if (tl_stream_gen == c->gen) { tl_stream_gen = 0; tl_stream = nullptr; }
    return LES_HIP_OK;
}

int les_hip_synchronize(les_hip_ctx* c)
{
    if (!c) return fail(LES_HIP_ERR_ARG, "null context");
    HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    return LES_HIP_OK;
}

int les_hip_refresh_volume(les_hip_ctx* c, int mode)
{
    if (!c) return fail(LES_HIP_ERR_ARG, "null argument");
    if (mode < 0 || mode > 1 || !c->v[mode].vol) return fail(LES_HIP_ERR_ARG, "view %d has no cost volume", mode);
    HIPCHECK(hipSetDevice(c->p.device));
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    if (!c->march || !c->v[mode].mstats) return LES_HIP_OK;          // strip kernel only: it keeps nothing derived from the volume
    const int rc = build_march_view(c, mode, nullptr);             // cost range -> fixed-point scales, tiled copy rebuilt; the guide's tables stay
    if (rc) return rc;
    HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    return LES_HIP_OK;
}

int les_hip_batch_create(les_hip_ctx* c, int n, const les_hip_rect* frs, const les_hip_rect* trs, int out_slabs, les_hip_batch** out)
{
    if (!c || !out || n < 0 || (n > 0 && (!frs || !trs))) return fail(LES_HIP_ERR_ARG, "null argument");
    if (out_slabs < 0) return fail(LES_HIP_ERR_ARG, "out_slabs must be 0 (one map) or the number of consecutive calls that share a slab");
    // (since round 4 out_slabs = k means "call i writes slab i / k"; before, any non-zero value meant k = 1.  A caller that still passes another
    // non-zero constant for "one slab per call" would get overlapping writes: only a k that divides n is a well-formed request)
    if (out_slabs > 1 && n % out_slabs != 0) return fail(LES_HIP_ERR_ARG, "out_slabs = %d does not divide the %d calls of the batch (slab i / out_slabs holds out_slabs consecutive calls; pass 1 for one slab per call)", out_slabs, n);
    *out = nullptr;
    std::vector<les::Job> jobs;
    int rc = build_jobs(c, n, frs, trs, out_slabs, jobs);
    if (rc) return rc;
    les_hip_batch* b = new les_hip_batch();
    b->n = n; b->njobs = (int)jobs.size(); b->out_slabs = out_slabs; b->R = c->R; b->device = c->p.device;
    b->targets.assign(trs, trs + n);
    {
        int max_area = 1;
        for (int i = 0; i < n; i++) max_area = std::max(max_area, trs[i].w * trs[i].h);
        b->wta_chunks = std::min(32, std::max(1, (max_area + 4095) / 4096));
    }
    b->graph_off.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        b->graph_off[i] = b->graph_nodes;
        b->graph_nodes += (long long)std::max(0, trs[i].w) * std::max(0, trs[i].h);
    }
    if (n > 0) {
        static_assert(sizeof(les::WtaJob) == sizeof(les_hip_rect), "rect layout");
        static_assert(sizeof(les::GraphCell) == sizeof(les_hip_rect), "rect layout");
        if (hipMalloc((void**)&b->d_graph_off, (size_t)n * sizeof(long long)) != hipSuccess ||
            hipMemcpy(b->d_graph_off, b->graph_off.data(), (size_t)n * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess ||
            hipMalloc((void**)&b->d_flow0, (size_t)n * b->wta_chunks * sizeof(double)) != hipSuccess) {
            les_hip_batch_destroy(b);
            return fail(LES_HIP_ERR_DEVICE, "upload of the graph offset table failed");
        }
        if (hipMalloc((void**)&b->d_targets, (size_t)n * sizeof(les::WtaJob)) != hipSuccess ||
            hipMemcpy(b->d_targets, trs, (size_t)n * sizeof(les::WtaJob), hipMemcpyHostToDevice) != hipSuccess) {
            les_hip_batch_destroy(b);
            return fail(LES_HIP_ERR_DEVICE, "upload of the target table failed");
        }
    }
    {
        std::vector<les::Job> mjobs;
        bool mok = false;
        build_march_jobs(c, n, frs, trs, out_slabs, mjobs, mok, b->mentry);
        if (mok && !mjobs.empty()) {
            if (hipMalloc((void**)&b->d_mjobs, mjobs.size() * sizeof(les::Job)) != hipSuccess ||
                hipMemcpy(b->d_mjobs, mjobs.data(), mjobs.size() * sizeof(les::Job), hipMemcpyHostToDevice) != hipSuccess) {
                les_hip_batch_destroy(b);
                return fail(LES_HIP_ERR_DEVICE, "upload of the march job table failed");
            }
            b->nmgroups = (int)(mjobs.size() / b->mentry->NJ);
            b->march_ok = true;
        }
        if (b->march_ok && c->naive) {
            // raw-cost patches: one per call, the size of its filterRect.  Batches whose patches would not fit the cap stay on the strip kernel.
            std::vector<les::RawCall> calls((size_t)n);
            std::vector<long long> offs((size_t)n);
            long long tot = 0, amax = 1;
            for (int i = 0; i < n; i++) {
                const bool live = trs[i].w > 0 && trs[i].h > 0;
                const long long a = live ? (long long)frs[i].w * frs[i].h : 0;
                calls[i] = les::RawCall{frs[i].x, frs[i].y, live ? frs[i].w : 0, live ? frs[i].h : 0, tot};
                offs[i] = tot;
                tot += a; amax = std::max(amax, a);
            }
            if (tot > kRawPatchCapFloats) {
                b->march_ok = false;
                note_fallback(c->fallback_seen, FB_PATCHES, "the raw-cost patches of one batch of the image-based energy exceed 4 GB");
            }
            else {
                b->raw_floats = tot;
                b->raw_chunks = (int)std::min<long long>(1024, std::max<long long>(1, (amax + 4095) / 4096));
                if (hipMalloc((void**)&b->d_rawcalls, (size_t)n * sizeof(les::RawCall)) != hipSuccess ||
                    hipMemcpy(b->d_rawcalls, calls.data(), (size_t)n * sizeof(les::RawCall), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMalloc((void**)&b->d_raw_off, (size_t)n * sizeof(long long)) != hipSuccess ||
                    hipMemcpy(b->d_raw_off, offs.data(), (size_t)n * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess) {
                    les_hip_batch_destroy(b);
                    return fail(LES_HIP_ERR_DEVICE, "upload of the raw-cost call table failed");
                }
            }
        }
    }
    if (!jobs.empty()) {
        if (hipMalloc((void**)&b->d_jobs, jobs.size() * sizeof(les::Job)) != hipSuccess) { les_hip_batch_destroy(b); return fail(LES_HIP_ERR_DEVICE, "hipMalloc(jobs) failed"); }
        if (hipMemcpy(b->d_jobs, jobs.data(), jobs.size() * sizeof(les::Job), hipMemcpyHostToDevice) != hipSuccess) {
            les_hip_batch_destroy(b); return fail(LES_HIP_ERR_DEVICE, "hipMemcpy(jobs) failed");
        }
    }
    *out = b;
    return LES_HIP_OK;
}

void les_hip_batch_destroy(les_hip_batch* b)
{
    if (!b) return;
    if (b->d_jobs) (void)hipFree(b->d_jobs);
    if (b->d_mjobs) (void)hipFree(b->d_mjobs);
    if (b->d_rawcalls) (void)hipFree(b->d_rawcalls);
    if (b->d_raw_off) (void)hipFree(b->d_raw_off);
    for (int m = 0; m < 2; m++) if (b->d_raw[m]) (void)hipFree(b->d_raw[m]);
    if (b->d_units) (void)hipFree(b->d_units);
    if (b->d_targets) (void)hipFree(b->d_targets);
    if (b->d_graph_off) (void)hipFree(b->d_graph_off);
    if (b->d_flow0) (void)hipFree(b->d_flow0);
    if (b->d_mt_tiles) (void)hipFree(b->d_mt_tiles);
    if (b->d_mt_tiles_per_cell) (void)hipFree(b->d_mt_tiles_per_cell);
    if (b->rs.disp) (void)hipFree(b->rs.disp);
    if (b->rs.idx) (void)hipFree(b->rs.idx);
    if (b->rs.state) (void)hipFree(b->rs.state);
    if (b->rs.noi) (void)hipFree(b->rs.noi);
    if (b->rs.no) (void)hipFree(b->rs.no);
    if (b->rs.refit) (void)hipFree(b->rs.refit);
    if (b->rs.cell) (void)hipFree(b->rs.cell);
    delete b;
}

int les_hip_batch_set_units(les_hip_ctx* c, les_hip_batch* b, const les_hip_rect* units)
{
    if (!c || !b || (b->n > 0 && !units)) return fail(LES_HIP_ERR_ARG, "null argument");
    int maxlen = 1;
    for (int i = 0; i < b->n; i++) {
        const les_hip_rect& u = units[i];
        if (u.w <= 0 || u.h <= 0 || u.x < 0 || u.y < 0 || u.x + u.w > c->p.W || u.y + u.h > c->p.H)
            return fail(LES_HIP_ERR_ARG, "unit rect %d empty or outside the image", i);
        maxlen = std::max(maxlen, u.w * u.h);
    }
    if (b->n == 0) return LES_HIP_OK;
    static_assert(sizeof(les::Rect4) == sizeof(les_hip_rect), "rect layout");
    if (!b->d_units) HIPCHECK(hipMalloc((void**)&b->d_units, (size_t)b->n * sizeof(les::Rect4)));
    HIPCHECK(hipMemcpy(b->d_units, units, (size_t)b->n * sizeof(les::Rect4), hipMemcpyHostToDevice));
    if (!b->rs.idx) {
        const size_t n = (size_t)b->n, S = kRansacMaxSam;
        HIPCHECK(hipMalloc((void**)&b->rs.idx, n * S * 3 * sizeof(int)));
        HIPCHECK(hipMalloc((void**)&b->rs.state, n * (S + 1) * sizeof(uint64_t)));
        HIPCHECK(hipMalloc((void**)&b->rs.noi, n * S * sizeof(int)));
        HIPCHECK(hipMalloc((void**)&b->rs.no, n * S * sizeof(int)));
        HIPCHECK(hipMalloc((void**)&b->rs.refit, n * S * 3 * sizeof(float)));
        HIPCHECK(hipMalloc((void**)&b->rs.cell, n * sizeof(les::RansacCell)));
    }
    if (b->rs.disp) HIPCHECK(hipFree(b->rs.disp));
    b->rs.disp = nullptr;
    b->rs.stride = maxlen;
    HIPCHECK(hipMalloc((void**)&b->rs.disp, (size_t)b->n * maxlen * sizeof(float)));
    return LES_HIP_OK;
}

int les_hip_batch_propose(les_hip_ctx* c, const les_hip_batch* b, int kind, int m, les_hip_plane* labels, uint64_t* rng,
                          les_hip_plane* planes)
{
    if (c) (void)hipSetDevice(c->p.device);                 // HIP's current device is per host thread
    if (!c || !b || !labels || !rng || !planes) return fail(LES_HIP_ERR_ARG, "null argument");
    if (b->n == 0) return LES_HIP_OK;
    if (!b->d_units) return fail(LES_HIP_ERR_ARG, "les_hip_batch_set_units was not called for this batch");
    float4* lab = reinterpret_cast<float4*>(labels);
    float4* pl = reinterpret_cast<float4*>(planes);
    const int n = b->n, W = c->p.W;
    const float mind = c->p.min_disparity, maxd = c->p.max_disparity;
    switch (kind) {
    case LES_HIP_PROPOSE_EXPANSION:
        hipLaunchKernelGGL(les::les_expansion_kernel, dim3((n + 63) / 64), dim3(64), 0, cur_stream(c), b->d_units, lab, W, rng, pl, n);
        break;
    case LES_HIP_PROPOSE_RANDOM:
        hipLaunchKernelGGL(les::les_random_kernel, dim3((n + 63) / 64), dim3(64), 0, cur_stream(c), b->d_units, lab, W, rng, pl, n, m, mind, maxd);
        break;
    case LES_HIP_PROPOSE_RANSAC:
        // RansacProposer(K, MAX_SAM = 500, conf = 0.95), threshold 1.0 (LES/Proposer.h:265,305)
        hipLaunchKernelGGL(les::les_ransac_snapshot_kernel, dim3(n), dim3(256), 0, cur_stream(c), b->d_units, lab, W, b->rs);
        // the reference's adaptive schedule (:193, :229-236): candidates in chunks, a cell whose loop has ended ignores the later launches
        // (no host round trip: the launches of dead chunks return at once)
        hipLaunchKernelGGL(les::les_ransac_begin_kernel, dim3((n + 63) / 64), dim3(64), 0, cur_stream(c), b->d_units, rng, b->rs, n, kRansacMaxSam, kRansacChunkEnds[0]);
        for (int k = 0, j0 = 0; k < kRansacChunks; k++) {
            const int j1 = kRansacChunkEnds[k], j2 = k + 1 < kRansacChunks ? kRansacChunkEnds[k + 1] : kRansacMaxSam;
            hipLaunchKernelGGL(les::les_ransac_eval_kernel, dim3(n, (j1 - j0 + 15) / 16), dim3(64), 0, cur_stream(c), b->d_units, b->rs, kRansacMaxSam, 1.0f, j0, j1);
            hipLaunchKernelGGL(les::les_ransac_walk_kernel, dim3((n + 63) / 64), dim3(64), 0, cur_stream(c), b->d_units, rng, pl, b->rs, n, kRansacMaxSam, 0.95f, j0, j1, j2);
            j0 = j1;
        }
        break;
    case LES_HIP_PROPOSE_INIT:
        hipLaunchKernelGGL(les::les_init_labels_kernel, dim3(n), dim3(64), 0, cur_stream(c), b->d_units, lab, W, rng, pl, mind, maxd);
        break;
    default:
        return fail(LES_HIP_ERR_ARG, "unknown proposer kind %d", kind);
    }
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int les_hip_batch_wta(les_hip_ctx* c, const les_hip_batch* b, const les_hip_plane* planes, float* cur, const float* prop,
                      les_hip_plane* labels)
{
    if (c) (void)hipSetDevice(c->p.device);                 // HIP's current device is per host thread
    if (!c || !b || !planes || !cur || !prop || !labels) return fail(LES_HIP_ERR_ARG, "null argument");
    if (b->n == 0) return LES_HIP_OK;
    if (!b->d_targets) return fail(LES_HIP_ERR_ARG, "batch has no target table");
    hipLaunchKernelGGL(les::les_wta_kernel, dim3(b->n, b->wta_chunks), dim3(256), 0, cur_stream(c), b->d_targets, reinterpret_cast<const float4*>(planes),
                       cur, prop, reinterpret_cast<float4*>(labels), c->p.W);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int les_hip_batch_num_jobs(const les_hip_batch* b) { return b ? (b->march_ok ? b->nmgroups : b->njobs) : 0; }

int les_hip_batch_kernel_kind(const les_hip_ctx* c, const les_hip_batch* b, int mode)
{
    if (!c || !b || mode < 0 || mode > 1) return -1;
    return (b->march_ok && c->march && c->v[mode].march_ok) ? 1 : 0;
}

long long les_hip_batch_graph_nodes(const les_hip_batch* b) { return b ? b->graph_nodes : 0; }

int les_hip_batch_graph_offsets(const les_hip_batch* b, long long* offsets)
{
    if (!b || !offsets) return fail(LES_HIP_ERR_ARG, "null argument");
    std::copy(b->graph_off.begin(), b->graph_off.end(), offsets);
    return LES_HIP_OK;
}

int les_hip_batch_expansion_graph(les_hip_ctx* c, const les_hip_batch* b, int mode, const les_hip_plane* d_planes, const les_hip_plane* d_labels,
                                  const float* d_cur, const float* d_prop, float lambda, float th_smooth, float omega, float epsilon,
                                  float* d_payload, double* flow0_host)
{
    if (!c || !b || !d_planes || !d_labels || !d_cur || !d_prop || !d_payload) return fail(LES_HIP_ERR_ARG, "null argument");
    if (mode < 0 || mode > 1 || !c->v[mode].ipk) return fail(LES_HIP_ERR_ARG, "view %d was not supplied at creation", mode);
    if (b->n == 0) return LES_HIP_OK;
    HIPCHECK(hipSetDevice(c->p.device));                     // the calling host thread may be new (one thread per view)
    std::unique_lock<std::mutex> lk(c->mu);
    if (c->pw_omega != omega || c->pw_epsilon != epsilon || !c->d_pw_tab) {
        // initSmoothnessCoeff (LES/StereoEnergy.h:131-163): max(epsilon, exp(-|dI|_1 / omega)) in float
        std::vector<float> tab(766);
        for (int k = 0; k < 766; k++) tab[k] = std::max(epsilon, std::exp(-(float)k / omega));
        if (!c->d_pw_tab) HIPCHECK(hipMalloc((void**)&c->d_pw_tab, tab.size() * sizeof(float)));
        HIPCHECK(hipMemcpyAsync(c->d_pw_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, cur_stream(c)));
        HIPCHECK(hipStreamSynchronize(cur_stream(c)));
        c->pw_omega = omega; c->pw_epsilon = epsilon;
    }
    lk.unlock();
    const les::PairwiseParams pp{c->p.H, c->p.W, lambda, th_smooth};
    const les::GraphCell* cells = reinterpret_cast<const les::GraphCell*>(b->d_targets);
    const long long* offs = b->d_graph_off;
    const float4 *pl = reinterpret_cast<const float4*>(d_planes), *lab = reinterpret_cast<const float4*>(d_labels);
    const uint32_t* ipk = c->v[mode].ipk;
    const float* wtab = c->d_pw_tab;
    double* flow0 = b->d_flow0;
    hipLaunchKernelGGL(les::les_expansion_graph_kernel, dim3(b->n, b->wta_chunks), dim3(256), 0, cur_stream(c), cells, offs, pl, lab, d_cur, d_prop, ipk, wtab,
                       pp, d_payload, flow0);
    HIPCHECK(hipGetLastError());
    if (flow0_host) {
        std::vector<double> part((size_t)b->n * b->wta_chunks);
        HIPCHECK(hipMemcpyAsync(part.data(), b->d_flow0, part.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream(c)));
        HIPCHECK(hipStreamSynchronize(cur_stream(c)));
        for (int i = 0; i < b->n; i++) {
            double s = 0;
            for (int k = 0; k < b->wta_chunks; k++) s += part[(size_t)i * b->wta_chunks + k];
            flow0_host[i] = s;
        }
    }
    return LES_HIP_OK;
}

long long les_hip_batch_max_cell_nodes(const les_hip_batch* b)
{
    long long m = 0;
    if (b) for (const les_hip_rect& t : b->targets) m = std::max(m, (long long)std::max(0, t.w) * std::max(0, t.h));
    return m;
}

int les_hip_batch_solve_graphs(les_hip_ctx* c, const les_hip_batch* b, const float* d_payload, unsigned char* d_masks, int* d_status, double* d_flows)
{
    return les_hip_batch_solve_graphs_counted(c, b, d_payload, d_masks, d_status, d_flows, nullptr);
}

int les_hip_batch_solve_graphs_counted(les_hip_ctx* c, const les_hip_batch* b, const float* d_payload, unsigned char* d_masks, int* d_status, double* d_flows,
                                       int* d_unsolved_total)
{
    if (c) (void)hipSetDevice(c->p.device);                 // HIP's current device is per host thread
    if (!c || !b || !d_payload || !d_masks || !d_status) return fail(LES_HIP_ERR_ARG, "null argument");
    if (b->n == 0) return LES_HIP_OK;
    const long long maxn = les_hip_batch_max_cell_nodes(b);
    if (maxn > LES_HIP_MAXFLOW_MAX_NODES) return fail(LES_HIP_ERR_ARG, "les_hip_batch_solve_graphs: a cell of %lld nodes exceeds the limit of %d", maxn, LES_HIP_MAXFLOW_MAX_NODES);
    static_assert(LES_HIP_MAXFLOW_MAX_NODES == les::kMfMaxNodes, "header constant out of date");
    const int np = (int)((std::max<long long>(maxn, 1) + 7) / 8) * 8;
    const size_t lds = les::mf_lds_bytes(np);
#if !defined(LES_SIM)
    // The opt-in to more than 64 KB of dynamic LDS is a per-DEVICE function attribute: it is set once per context (a context is bound
    // to one device), under the context's mutex, with that device current -- a process-wide flag would leave the second GPU of a
    // process that drives two without it.  A failure is reported with its own message so that callers can cut on the host instead.
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->maxflow_lds_ready) {
            HIPCHECK(hipSetDevice(c->p.device));
            hipError_t arc = hipFuncSetAttribute(reinterpret_cast<const void*>(les::les_maxflow_kernel<2, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)les::mf_lds_bytes(les::kMfMaxNodes));
            if (arc == hipSuccess)
                arc = hipFuncSetAttribute(reinterpret_cast<const void*>(les::les_maxflow_kernel<5, 512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)les::mf_lds_bytes(les::kMfMaxNodes));
            if (arc != hipSuccess) return fail(LES_HIP_ERR_DEVICE, "les_hip_batch_solve_graphs: device max-flow unavailable on device %d (hipFuncSetAttribute max dynamic LDS: %s); cut on the host",
                                               c->p.device, hipGetErrorString(arc));
            c->maxflow_lds_ready = true;
        }
    }
#endif
    const les::GraphCellMf* cells = reinterpret_cast<const les::GraphCellMf*>(b->d_targets);
    int max_iter = les::kMfMaxIter;
    if (const char* ev = getenv("LES_HIP_MAXFLOW_MAX_ITER")) max_iter = std::max(0, atoi(ev));      // tests of the callers' host fall-back
    if (maxn <= 2048)
        hipLaunchKernelGGL((les::les_maxflow_kernel<2, 1024>), dim3(b->n), dim3(1024), lds, cur_stream(c), cells, b->d_graph_off, d_payload, np, max_iter, d_masks, d_status, d_flows, d_unsolved_total);
    else
        hipLaunchKernelGGL((les::les_maxflow_kernel<5, 512>), dim3(b->n), dim3(512), lds, cur_stream(c), cells, b->d_graph_off, d_payload, np, max_iter, d_masks, d_status, d_flows, d_unsolved_total);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

// ---- tiled device max-flow: cells of any size (les_maxflow_tiled.h) ------------------------------------------------------------
namespace {
// Cuts a w x h cell into tiles of at most kMtMaxTileNodes nodes and kMtMaxSide a side: the fewest tiles, then the most square ones.
void mt_partition(int w, int h, int& tw, int& th)
{
    long long best = -1;
    tw = th = 1;
    for (int ntx = (w + les::kMtMaxSide - 1) / les::kMtMaxSide; ntx <= w; ntx++) {
        const int cw = (w + ntx - 1) / ntx;
        const int chmax = std::min(les::kMtMaxSide, les::kMtMaxTileNodes / cw);
        if (chmax < 1) continue;
        const int nty = (h + chmax - 1) / chmax;
        const int ch = (h + nty - 1) / nty;
        const long long tiles = (long long)ntx * nty;
        const long long score = tiles * 1000 + std::abs(cw - ch);            // fewest tiles first
        if (best < 0 || score < best) { best = score; tw = cw; th = ch; }
        if (cw * 2 < ch) break;                                              // narrower tiles only get worse from here
    }
}
int mt_build_tiles(const les_hip_batch* b)
{
    std::lock_guard<std::mutex> lk(b->mt_mu);
    if (b->mt_ntiles >= 0) return LES_HIP_OK;
    std::vector<les::MtTile> tiles;
    std::vector<int> per_cell((size_t)std::max(1, b->n), 0);
    // tiles of one cell are neighbours in the launch order (they share halos in L2), cells in batch order
    for (int i = 0; i < b->n; i++) {
        const int w = std::max(0, b->targets[i].w), h = std::max(0, b->targets[i].h);
        if (w == 0 || h == 0) continue;
        int tw, th;
        mt_partition(w, h, tw, th);
        for (int y0 = 0; y0 < h; y0 += th)
            for (int x0 = 0; x0 < w; x0 += tw) {
                tiles.push_back(les::MtTile{i, x0, y0, std::min(tw, w - x0), std::min(th, h - y0), w, h, 0, b->graph_off[i], 0});
                per_cell[i]++;
            }
    }
    les::MtTile* d_t = nullptr;
    int* d_p = nullptr;
    if (hipMalloc((void**)&d_t, std::max<size_t>(1, tiles.size()) * sizeof(les::MtTile)) != hipSuccess ||
        hipMalloc((void**)&d_p, per_cell.size() * sizeof(int)) != hipSuccess ||
        (tiles.size() && hipMemcpy(d_t, tiles.data(), tiles.size() * sizeof(les::MtTile), hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemcpy(d_p, per_cell.data(), per_cell.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        if (d_t) (void)hipFree(d_t);
        if (d_p) (void)hipFree(d_p);
        return fail(LES_HIP_ERR_DEVICE, "les_hip_batch_solve_graphs_tiled: tile table allocation failed");
    }
    b->d_mt_tiles = d_t;
    b->d_mt_tiles_per_cell = d_p;
    b->mt_ntiles = (int)tiles.size();
    return LES_HIP_OK;
}
}  // namespace

long long les_hip_batch_tiled_workspace_bytes(const les_hip_batch* b)
{
    if (!b) return 0;
    return (long long)les::mt_layout(std::max<long long>(1, b->graph_nodes), std::max(1, b->n)).total;
}

int les_hip_batch_solve_graphs_tiled(les_hip_ctx* c, const les_hip_batch* b, const float* d_payload, unsigned char* d_masks, int* d_status, double* d_flows,
                                     void* d_workspace, long long workspace_bytes, int* launches_out, int* unsolved_out)
{
    les_hip_tiled_stats st;
    const int rc = les_hip_batch_solve_graphs_tiled_stats(c, b, d_payload, d_masks, d_status, d_flows, d_workspace, workspace_bytes, &st);
    if (launches_out) *launches_out = st.launches;
    if (unsolved_out) *unsolved_out = st.unsolved;
    return rc;
}

int les_hip_batch_solve_graphs_tiled_stats(les_hip_ctx* c, const les_hip_batch* b, const float* d_payload, unsigned char* d_masks, int* d_status, double* d_flows,
                                           void* d_workspace, long long workspace_bytes, les_hip_tiled_stats* stats)
{
    int launches_tmp = 0, unsolved_tmp = 0, handed_tmp = 0;
    int *launches_out = &launches_tmp, *unsolved_out = &unsolved_tmp, *handed_out = &handed_tmp;
    long long handed_nodes = 0;
    double host_ms = 0.0;
    struct Report {                                       // fills *stats on every exit path
        les_hip_tiled_stats* s; int *l, *u, *h; long long* hn; double* ms;
        ~Report() { if (s) { s->launches = *l; s->unsolved = *u; s->handed_cells = *h; s->handed_nodes = *hn; s->host_ms = *ms; } }
    } report{stats, launches_out, unsolved_out, handed_out, &handed_nodes, &host_ms};
    if (c) (void)hipSetDevice(c->p.device);                 // HIP's current device is per host thread
    if (!c || !b || !d_payload || !d_masks || !d_status || !d_workspace) return fail(LES_HIP_ERR_ARG, "null argument");
    if (b->n == 0) return LES_HIP_OK;
    if (workspace_bytes < les_hip_batch_tiled_workspace_bytes(b))
        return fail(LES_HIP_ERR_ARG, "les_hip_batch_solve_graphs_tiled: workspace of %lld bytes, %lld needed (les_hip_batch_tiled_workspace_bytes)", workspace_bytes,
                    les_hip_batch_tiled_workspace_bytes(b));
    if (((uintptr_t)d_workspace & 255) != 0) return fail(LES_HIP_ERR_ARG, "les_hip_batch_solve_graphs_tiled: the workspace must be 256-byte aligned");
    if (b->graph_nodes >= (1ll << 31) - 16) return fail(LES_HIP_ERR_ARG, "les_hip_batch_solve_graphs_tiled: %lld graph nodes exceed the 32-bit height range", b->graph_nodes);
    int rc = mt_build_tiles(b);
    if (rc) return rc;
#if !defined(LES_SIM)
    {
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->maxflow_tiled_lds_ready) {
            HIPCHECK(hipSetDevice(c->p.device));
            const hipError_t arc = hipFuncSetAttribute(reinterpret_cast<const void*>(les::les_maxflow_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)les::kMtLdsBytes);
            if (arc != hipSuccess) return fail(LES_HIP_ERR_DEVICE, "les_hip_batch_solve_graphs_tiled: device max-flow unavailable on device %d (hipFuncSetAttribute max dynamic LDS: %s); cut on the host",
                                               c->p.device, hipGetErrorString(arc));
            c->maxflow_tiled_lds_ready = true;
        }
    }
#endif
    hipStream_t st = cur_stream(c);
    les::MtArgs a;
    a.cells = reinterpret_cast<const les::GraphCellMf*>(b->d_targets);
    a.offsets = b->d_graph_off;
    a.payload = d_payload;
    a.tiles = b->d_mt_tiles;
    a.ws = reinterpret_cast<char*>(d_workspace);
    a.nodes = b->graph_nodes;
    a.ncells = b->n;
    a.K = 8; a.S = 12;                                     // short sweeps, a dozen of them between exact relabellings (measured: K = 8 / 16 / 32 / 64 -> 4.5 / 5.2 / 6.8 / 9.7 ms on a hard layer-1 lock-step)
    a.max_launches = 2000;
    a.K2 = a.K; a.S2 = a.S;
    if (const char* ev = getenv("LES_HIP_MAXFLOW_TILED_K")) a.K = a.K2 = std::max(1, atoi(ev));
    if (const char* ev = getenv("LES_HIP_MAXFLOW_TILED_S")) a.S = a.S2 = std::max(1, atoi(ev));
    if (const char* ev = getenv("LES_HIP_MAXFLOW_TILED_K2")) a.K2 = std::max(1, atoi(ev));
    if (const char* ev = getenv("LES_HIP_MAXFLOW_TILED_S2")) a.S2 = std::max(1, atoi(ev));
    if (const char* ev = getenv("LES_HIP_MAXFLOW_MAX_ITER")) a.max_launches = std::max(1, atoi(ev));      // tests of the callers' host fall-back
    a.masks = d_masks;
    a.status = d_status;
    a.flows = d_flows;
    MtHostLease lease{c, nullptr};
    rc = mt_host_acquire(c, &lease.m);
    if (rc) return rc;
    MtHost& hf = *lease.m;
    volatile int* flags = hf.h_flags;
    flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0;   // (nothing in flight writes them: the previous user of this MtHost has synchronised)
    a.host_flags = hf.d_flags;
    hipLaunchKernelGGL(les::les_maxflow_tiled_init_kernel, dim3((b->n + 255) / 256), dim3(256), 0, st, a.ws, a.nodes, a.ncells, b->d_mt_tiles_per_cell, d_status, d_flows, a.host_flags);
    HIPCHECK(hipGetLastError());
    if (b->mt_ntiles == 0) {                                // every target rect is empty: the init kernel has closed all cells
        HIPCHECK(hipStreamSynchronize(st));
        return LES_HIP_OK;
    }
    // Hand-over policy (les_maxflow_tiled.h, host/ResidualCut.h): after `hand_after` launches, as soon as at most `hand_cells` cells of at most
    // `hand_nodes` nodes in total are still open, the host cores finish them from their residual graphs.  LES_HIP_MAXFLOW_HANDOVER=0 switches it off
    // (every cell is then cut by launches alone, as in round 5); ..._AFTER / _CELLS / _NODES override the thresholds (A/B measurements).
    // A lock-step that is still running after `hand_all_after` launches hands over whatever is open (the launches would go on for hundreds more: the
    // status-1 exit of round 5 at 2 000 launches remains for a hand-over that is switched off).
    int hand_after = 28, hand_cells = 8, hand_all_after = 300;
    long long hand_nodes = 400000;
    bool hand = true;
    if (const char* ev = getenv("LES_HIP_MAXFLOW_HANDOVER")) hand = atoi(ev) != 0;
    if (const char* ev = getenv("LES_HIP_MAXFLOW_HANDOVER_AFTER")) hand_after = std::max(1, atoi(ev));
    if (const char* ev = getenv("LES_HIP_MAXFLOW_HANDOVER_CELLS")) hand_cells = std::max(1, atoi(ev));
    if (const char* ev = getenv("LES_HIP_MAXFLOW_HANDOVER_NODES")) hand_nodes = std::max(1ll, atoll(ev));
    if (const char* ev = getenv("LES_HIP_MAXFLOW_HANDOVER_ALL_AFTER")) hand_all_after = std::max(1, atoi(ev));
    // Launches are enqueued in groups; after each group the host reads "cells done" (the only synchronisation).  Launches that come
    // after the last cell finished return at once.
    int total = 0, group = 12, handed = 0;
    for (;;) {
        for (int i = 0; i < group; i++)
            hipLaunchKernelGGL(les::les_maxflow_tiled_kernel, dim3(b->mt_ntiles), dim3(les::kMtThreads), les::kMtLdsBytes, st, a);
        total += group;
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(st));
        const int done = flags[0];
        if (done >= b->n) break;
        const bool everything = total >= hand_all_after;
        if (hand && total >= hand_after && (b->n - done <= hand_cells || everything)) {
            const int want_cells = everything ? b->n - done : hand_cells;
            const long long want_nodes = everything ? b->graph_nodes : std::min<long long>(hand_nodes, b->graph_nodes);
            rc = mt_host_stage(&hf, want_nodes, want_cells);
            if (rc) return rc;
            les::MtHandArgs ha;
            ha.tiles = b->d_mt_tiles; ha.ws = a.ws; ha.nodes = a.nodes; ha.ncells = a.ncells; ha.cells = a.cells;
            ha.max_cells = want_cells; ha.max_nodes = want_nodes;
            ha.list = reinterpret_cast<les::MtHandCell*>(hf.d_stage + hf.off_list());
            ha.rc8 = reinterpret_cast<float*>(hf.d_stage);
            ha.ex = reinterpret_cast<float*>(hf.d_stage + hf.off_ex());
            ha.hmasks = reinterpret_cast<const uint8_t*>(hf.d_stage + hf.off_masks());
            ha.hflows = reinterpret_cast<const double*>(hf.d_stage + hf.off_flows());
            ha.masks = d_masks; ha.status = d_status; ha.flows = d_flows; ha.host_flags = hf.d_flags;
            hipLaunchKernelGGL(les::les_maxflow_tiled_collect_kernel, dim3(1), dim3(64), 0, st, ha);
            hipLaunchKernelGGL(les::les_maxflow_tiled_pack_kernel, dim3(b->mt_ntiles), dim3(les::kMtThreads), 0, st, ha);      // (nothing to pack when the policy said no)
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipStreamSynchronize(st));
            handed = flags[2];
            if (handed > 0) {
                handed_nodes = flags[3];
                const auto h0 = std::chrono::steady_clock::now();
                const float* h_rc8 = reinterpret_cast<const float*>(hf.h_stage);
                const float* h_ex = reinterpret_cast<const float*>(hf.h_stage + hf.off_ex());
                uint8_t* h_masks = reinterpret_cast<uint8_t*>(hf.h_stage + hf.off_masks());
                const les::MtHandCell* list = reinterpret_cast<const les::MtHandCell*>(hf.h_stage + hf.off_list());
                double* hflows = reinterpret_cast<double*>(hf.h_stage + hf.off_flows());
                const std::vector<les_hip_rect>& tg = b->targets;
                const char* sev = getenv("LES_HIP_MAXFLOW_HANDOVER_SOLVER");      // 1 (default): FIFO push-relabel; 0: search trees with the push-relabel continuation (measured slower on what is handed over: tools/residual_probe.py)
                const int solver = sev ? atoi(sev) : 1;
                if (const char* dump = getenv("LES_HIP_MAXFLOW_HANDOVER_DUMP")) {      // tooling: the residual graphs as handed over (tools/residual_probe.py)
                    if (FILE* f = fopen(dump, "wb")) {
                        const long long hn = flags[3];
                        fwrite(&handed, sizeof(int), 1, f); fwrite(&hn, sizeof(long long), 1, f);
                        for (int q = 0; q < handed; q++) { const int wh[2] = {tg[(size_t)list[q].cell].w, tg[(size_t)list[q].cell].h}; fwrite(wh, sizeof(int), 2, f); fwrite(&list[q].hoff, sizeof(long long), 1, f); }
                        fwrite(h_rc8, sizeof(float), (size_t)hn * 8, f); fwrite(h_ex, sizeof(float), (size_t)hn, f);
                        fclose(f);
                    }
                }
                // one host thread per cell, at most as many as the process may keep busy (a persistent team owned by the calling thread); the large
                // cells split their phases over row bands as the host cuts do
                const int team = std::max(1, std::min(handed, les_host::cpuBudget()));
                std::atomic<int> next{0};
                les_host::BandPool::mine().run(team, [&](int) {
                    for (int q = next.fetch_add(1); q < handed; q = next.fetch_add(1)) {
                        const int cell = list[q].cell;
                        const int w = tg[(size_t)cell].w, h = tg[(size_t)cell].h;
                        hflows[q] = les_host::finishResidualCut(h_rc8 + 8 * list[q].hoff, h_ex + list[q].hoff, w, h, h_masks + list[q].hoff, les_host::residualBands(w, h), solver);
                    }
                });
                host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
                hipLaunchKernelGGL(les::les_maxflow_tiled_unpack_kernel, dim3(b->mt_ntiles), dim3(256), 0, st, ha);
                HIPCHECK(hipGetLastError());
                // the staging belongs to the next caller as soon as this MtHost is back in the pool: the unpack kernel must have read it
                HIPCHECK(hipStreamSynchronize(st));
                if (done + handed >= b->n) break;
            }
        }
        if (total >= a.max_launches + group) return fail(LES_HIP_ERR_DEVICE, "les_hip_batch_solve_graphs_tiled: %d of %d cells still open after %d launches", b->n - done, b->n, total);
        group = 16;
    }
    *launches_out = total;
    *unsolved_out = flags[1];
    *handed_out = handed;
    return LES_HIP_OK;
}

int les_hip_batch_apply_masks(les_hip_ctx* c, const les_hip_batch* b, const les_hip_plane* d_planes, const unsigned char* d_masks, float* d_cur,
                              const float* d_prop, les_hip_plane* d_labels)
{
    if (c) (void)hipSetDevice(c->p.device);                 // HIP's current device is per host thread
    if (!c || !b || !d_planes || !d_masks || !d_cur || !d_prop || !d_labels) return fail(LES_HIP_ERR_ARG, "null argument");
    if (b->n == 0) return LES_HIP_OK;
    const les::GraphCell* cells = reinterpret_cast<const les::GraphCell*>(b->d_targets);
    const long long* offs = b->d_graph_off;
    const float4* pl = reinterpret_cast<const float4*>(d_planes);
    float4* lab = reinterpret_cast<float4*>(d_labels);
    hipLaunchKernelGGL(les::les_apply_masks_kernel, dim3(b->n, b->wta_chunks), dim3(256), 0, cur_stream(c), cells, offs, pl, d_masks, d_cur, d_prop, lab, c->p.W);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int les_hip_batch_run(les_hip_ctx* c, const les_hip_batch* b, int mode, const les_hip_plane* planes, int planes_on_device,
                      float* out_dev, int check)
{
    if (c) (void)hipSetDevice(c->p.device);                 // HIP's current device is per host thread
    if (!c || !b || !out_dev || (b->n > 0 && !planes)) return fail(LES_HIP_ERR_ARG, "null argument");
    if (b->R != c->R) return fail(LES_HIP_ERR_ARG, "batch was prepared for a different context");
    const float4* d_planes = reinterpret_cast<const float4*>(planes);
    if (!planes_on_device) {
        int rc = ensure_planes(c, (size_t)b->n);
        if (rc) return rc;
        HIPCHECK(hipMemcpyAsync(c->d_planes, planes, (size_t)b->n * sizeof(float4), hipMemcpyHostToDevice, cur_stream(c)));
        d_planes = c->d_planes;
    }
    if (b->march_ok && mode >= 0 && mode <= 1 && c->march && c->v[mode].march_ok) {
        if (!c->naive) return launch_march(c, b->mentry, mode, b->d_mjobs, b->nmgroups, d_planes, out_dev, check, cur_stream(c));
        {
            std::lock_guard<std::mutex> lk(c->mu);
            if (!b->d_raw[mode]) HIPCHECK(hipMalloc((void**)&b->d_raw[mode], (size_t)std::max<long long>(b->raw_floats, 1) * sizeof(float)));
        }
        const RawPatches rp{b->d_rawcalls, b->d_raw_off, b->d_raw[mode], b->n, b->raw_chunks};
        return launch_march(c, b->mentry, mode, b->d_mjobs, b->nmgroups, d_planes, out_dev, check, cur_stream(c), &rp);
    }
    return launch_strips(c, mode, b->d_jobs, b->njobs, d_planes, out_dev, check, cur_stream(c));
}

int les_hip_unary_batch(les_hip_ctx* c, int mode, int n, const les_hip_rect* frs, const les_hip_rect* trs,
                        const les_hip_plane* planes, float* cost_map, int check)
{
    if (!c || !cost_map) return fail(LES_HIP_ERR_ARG, "null argument");
    les_hip_batch* b = nullptr;
    int rc = les_hip_batch_create(c, n, frs, trs, 0, &b);
    if (rc) return rc;
    rc = les_hip_batch_run(c, b, mode, planes, 0, c->d_map, check);
    if (rc == LES_HIP_OK) {
        // copy back only the target rects (the reference writes nothing else, LES/CostVolumeEnergy.h:169-171)
        for (int i = 0; i < n && rc == LES_HIP_OK; i++) {
            const les_hip_rect& t = trs[i];
            if (t.w <= 0 || t.h <= 0) continue;
            size_t off = (size_t)t.y * c->p.W + t.x;
            hipError_t e = hipMemcpy2DAsync(cost_map + off, (size_t)c->p.W * sizeof(float), c->d_map + off, (size_t)c->p.W * sizeof(float),
                                            (size_t)t.w * sizeof(float), (size_t)t.h, hipMemcpyDeviceToHost, cur_stream(c));
            if (e != hipSuccess) rc = fail(LES_HIP_ERR_DEVICE, "hipMemcpy2DAsync failed: %s", hipGetErrorString(e));
        }
        if (rc == LES_HIP_OK && hipStreamSynchronize(cur_stream(c)) != hipSuccess) rc = fail(LES_HIP_ERR_DEVICE, "stream synchronize failed");
    }
    les_hip_batch_destroy(b);
    return rc;
}

int les_hip_scratch_create(les_hip_ctx* c, les_hip_scratch** out)
{
    if (!c || !out) return fail(LES_HIP_ERR_ARG, "null argument");
    *out = nullptr;
    HIPCHECK(hipSetDevice(c->p.device));
    les_hip_scratch* s = new les_hip_scratch();
    s->c = c;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&s->d_plane, sizeof(float4)) != hipSuccess || hipHostMalloc((void**)&s->h_plane, sizeof(float4), hipHostMallocDefault) != hipSuccess) {
        les_hip_scratch_destroy(s);
        return fail(LES_HIP_ERR_DEVICE, "scratch allocation failed");
    }
    *out = s;
    return LES_HIP_OK;
}

void les_hip_scratch_destroy(les_hip_scratch* s)
{
    if (!s) return;
    if (s->c) (void)hipSetDevice(s->c->p.device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    for (auto& e : s->cache) if (e.d_jobs) (void)hipFree(e.d_jobs);
    if (s->d_tile) (void)hipFree(s->d_tile);
    if (s->h_tile) (void)hipHostFree(s->h_tile);
    if (s->d_raw) (void)hipFree(s->d_raw);
    if (s->d_rawcall) (void)hipFree(s->d_rawcall);
    if (s->d_raw_off) (void)hipFree(s->d_raw_off);
    if (s->d_plane) (void)hipFree(s->d_plane);
    if (s->h_plane) (void)hipHostFree(s->h_plane);
    delete s;
}

int les_hip_unary_one_scratch(les_hip_ctx* c, les_hip_scratch* s, int mode, const les_hip_rect* fr, const les_hip_rect* tr,
                              const les_hip_plane* plane, float* costs, int row_stride, int check)
{
    if (!c || !s || !fr || !tr || !plane || !costs) return fail(LES_HIP_ERR_ARG, "null argument");
    if (s->c != c) return fail(LES_HIP_ERR_ARG, "scratch belongs to another context");
    if (mode < 0 || mode > 1 || !c->v[mode].stats || (c->naive ? !c->v[1 - mode].feat : !c->v[mode].vol))
        return fail(LES_HIP_ERR_ARG, "view %d was not supplied at creation", mode);
    HIPCHECK(hipSetDevice(c->p.device));                  // HIP's current device is per host thread
    int rc = check_rects(c, *fr, *tr);
    if (rc) return rc;
    if (tr->w <= 0 || tr->h <= 0) return LES_HIP_OK;
    // ---- job table of this rect pair (built and uploaded the first time it is seen; 16 pairs are remembered)
    const int want_march = (c->march && c->v[mode].march_ok) ? 1 : 0;      // per view: the march kernel needs a finite, bounded volume
    les_hip_scratch::Entry* e = nullptr;
    for (auto& x : s->cache)
        if (x.want_march == want_march && !memcmp(&x.f, fr, sizeof *fr) && !memcmp(&x.t, tr, sizeof *tr)) { e = &x; break; }
    if (!e) {
        std::vector<les::Job> jobs;
        bool mok = false;
        const MarchEntry* me = nullptr;
        if (want_march) build_march_jobs(c, 1, fr, tr, 0, jobs, mok, me);
        const bool use_march = want_march && mok && !jobs.empty();
        if (!use_march) {
            rc = build_jobs(c, 1, fr, tr, 0, jobs);
            if (rc) return rc;
            me = nullptr;
        }
        for (auto& j : jobs) {                           // compact tile: row stride = target width, origin = target corner
            j.out_off = (long long)(j.ty0 - tr->y) * tr->w + (j.tx0 - tr->x);
            j.out_stride = tr->w;
        }
        les_hip_scratch::Entry ne{*fr, *tr, want_march, me, (int)jobs.size(), me ? (int)(jobs.size() / me->NJ) : 0, nullptr, 0};
        if (s->cache.size() >= 16) {                     // evict the least recently used pair
            size_t k = 0;
            for (size_t i = 1; i < s->cache.size(); i++) if (s->cache[i].stamp < s->cache[k].stamp) k = i;
            HIPCHECK(hipStreamSynchronize(s->stream));
            if (s->cache[k].d_jobs) HIPCHECK(hipFree(s->cache[k].d_jobs));
            s->cache.erase(s->cache.begin() + (long)k);
        }
        HIPCHECK(hipMalloc((void**)&ne.d_jobs, jobs.size() * sizeof(les::Job)));
        if (hipMemcpy(ne.d_jobs, jobs.data(), jobs.size() * sizeof(les::Job), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(ne.d_jobs);
            return fail(LES_HIP_ERR_DEVICE, "upload of the job table failed");
        }
        s->cache.push_back(ne);
        e = &s->cache.back();
    }
    e->stamp = ++s->clock;
    const size_t need = (size_t)tr->w * tr->h;
    if (need > s->tile_cap) {
        HIPCHECK(hipStreamSynchronize(s->stream));
        if (s->d_tile) HIPCHECK(hipFree(s->d_tile));
        if (s->h_tile) HIPCHECK(hipHostFree(s->h_tile));
        s->d_tile = nullptr; s->h_tile = nullptr; s->tile_cap = 0;
        const size_t cap = std::max(need, (size_t)256 * 256);
        HIPCHECK(hipMalloc((void**)&s->d_tile, cap * sizeof(float)));
        HIPCHECK(hipHostMalloc((void**)&s->h_tile, cap * sizeof(float), hipHostMallocDefault));
        s->tile_cap = cap;
    }
    *s->h_plane = make_float4(plane->a, plane->b, plane->c, plane->v);
    HIPCHECK(hipMemcpyAsync(s->d_plane, s->h_plane, sizeof(float4), hipMemcpyHostToDevice, s->stream));
    if (e->march && c->naive) {
        // raw-cost patch of this filterRect (the one-entry call table is rewritten when the rect changes; everything is ordered on the scratch's stream)
        const size_t rneed = (size_t)fr->w * fr->h;
        if (rneed > s->raw_cap || !s->d_rawcall) {
            HIPCHECK(hipStreamSynchronize(s->stream));
            if (s->d_raw) HIPCHECK(hipFree(s->d_raw));
            s->d_raw = nullptr; s->raw_cap = 0;
            const size_t cap = std::max(rneed, (size_t)256 * 256);
            HIPCHECK(hipMalloc((void**)&s->d_raw, cap * sizeof(float)));
            s->raw_cap = cap;
            if (!s->d_rawcall) {
                HIPCHECK(hipMalloc((void**)&s->d_rawcall, sizeof(les::RawCall)));
                HIPCHECK(hipMalloc((void**)&s->d_raw_off, sizeof(long long)));
                const long long zero = 0;
                HIPCHECK(hipMemcpy(s->d_raw_off, &zero, sizeof zero, hipMemcpyHostToDevice));
            }
            s->raw_f = les_hip_rect{-1, -1, -1, -1};
        }
        if (memcmp(&s->raw_f, fr, sizeof *fr)) {
            const les::RawCall call{fr->x, fr->y, fr->w, fr->h, 0};
            HIPCHECK(hipStreamSynchronize(s->stream));
            HIPCHECK(hipMemcpy(s->d_rawcall, &call, sizeof call, hipMemcpyHostToDevice));
            s->raw_f = *fr;
        }
        const RawPatches rp{s->d_rawcall, s->d_raw_off, s->d_raw, 1, (int)std::min<size_t>(1024, (rneed + 4095) / 4096)};
        rc = launch_march(c, static_cast<const MarchEntry*>(e->march), mode, e->d_jobs, e->ngroups, s->d_plane, s->d_tile, check, s->stream, &rp);
    }
    else if (e->march) rc = launch_march(c, static_cast<const MarchEntry*>(e->march), mode, e->d_jobs, e->ngroups, s->d_plane, s->d_tile, check, s->stream);
    else rc = launch_strips(c, mode, e->d_jobs, e->njobs, s->d_plane, s->d_tile, check, s->stream);
    if (rc) return rc;
    HIPCHECK(hipMemcpyAsync(s->h_tile, s->d_tile, need * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHECK(hipStreamSynchronize(s->stream));
    // costs(targetRect - filterRect.tl()), LES/CostVolumeEnergy.h:169
    float* dst = costs + (size_t)(tr->y - fr->y) * row_stride + (tr->x - fr->x);
    for (int y = 0; y < tr->h; y++) memcpy(dst + (size_t)y * row_stride, s->h_tile + (size_t)y * tr->w, (size_t)tr->w * sizeof(float));
    return LES_HIP_OK;
}

int les_hip_unary_one(les_hip_ctx* c, int mode, const les_hip_rect* fr, const les_hip_rect* tr, const les_hip_plane* plane,
                      float* costs, int row_stride, int check)
{
    if (!c || !fr || !tr || !plane || !costs) return fail(LES_HIP_ERR_ARG, "null argument");
    // One scratch per (calling thread, context), created on the thread's first call and owned by the context.  The thread-local entry
    // is keyed by the context's generation id (never by its address); when the thread exits, its scratches go back to their contexts'
    // idle lists -- if those contexts are still alive -- so short-lived caller threads recycle a bounded set instead of piling up.
    struct Mine {
        std::vector<std::pair<unsigned long long, les_hip_scratch*>> v;
        ~Mine() { for (auto& e : v) release_hidden_scratch(e.first, e.second); }
    };
    thread_local Mine mine;
    les_hip_scratch* s = nullptr;
    for (auto& e : mine.v) if (e.first == c->gen) { s = e.second; break; }
    if (!s) {
        {
            std::lock_guard<std::mutex> lk(c->mu);
            if (!c->idle_scratch.empty()) { s = c->idle_scratch.back(); c->idle_scratch.pop_back(); }
        }
        if (!s) {
            int rc = les_hip_scratch_create(c, &s);
            if (rc) return rc;
            std::lock_guard<std::mutex> lk(c->mu);
            c->own_scratch.push_back(s);
        }
        mine.v.emplace_back(c->gen, s);
    }
    return les_hip_unary_one_scratch(c, s, mode, fr, tr, plane, costs, row_stride, check);
}

int les_hip_wta_update(les_hip_ctx* c, int n, const les_hip_rect* rects, const les_hip_plane* planes, int planes_on_device,
                       float* cur, const float* prop, les_hip_plane* labels)
{
    if (!c || n < 0 || (n > 0 && (!rects || !planes || !cur || !prop || !labels))) return fail(LES_HIP_ERR_ARG, "null argument");
    if (n == 0) return LES_HIP_OK;
    for (int i = 0; i < n; i++)
        if (rects[i].x < 0 || rects[i].y < 0 || rects[i].w < 0 || rects[i].h < 0 || rects[i].x + rects[i].w > c->p.W || rects[i].y + rects[i].h > c->p.H)
            return fail(LES_HIP_ERR_ARG, "rect outside the image");
    if ((size_t)n > c->wta_cap) {
        if (c->d_wta) HIPCHECK(hipFree(c->d_wta));
        c->d_wta = nullptr; c->wta_cap = 0;
        HIPCHECK(hipMalloc((void**)&c->d_wta, std::max<size_t>(n, 1024) * sizeof(les::WtaJob)));
        c->wta_cap = std::max<size_t>(n, 1024);
    }
    static_assert(sizeof(les::WtaJob) == sizeof(les_hip_rect), "rect layout");
    HIPCHECK(hipMemcpyAsync(c->d_wta, rects, (size_t)n * sizeof(les::WtaJob), hipMemcpyHostToDevice, cur_stream(c)));
    const float4* d_planes = reinterpret_cast<const float4*>(planes);
    if (!planes_on_device) {
        if ((size_t)n > c->wta_planes_cap) {
            if (c->d_wta_planes) HIPCHECK(hipFree(c->d_wta_planes));
            c->d_wta_planes = nullptr; c->wta_planes_cap = 0;
            HIPCHECK(hipMalloc((void**)&c->d_wta_planes, std::max<size_t>(n, 1024) * sizeof(float4)));
            c->wta_planes_cap = std::max<size_t>(n, 1024);
        }
        HIPCHECK(hipMemcpyAsync(c->d_wta_planes, planes, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, cur_stream(c)));
        d_planes = c->d_wta_planes;
    }
    int max_area = 1;
    for (int i = 0; i < n; i++) max_area = std::max(max_area, rects[i].w * rects[i].h);
    const int chunks = std::min(32, std::max(1, (max_area + 4095) / 4096));
    hipLaunchKernelGGL(les::les_wta_kernel, dim3(n, chunks), dim3(256), 0, cur_stream(c), c->d_wta, d_planes, cur, prop,
                       reinterpret_cast<float4*>(labels), c->p.W);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int les_hip_fill_out_of_view(float* vol, int D, int H, int W, int mode, int device, void* stream)
{
    if (!vol || D <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 1) return fail(LES_HIP_ERR_ARG, "bad argument");
    HIPCHECK(hipSetDevice(device));
    hipLaunchKernelGGL(les::les_fill_out_of_view_kernel, dim3((W + 255) / 256, H, D), dim3(256), 0, (hipStream_t)stream, vol, D, H, W, mode);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int les_hip_convert_volume_l2r(const float* src, float* dst, int D, int H, int W, int device, void* stream)
{
    if (!src || !dst || src == dst || D <= 0 || H <= 0 || W <= 0) return fail(LES_HIP_ERR_ARG, "bad argument");
    HIPCHECK(hipSetDevice(device));
    hipLaunchKernelGGL(les::les_convert_l2r_kernel, dim3((W + 255) / 256, H, D), dim3(256), 0, (hipStream_t)stream, src, dst, D, H, W);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

// ---- dual-view post-processing (LES/PMStereoBase.h:111-256)
namespace {
struct PostScratch {
    float* disp[2] = {nullptr, nullptr};
    uint8_t *fail = nullptr, *failb = nullptr, *fail2 = nullptr;
    float4* copy = nullptr;
    float* wtab = nullptr;
    PostScratch() = default;
    PostScratch(const PostScratch&) = delete;              // launches must capture the raw pointers, not this owner
    PostScratch& operator=(const PostScratch&) = delete;
    ~PostScratch()
    {
        for (float* d : disp) if (d) (void)hipFree(d);
        if (fail) (void)hipFree(fail);
        if (failb) (void)hipFree(failb);
        if (fail2) (void)hipFree(fail2);
        if (copy) (void)hipFree(copy);
        if (wtab) (void)hipFree(wtab);
    }
};
int post_disparities(les_hip_ctx* c, PostScratch& ps, const les_hip_plane* const labels[2])
{
    const int H = c->p.H, W = c->p.W;
    const size_t P = (size_t)H * W;
    for (int m = 0; m < 2; m++) {
        if (!ps.disp[m]) HIPCHECK(hipMalloc((void**)&ps.disp[m], P * sizeof(float)));
        const float4* lab = (const float4*)labels[m];
        float* disp = ps.disp[m];
        hipLaunchKernelGGL(les::les_disparity_kernel, dim3((W + 255) / 256, H), dim3(256), 0, cur_stream(c), lab, disp, H, W);
    }
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}
}  // namespace

int les_hip_consistency_check(les_hip_ctx* c, const les_hip_plane* d_labelsL, const les_hip_plane* d_labelsR, float threshold,
                              unsigned char* d_failL, unsigned char* d_failR)
{
    if (!c || !d_labelsL || !d_labelsR || !d_failL || !d_failR) return fail(LES_HIP_ERR_ARG, "null argument");
    HIPCHECK(hipSetDevice(c->p.device));
    const int H = c->p.H, W = c->p.W;
    PostScratch ps;
    const les_hip_plane* labels[2] = {d_labelsL, d_labelsR};
    int rc = post_disparities(c, ps, labels);
    if (rc) return rc;
    unsigned char* out[2] = {d_failL, d_failR};
    for (int m = 0; m < 2; m++) {
        const float *d_self = ps.disp[m], *d_other = ps.disp[1 - m];
        unsigned char* o = out[m];
        const float sign = m ? -1.0f : 1.0f;
        hipLaunchKernelGGL(les::les_lr_check_kernel, dim3((W + 255) / 256, H), dim3(256), 0, cur_stream(c), d_self, d_other, o, H, W, sign, threshold);
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    return LES_HIP_OK;
}

int les_hip_post_process(les_hip_ctx* c, les_hip_plane* d_labelsL, les_hip_plane* d_labelsR, float threshold, float omega)
{
    if (!c || !d_labelsL || !d_labelsR) return fail(LES_HIP_ERR_ARG, "null argument");
    if (!c->v[0].ipk || !c->v[1].ipk) return fail(LES_HIP_ERR_ARG, "post-processing needs both views' images");
    const int windR = c->p.windR;
    if (windR > 31) return fail(LES_HIP_ERR_UNSUPPORTED, "weighted median window radius %d > 31", windR);
    HIPCHECK(hipSetDevice(c->p.device));
    const int H = c->p.H, W = c->p.W;
    const size_t P = (size_t)H * W;
    PostScratch ps;
    les_hip_plane* labels[2] = {d_labelsL, d_labelsR};
    int rc = post_disparities(c, ps, labels);
    if (rc) return rc;
    HIPCHECK(hipMalloc((void**)&ps.fail, P));
    HIPCHECK(hipMalloc((void**)&ps.failb, 2 * P));
    HIPCHECK(hipMalloc((void**)&ps.fail2, P));
    HIPCHECK(hipMalloc((void**)&ps.copy, P * sizeof(float4)));
    {
        // computePatchWeight (LES/StereoEnergy.h:251-257): exp(-|dI|_1 / omega) in float; |dI|_1 of 8-bit colours is an integer
        std::vector<float> tab(766);
        for (int k = 0; k < 766; k++) tab[k] = std::exp(-(float)k / omega);
        HIPCHECK(hipMalloc((void**)&ps.wtab, tab.size() * sizeof(float)));
        HIPCHECK(hipMemcpyAsync(ps.wtab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, cur_stream(c)));
        HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    }
    const dim3 g((W + 255) / 256, H), b(256);
    // both fail masks come from the labels before any of them is modified (LES/PMStereoBase.h:158-164)
    uint8_t *failm = ps.fail, *fail2 = ps.fail2;
    const float* wtab = ps.wtab;
    float4* copy = ps.copy;
    for (int m = 0; m < 2; m++) {
        const float *d_self = ps.disp[m], *d_other = ps.disp[1 - m];
        uint8_t* failb = ps.failb + m * P;
        float4* lab = (float4*)labels[m];
        const float sign = m ? -1.0f : 1.0f;
        hipLaunchKernelGGL(les::les_lr_check_kernel, g, b, 0, cur_stream(c), d_self, d_other, failm, H, W, sign, threshold);
        hipLaunchKernelGGL(les::les_fail_dilate_kernel, g, b, 0, cur_stream(c), failm, failb, fail2, H, W);
        hipLaunchKernelGGL(les::les_nn_fill_kernel, g, b, 0, cur_stream(c), failb, fail2, lab, H, W);
    }
    for (int m = 0; m < 2; m++) {
        HIPCHECK(hipMemcpyAsync(ps.copy, labels[m], P * sizeof(float4), hipMemcpyDeviceToDevice, cur_stream(c)));
        const dim3 gp(W, H);
        const int area = (2 * windR + 1) * (2 * windR + 1);
        const uint8_t* failb = ps.failb + m * P;
        float4* lab = (float4*)labels[m];
        const uint32_t* ipk = c->v[m].ipk;
        if (area <= 256)
            hipLaunchKernelGGL((les::les_weighted_median_kernel<256, 64>), gp, dim3(64), 0, cur_stream(c), failb, copy, lab, ipk, wtab, H, W, windR);
        else if (area <= 1024)
            hipLaunchKernelGGL((les::les_weighted_median_kernel<1024, 256>), gp, dim3(256), 0, cur_stream(c), failb, copy, lab, ipk, wtab, H, W, windR);
        else if (area <= 2048)
            hipLaunchKernelGGL((les::les_weighted_median_kernel<2048, 256>), gp, dim3(256), 0, cur_stream(c), failb, copy, lab, ipk, wtab, H, W, windR);
        else
            hipLaunchKernelGGL((les::les_weighted_median_kernel<4096, 256>), gp, dim3(256), 0, cur_stream(c), failb, copy, lab, ipk, wtab, H, W, windR);
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    return LES_HIP_OK;
}

int les_hip_calib_copy(const float* d_src, float* d_dst, size_t n, int device, void* stream)
{
    if (!d_src || !d_dst || n == 0) return fail(LES_HIP_ERR_ARG, "bad argument");
    HIPCHECK(hipSetDevice(device));
    hipLaunchKernelGGL(les::les_calib_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_src, d_dst, n);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

// ---- multi-GPU tile exchange (include/localexp_hip.h): plan = rect table of every rank, slot layout; pack / unpack kernels; the
// all-gather itself over an ncclComm_t handed in by the host (RCCL is resolved at run time: a single-GPU user needs no librccl)
struct les_hip_exchange {
    les_hip_ctx* c = nullptr;
    int rank = 0, world = 1, nrects = 0, own_first = 0, own_n = 0, lmax = 0, max_px = 0;
    les::XchgRect* d_rects = nullptr;
    float* d_send = nullptr; float* d_recv = nullptr;       // buffers of les_hip_exchange_tiles, allocated on its first call
};

namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
nccl_allgather_fn load_nccl_allgather()
{
    static std::once_flag once;
    static nccl_allgather_fn fn = nullptr;
#if !defined(LES_SIM)
    std::call_once(once, [] {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                fn = reinterpret_cast<nccl_allgather_fn>(dlsym(h, "ncclAllGather"));
                if (fn) break;
            }
        }
    });
#endif
    return fn;
}
int xchg_chunks(int max_px) { return std::max(1, std::min(64, max_px / 4096)); }
}  // namespace

int les_hip_exchange_create(les_hip_ctx* c, int rank, int world, int n, const les_hip_rect* rects, const int* first, les_hip_exchange** out)
{
    if (!c || !out || world < 1 || rank < 0 || rank >= world || n < 0 || (n > 0 && !rects) || !first) return fail(LES_HIP_ERR_ARG, "les_hip_exchange_create: bad argument");
    *out = nullptr;
    if (first[0] != 0 || first[world] != n) return fail(LES_HIP_ERR_ARG, "les_hip_exchange_create: first[] must run from 0 to n");
    std::vector<les::XchgRect> tab((size_t)n);
    long long lmax = 0;
    int max_px = 0;
    for (int r = 0; r < world; r++) {
        if (first[r + 1] < first[r]) return fail(LES_HIP_ERR_ARG, "les_hip_exchange_create: first[] must not decrease");
        long long off = 0;
        for (int i = first[r]; i < first[r + 1]; i++) {
            const les_hip_rect& q = rects[i];
            if (q.w < 0 || q.h < 0 || (q.w > 0 && q.h > 0 && (q.x < 0 || q.y < 0 || q.x + q.w > c->p.W || q.y + q.h > c->p.H)))
                return fail(LES_HIP_ERR_ARG, "les_hip_exchange_create: rect %d outside the image", i);
            tab[(size_t)i] = les::XchgRect{q.x, q.y, q.w, q.h, (int)off, r};
            off += (long long)q.w * q.h;
            max_px = std::max(max_px, q.w * q.h);
        }
        lmax = std::max(lmax, off);
    }
    lmax = (lmax + 3) / 4 * 4;                                  // the cost block of a slot stays 16-byte aligned
    if (lmax * 5 * world >= (1ll << 31)) return fail(LES_HIP_ERR_ARG, "les_hip_exchange_create: exchange buffer too large");
    HIPCHECK(hipSetDevice(c->p.device));
    les_hip_exchange* x = new les_hip_exchange();
    x->c = c; x->rank = rank; x->world = world; x->nrects = n; x->own_first = first[rank]; x->own_n = first[rank + 1] - first[rank];
    x->lmax = (int)lmax; x->max_px = max_px;
    if (n > 0) {
        if (hipMalloc((void**)&x->d_rects, (size_t)n * sizeof(les::XchgRect)) != hipSuccess ||
            hipMemcpy(x->d_rects, tab.data(), (size_t)n * sizeof(les::XchgRect), hipMemcpyHostToDevice) != hipSuccess) {
            les_hip_exchange_destroy(x);
            return fail(LES_HIP_ERR_DEVICE, "les_hip_exchange_create: upload of the rect table failed");
        }
    }
    *out = x;
    return LES_HIP_OK;
}

void les_hip_exchange_destroy(les_hip_exchange* x)
{
    if (!x) return;
    if (x->c) (void)hipSetDevice(x->c->p.device);
    if (x->d_rects) (void)hipFree(x->d_rects);
    if (x->d_send) (void)hipFree(x->d_send);
    if (x->d_recv) (void)hipFree(x->d_recv);
    delete x;
}

long long les_hip_exchange_slot_floats(const les_hip_exchange* x) { return x ? 5ll * x->lmax : 0; }

int les_hip_exchange_pack(les_hip_ctx* c, const les_hip_exchange* x, const les_hip_plane* d_labels, const float* d_cost, float* d_slot)
{
    if (!c || !x || x->c != c || !d_labels || !d_cost || !d_slot) return fail(LES_HIP_ERR_ARG, "les_hip_exchange_pack: bad argument");
    if (x->own_n <= 0) return LES_HIP_OK;
    hipLaunchKernelGGL(les::les_xchg_pack_kernel, dim3(x->own_n, xchg_chunks(x->max_px)), dim3(256), 0, cur_stream(c), x->d_rects, x->own_first,
                       reinterpret_cast<const float4*>(d_labels), d_cost, d_slot, x->lmax, c->p.W);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int les_hip_exchange_unpack(les_hip_ctx* c, const les_hip_exchange* x, const float* d_recv, les_hip_plane* d_labels, float* d_cost)
{
    if (!c || !x || x->c != c || !d_labels || !d_cost || !d_recv) return fail(LES_HIP_ERR_ARG, "les_hip_exchange_unpack: bad argument");
    if (x->nrects <= 0 || x->world == 1) return LES_HIP_OK;
    hipLaunchKernelGGL(les::les_xchg_unpack_kernel, dim3(x->nrects, xchg_chunks(x->max_px)), dim3(256), 0, cur_stream(c), x->d_rects, d_recv,
                       reinterpret_cast<float4*>(d_labels), d_cost, x->lmax, c->p.W, x->rank);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int les_hip_exchange_tiles(les_hip_ctx* c, les_hip_exchange* x, void* nccl_comm, les_hip_plane* d_labels, float* d_cost)
{
    if (!c || !x || x->c != c || !d_labels || !d_cost) return fail(LES_HIP_ERR_ARG, "les_hip_exchange_tiles: bad argument");
    if (x->world == 1 && !nccl_comm) return LES_HIP_OK;                                  // one rank: nothing to publish
    if (!nccl_comm) return fail(LES_HIP_ERR_ARG, "les_hip_exchange_tiles: %d ranks need an ncclComm_t", x->world);
    nccl_allgather_fn allgather = load_nccl_allgather();
    if (!allgather) return fail(LES_HIP_ERR_DEVICE, "les_hip_exchange_tiles: librccl.so (ncclAllGather) not found");
    const size_t slot = (size_t)5 * x->lmax;
    if (slot == 0) return LES_HIP_OK;
    if (!x->d_send) {
        HIPCHECK(hipSetDevice(c->p.device));
        HIPCHECK(hipMalloc((void**)&x->d_send, slot * sizeof(float)));
        HIPCHECK(hipMalloc((void**)&x->d_recv, slot * sizeof(float) * (size_t)x->world));
    }
    // pack -> all-gather -> unpack, all enqueued on the calling thread's stream: no host synchronisation anywhere
    int rc = les_hip_exchange_pack(c, x, d_labels, d_cost, x->d_send);
    if (rc) return rc;
    const int nrc = allgather(x->d_send, x->d_recv, slot, 7 /* ncclFloat32 */, nccl_comm, cur_stream(c));
    if (nrc != 0) return fail(LES_HIP_ERR_DEVICE, "les_hip_exchange_tiles: ncclAllGather failed with ncclResult_t %d", nrc);
    return les_hip_exchange_unpack(c, x, x->d_recv, d_labels, d_cost);
}

int les_hip_calib_copy_wide(const float* d_src, float* d_dst, size_t n, int device, void* stream)
{
    if (!d_src || !d_dst || n == 0 || (n & 3) || (((uintptr_t)d_src | (uintptr_t)d_dst) & 15)) return fail(LES_HIP_ERR_ARG, "bad argument (n must be a multiple of 4, pointers 16-byte aligned)");
    HIPCHECK(hipSetDevice(device));
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(les::les_calib_copy_wide_kernel, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(d_src), reinterpret_cast<float4*>(d_dst), n4);
    HIPCHECK(hipGetLastError());
    return LES_HIP_OK;
}

int les_hip_malloc(les_hip_ctx* c, void** p, size_t bytes)
{
    if (!c || !p) return fail(LES_HIP_ERR_ARG, "null argument");
    HIPCHECK(hipMalloc(p, bytes));
    return LES_HIP_OK;
}
int les_hip_free(les_hip_ctx* c, void* p)
{
    if (!c) return fail(LES_HIP_ERR_ARG, "null argument");
    if (p) HIPCHECK(hipFree(p));
    return LES_HIP_OK;
}
int les_hip_memcpy_h2d(les_hip_ctx* c, void* d, const void* s, size_t bytes)
{
    if (!c) return fail(LES_HIP_ERR_ARG, "null argument");
    HIPCHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, cur_stream(c)));
    HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    return LES_HIP_OK;
}
int les_hip_memcpy_d2h(les_hip_ctx* c, void* d, const void* s, size_t bytes)
{
    if (!c) return fail(LES_HIP_ERR_ARG, "null argument");
    HIPCHECK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, cur_stream(c)));
    HIPCHECK(hipStreamSynchronize(cur_stream(c)));
    return LES_HIP_OK;
}
int les_hip_memset(les_hip_ctx* c, void* d, int value, size_t bytes)
{
    if (!c) return fail(LES_HIP_ERR_ARG, "null argument");
    HIPCHECK(hipMemsetAsync(d, value, bytes, cur_stream(c)));
    return LES_HIP_OK;
}

size_t les_hip_tiled_volume_bytes(les_hip_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 1 || !c->v[mode].vol_t) return 0;
    return (size_t)c->p.H * (size_t)((c->p.W + 7) / 8) * 8u * (size_t)c->p.D * sizeof(float);
}

int les_hip_get_stats(les_hip_ctx* c, int mode, float* out)
{
    if (!c || !out || mode < 0 || mode > 1 || !c->v[mode].stats) return fail(LES_HIP_ERR_ARG, "bad argument");
    HIPCHECK(hipMemcpy(out, c->v[mode].stats, (size_t)c->p.H * c->p.W * 12 * sizeof(float), hipMemcpyDeviceToHost));
    return LES_HIP_OK;
}

}  // extern "C"
