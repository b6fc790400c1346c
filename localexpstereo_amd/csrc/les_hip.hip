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
    // tables (d_hs == nullptr: les_hip_refresh_volume -- the guide has not changed, its tables and the bound on the inverse covariance are kept)
    // (built BEFORE the checks of the volume below: a context created on a placeholder or out-of-range volume keeps the guide's tables, so that
    //  les_hip_refresh_volume can move it onto the march kernel after a valid refill)
    if (d_hs) {
        unsigned dbits = 0;
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
    auto drop_tiled = [&]() { if (v.vol_t) { (void)hipFree(v.vol_t); v.vol_t = nullptr; } };      // a stale copy (GBs) must not outlive a fall-back of a refresh
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
        drop_tiled();
        return LES_HIP_OK;
    }
    vmin = std::min(vmin, 0.5f * th);                                // a volume entirely above th_col: p == th_col everywhere
    const double range = (double)th - (double)vmin;
    if (!(range <= 8.0 * (double)th)) {                              // the 20-bit fixed point would resolve th_col too coarsely
        note_fallback(c->fallback_seen, FB_RANGE, "view %d: costs reach %g below the truncation threshold %g (more than 8 x the threshold)", m, range, (double)th);
        drop_tiled();
        return LES_HIP_OK;
    }
    const unsigned dbits = v.dmax_bits;
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

// ---- the C ABI (include/localexp_hip.h), by concern; one translation unit (the kernels are templates in headers, the context and batch structs above are
// shared by every part)
#include "les_hip_context.inc"
#include "les_hip_batch.inc"
#include "les_hip_cuts.inc"
#include "les_hip_ingest_post.inc"
#include "les_hip_exchange.inc"
