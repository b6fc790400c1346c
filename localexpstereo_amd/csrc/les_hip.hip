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
#include "les_maxflow_cell.h"

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

#include "les_hip_march_tables.inc"      // which kernel instantiations exist (radius -> geometry): part of what bench.py hashes as "the kernel sources"

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
    int graph_chunks = 1;                // ... in the graph-construction kernel: about one node per thread (its loads are dependent: occupancy hides them)
    // expansion-graph payload layout (les_hip_batch_expansion_graph): node offset of every target, total node count
    std::vector<long long> graph_off;
    long long graph_nodes = 0;
    long long* d_graph_off = nullptr;
    double* d_flow0 = nullptr;           // n * graph_chunks partial sums
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

#include "les_hip_march.inc"             // job tables, launches and per-view set-up of the unary-cost kernels (hashed with the kernel headers by bench.py)

float naive_alpha(const les_hip_ctx* c) { return c->naive_alpha; }

}  // namespace

// ---- the C ABI (include/localexp_hip.h), by concern; one translation unit (the kernels are templates in headers, the context and batch structs above are
// shared by every part)
#include "les_hip_context.inc"
#include "les_hip_batch.inc"
#include "les_hip_cuts.inc"
#include "les_hip_ingest_post.inc"
#include "les_hip_exchange.inc"
