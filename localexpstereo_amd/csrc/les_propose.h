// les_propose.h -- PatchMatch-style hypothesis generation on the device (reference: LES/Proposer.h,
// LES/StereoEnergy.h:120-129, LES/Utilities.hpp:254-261, LES/FastGCStereo.h:94-115,231-238).
//
// One lock-step of the local-expansion loop draws one proposal per grid cell of a disjoint set
// (LES/FastGCStereo.h:41-48).  Here every cell owns a cv::RNG-compatible generator state
// (multiply-with-carry, [recollection] of OpenCV 3.1) kept in device memory, so a proposal for a cell
// is a pure function of (label map, cell rect, generator state) and can be checked against the CPU
// restatement draw for draw.  The reference seeds its per-thread generators from time(NULL)
// (LES/main.cpp:430,444-450): trajectories are not reproducible there, only the distributions are.
//
// Kernels:
//   les_expansion_kernel : ExpansionProposer::getNextProposal      (LES/Proposer.h:69-75)
//   les_random_kernel    : RandomProposer::getNextProposal         (LES/Proposer.h:120-148)
//   les_ransac_{snapshot,draw,eval,walk}_kernel : RansacProposer::startIterations + RANSACPlane
//                          (LES/Proposer.h:177-301), one lane per (cell, candidate)
//   les_init_labels_kernel : FastGCStereo::initCurrentFast label part (LES/FastGCStereo.h:105-109)
#pragma once

#include "les_simt.h"

namespace les {

struct Rect4 { int x, y, w, h; };

// ---- cv::RNG [recollection]
struct Rng {
    uint64_t state;
    __device__ __forceinline__ uint32_t next()
    {
        state = (uint64_t)(uint32_t)state * 4164903690ULL + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    __device__ __forceinline__ int uniform_int(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
    __device__ __forceinline__ float uniform_float(float a, float b)
    {
        float f = next() * 2.3283064365386962890625e-10f;
        return f * (b - a) + a;
    }
    __device__ __forceinline__ double uniform_double(double a, double b)
    {
        uint32_t t = next();
        double d = (double)(((uint64_t)t << 32) | next()) * 5.4210108624275221700372640043497e-20;
        return d * (b - a) + a;
    }
};

// LES/Plane.h:23-31
__device__ __forceinline__ float4 plane_create(float nx, float ny, float nz, float z, float x, float y, float v)
{
    float4 p;
    p.x = -nx / nz;
    p.y = -ny / nz;
    p.z = z - p.x * x - p.y * y;
    p.w = v;
    return p;
}
// LES/Plane.h:42-50 (sqrt in double, then cast)
__device__ __forceinline__ void plane_normal(const float4& p, float n[3])
{
    float nz = (float)(1.0 / sqrt(1.0 + p.x * p.x + p.y * p.y));
    n[0] = -p.x * nz;
    n[1] = -p.y * nz;
    n[2] = nz;
}
// LES/Utilities.hpp:254-261
__device__ __forceinline__ void random_unit_vector(Rng& r, double thetaRange, double n[3])
{
    const double PI = 3.1415926535897932384626433832795;
    double theta = r.uniform_double(0.0, thetaRange);
    double phi = r.uniform_double(0.0, PI * 2.0);
    double cosT = cos(theta), sinT = sin(theta);
    double cosP = cos(phi), sinP = sin(phi);
    n[0] = sinT * cosP; n[1] = sinT * sinP; n[2] = cosT;
}
// (MAX - MIN) * pow(0.5f, m + 1): exact powers of two (LES/Proposer.h:93-96)
__device__ __forceinline__ float perturbation_width(float mind, float maxd, int m)
{
    return (float)((double)(maxd - mind) * ldexp(1.0, -(m + 1)));
}

// ---------------------------------------------------------------------------------------------------
__global__ void les_expansion_kernel(const Rect4* __restrict__ units, const float4* __restrict__ labels, int W,
                                     uint64_t* __restrict__ rng, float4* __restrict__ planes, int n)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const Rect4 u = units[i];
    Rng r{rng[i]};
    int k = r.uniform_int(0, u.h * u.w);                              // LES/Proposer.h:37-44
    int px = k % u.w, py = k / u.w;
    planes[i] = labels[(size_t)(u.y + py) * W + u.x + px];            // :72
    rng[i] = r.state;
}

__global__ void les_random_kernel(const Rect4* __restrict__ units, const float4* __restrict__ labels, int W,
                                  uint64_t* __restrict__ rng, float4* __restrict__ planes, int n, int m,
                                  float mind, float maxd)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const double PI = 3.1415926535897932384626433832795;
    const Rect4 u = units[i];
    Rng r{rng[i]};
    int k = r.uniform_int(0, u.h * u.w);                              // :122
    int px = k % u.w, py = k / u.w;
    const float4 in = labels[(size_t)(u.y + py) * W + u.x + px];      // :123
    const int sx = u.x + px, sy = u.y + py;                           // :127
    float zs = in.x * (float)sx + in.y * (float)sy + in.z;            // :128 Plane::GetZ
    const float dz = perturbation_width(mind, maxd, m);               // :129
    const float minz = fmaxf(mind, zs - dz);                          // :130
    const float maxz = fminf(maxd, zs + dz);                          // :131
    zs = r.uniform_float(minz, maxz);                                 // :132
    const float nr = (float)ldexp(1.0, -m);                           // :142 randomNmax * pow(0.5f, m)
    float n0[3];
    plane_normal(in, n0);
    double uv[3];
    random_unit_vector(r, PI, uv);                                    // :143
    float nv[3];
    for (int c = 0; c < 3; c++) nv[c] = n0[c] + (float)uv[c] * nr;
    const double dd = (double)nv[0] * nv[0] + (double)nv[1] * nv[1] + (double)nv[2] * nv[2];
    const double inv = 1. / sqrt(dd);                                 // :145
    for (int c = 0; c < 3; c++) nv[c] = (float)(nv[c] * inv);
    planes[i] = plane_create(nv[0], nv[1], nv[2], zs, (float)sx, (float)sy, in.w);   // :147
    rng[i] = r.state;
}

// createRandomLabel + fill of the unit region (LES/FastGCStereo.h:105-109, LES/StereoEnergy.h:120-129)
__global__ void les_init_labels_kernel(const Rect4* __restrict__ units, float4* __restrict__ labels, int W,
                                       uint64_t* __restrict__ rng, float4* __restrict__ planes, float mind, float maxd)
{
    const double PI = 3.1415926535897932384626433832795;
    const int i = (int)blockIdx.x;
    const Rect4 u = units[i];
    __shared__ float4 s_plane;
    if (threadIdx.x == 0) {
        Rng r{rng[i]};
        int k = r.uniform_int(0, u.h * u.w);                          // selectRandomPixelInRect, LES/FastGCStereo.h:231-238
        int sx = u.x + k % u.w, sy = u.y + k / u.w;
        float zs = r.uniform_float(mind, maxd);                       // LES/StereoEnergy.h:122
        double nn[3];
        random_unit_vector(r, PI / 3, nn);                            // :126
        s_plane = plane_create((float)nn[0], (float)nn[1], (float)nn[2], zs, (float)sx, (float)sy, 0.0f);
        planes[i] = s_plane;
        rng[i] = r.state;
    }
    __syncthreads();
    const float4 pl = s_plane;
    for (int idx = (int)threadIdx.x; idx < u.w * u.h; idx += (int)blockDim.x)
        labels[(size_t)(u.y + idx / u.w) * W + u.x + idx % u.w] = pl;  // currentLabeling(unit) = label, :109
}

// ---------------------------------------------------------------------------------------------------
// RANSAC.  cv::solve(A, b, x, DECOMP_SVD) of an m x 3 system is restated as the pseudo-inverse through
// the 3x3 eigen-decomposition of A^T A in double (cyclic Jacobi; LES/Proposer.h:203,224).
// ---------------------------------------------------------------------------------------------------
__device__ inline void solve_normal_3x3(double M[3][3], const double rhs[3], float x[3])
{
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 16; sweep++) {
        double off = fabs(M[0][1]) + fabs(M[0][2]) + fabs(M[1][2]);
        if (off <= 1e-15 * (fabs(M[0][0]) + fabs(M[1][1]) + fabs(M[2][2]))) break;   // off-diagonals at the rounding floor of the diagonal
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (fabs(M[p][q]) < 1e-300) continue;
                double theta = (M[q][q] - M[p][p]) / (2 * M[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; k++) {
                    double mkp = M[k][p], mkq = M[k][q];
                    M[k][p] = c * mkp - s * mkq; M[k][q] = s * mkp + c * mkq;
                }
                for (int k = 0; k < 3; k++) {
                    double mpk = M[p][k], mqk = M[q][k];
                    M[p][k] = c * mpk - s * mqk; M[q][k] = s * mpk + c * mqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    double w[3], wsum = 0;
    for (int k = 0; k < 3; k++) { w[k] = sqrt(M[k][k] > 0.0 ? M[k][k] : 0.0); wsum += w[k]; }
    const double thr = wsum * 2 * 1.1920929e-07;
    double out[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
        if (w[k] <= thr) continue;
        double proj = (V[0][k] * rhs[0] + V[1][k] * rhs[1] + V[2][k] * rhs[2]) / (w[k] * w[k]);
        for (int r = 0; r < 3; r++) out[r] += V[r][k] * proj;
    }
    for (int r = 0; r < 3; r++) x[r] = (float)out[r];
}

// LES/Proposer.h:243-262
__device__ inline int ransac_sample_count(int ni, int ptNum, int pf, double conf)
{
    double q = 1.0;
    for (double a = (ni - pf + 1), b = (ptNum - pf + 1); a <= ni; a += 1.0, b += 1.0) q *= (a / b);
    int cnt;
    if ((1.0 - q) < 1e-4) cnt = 1;
    else cnt = (int)(log(1.0 - conf) / log(1.0 - q));
    return cnt < 1 ? 1 : cnt;
}

// RANSACPlane (LES/Proposer.h:177-240) with the reference's sequential semantics AND its adaptive schedule (`while (no_sam < max_sam)`,
// :193; `max_sam = min(max_sam, computeSampleCount(...))`, :229-236), split so that the whole GPU works even when a disjoint set has
// only a handful of (large) cells.  The MAX_SAM candidates are processed in CHUNKS (16, then 48, 64, 128, 244: kRansacChunkEnds);
// a cell whose loop has ended is marked done and every later launch returns at once for it, so a planar region costs one chunk of 16
// candidates instead of 500 (round 6; rounds 1-5 evaluated all 500 with a refit each before looking at the stop rule):
//   snapshot : disparities of the unit region under the current labelling (startIterations, :283-301)
//   begin    : one lane per cell: loop state (:180-185), draws the sample triples of the first chunk in order (the generator is
//              sequential) and records the generator state after every sample
//   eval     : one quad of lanes per (cell, candidate of the chunk that the loop can still reach: j < max_sam): solves the 3-point
//              plane, counts its inliers and -- only if the candidate can still trigger the "better than max_i" branch (no_i > the
//              cell's max_i at the start of the chunk; max_i only grows, :234) -- the least-squares refit on the inliers among the
//              first no_i points and the refit's inlier count.  None of this depends on the state evolving INSIDE the chunk.
//   walk     : one lane per cell replays the reference's acceptance / adaptive-termination logic over the chunk's candidates IN
//              ORDER; when the loop ends it rewinds the generator to the last consumed sample and writes the plane, otherwise it
//              draws the next chunk.
// Result and generator state are exactly those of the sequential algorithm (and of the all-candidates schedule of rounds 1-5).
struct RansacCell { int max_i, max_sam, no_sam, no_i_c; float result[3]; int done; };     // loop state of one cell (:180-185); 32 B
struct RansacScratch {
    float* disp;       // [n][stride]            disparity snapshot
    int* idx;          // [n][MAX_SAM][3]        sample triples
    uint64_t* state;   // [n][MAX_SAM + 1]       generator state before sample j
    int* noi;          // [n][MAX_SAM]           inliers of the 3-point plane
    int* no;           // [n][MAX_SAM]           inliers of the refit (-1: not computed)
    float* refit;      // [n][MAX_SAM][3]        refitted plane
    int stride;
    RansacCell* cell;  // [n]
};

__global__ void les_ransac_snapshot_kernel(const Rect4* __restrict__ units, const float4* __restrict__ labels, int W, RansacScratch sc)
{
    const int cell = (int)blockIdx.x;
    const Rect4 u = units[cell];
    float* disp = sc.disp + (size_t)cell * sc.stride;
    for (int i = (int)threadIdx.x; i < u.w * u.h; i += (int)blockDim.x) {
        const int yy = i / u.w, xx = i - yy * u.w;
        const float c0 = (float)xx + u.x, c1 = (float)yy + u.y;
        const float4 v = labels[(size_t)(yy + u.y) * W + xx + u.x];
        disp[i] = v.x * c0 + v.y * c1 + v.z;                           // :297
    }
}

// x mod d for 32-bit x and d >= 1 with the precomputed M = floor((2^64 - 1) / d) + 1 (Lemire, Kaser, Kurz 2019: exact for all 32-bit operands): two
// multiplications instead of the ~35-instruction division sequence -- the draw loop below is ONE lane's dependent instruction stream
__device__ __forceinline__ uint32_t fastmod_u32(uint32_t x, uint64_t M, uint32_t d)
{
    const uint64_t low = M * (uint64_t)x;
#if defined(LES_SIM)
    return (uint32_t)(((unsigned __int128)low * d) >> 64);
#else
    return (uint32_t)__umul64hi(low, (uint64_t)d);
#endif
}

// samples j0 <= j < j1 of one cell, continuing the generator from the recorded state before sample j0
__device__ inline void ransac_draw(const Rect4& u, RansacScratch& sc, int cell, int MAX_SAM, int j0, int j1)
{
    const int len = u.w * u.h;
    int* idxp = sc.idx + (size_t)cell * MAX_SAM * 3;
    uint64_t* st = sc.state + (size_t)cell * (MAX_SAM + 1);
    Rng r{st[j0]};
    const uint64_t M = len > 0 ? ~0ull / (uint32_t)len + 1 : 0;          // uniform(0, len) = next() % len (cv::RNG), as fastmod_u32
    // one sample the exact way: three distinct uniformly random indices, the first three entries of randperm (:163-174,196-201)
    auto draw_one = [&](int j) {
        int idx[3];
        for (int i = 0; i < 3; i++) {
            bool again;
            do {
                idx[i] = len > 0 ? (int)fastmod_u32(r.next(), M, (uint32_t)len) : 0;
                again = false;
                for (int q = 0; q < i; q++) if (idx[q] == idx[i] && len > i) again = true;
            } while (again);
        }
        idxp[j * 3 + 0] = idx[0]; idxp[j * 3 + 1] = idx[1]; idxp[j * 3 + 2] = idx[2];
        st[j + 1] = r.state;
    };
    int j = j0;
    // Four samples at a time on the assumption that no draw repeats an index of its sample (a repeat has probability ~3 / len per sample): the twelve
    // generator steps are one short chain and the twelve reductions mod len independent of each other, instead of one dependent stream of
    // step -> reduce -> compare -> branch per draw.  A batch with a repeat is drawn again the exact way, from the state it started with.
    for (; len > 2 && j + 4 <= j1; j += 4) {
        Rng t{r.state};
        uint64_t s[12];
        int id[12];
#pragma unroll
        for (int q = 0; q < 12; q++) { id[q] = (int)t.next(); s[q] = t.state; }
#pragma unroll
        for (int q = 0; q < 12; q++) id[q] = (int)fastmod_u32((uint32_t)id[q], M, (uint32_t)len);
        bool distinct = true;
#pragma unroll
        for (int q = 0; q < 12; q += 3) distinct = distinct && id[q] != id[q + 1] && id[q] != id[q + 2] && id[q + 1] != id[q + 2];
        if (distinct) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                idxp[(j + q) * 3 + 0] = id[3 * q]; idxp[(j + q) * 3 + 1] = id[3 * q + 1]; idxp[(j + q) * 3 + 2] = id[3 * q + 2];
                st[j + q + 1] = s[3 * q + 2];
            }
            r.state = s[11];
        } else {
            for (int q = 0; q < 4; q++) draw_one(j + q);
        }
    }
    for (; j < j1; j++) draw_one(j);
}

__global__ void les_ransac_begin_kernel(const Rect4* __restrict__ units, const uint64_t* __restrict__ rng, RansacScratch sc, int n, int MAX_SAM, int first)
{
    const int cell = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (cell >= n) return;
    RansacCell c;
    c.max_i = 3; c.max_sam = MAX_SAM; c.no_sam = 0; c.no_i_c = 0;      // :180-185
    c.result[0] = c.result[1] = c.result[2] = 0.0f; c.done = 0;
    sc.cell[cell] = c;
    sc.state[(size_t)cell * (MAX_SAM + 1)] = rng[cell];
    ransac_draw(units[cell], sc, cell, MAX_SAM, 0, first < MAX_SAM ? first : MAX_SAM);
}

// Candidates j0 <= j < j1 of every cell that is still running.  grid = (cells, ceil((j1 - j0) / 16)), block = 1024: ONE WAVE per (cell,
// candidate), sixteen candidates per workgroup (round 6; rounds 1-5: a quad of lanes per candidate -- a scan of a 129 x 129 unit region by four
// lanes is a chain of 4 160 dependent iterations, 150 us whatever the rest of the GPU does).  Lane l of the wave visits the points with
// yy = l / 16 (mod 4) and xx = l % 16 (mod 16) in increasing (row, column) order.
// The 3-point planes of the workgroup's sixteen candidates are solved first, one lane each (the 3 x 3 eigen-solve is a long scalar chain: done
// by every lane of sixteen waves it would occupy sixteen times the issue slots).  Inlier counts are integer sums over the wave.  The normal
// equations of the refit are accumulated per lane in that order and combined by the butterfly l ^ 1, l ^ 2, ... l ^ 32: a DEFINED order,
// shared with the host RansacProposer (host/Proposer.h: refitSums), so that the per-call drop-in loop and the device proposals are
// bit-identical.  (The test oracle sums in the natural row order; the comparison with it is to float round-off.)
constexpr int kRansacCandPerBlock = 16;
__global__ void __launch_bounds__(64 * kRansacCandPerBlock)
les_ransac_eval_kernel(const Rect4* __restrict__ units, RansacScratch sc, int MAX_SAM, float threshold, int j0, int j1)
{
#if defined(LES_SIM)
    static thread_local float sN[kRansacCandPerBlock][3];
#else
    __shared__ float sN[kRansacCandPerBlock][3];
#endif
    const int cell = (int)blockIdx.x;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const RansacCell cs = sc.cell[cell];
    const int jend = j1 < cs.max_sam ? j1 : cs.max_sam;                    // the loop never reaches a candidate at or beyond max_sam (it only shrinks)
    const int jb = j0 + (int)blockIdx.y * kRansacCandPerBlock;
    if (cs.done || jb >= jend) return;                                      // (uniform over the workgroup)
    const Rect4 u = units[cell];
    const int len = u.w * u.h;
    const float* disp = sc.disp + (size_t)cell * sc.stride;

    if (tid < kRansacCandPerBlock && jb + tid < jend) {
        const int* idxp = sc.idx + ((size_t)cell * MAX_SAM + jb + tid) * 3;
        double M[3][3] = {{0}}, rhs[3] = {0, 0, 0};
        for (int i = 0; i < 3; i++) {
            const int id = idxp[i];
            const int yy = id / u.w, xx = id - yy * u.w;
            const double c[3] = {(double)((float)xx + u.x), (double)((float)yy + u.y), 1.0};
            const double d = disp[id];
            for (int a = 0; a < 3; a++) {
                rhs[a] += c[a] * d;
                for (int b = 0; b < 3; b++) M[a][b] += c[a] * c[b];
            }
        }
        float N3[3];
        solve_normal_3x3(M, rhs, N3);                                   // cv::solve(ranpts, div, N, DECOMP_SVD) :203
        sN[tid][0] = N3[0]; sN[tid][1] = N3[1]; sN[tid][2] = N3[2];
    }
    __syncthreads();
    const int j = jb + wave;                                            // candidate of this wave
    if (j >= jend) return;                                              // (whole waves; no barrier follows)
    const int rp = lane >> 4, cp = lane & 15;                           // row / column phase of this lane

    // visits this lane's points among the first `upto` in index order: f(x, y, disparity, is_inlier_of_N)
    auto scan = [&](int upto, const float N[3], auto&& f) {
        for (int yy = rp; yy < u.h; yy += 4) {
            const int row = yy * u.w;
            if (row >= upto) break;
            const float y = (float)yy + u.y;
            const double ty = (double)y * N[1];
            for (int xx = cp; xx < u.w; xx += 16) {
                const int i = row + xx;
                if (i >= upto) break;
                const float x = (float)xx + u.x;
                const float d = disp[i];
                const float dot = (float)(((double)x * N[0] + ty) + (double)1.0f * N[2]);   // pts * N (:204): x*N0 + y*N1 + 1*N2
                f(x, y, d, fabsf(dot - d) < threshold);
            }
        }
    };

    const float N[3] = {sN[wave][0], sN[wave][1], sN[wave][2]};
    int cnt = 0;
    scan(len, N, [&](float, float, float, bool in) { cnt += in; });   // :204-206
    const int no_i = wave_sum_tree(cnt);
    int no = -1;
    float N2[3] = {0, 0, 0};
    // max_i starts at 3 and only grows (:180,:234): a candidate with no_i <= the cell's max_i at the start of this chunk can never
    // enter the refit branch.  no_i is uniform over the wave.
    if (no_i > cs.max_i) {
        // least-squares refit on the inliers among the FIRST no_i points (the reference's loop bound quirk, :216)
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                   // xx xy x yy y 1 | xd yd d
        scan(no_i, N, [&](float x, float y, float d, bool in) {
            if (in) {
                const double dx = x, dy = y, dd = d;
                acc[0] += dx * dx; acc[1] += dx * dy; acc[2] += dx; acc[3] += dy * dy; acc[4] += dy; acc[5] += 1.0;
                acc[6] += dx * dd; acc[7] += dy * dd; acc[8] += dd;
            }
        });
        double t[9];
        for (int k = 0; k < 9; k++) t[k] = wave_sum_tree(acc[k]);
        double A[3][3] = {{t[0], t[1], t[2]}, {t[1], t[3], t[4]}, {t[2], t[4], t[5]}};
        double r3[3] = {t[6], t[7], t[8]};
        solve_normal_3x3(A, r3, N2);                                   // :224 (every lane: the same result)
        int c2 = 0;
        scan(len, N2, [&](float, float, float, bool in) { c2 += in; });   // :225-227
        no = wave_sum_tree(c2);
    }
    if (lane == 0) {
        const size_t o = (size_t)cell * MAX_SAM + j;
        sc.noi[o] = no_i;
        sc.no[o] = no;
        sc.refit[o * 3 + 0] = N2[0]; sc.refit[o * 3 + 1] = N2[1]; sc.refit[o * 3 + 2] = N2[2];
    }
}

// the loop of :193-237 over the candidates j0 <= j < j1 of every running cell; draws the candidates j1 <= j < j2 when the loop goes on
__global__ void les_ransac_walk_kernel(const Rect4* __restrict__ units, uint64_t* __restrict__ rng, float4* __restrict__ planes,
                                       RansacScratch sc, int n, int MAX_SAM, float conf, int j0, int j1, int j2)
{
    const int cell = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (cell >= n) return;
    RansacCell c = sc.cell[cell];
    if (c.done) return;
    const Rect4 u = units[cell];
    const int len = u.w * u.h;
    const size_t base = (size_t)cell * MAX_SAM;
    // (one thread walks one cell: a chain of dependent loads unless the inlier counts of the next eight candidates are fetched together)
    while (c.no_sam < c.max_sam && c.no_sam < j1) {                    // :193
        const int jb = c.no_sam;
        int noi8[8];
#pragma unroll
        for (int q = 0; q < 8; q++) noi8[q] = jb + q < j1 ? sc.noi[base + jb + q] : 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (!(c.no_sam < c.max_sam && c.no_sam < j1)) break;
            const int j = c.no_sam;
            c.no_sam++;
            const int no_i = noi8[q];
            if (c.max_i < no_i) {                                      // :208
                const int no = sc.no[base + j];
                if (no > c.no_i_c) {                                   // :229-236
                    c.result[0] = sc.refit[(base + j) * 3]; c.result[1] = sc.refit[(base + j) * 3 + 1]; c.result[2] = sc.refit[(base + j) * 3 + 2];
                    c.no_i_c = no;
                    c.max_i = no_i;
                    const int cnt = ransac_sample_count(no, len, 3, conf);
                    c.max_sam = c.max_sam < cnt ? c.max_sam : cnt;
                }
            }
        }
    }
    if (c.no_sam >= c.max_sam) {
        c.done = 1;
        planes[cell] = make_float4(c.result[0], c.result[1], c.result[2], 0.0f);   // :239
        rng[cell] = sc.state[(size_t)cell * (MAX_SAM + 1) + c.no_sam];              // state after the last consumed sample
    } else {
        ransac_draw(u, sc, cell, MAX_SAM, j1, j2 < c.max_sam ? j2 : c.max_sam);
    }
    sc.cell[cell] = c;
}

}  // namespace les
