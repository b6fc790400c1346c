// les_pairwise.h -- pairwise smoothness terms of the expansion moves on the device ("next" row N1 of SURVEY.md
// section 8(f)): for every cell of a lock-step the kernel produces the ready-made capacities of the cell's graph, so the
// D2H payload is what the max-flow consumes and the host no longer touches labels, images or cost maps.
//
// Reference: StereoEnergy::initSmoothnessCoeff / computeSmoothnessTerm / computeSmoothnessTermsExpansion
// (LES/StereoEnergy.h:131-163, 225-230, 398-453) and the graph construction of FastGCStereo::expansionMoveBK
// (LES/FastGCStereo.h:425-551).  One thread per graph node replays, in the reference's program order, exactly the
// sequence of add_tweights calls that touch its node (unary, border terms k = 0..7, then for each forward direction
// the contribution as `j` of the preceding pixel followed by its own as `i`), so the terminal capacity and the four
// forward arc capacities are bit-identical to the host construction (localexpstereo_amd/host/ExpansionMove.h).
//
// Payload per node (5 floats, AoS): { tr = source - sink residual, cap E, cap S, cap SW, cap SE }; cell i starts at
// 5 * offset[i] floats, nodes row-major over the cell's region.  flow0 (per cell, double): the flow already routed
// source -> node -> sink by the t-links (sum over nodes, in node order within a lane's partial sums -- it is only used by
// the optional flow == energy self-check).
#pragma once

#include "les_simt.h"

namespace les {

struct PairwiseParams {
    int H, W;
    float lambda, th_smooth;
};

// smoothness coefficient of pixel (x, y) towards neighbour k: max(epsilon, exp(-|dI|_1 / omega)) via the 766-entry table
// (|dI|_1 of 8-bit colours is an integer), 0 for pairs that leave the image  (LES/StereoEnergy.h:131-163)
__device__ __forceinline__ float pw_coeff(const uint32_t* __restrict__ ipk, const float* __restrict__ wtab, int W, int H, int x, int y, int dx, int dy)
{
    const int xn = x + dx, yn = y + dy;
    if (xn < 0 || xn >= W || yn < 0 || yn >= H) return 0.0f;
    const uint32_t a = ipk[(size_t)y * W + x], b = ipk[(size_t)yn * W + xn];
    const int ad = abs((int)(a & 255) - (int)(b & 255)) + abs((int)((a >> 8) & 255) - (int)((b >> 8) & 255)) + abs((int)((a >> 16) & 255) - (int)((b >> 16) & 255));
    return wtab[ad];
}
// std::min(a, b) of the reference / host code: (b < a) ? b : a  -- NaN in `a` propagates, unlike fminf
__device__ __forceinline__ float pw_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float pw_getz(float4 l, int x, int y) { return (l.x * (float)x + l.y * (float)y) + l.z; }          // Plane::GetZ
__device__ __forceinline__ float pw_dot(float4 l, float x, float y) { return ((l.x * x + l.y * y) + l.z * 1.0f) + l.w * 0.0f; }   // channelDot order

// computeSmoothnessTerm (LES/StereoEnergy.h:225-230)
__device__ __forceinline__ float pw_term(float coeff, float4 ls, float4 lt, int x, int y, int xt, int yt, const PairwiseParams& p)
{
    const float d = fabsf(pw_getz(ls, x, y) - pw_getz(lt, x, y)) + fabsf(pw_getz(ls, xt, yt) - pw_getz(lt, xt, yt));
    return coeff * pw_min(d, p.th_smooth) * p.lambda;
}

struct PwTerms { float c00, c01, c10; };
// computeSmoothnessTermsExpansion for pixel ee = (ex, ey) and forward neighbour (dx, dy)  (LES/StereoEnergy.h:398-453)
__device__ __forceinline__ PwTerms pw_expansion_terms(const float4* __restrict__ labels, const uint32_t* __restrict__ ipk, const float* __restrict__ wtab,
                                                      float4 label1, int ex, int ey, int dx, int dy, const PairwiseParams& p)
{
    const int lx = ex + dx, ly = ey + dy;
    const bool inside = lx >= 0 && lx < p.W && ly >= 0 && ly < p.H;
    const float4 l0_ee = labels[(size_t)ey * p.W + ex];
    const float4 l0_le = inside ? labels[(size_t)ly * p.W + lx] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float fx = (float)ex, fy = (float)ey, gx = (float)lx, gy = (float)ly;
    const float d0_ee_at_ee = pw_dot(l0_ee, fx, fy), d0_le_at_ee = pw_dot(l0_le, fx, fy);
    const float d0_ee_at_le = pw_dot(l0_ee, gx, gy), d0_le_at_le = pw_dot(l0_le, gx, gy);
    const float d1_at_ee = pw_dot(label1, fx, fy), d1_at_le = pw_dot(label1, gx, gy);
    const float w = pw_coeff(ipk, wtab, p.W, p.H, ex, ey, dx, dy);
    PwTerms t;
    t.c00 = pw_min(fabsf(d0_ee_at_ee - d0_le_at_ee) + fabsf(d0_ee_at_le - d0_le_at_le), p.th_smooth) * w * p.lambda;
    t.c01 = pw_min(fabsf(d0_ee_at_ee - d1_at_ee) + fabsf(d0_ee_at_le - d1_at_le), p.th_smooth) * w * p.lambda;
    t.c10 = pw_min(fabsf(d1_at_ee - d0_le_at_ee) + fabsf(d1_at_le - d0_le_at_le), p.th_smooth) * w * p.lambda;
    return t;
}

struct TLink {
    float tr = 0.0f;
    double flow = 0.0;
    // MaxFlow add_tweights: capacities accumulate; the common part of source and sink capacity is flow
    __device__ __forceinline__ void add(float cap_source, float cap_sink)
    {
        const float delta = tr;
        if (delta > 0) cap_source += delta;
        else cap_sink -= delta;
        flow += (double)((cap_source < cap_sink) ? cap_source : cap_sink);
        tr = cap_source - cap_sink;
    }
};

struct GraphCell { int x, y, w, h; };

// grid = (cells, chunks); block = 256
__global__ void les_expansion_graph_kernel(const GraphCell* __restrict__ cells, const long long* __restrict__ offsets, const float4* __restrict__ planes,
                                           const float4* __restrict__ labels, const float* __restrict__ cur, const float* __restrict__ prop,
                                           const uint32_t* __restrict__ ipk, const float* __restrict__ wtab, PairwiseParams p,
                                           float* __restrict__ payload, double* __restrict__ flow0)
{
    const GraphCell c = cells[blockIdx.x];
    const float4 label1 = planes[blockIdx.x];
    float* out = payload + 5 * offsets[blockIdx.x];
    const int N = c.w * c.h;
    // neighbour table of the reference (LES/StereoEnergy.h:99-110): LE GE EL EG LL GL LG GG
    const int nbx[8] = {-1, +1, 0, 0, -1, +1, -1, +1}, nby[8] = {0, 0, -1, +1, -1, -1, +1, +1};
    // forward directions in the order the graph is linked: GE, EG, LG, GG
    const int fdx[4] = {+1, 0, -1, +1}, fdy[4] = {0, +1, +1, +1};
    double flow_acc = 0.0;
    for (int idx = (int)(blockIdx.y * blockDim.x + threadIdx.x); idx < N; idx += (int)(blockDim.x * gridDim.y)) {
        const int y = idx / c.w, x = idx - y * c.w;
        const int X = c.x + x, Y = c.y + y;
        const size_t px = (size_t)Y * p.W + X;
        TLink t;
        t.add(cur[px], prop[px]);                                            // LES/FastGCStereo.h:433
        if (x == 0 || x == c.w - 1 || y == 0 || y == c.h - 1) {              // :455-475 terms towards fixed neighbours outside the region
            const float4 lps = labels[px];
            for (int k = 0; k < 8; k++) {
                const int xt = X + nbx[k], yt = Y + nby[k];
                const bool in_region = xt >= c.x && xt < c.x + c.w && yt >= c.y && yt < c.y + c.h;
                if (in_region || xt < 0 || xt >= p.W || yt < 0 || yt >= p.H) continue;
                const float coeff = pw_coeff(ipk, wtab, p.W, p.H, X, Y, nbx[k], nby[k]);
                const float4 lpt = labels[(size_t)yt * p.W + xt];
                t.add(pw_term(coeff, lps, lpt, X, Y, xt, yt, p), pw_term(coeff, label1, lpt, X, Y, xt, yt, p));
            }
        }
        float cap[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < 4; d++) {
            // as `j` of the preceding pixel (x - dx, y - dy), if that pixel links in this direction
            const int xs = x - fdx[d], ys = y - fdy[d];
            if (xs >= 0 && xs < c.w && ys >= 0 && ys < c.h) {
                const PwTerms s = pw_expansion_terms(labels, ipk, wtab, label1, c.x + xs, c.y + ys, fdx[d], fdy[d], p);
                t.add(s.c00 - s.c01, 0.0f);                                  // add_tweights(j, D - C, 0)
            }
            // as `i`: its own pair, if the neighbour is inside the region
            const int xn = x + fdx[d], yn = y + fdy[d];
            if (xn >= 0 && xn < c.w && yn < c.h) {
                const PwTerms s = pw_expansion_terms(labels, ipk, wtab, label1, X, Y, fdx[d], fdy[d], p);
                const float bcd = s.c10 + s.c01 - s.c00;
                cap[d] = (0.0f < bcd) ? bcd : 0.0f;                          // add_edge(i, j, std::max(0.f, B + C - D), 0)
                t.add(s.c01, 0.0f);                                          // add_tweights(i, C, 0)
            }
        }
        float* o = out + 5 * (size_t)idx;
        o[0] = t.tr; o[1] = cap[0]; o[2] = cap[1]; o[3] = cap[2]; o[4] = cap[3];
        flow_acc += t.flow;
    }
    // per-cell flow already routed through the t-links (diagnostic quantity)
    __shared__ double s_red[256];
    s_red[threadIdx.x] = flow_acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_red[threadIdx.x] += s_red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) flow0[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = s_red[0];
}

// subProposalCost.copyTo(subCurrentCost, updateMask); subCurrentLabeling.setTo(label, updateMask)  (LES/FastGCStereo.h:61-62)
// with the masks of a lock-step in graph-node order (cell i at offsets[i], row-major over its region).  grid = (cells, chunks)
__global__ void les_apply_masks_kernel(const GraphCell* __restrict__ cells, const long long* __restrict__ offsets, const float4* __restrict__ planes,
                                       const uint8_t* __restrict__ masks, float* __restrict__ cur, const float* __restrict__ prop,
                                       float4* __restrict__ labels, int W)
{
    const GraphCell c = cells[blockIdx.x];
    const float4 pl = planes[blockIdx.x];
    const uint8_t* m = masks + offsets[blockIdx.x];
    for (int idx = (int)(blockIdx.y * blockDim.x + threadIdx.x); idx < c.w * c.h; idx += (int)(blockDim.x * gridDim.y)) {
        if (!m[idx]) continue;
        const int y = idx / c.w, x = idx - y * c.w;
        const size_t px = (size_t)(c.y + y) * W + c.x + x;
        cur[px] = prop[px];
        labels[px] = pl;
    }
}

}  // namespace les
