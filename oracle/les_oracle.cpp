/*
 * les_oracle.cpp -- CPU restatement of the LocalExpStereo matching-cost hot path (see les_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (no reference golden vectors exist; see header).
 *
 * Build: g++ -O2 -std=c++17 -fopenmp -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off matters: the reference is built by MSVC x64 /O2 without FMA contraction
 * ("FMA contraction"), so every a*b+c below is two roundings.
 *
 * Citations: LES/<file>:<line> == /root/reference/LocalExpansionStereo/<file>:<line>.
 * "[recollection]" marks restated OpenCV 3.1 behaviour whose source is not under /root/reference.
 */
#include "les_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

// =====================================================================================================
// cv::RNG  [recollection: OpenCV 3.1 modules/core/include/opencv2/core/operations.hpp]
//   state = (uint32)state * 4164903690 + (state >> 32);  next() = (uint32)state
// =====================================================================================================
extern "C" void les_rng_seed(les_rng* r, uint64_t seed) { r->state = seed ? seed : 0xffffffffULL; }

extern "C" uint32_t les_rng_next(les_rng* r)
{
    r->state = (uint64_t)(uint32_t)r->state * 4164903690ULL + (uint32_t)(r->state >> 32);
    return (uint32_t)r->state;
}
extern "C" int les_rng_uniform_int(les_rng* r, int a, int b)
{
    return a == b ? a : (int)(les_rng_next(r) % (uint32_t)(b - a) + a);
}
extern "C" float les_rng_uniform_float(les_rng* r, float a, float b)
{
    float f = les_rng_next(r) * 2.3283064365386962890625e-10f;
    return f * (b - a) + a;
}
extern "C" double les_rng_uniform_double(les_rng* r, double a, double b)
{
    uint32_t t = les_rng_next(r);
    double d = (double)(((uint64_t)t << 32) | les_rng_next(r)) * 5.4210108624275221700372640043497e-20;
    return d * (b - a) + a;
}

// =====================================================================================================
// Plane  (LES/Plane.h)
// =====================================================================================================
extern "C" les_plane les_plane_create(float nx, float ny, float nz, float z, float x, float y, float v)
{
    les_plane p;                 // LES/Plane.h:23-31
    p.a = -nx / nz;
    p.b = -ny / nz;
    p.c = z - p.a * x - p.b * y;
    p.v = v;
    return p;
}
extern "C" void les_plane_normal(const les_plane* p, float n[3])
{
    // LES/Plane.h:42-50: "Calc sqrt in double then cast to float."  1.0 + a*a + b*b: a*a and b*b are
    // float products promoted to double for the additions.
    float nz = float(1.0 / sqrt(1.0 + p->a * p->a + p->b * p->b));
    n[0] = -p->a * nz;
    n[1] = -p->b * nz;
    n[2] = nz;
}
extern "C" float les_plane_z(const les_plane* p, float x, float y)
{
    return p->a * x + p->b * y + p->c;   // LES/Plane.h:51-54
}

// =====================================================================================================
// LayerManager::addLayer  (LES/LayerManager.h:88-185, the #else branch that is compiled)
// =====================================================================================================
struct les_layer {
    int heightBlocks, widthBlocks, unit;
    std::vector<les_rect> unitRegions, sharedRegions, filterRegions;
    std::vector<std::vector<int>> sets;
};

static les_rect rect_and(les_rect a, les_rect b)
{   // cv::Rect operator& [recollection]: intersection, empty -> Rect()
    int x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
    int x2 = std::min(a.x + a.w, b.x + b.w), y2 = std::min(a.y + a.h, b.y + b.h);
    les_rect r = {x1, y1, x2 - x1, y2 - y1};
    if (r.w <= 0 || r.h <= 0) r = les_rect{0, 0, 0, 0};
    return r;
}

extern "C" les_layer* les_layer_create(int width, int height, int windowR, int unitRegionSize)
{
    les_layer* L = new les_layer();
    les_layer& layer = *L;
    layer.unit = unitRegionSize;                                           // :92
    int minsize = std::max(2, unitRegionSize / 2);                         // :93
    int frac_h = height % unitRegionSize;                                  // :94
    int frac_w = width % unitRegionSize;                                   // :95
    int split_h = frac_h >= minsize ? 1 : 0;                               // :96
    int split_w = frac_w >= minsize ? 1 : 0;                               // :97
    layer.heightBlocks = (height / unitRegionSize) + split_h;              // :99
    layer.widthBlocks = (width / unitRegionSize) + split_w;                // :100
    int n = layer.heightBlocks * layer.widthBlocks;
    layer.sharedRegions.resize(n);
    layer.filterRegions.resize(n);
    layer.unitRegions.resize(n);
    les_rect imageDomain = {0, 0, width, height};                          // :107
    for (int i = 0; i < layer.heightBlocks; i++)
        for (int j = 0; j < layer.widthBlocks; j++) {
            int r = i * layer.widthBlocks + j;
            les_rect u = {j * unitRegionSize, i * unitRegionSize, unitRegionSize, unitRegionSize};   // :117-120
            layer.unitRegions[r] = rect_and(u, imageDomain);                                         // :121
            les_rect s = {(j - 1) * unitRegionSize, (i - 1) * unitRegionSize, unitRegionSize * 3, unitRegionSize * 3};
            layer.sharedRegions[r] = rect_and(s, imageDomain);                                       // :123-127
            les_rect f = {(j - 1) * unitRegionSize - windowR, (i - 1) * unitRegionSize - windowR,
                          unitRegionSize * 3 + windowR * 2, unitRegionSize * 3 + windowR * 2};
            layer.filterRegions[r] = rect_and(f, imageDomain);                                       // :129-133
        }
    if (split_w == 0) {                                                    // :138-151
        for (int i = 0; i < layer.heightBlocks; i++) {
            int x1 = i * layer.widthBlocks + layer.widthBlocks - 1;
            layer.unitRegions[x1].w += frac_w;
        }
        for (int i = 0; i < layer.heightBlocks; i++) {
            int x1 = i * layer.widthBlocks + layer.widthBlocks - 2;
            if (layer.widthBlocks - 2 < 0) continue;  // (reference would index out of range; never hit for shipped sizes)
            layer.sharedRegions[x1].w += frac_w;
            layer.filterRegions[x1].w += frac_w;
            layer.filterRegions[x1] = rect_and(layer.filterRegions[x1], imageDomain);
        }
    }
    if (split_h == 0) {                                                    // :152-165
        for (int j = 0; j < layer.widthBlocks; j++) {
            int y1 = (layer.heightBlocks - 1) * layer.widthBlocks + j;
            layer.unitRegions[y1].h += frac_h;
        }
        for (int j = 0; j < layer.widthBlocks; j++) {
            if (layer.heightBlocks - 2 < 0) continue;
            int y1 = (layer.heightBlocks - 2) * layer.widthBlocks + j;
            layer.sharedRegions[y1].h += frac_h;
            layer.filterRegions[y1].h += frac_h;
            layer.filterRegions[y1] = rect_and(layer.filterRegions[y1], imageDomain);
        }
    }
    std::vector<std::vector<int>> sets(16);                                // :106, :168-173
    for (int i = 0; i < layer.heightBlocks; i++)
        for (int j = 0; j < layer.widthBlocks; j++)
            sets[(i % 4) * 4 + (j % 4)].push_back(i * layer.widthBlocks + j);
    for (auto& s : sets)                                                   // :174-182 (erase empty sets)
        if (!s.empty()) layer.sets.push_back(s);
    return L;
}
extern "C" void les_layer_destroy(les_layer* L) { delete L; }
extern "C" int les_layer_height_blocks(const les_layer* L) { return L->heightBlocks; }
extern "C" int les_layer_width_blocks(const les_layer* L) { return L->widthBlocks; }
extern "C" int les_layer_num_cells(const les_layer* L) { return (int)L->unitRegions.size(); }
extern "C" void les_layer_rects(const les_layer* L, les_rect* unit, les_rect* shared, les_rect* filter)
{
    size_t n = L->unitRegions.size();
    if (unit) memcpy(unit, L->unitRegions.data(), n * sizeof(les_rect));
    if (shared) memcpy(shared, L->sharedRegions.data(), n * sizeof(les_rect));
    if (filter) memcpy(filter, L->filterRegions.data(), n * sizeof(les_rect));
}
extern "C" int les_layer_num_sets(const les_layer* L) { return (int)L->sets.size(); }
extern "C" int les_layer_set_size(const les_layer* L, int s) { return (int)L->sets[s].size(); }
extern "C" void les_layer_set_cells(const les_layer* L, int s, int* cells)
{
    memcpy(cells, L->sets[s].data(), L->sets[s].size() * sizeof(int));
}

// =====================================================================================================
// cv::boxFilter(src, dst, -1, Size(2R+1,2R+1), Point(-1,-1), normalize=false, BORDER_CONSTANT)
// (call site LES/GuidedFilter.h:43).  [recollection] OpenCV 3.1 imgproc/smooth.cpp: separable
// RowSum<T,double> followed by ColumnSum<double,T>; for CV_32F/CV_64F sources the sum type is
// CV_64F; border value 0; anchor at the kernel centre.  Every Mat the reference passes here is a
// standalone (non-ROI) matrix, so the zero border sits exactly at the matrix edge.
// The running-sum order below follows RowSum ("s += S[i+ksz] - S[i]") and ColumnSum
// ("s0 = SUM[i] + Sp[i]; D[i] = s0; SUM[i] = s0 - Sm[i]").
// =====================================================================================================
template <typename T>
static void boxfilter(const T* src, T* dst, int rows, int cols, int R)
{
    const int ksz = 2 * R + 1;
    // scratch is thread-local and reused across calls (allocation only, no effect on results)
    static thread_local std::vector<double> rowsum, ext, SUM;
    rowsum.resize((size_t)rows * cols);
    ext.resize(cols + 2 * R);
    for (int y = 0; y < rows; y++) {
        for (int i = 0; i < R; i++) ext[i] = 0.0, ext[cols + R + i] = 0.0;
        for (int x = 0; x < cols; x++) ext[R + x] = (double)src[(size_t)y * cols + x];
        double s = 0;
        for (int i = 0; i < ksz; i++) s += ext[i];
        double* D = &rowsum[(size_t)y * cols];
        D[0] = s;
        for (int i = 0; i < cols - 1; i++) {
            s += ext[i + ksz] - ext[i];
            D[i + 1] = s;
        }
    }
    SUM.assign(cols, 0.0);
    auto rowptr = [&](int y) -> const double* {   // rows outside [0,rows) are the zero border
        return (y < 0 || y >= rows) ? nullptr : &rowsum[(size_t)y * cols];
    };
    for (int y = -R; y < R; y++) {                // first ksize-1 rows
        const double* Sp = rowptr(y);
        if (Sp) for (int i = 0; i < cols; i++) SUM[i] += Sp[i];
    }
    for (int y = 0; y < rows; y++) {
        const double* Sp = rowptr(y + R);
        const double* Sm = rowptr(y - R);
        T* D = dst + (size_t)y * cols;
        for (int i = 0; i < cols; i++) {
            double s0 = SUM[i] + (Sp ? Sp[i] : 0.0);
            D[i] = (T)s0;
            SUM[i] = s0 - (Sm ? Sm[i] : 0.0);
        }
    }
}

// =====================================================================================================
// GuidedImageFilter<T> / FastGuidedImageFilter<T>  (LES/GuidedFilter.h:28-327)
// =====================================================================================================
template <typename T>
struct GuideStats {
    int H = 0, W = 0, R = 0;
    double eps = 0;
    std::vector<T> I[3], mean_I[3], invrr, invrg, invrb, invgg, invgb, invbb, N;

    // LES/GuidedFilter.h:58-102.  `img` is H x W x 3 uint8 interleaved (channel 0 first).
    void build(const uint8_t* img, int H_, int W_, int R_, double eps_, double scaling)
    {
        H = H_; W = W_; R = R_; eps = eps_;
        size_t P = (size_t)H * W;
        for (int c = 0; c < 3; c++) {
            I[c].resize(P);
            // :65 I.convertTo(realI, DEPTH, scaling): [recollection] for a CV_64F destination the
            // scale is applied in double, for CV_32F in float.
            for (size_t i = 0; i < P; i++) {
                if (sizeof(T) == 8) I[c][i] = (T)((double)img[i * 3 + c] * scaling);
                else                I[c][i] = (T)((float)img[i * 3 + c] * (float)scaling);
            }
        }
        std::vector<T> ones(P, (T)1), tmp(P), prod(P);
        N.resize(P);
        boxfilter(ones.data(), N.data(), H, W, R);                                   // :69
        for (int c = 0; c < 3; c++) {                                                // :70-72
            mean_I[c].resize(P);
            boxfilter(I[c].data(), tmp.data(), H, W, R);
            for (size_t i = 0; i < P; i++) mean_I[c][i] = tmp[i] / N[i];
        }
        auto var = [&](int a, int b, bool diag) {                                    // :79-84
            std::vector<T> v(P);
            for (size_t i = 0; i < P; i++) prod[i] = I[a][i] * I[b][i];
            boxfilter(prod.data(), tmp.data(), H, W, R);
            for (size_t i = 0; i < P; i++) {
                // cv::MatExpr evaluation order: (box/N - mean.mul(mean)) + eps, all in T
                T t = tmp[i] / N[i] - mean_I[a][i] * mean_I[b][i];
                v[i] = diag ? (T)(t + (T)eps) : t;   // Mat + double scalar: [recollection] scalar cast to T's work type
            }
            return v;
        };
        std::vector<T> rr = var(0, 0, true), rg = var(0, 1, false), rb = var(0, 2, false);
        std::vector<T> gg = var(1, 1, true), gb = var(1, 2, false), bb = var(2, 2, true);
        invrr.resize(P); invrg.resize(P); invrb.resize(P); invgg.resize(P); invgb.resize(P); invbb.resize(P);
        for (size_t i = 0; i < P; i++) {                                             // :87-101
            T irr = gg[i] * bb[i] - gb[i] * gb[i];
            T irg = gb[i] * rb[i] - rg[i] * bb[i];
            T irb = rg[i] * gb[i] - gg[i] * rb[i];
            T igg = rr[i] * bb[i] - rb[i] * rb[i];
            T igb = rb[i] * rg[i] - rr[i] * gb[i];
            T ibb = rr[i] * gg[i] - rg[i] * rg[i];
            T covDet = irr * rr[i] + irg * rg[i] + irb * rb[i];
            invrr[i] = irr / covDet; invrg[i] = irg / covDet; invrb[i] = irb / covDet;
            invgg[i] = igg / covDet; invgb[i] = igb / covDet; invbb[i] = ibb / covDet;
        }
    }

    // createSubregionFilter(rect) (:301-326) + filter(p) (:248-266) -> filter_raw (:142-247).
    // p, q are dense rect.h x rect.w float images.
    void filter_subregion(les_rect rect, const float* p_in, float* q_out) const
    {
        const int rows = rect.h, cols = rect.w;
        const size_t P = (size_t)rows * cols;
        // thread-local scratch reused across calls (allocation only, no effect on results)
        static thread_local std::vector<T> Nloc, ones, p, mean_p, mIp[3], a[3], b, tmp, Ba[3], Bb;
        Nloc.resize(P); ones.assign(P, (T)1); p.resize(P); mean_p.resize(P); b.resize(P); tmp.resize(P); Bb.resize(P);
        boxfilter(ones.data(), Nloc.data(), rows, cols, R);                          // :324
        for (size_t i = 0; i < P; i++) p[i] = (T)p_in[i];                            // :251
        boxfilter(p.data(), mean_p.data(), rows, cols, R);                           // :145
        auto g = [&](const std::vector<T>& plane, int i, int j) -> T {               // ROI slice plane(rect)
            return plane[(size_t)(rect.y + i) * W + rect.x + j];
        };
        for (int c = 0; c < 3; c++) {                                                // :151-172
            mIp[c].resize(P);
            for (int i = 0; i < rows; i++)
                for (int j = 0; j < cols; j++) tmp[(size_t)i * cols + j] = g(I[c], i, j) * p[(size_t)i * cols + j];
            boxfilter(tmp.data(), mIp[c].data(), rows, cols, R);
            a[c].resize(P);
        }
        for (int i = 0; i < rows; i++)                                               // :180-222
            for (int j = 0; j < cols; j++) {
                size_t k = (size_t)i * cols + j;
                T n = Nloc[k];
                T mp = mean_p[k] / n;
                T mIr = g(mean_I[0], i, j), mIg = g(mean_I[1], i, j), mIb = g(mean_I[2], i, j);
                T cov_r = mIp[0][k] / n - mIr * mp;
                T cov_g = mIp[1][k] / n - mIg * mp;
                T cov_b = mIp[2][k] / n - mIb * mp;
                T ar = g(invrr, i, j) * cov_r + g(invrg, i, j) * cov_g + g(invrb, i, j) * cov_b;
                T ag = g(invrg, i, j) * cov_r + g(invgg, i, j) * cov_g + g(invgb, i, j) * cov_b;
                T ab = g(invrb, i, j) * cov_r + g(invgb, i, j) * cov_g + g(invbb, i, j) * cov_b;
                a[0][k] = ar; a[1][k] = ag; a[2][k] = ab;
                b[k] = mp - ar * mIr - ag * mIg - ab * mIb;
            }
        for (int c = 0; c < 3; c++) { Ba[c].resize(P); boxfilter(a[c].data(), Ba[c].data(), rows, cols, R); }   // :224-226
        boxfilter(b.data(), Bb.data(), rows, cols, R);                                                           // :227
        for (int i = 0; i < rows; i++)                                               // :229-245
            for (int j = 0; j < cols; j++) {
                size_t k = (size_t)i * cols + j;
                T q = (Bb[k] + Ba[0][k] * g(I[0], i, j) + Ba[1][k] * g(I[1], i, j) + Ba[2][k] * g(I[2], i, j)) / Nloc[k];
                q_out[k] = (float)q;                                                 // :260-261
            }
    }
};

// =====================================================================================================
// CostVolumeEnergy  (LES/CostVolumeEnergy.h) + StereoEnergy::IsValiLabel (LES/StereoEnergy.h:560-610)
// =====================================================================================================
struct les_oracle {
    int H, W, D, windR, use_float;
    float th_col, MAXD, MIND;
    const float* vol[2];
    GuideStats<double> gd[2];
    GuideStats<float> gf[2];
    // NaiveStereoEnergy (LES/StereoEnergy.h:629-764): kind == 1
    int kind = 0;
    std::vector<float> ExI[2];        // H x W x 4: (1-alpha)*B, (1-alpha)*G, (1-alpha)*R, alpha*Gx
    float thresh_color = 0, thresh_gradient = 0;
};

extern "C" les_oracle* les_oracle_create(const uint8_t* imL, const uint8_t* imR, int H, int W,
                                         const float* volL, const float* volR, int D, int windR, double eps,
                                         float th_col, float max_disparity, float min_disparity, int use_float)
{
    les_oracle* o = new les_oracle();
    o->H = H; o->W = W; o->D = D; o->windR = windR; o->use_float = use_float;
    o->th_col = th_col; o->MAXD = max_disparity; o->MIND = min_disparity;
    o->vol[0] = volL; o->vol[1] = volR;
    const uint8_t* im[2] = {imL, imR};
    for (int m = 0; m < 2; m++) {
        if (!im[m]) continue;
        // LES/CostVolumeEnergy.h:30-31 / :35-36: radius windR/2, eps = filter_param1, scaling 1/255
        if (use_float) o->gf[m].build(im[m], H, W, windR / 2, eps, 1.0 / 255);
        else           o->gd[m].build(im[m], H, W, windR / 2, eps, 1.0 / 255);
    }
    return o;
}
// NaiveStereoEnergy constructor, LES/StereoEnergy.h:638-689 (filterName "GF").  [recollection] of the OpenCV calls:
// convertTo(CV_32F) of the 8-bit image; cvtColor(BGR2GRAY) on float = 0.114 B + 0.587 G + 0.299 R;
// Sobel(dx=1, dy=0, ksize=1, scale=0.5, BORDER_REPLICATE) = 0.5 * (g(x+1) - g(x-1)).
extern "C" les_oracle* les_oracle_create_naive(const uint8_t* imL, const uint8_t* imR, int H, int W, int windR, double eps,
                                               float alpha, float th_col, float th_grad, float max_disparity, float min_disparity)
{
    les_oracle* o = new les_oracle();
    o->kind = 1;
    o->H = H; o->W = W; o->D = 0; o->windR = windR; o->use_float = 0;
    o->th_col = th_col; o->MAXD = max_disparity; o->MIND = min_disparity;
    o->vol[0] = o->vol[1] = nullptr;
    o->thresh_color = th_col * (1.0f - alpha);                                      // :662
    o->thresh_gradient = th_grad * alpha;                                           // :663
    const uint8_t* im[2] = {imL, imR};
    const size_t P = (size_t)H * W;
    for (int m = 0; m < 2; m++) {
        o->gd[m].build(im[m], H, W, windR / 2, eps, 1.0 / 255);                     // :673-674
        std::vector<float> gray(P);
        for (size_t i = 0; i < P; i++)
            gray[i] = (float)im[m][i * 3] * 0.114f + (float)im[m][i * 3 + 1] * 0.587f + (float)im[m][i * 3 + 2] * 0.299f;   // :651
        o->ExI[m].resize(P * 4);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t i = (size_t)y * W + x;
                const float gx = 0.5f * (gray[(size_t)y * W + std::min(x + 1, W - 1)] - gray[(size_t)y * W + std::max(x - 1, 0)]);   // :654
                for (int c = 0; c < 3; c++) o->ExI[m][i * 4 + c] = (float)((double)im[m][i * 3 + c] * (1.0 - (double)alpha));          // :658
                o->ExI[m][i * 4 + 3] = gx * alpha;                                                                                     // :659
            }
    }
    return o;
}
extern "C" void les_oracle_destroy(les_oracle* o) { delete o; }

// NaiveStereoEnergy::ComputeUnaryPotentialWithoutCheck raw-cost part, LES/StereoEnergy.h:702-742.  The affine map
// through the three corner points reproduces the plane warp x_src = x - sign * (a x + b y + c), y_src = y exactly
// (the plane is affine).  [recollection] cv::warpAffine(INTER_LINEAR, BORDER_REPLICATE) on float images: source
// coordinates are quantised to 1/32 pixel, bilinear weights from that fraction.  Tolerance-level restatement only.
static void naive_raw(const les_oracle* o, int mode, les_rect fr, les_plane plane, float* raw)
{
    const float sign = mode ? -1.f : 1.f;
    const std::vector<float>& I0 = o->ExI[mode];
    const std::vector<float>& I1 = o->ExI[1 - mode];
    const int W = o->W;
    for (int y = 0; y < fr.h; y++)
        for (int x = 0; x < fr.w; x++) {
            const int X = fr.x + x, Y = fr.y + y;
            const float z = plane.a * (float)X + plane.b * (float)Y + plane.c;
            const double sx = (double)X - (double)sign * z;
            const double q = std::floor(sx * 32.0 + 0.5) / 32.0;                     // INTER_BITS = 5
            const double fl = std::floor(q);
            const float w1 = (float)(q - fl), w0 = 1.0f - w1;
            const int x0 = (int)std::fmin(std::fmax(fl, -2.0), (double)W + 1.0);     // defined conversion for NaN / huge planes
            const int xa = std::min(std::max(x0, 0), W - 1), xb = std::min(std::max(x0 + 1, 0), W - 1);
            const float* a = &I1[((size_t)Y * W + xa) * 4];
            const float* b = &I1[((size_t)Y * W + xb) * 4];
            const float* p0 = &I0[((size_t)Y * W + X) * 4];
            float v[4];
            for (int c = 0; c < 4; c++) v[c] = w0 * a[c] + w1 * b[c];
            const float col = (std::fabs(p0[0] - v[0]) + std::fabs(p0[1] - v[1])) + std::fabs(p0[2] - v[2]);
            raw[(size_t)y * fr.w + x] = std::min(o->thresh_color, col) + std::min(o->thresh_gradient, std::fabs(p0[3] - v[3]));   // :738-740
        }
}

extern "C" void les_oracle_get_stats(const les_oracle* o, int mode, double* out)
{
    size_t P = (size_t)o->H * o->W;
    auto put = [&](int k, auto& v) { for (size_t i = 0; i < P; i++) out[k * P + i] = (double)v[i]; };
    if (o->use_float) {
        const auto& g = o->gf[mode];
        for (int c = 0; c < 3; c++) { put(c, g.I[c]); put(3 + c, g.mean_I[c]); }
        put(6, g.invrr); put(7, g.invrg); put(8, g.invrb); put(9, g.invgg); put(10, g.invgb); put(11, g.invbb); put(12, g.N);
    } else {
        const auto& g = o->gd[mode];
        for (int c = 0; c < 3; c++) { put(c, g.I[c]); put(3 + c, g.mean_I[c]); }
        put(6, g.invrr); put(7, g.invrg); put(8, g.invrb); put(9, g.invgg); put(10, g.invgb); put(11, g.invbb); put(12, g.N);
    }
}

// LES/CostVolumeEnergy.h:64-98 (interpolate == 1, the only mode ever selected: ctor :18).
extern "C" void les_oracle_gather(const les_oracle* o, int mode, les_rect fr, les_plane plane, float* raw)
{
    const float* vol = o->vol[mode];
    const size_t HW = (size_t)o->H * o->W;
    const int D = o->D;
    const int D0 = int(-o->MIND);                                                    // :67
    const float MIN_DISPARITY = o->MIND, MAX_DISPARITY = o->MAXD;
    const int y0 = fr.y + fr.h, x0 = fr.x + fr.w;
    for (int y = fr.y; y < y0; y++) {
        float* pC = raw + (size_t)(y - fr.y) * fr.w;
        float d_base = plane.b * y + plane.c;                                        // :73
        for (int x = fr.x; x < x0; x++) {
            float d = plane.a * x + d_base;                                          // :76
            float C;
            size_t px = (size_t)y * o->W + x;
            if (d < MIN_DISPARITY) C = vol[px];                                      // :78
            else if (d >= MAX_DISPARITY) C = vol[(size_t)(D - 1) * HW + px];         // :79
            else if (std::isnan(d) || std::isinf(d)) C = LES_COST_FOR_INVALID;       // :80
            else {
                int d0 = int(d) + D0;                                                // :83
                int d1 = d0 + 1;
                float f1 = d - std::floor(d);                                        // :85
                float f0 = 1.0f - f1;
                if (d1 >= D || d0 < 0) C = LES_COST_FOR_INVALID;                     // :87-90 (diagnostic printf omitted)
                else C = f0 * vol[(size_t)d0 * HW + px] + f1 * vol[(size_t)d1 * HW + px];   // :92
            }
            pC[x - fr.x] = (o->th_col < C) ? o->th_col : C;                          // :96 std::min(C, th_col)
        }
    }
}

extern "C" void les_oracle_filter_subregion(const les_oracle* o, int mode, les_rect fr, const float* p, float* q)
{
    if (o->use_float) o->gf[mode].filter_subregion(fr, p, q);
    else              o->gd[mode].filter_subregion(fr, p, q);
}

static bool valid_at(const les_oracle* o, float ds, float a5, float b5)
{   // LES/StereoEnergy.h:567-573 == :600-605
    const float MIN_DISPARITY = o->MIND, MAX_DISPARITY = o->MAXD;
    float d;
    return (ds >= MIN_DISPARITY && ds <= MAX_DISPARITY
            && ((d = ds + a5 + b5) >= MIN_DISPARITY) && d <= MAX_DISPARITY
            && ((d = ds + a5 - b5) >= MIN_DISPARITY) && d <= MAX_DISPARITY
            && ((d = ds - a5 + b5) >= MIN_DISPARITY) && d <= MAX_DISPARITY
            && ((d = ds - a5 - b5) >= MIN_DISPARITY) && d <= MAX_DISPARITY);
}

extern "C" void les_oracle_valid_mask(const les_oracle* o, les_rect pos, les_plane label, uint8_t* mask)
{
    float a5 = label.a * 5;                                                          // :563 / :586
    float b5 = label.b * 5;
    for (int y = 0; y < pos.h; y++)
        for (int x = 0; x < pos.w; x++) {
            float fx = (float)(pos.x + x), fy = (float)(pos.y + y);
            float ds;
            if (pos.w == 1 && pos.h == 1) {
                ds = label.a * fx + label.b * fy + label.c;                          // :562 via Plane::GetZ(cv::Point) LES/Plane.h:55-58
            } else {
                // :592 channelSum(coordinates(pos).mul(label.toScalar())): coordinates = (x, y, 1, 0)
                // (:114); [recollection] Mat.mul(Scalar) multiplies in float, cv::reduce(SUM) over the
                // 4 channels accumulates left to right in float.
                ds = ((fx * label.a + fy * label.b) + 1.0f * label.c) + 0.0f * label.v;
            }
            mask[(size_t)y * pos.w + x] = valid_at(o, ds, a5, b5) ? 255 : 0;
        }
}

extern "C" void les_oracle_unary_nocheck(const les_oracle* o, int mode, les_rect fr, les_rect tr,
                                         float* costs, int stride, les_plane plane)
{
    // LES/CostVolumeEnergy.h:55-174.  The Reusable scratch (pIL + sub-region filter, :57-62) is
    // recreated per call here; it only caches, it does not change results.
    static thread_local std::vector<float> pIL, q;
    pIL.resize((size_t)fr.w * fr.h); q.resize((size_t)fr.w * fr.h);
    if (o->kind == 1) naive_raw(o, mode, fr, plane, pIL.data());                     // LES/StereoEnergy.h:730-742
    else les_oracle_gather(o, mode, fr, plane, pIL.data());
    les_oracle_filter_subregion(o, mode, fr, pIL.data(), q.data());                  // :171 / LES/StereoEnergy.h:747
    int sx = tr.x - fr.x, sy = tr.y - fr.y;                                          // :169 subrect = targetRect - filterRect.tl()
    for (int y = 0; y < tr.h; y++)
        for (int x = 0; x < tr.w; x++)
            costs[(size_t)(sy + y) * stride + sx + x] = q[(size_t)(sy + y) * fr.w + sx + x];
}

extern "C" void les_oracle_unary(const les_oracle* o, int mode, les_rect fr, les_rect tr,
                                 float* costs, int stride, les_plane plane)
{
    les_oracle_unary_nocheck(o, mode, fr, tr, costs, stride, plane);                 // :178
    std::vector<uint8_t> mask((size_t)tr.w * tr.h);
    les_oracle_valid_mask(o, tr, plane, mask.data());                                // :181
    int sx = tr.x - fr.x, sy = tr.y - fr.y;
    for (int y = 0; y < tr.h; y++)                                                   // :182 setTo(COST_FOR_INVALID, ~validMask)
        for (int x = 0; x < tr.w; x++)
            if (!mask[(size_t)y * tr.w + x]) costs[(size_t)(sy + y) * stride + sx + x] = LES_COST_FOR_INVALID;
}

extern "C" void les_oracle_unary_batch(const les_oracle* o, int mode, int n, const les_rect* frs,
                                       const les_rect* trs, const les_plane* planes, float* cost_map,
                                       int check, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    #pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        float* origin = cost_map + (size_t)frs[i].y * o->W + frs[i].x;               // proposalCost(filterRect), LES/FastGCStereo.h:49
        if (check) les_oracle_unary(o, mode, frs[i], trs[i], origin, o->W, planes[i]);
        else       les_oracle_unary_nocheck(o, mode, frs[i], trs[i], origin, o->W, planes[i]);
    }
}

extern "C" void les_oracle_aggregate_planes(const les_oracle* o, int mode, int n, const les_plane* planes,
                                            float* out, int check, int nthreads)
{
    // whole-image aggregation of n hypothesis planes into out[n][H][W]: n calls of
    // ComputeUnaryPotential with filterRect = targetRect = image, OpenMP over planes.
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const les_rect full = {0, 0, o->W, o->H};
    #pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        float* slab = out + (size_t)i * o->H * o->W;
        if (check) les_oracle_unary(o, mode, full, full, slab, o->W, planes[i]);
        else       les_oracle_unary_nocheck(o, mode, full, full, slab, o->W, planes[i]);
    }
}

extern "C" void les_oracle_wta_update(int W, les_rect r, float* cur, const float* prop, les_plane* labels, les_plane plane)
{
    for (int y = r.y; y < r.y + r.h; y++)                                            // LES/FastGCStereo.h:57-60
        for (int x = r.x; x < r.x + r.w; x++) {
            size_t k = (size_t)y * W + x;
            if (cur[k] > prop[k]) { cur[k] = prop[k]; labels[k] = plane; }
        }
}

// =====================================================================================================
// Label generation
// =====================================================================================================
extern "C" void les_random_unit_vector(les_rng* r, double thetaRange, double n[3])
{   // LES/Utilities.hpp:254-261
    double theta = les_rng_uniform_double(r, 0.0, thetaRange);
    double phi = les_rng_uniform_double(r, 0.0, M_PI * 2.0);
    double cosT = cos(theta), sinT = sin(theta);
    double cosP = cos(phi), sinP = sin(phi);
    n[0] = sinT * cosP; n[1] = sinT * sinP; n[2] = cosT;
}

extern "C" les_plane les_create_random_label(les_rng* r, float min_disp, float max_disp, int sx, int sy)
{   // LES/StereoEnergy.h:120-129 (MAX_VDISPARITY == 0 -> vs = 0, no draw)
    float zs = les_rng_uniform_float(r, min_disp, max_disp);
    double n[3];
    les_random_unit_vector(r, M_PI / 3, n);
    // Plane::CreatePlane(cv::Vec<float,3> n, ...): the Vec3d is converted to Vec3f (LES/Plane.h:32-35)
    return les_plane_create((float)n[0], (float)n[1], (float)n[2], zs, (float)sx, (float)sy, 0.0f);
}

extern "C" void les_select_random_pixel(les_rng* r, les_rect rect, int* px, int* py)
{   // LES/Proposer.h:37-44 == LES/FastGCStereo.h:231-238
    int n = les_rng_uniform_int(r, 0, rect.h * rect.w);
    *px = rect.x + n % rect.w;
    *py = rect.y + n / rect.w;
}

extern "C" les_plane les_expansion_proposal(les_rng* r, const les_plane* labels, int W, les_rect unit)
{   // LES/Proposer.h:69-75: labeling = labeling(unitRegion) (:64), pixel drawn in Rect(0,0,cols,rows)
    int px, py;
    les_select_random_pixel(r, les_rect{0, 0, unit.w, unit.h}, &px, &py);
    return labels[(size_t)(unit.y + py) * W + unit.x + px];
}

extern "C" float les_random_perturbation_width(float min_disp, float max_disp, int m)
{   // LES/Proposer.h:93-96: (MAX - MIN) * pow(0.5f, m + 1); pow(float,int) evaluates in double
    return (float)((max_disp - min_disp) * pow((double)0.5f, (double)(m + 1)));
}

extern "C" int les_random_is_continued(int iter, int K, int outerIter, float min_disp, float max_disp)
{   // LES/Proposer.h:149-152 (doEarlyStop = true)
    return (iter < K) && !(les_random_perturbation_width(min_disp, max_disp, outerIter + iter) < 0.1);
}

extern "C" les_plane les_random_proposal(les_rng* r, const les_plane* labels, int W, les_rect unit, int m,
                                         float MIN_DISPARITY, float MAX_DISPARITY)
{   // LES/Proposer.h:120-148 (MAX_VDISPARITY == 0)
    int px, py;
    les_select_random_pixel(r, les_rect{0, 0, unit.w, unit.h}, &px, &py);            // :122
    les_plane in = labels[(size_t)(unit.y + py) * W + unit.x + px];                  // :123
    int sx = unit.x + px, sy = unit.y + py;                                          // :127
    float zs = les_plane_z(&in, float(sx), float(sy));                               // :128
    float dz = les_random_perturbation_width(MIN_DISPARITY, MAX_DISPARITY, m);       // :129
    float minz = std::max(MIN_DISPARITY, zs - dz);                                   // :130
    float maxz = std::min(MAX_DISPARITY, zs + dz);                                   // :131
    zs = les_rng_uniform_float(r, minz, maxz);                                       // :132
    float vs = in.v;                                                                 // :134
    float nr = (float)(1.0f * pow((double)0.5f, (double)m));                         // :142 randomNmax = 1.0
    float n0[3];
    les_plane_normal(&in, n0);
    double u[3];
    les_random_unit_vector(r, M_PI, u);                                              // :143 default thetaRange = CV_PI
    float nv[3];
    for (int i = 0; i < 3; i++) nv[i] = n0[i] + (float)u[i] * nr;                    // (Vec3f)Vec3d * nr
    double dd = (double)nv[0] * nv[0] + (double)nv[1] * nv[1] + (double)nv[2] * nv[2];   // ddot
    double inv = 1. / sqrt(dd);                                                      // :145 Matx / double -> * (1./alpha) [recollection]
    for (int i = 0; i < 3; i++) nv[i] = (float)(nv[i] * inv);
    return les_plane_create(nv[0], nv[1], nv[2], zs, float(sx), float(sy), vs);      // :147
}

extern "C" int les_ransac_sample_count(int ni, int ptNum, int pf, double conf)
{   // LES/Proposer.h:243-262
    int SampleCnt;
    double q = 1.0;
    for (double a = (ni - pf + 1), b = (ptNum - pf + 1); a <= ni; a += 1.0, b += 1.0) q *= (a / b);
    const double eps = 1e-4;
    if ((1.0 - q) < eps) SampleCnt = 1;
    else SampleCnt = int(log(1.0 - conf) / log(1.0 - q));
    if (SampleCnt < 1) SampleCnt = 1;
    return SampleCnt;
}

// cv::solve(A, b, x, DECOMP_SVD) for an m x 3 float system  [recollection]: least-squares / minimum-norm
// solution through the SVD with singular values below (sum of w) * 2*FLT_EPSILON treated as zero.
// Restated through the 3x3 eigen-decomposition of A^T A in double (same pseudo-inverse; OpenCV runs a
// one-sided Jacobi SVD in float, so agreement is to float round-off, not bitwise).
static void solve_svd_mx3(const float* A, const float* b, int m, float x[3])
{
    // Normal equations in double, rows accumulated in their natural order.  (The reference delegates to OpenCV's float SVD, whose
    // internal order is unknowable here; round 4 removed an accumulation order that had been shaped after the device kernel's -- the
    // product's RANSAC is now compared with this restatement to float round-off, and this solve with an order-free numpy
    // least-squares solve in double, tests/test_oracle_proposers.py.)
    double M[3][3] = {{0}}, rhs[3] = {0};
    for (int i = 0; i < m; i++) {
        const double c[3] = {A[i * 3], A[i * 3 + 1], A[i * 3 + 2]};
        const double d = b[i];
        for (int r = 0; r < 3; r++) {
            rhs[r] += c[r] * d;
            for (int q = 0; q < 3; q++) M[r][q] += c[r] * c[q];
        }
    }
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
    for (int k = 0; k < 3; k++) { w[k] = sqrt(std::max(M[k][k], 0.0)); wsum += w[k]; }
    double thr = wsum * 2 * 1.1920929e-07;
    double out[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
        if (w[k] <= thr) continue;
        double proj = (V[0][k] * rhs[0] + V[1][k] * rhs[1] + V[2][k] * rhs[2]) / (w[k] * w[k]);
        for (int r = 0; r < 3; r++) out[r] += V[r][k] * proj;
    }
    for (int r = 0; r < 3; r++) x[r] = (float)out[r];
}

// cv::solve(A, b, x, DECOMP_SVD) of an m x 3 float system as restated above (exported for the order-free check against numpy)
extern "C" void les_oracle_solve_mx3(const float* A, const float* b, int m, float* x) { solve_svd_mx3(A, b, m, x); }

extern "C" les_plane les_ransac_proposal(les_rng* r, const les_plane* labels, int W, les_rect unit,
                                         int MAX_SAM, float conf, float threshold)
{
    // startIterations, LES/Proposer.h:283-301: snapshot coordinates and disparities of the unit region
    const int len = unit.w * unit.h;
    std::vector<float> pts((size_t)len * 3), disp(len);
    for (int y = 0; y < unit.h; y++)
        for (int x = 0; x < unit.w; x++) {
            float c0 = (float)x + unit.x, c1 = (float)y + unit.y;
            const les_plane& v = labels[(size_t)(y + unit.y) * W + x + unit.x];
            int k = y * unit.w + x;
            pts[k * 3 + 0] = c0; pts[k * 3 + 1] = c1; pts[k * 3 + 2] = 1.0f;
            disp[k] = v.a * c0 + v.b * c1 + v.c;                                     // :297
        }
    // RANSACPlane, :177-240
    int max_i = 3, max_sam = MAX_SAM, no_sam = 0, no_i_c = 0;
    float N[3] = {0, 0, 0}, result[3] = {0, 0, 0};
    std::vector<uint8_t> v(len);
    std::vector<float> A, b;
    auto count_inliers = [&](const float* n) {
        int cnt = 0;
        for (int i = 0; i < len; i++) {
            // cv::abs(pts * N - disp) < threshold : float GEMM (accumulated in double by cv::gemm
            // [recollection]) then float subtract
            float dot = (float)((double)pts[i * 3] * n[0] + (double)pts[i * 3 + 1] * n[1] + (double)pts[i * 3 + 2] * n[2]);
            v[i] = std::fabs(dot - disp[i]) < threshold;
            cnt += v[i];
        }
        return cnt;
    };
    while (no_sam < max_sam) {
        no_sam = no_sam + 1;
        // randperm(len) (:163-174) draws a full std::random_shuffle (rand()) and uses only its first three
        // entries: three distinct uniformly random indices.  Restated with a partial Fisher-Yates on `r`.
        int idx[3];
        for (int i = 0; i < 3; i++) {
            bool again;
            do {
                idx[i] = len > 0 ? les_rng_uniform_int(r, 0, len) : 0;
                again = false;
                for (int j = 0; j < i; j++) if (idx[j] == idx[i] && len > i) again = true;
            } while (again);
        }
        float ranpts[9], div[3];
        for (int i = 0; i < 3; i++) {
            for (int c = 0; c < 3; c++) ranpts[i * 3 + c] = pts[idx[i] * 3 + c];     // :199
            div[i] = disp[idx[i]];                                                   // :200
        }
        solve_svd_mx3(ranpts, div, 3, N);                                            // :203
        int no_i = count_inliers(N);                                                 // :204-206
        if (max_i < no_i) {
            // :211-222 -- QUIRK kept: the copy loop runs i < no_i (not i < len), so only inliers among
            // the first no_i points are used (compacted to rows j = 0, 1, ...) and the remaining rows of A, b stay zero.
            A.assign((size_t)no_i * 3, 0.0f);
            b.assign(no_i, 0.0f);
            for (int i = 0, j = 0; i < no_i; i++)
                if (v[i]) {
                    for (int c = 0; c < 3; c++) A[j * 3 + c] = pts[i * 3 + c];
                    b[j] = disp[i];
                    j++;
                }
            solve_svd_mx3(A.data(), b.data(), no_i, N);                              // :224
            int no = count_inliers(N);                                               // :225-227
            if (no > no_i_c) {                                                       // :229-236
                result[0] = N[0]; result[1] = N[1]; result[2] = N[2];
                no_i_c = no;
                max_i = no_i;
                max_sam = std::min(max_sam, les_ransac_sample_count(no, len, 3, conf));
            }
        }
    }
    return les_plane{result[0], result[1], result[2], 0.0f};                         // :239
}

// =====================================================================================================
// One disjoint set of cells of the PatchMatch-style loop, in the REFERENCE's own order (LES/FastGCStereo.h:30-64
// with doGC == false): OpenMP over cells; per cell: for each proposer, for each of its proposals:
// propose -> ComputeUnaryPotential(filter, shared) -> WTA update of cost and labels over the shared region.
// kinds[j] in {0: Expansion, 1: Random, 2: Ransac}, Ks[j] proposals each.  states: one generator per cell.
// prop_cost: scratch H x W map (like proposalCost, :25).
// =====================================================================================================
extern "C" void les_oracle_pm_set(const les_oracle* o, int mode, int n, const les_rect* units, const les_rect* shared,
                                  const les_rect* filter, uint64_t* states, int nprop, const int* kinds, const int* Ks,
                                  les_plane* labels, float* cur_cost, float* prop_cost, int iteration, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int W = o->W;
    #pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n; i++) {
        les_rng r{states[i]};
        for (int j = 0; j < nprop; j++) {
            for (int it = 0; it < Ks[j]; it++) {
                if (kinds[j] == 1 && !les_random_is_continued(it, Ks[j], iteration, o->MIND, o->MAXD)) break;   // LES/Proposer.h:149-152
                les_plane label;
                if (kinds[j] == 0) label = les_expansion_proposal(&r, labels, W, units[i]);
                else if (kinds[j] == 1) label = les_random_proposal(&r, labels, W, units[i], iteration + it, o->MIND, o->MAXD);
                else label = les_ransac_proposal(&r, labels, W, units[i], 500, 0.95f, 1.0f);
                les_oracle_unary(o, mode, filter[i], shared[i], prop_cost + (size_t)filter[i].y * W + filter[i].x, W, label);   // :49
                les_oracle_wta_update(W, shared[i], cur_cost, prop_cost, labels, label);                                        // :57-60
            }
        }
        states[i] = r.state;
    }
}

// initCurrentFast, LES/FastGCStereo.h:94-115: random label per layer-0 cell, cost of its unit region
extern "C" void les_oracle_pm_init(const les_oracle* o, int mode, int n, const les_rect* units, uint64_t* states,
                                   les_plane* labels, float* cur_cost, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int W = o->W, H = o->H, R = o->windR;
    #pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; i++) {
        les_rng r{states[i]};
        int px, py;
        les_select_random_pixel(&r, units[i], &px, &py);                              // :107
        les_plane label = les_create_random_label(&r, o->MIND, o->MAXD, px, py);     // :108
        for (int y = 0; y < units[i].h; y++)
            for (int x = 0; x < units[i].w; x++) labels[(size_t)(units[i].y + y) * W + units[i].x + x] = label;   // :109
        les_rect f = {units[i].x - R, units[i].y - R, units[i].w + 2 * R, units[i].h + 2 * R};
        f = rect_and(f, les_rect{0, 0, W, H});                                        // :112
        les_oracle_unary(o, mode, f, units[i], cur_cost + (size_t)f.y * W + f.x, W, label);   // :113
        states[i] = r.state;
    }
}

// =====================================================================================================
// Volume preparation (LES/main.cpp:146-199, margin = 0 as shipped: interp_margin = 0, :359)
// =====================================================================================================
extern "C" void les_fill_out_of_view(float* vol, int D, int H, int W, int mode)
{
    for (int d = 0; d < D; d++)
        for (int y = 0; y < H; y++) {
            float* row = vol + ((size_t)d * H + y) * W;
            if (mode == 0) {                                                         // :152-163
                int q = std::min(d, W - 1);   // (reference assumes d < W)
                float v = row[q];
                for (int x = 0; x < q; x++) row[x] = v;
            } else {                                                                 // :165-175
                int p = W - d;                // q - d - margin
                if (p < 1) p = 1;
                float v = row[p - 1];
                for (int x = p; x < W; x++) row[x] = v;
            }
        }
}

extern "C" void les_convert_volume_l2r(const float* src, float* dst, int D, int H, int W)
{   // LES/main.cpp:178-199 with margin = 0
    memcpy(dst, src, (size_t)D * H * W * sizeof(float));                             // :183 clone
    for (int d = 0; d < D && d < W; d++)
        for (int y = 0; y < H; y++) {
            const float* s0 = src + ((size_t)d * H + y) * W;
            float* s1 = dst + ((size_t)d * H + y) * W;
            for (int x = 0; x < W - d; x++) s1[x] = s0[x + d];                       // :189
            float edge1 = s0[W - 1];                                                 // :191
            for (int x = W - 1 - d; x < W; x++) s1[x] = edge1;                       // :193-194
        }
}

// =====================================================================================================
// Dual-view post-processing  (LES/PMStereoBase.h:111-256)
// =====================================================================================================
extern "C" void les_consistency_check(const float* dispL, const float* dispR, int H, int W, float dispThreshold, uint8_t* failL, uint8_t* failR)
{   // doConsistencyCheck, LES/PMStereoBase.h:111-144 (isValid is constant true, :80-83)
    const float* disp[2] = {dispL, dispR};
    uint8_t* fail[2] = {failL, failR};
    for (int i = 0; i < 2; i++) {
        const float sign = (i ? -1.0f : 1.0f);                                       // :121
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float ds = disp[i][(size_t)y * W + x];                         // :126
                const float v = (float)x - ds * sign + 0.5f;                         // :127
                uint8_t f = 128;                                                     // :135-138
                if (v > -1.0e9f && v < 1.0e9f) {                                     // int(v) is undefined otherwise; such pixels are "outside"
                    const int rx = int(v);
                    if (rx >= 0 && rx < W) {                                         // :130 imageDomain.contains(q)
                        const float dsr = disp[1 - i][(size_t)y * W + rx];
                        f = (std::fabs(dsr - ds) > dispThreshold) ? 255 : 0;         // :131-133
                    }
                }
                fail[i][(size_t)y * W + x] = f;
            }
    }
}

extern "C" void les_post_process(les_plane* labelsL, les_plane* labelsR, const uint8_t* imL, const uint8_t* imR, int H, int W, int windR,
                                 float threshold, float omega)
{   // postProcess, LES/PMStereoBase.h:146-256
    les_plane* LR[2] = {labelsL, labelsR};
    const uint8_t* im[2] = {imL, imR};
    const size_t P = (size_t)H * W;
    std::vector<float> disp[2];
    for (int i = 0; i < 2; i++) {
        disp[i].resize(P);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const les_plane& l = LR[i][(size_t)y * W + x];
                disp[i][(size_t)y * W + x] = l.a * (float)x + l.b * (float)y + l.c;   // computeDisparities, LES/StereoEnergy.h:269-272
            }
    }
    std::vector<uint8_t> fail[2], fail2[2];
    for (int i = 0; i < 2; i++) { fail[i].assign(P, 0); fail2[i].assign(P, 0); }
    les_consistency_check(disp[0].data(), disp[1].data(), H, W, threshold, fail[0].data(), fail[1].data());   // :160
    for (int i = 0; i < 2; i++) {
        for (auto& f : fail[i]) f = f > 0 ? 255 : 0;                                  // :161-162
        for (int y = 0; y < H; y++)                                                   // :164-165 cv::dilate, 3x3
            for (int x = 0; x < W; x++) {
                uint8_t m = 0;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        const int xx = x + dx, yy = y + dy;
                        if (xx >= 0 && xx < W && yy >= 0 && yy < H) m = std::max(m, fail[i][(size_t)yy * W + xx]);
                    }
                fail2[i][(size_t)y * W + x] = m;
            }
    }
    auto getz = [](const les_plane& l, int x, int y) { return l.a * (float)x + l.b * (float)y + l.c; };
    // horizontal NN-interpolation, :167-201
    for (int i = 0; i < 2; i++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t row = (size_t)y * W;
                if (fail[i][row + x] == 0) continue;
                const les_plane *pl = nullptr, *pr = nullptr;
                int xx;
                for (xx = x; xx >= 0 && fail2[i][row + xx] == 255; xx--);
                if (xx >= 0) pl = &LR[i][row + xx];
                for (xx = x; xx < W && fail2[i][row + xx] == 255; xx++);
                if (xx < W) pr = &LR[i][row + xx];
                if (pl == nullptr && pr == nullptr) continue;
                else if (pl == nullptr) LR[i][row + x] = *pr;
                else if (pr == nullptr) LR[i][row + x] = *pl;
                else if (getz(*pl, x, y) < getz(*pr, x, y)) LR[i][row + x] = *pl;
                else LR[i][row + x] = *pr;
            }
    // weighted median filter, :207-250.  std::sort is not stable; equal disparities are ordered here by window scan
    // position (the stable order), which is what the device kernel implements as well.
    struct Cand { les_plane l; float w; float z; };
    for (int i = 0; i < 2; i++) {
        std::vector<les_plane> copy(LR[i], LR[i] + P);                                // :209
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < H; y++) {
            std::vector<Cand> median;
            for (int x = 0; x < W; x++) {
                if (fail[i][(size_t)y * W + x] == 0) continue;
                median.clear();
                double sumw = 0;
                const int x0 = std::max(x - windR, 0), y0 = std::max(y - windR, 0), x1 = std::min(x + windR + 1, W), y1 = std::min(y + windR + 1, H);
                const uint8_t* ip = im[i] + ((size_t)y * W + x) * 3;
                for (int yy = y0; yy < y1; yy++)
                    for (int xx = x0; xx < x1; xx++) {
                        const uint8_t* iq = im[i] + ((size_t)yy * W + xx) * 3;
                        // computePatchWeight, LES/StereoEnergy.h:251-257
                        const float absdiff = std::fabs((float)ip[0] - (float)iq[0]) + std::fabs((float)ip[1] - (float)iq[1]) + std::fabs((float)ip[2] - (float)iq[2]);
                        const float w = std::exp(-absdiff / omega);
                        sumw += w;
                        const les_plane& l = copy[(size_t)yy * W + xx];
                        median.push_back(Cand{l, w, getz(l, x, y)});
                    }
                std::stable_sort(median.begin(), median.end(), [](const Cand& a, const Cand& b) { return a.z < b.z; });
                const double center = sumw / 2.0;
                sumw = 0;
                for (size_t j = 0; j < median.size(); j++) {
                    sumw += median[j].w;
                    if (sumw > center) { LR[i][(size_t)y * W + x] = median[j].l; break; }
                }
            }
        }
    }
}


// =====================================================================================================================
// Pairwise terms + graph construction of one expansion move (see les_oracle.h).  Written the way the reference computes
// it: padded whole-image coefficient maps, padded coordinate / label maps, one matrix per forward neighbour, then the
// graph in the reference's loop order.  Deliberately NOT the per-node fused form of the product (csrc/les_pairwise.h,
// host/ExpansionMove.h): the two are compared node by node in tests/.
// =====================================================================================================================
namespace {

struct OGraph {                       // the part of Graph<float,float,double> the construction touches [recollection: maxflow-v3.01 graph.h]
    std::vector<float> tr;            // tr_cap of every node: source capacity - sink capacity
    double flow = 0.0;
    explicit OGraph(int n) : tr((size_t)n, 0.0f) {}
    void add_tweights(int i, float cap_source, float cap_sink)
    {
        float delta = tr[(size_t)i];
        if (delta > 0) cap_source += delta;
        else cap_sink -= delta;
        flow += (cap_source < cap_sink) ? cap_source : cap_sink;
        tr[(size_t)i] = cap_source - cap_sink;
    }
};

struct Vec4 { float v[4]; };

// cvutils::channelDot / channelSum: element-wise product, then cv::reduce(REDUCE_SUM) over the 4 channels (LES/Utilities.hpp:215-229)
inline float channel_dot(const Vec4& a, const Vec4& b)
{
    float m[4] = {a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3]};
    float s = m[0];
    s += m[1]; s += m[2]; s += m[3];
    return s;
}

}  // namespace

extern "C" void les_oracle_expansion_graph(const uint8_t* img, int H, int W, const les_plane* labels, const float* cur, const float* prop,
                                            les_rect region, les_plane label1, float lambda, float th_smooth, float omega, float epsilon,
                                            float* payload, double* flow_out)
{
    const int M = 1;                                                        // LES/StereoEnergy.h:87
    const int nbx[8] = {-1, +1, 0, 0, -1, +1, -1, +1}, nby[8] = {0, 0, -1, +1, -1, -1, +1, +1};   // :99-110
    enum { NB_LE = 0, NB_GE = 1, NB_EL = 2, NB_EG = 3, NB_LL = 4, NB_GL = 5, NB_LG = 6, NB_GG = 7 };
    const int Wm = W + 2 * M, Hm = H + 2 * M;
    // ---- initSmoothnessCoeff (:131-163): I is the image as float, padded with a zero border
    std::vector<float> Im((size_t)Hm * Wm * 3, 0.0f);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) Im[((size_t)(y + M) * Wm + x + M) * 3 + c] = (float)img[((size_t)y * W + x) * 3 + c];
    std::vector<std::vector<float>> coeff(8, std::vector<float>((size_t)Hm * Wm, 0.0f));     // padded again with zeros (:158-160)
    for (int i = 0; i < 8; i++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float* ee = &Im[((size_t)(y + M) * Wm + x + M) * 3];
                const float* nb = &Im[((size_t)(y + M + nby[i]) * Wm + x + M + nbx[i]) * 3];
                float s = std::fabs(nb[0] - ee[0]);                          // absdiff, then channelSum
                s += std::fabs(nb[1] - ee[1]); s += std::fabs(nb[2] - ee[2]);
                float w = std::exp(-s / omega);
                w = std::max(epsilon, w);                                    // cv::max(params.epsilon, .)
                // "set invalid pairwise terms to zero" (:147-155)
                if (nbx[i] < 0 && x < -nbx[i]) w = 0.0f;
                if (nbx[i] > 0 && x >= W - nbx[i]) w = 0.0f;
                if (nby[i] < 0 && y < -nby[i]) w = 0.0f;
                if (nby[i] > 0 && y >= H - nby[i]) w = 0.0f;
                coeff[i][(size_t)(y + M) * Wm + x + M] = w;
            }
    // ---- coordinates_m (zeros in the margin, (x, y, 1, 0) inside, :88-116) and labeling_m (zero margin, LES/PMStereoBase.h:44-47)
    std::vector<Vec4> coord((size_t)Hm * Wm, Vec4{{0, 0, 0, 0}}), lab((size_t)Hm * Wm, Vec4{{0, 0, 0, 0}});
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            coord[(size_t)(y + M) * Wm + x + M] = Vec4{{(float)x, (float)y, 1.0f, 0.0f}};
            const les_plane& l = labels[(size_t)y * W + x];
            lab[(size_t)(y + M) * Wm + x + M] = Vec4{{l.a, l.b, l.c, l.v}};
        }
    const Vec4 sc{{label1.a, label1.b, label1.c, label1.v}};                // label1.toScalar()
    // ---- computeSmoothnessTermsExpansion(labeling_m, label1, region, ..., onlyForward = true) (:398-453)
    const int rw = region.w, rh = region.h, N = rw * rh;
    auto at = [&](int x, int y) { return (size_t)(y + M + region.y) * Wm + (x + M + region.x); };   // rect_ee element
    std::vector<float> d0_ee_at_ee((size_t)N), d1_at_ee((size_t)N);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            d0_ee_at_ee[(size_t)y * rw + x] = channel_dot(lab[at(x, y)], coord[at(x, y)]);
            d1_at_ee[(size_t)y * rw + x] = channel_dot(coord[at(x, y)], sc);                   // channelSum(coord_ee.mul(sc))
        }
    std::vector<std::vector<float>> cost00(8), cost01(8), cost10(8);
    for (int i = 0; i < 8; i++) {
        if (nby[i] * W + nbx[i] <= 0) continue;                             // onlyForward
        cost00[i].resize((size_t)N); cost01[i].resize((size_t)N); cost10[i].resize((size_t)N);
        for (int y = 0; y < rh; y++)
            for (int x = 0; x < rw; x++) {
                const size_t ee = at(x, y), le = at(x + nbx[i], y + nby[i]);
                const float d0_le_at_ee = channel_dot(lab[le], coord[ee]);
                const float d0_ee_at_le = channel_dot(lab[ee], coord[le]);
                const float d0_le_at_le = channel_dot(lab[le], coord[le]);
                const float d1_at_le = channel_dot(coord[le], sc);
                const float w = coeff[i][ee];
                const size_t k = (size_t)y * rw + x;
                float c;
                c = std::fabs(d0_ee_at_ee[k] - d0_le_at_ee) + std::fabs(d0_ee_at_le - d0_le_at_le);
                c = (c > th_smooth) ? th_smooth : c;                         // cv::threshold(THRESH_TRUNC)
                cost00[i][k] = c * w * lambda;                               // .mul(coeff, lambda): saturate_cast<float>(a * b * scale)
                c = std::fabs(d0_ee_at_ee[k] - d1_at_ee[k]) + std::fabs(d0_ee_at_le - d1_at_le);
                c = (c > th_smooth) ? th_smooth : c;
                cost01[i][k] = c * w * lambda;
                c = std::fabs(d1_at_ee[k] - d0_le_at_ee) + std::fabs(d1_at_le - d0_le_at_le);
                c = (c > th_smooth) ? th_smooth : c;
                cost10[i][k] = c * w * lambda;
            }
    }
    // computeSmoothnessTerm(ls, lt, ps, neighborId) (:225-230) with Plane::GetZ (LES/Plane.h:51-58)
    auto getz = [](const les_plane& l, int x, int y) { return l.a * (float)x + l.b * (float)y + l.c; };
    auto term = [&](const les_plane& ls, const les_plane& lt, int px, int py, int k) {
        const int tx = px + nbx[k], ty = py + nby[k];
        const float d = std::fabs(getz(ls, px, py) - getz(lt, px, py)) + std::fabs(getz(ls, tx, ty) - getz(lt, tx, ty));
        return coeff[k][(size_t)(py + M) * Wm + px + M] * std::min(d, th_smooth) * lambda;
    };
    // ---- graph construction (LES/FastGCStereo.h:425-551)
    OGraph graph(N);
    std::vector<float> cap((size_t)N * 4, 0.0f);                            // arcs i -> j per forward direction GE, EG, LG, GG
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            const int s = y * rw + x;
            const int X = region.x + x, Y = region.y + y;
            graph.add_tweights(s, cur[(size_t)Y * W + X], prop[(size_t)Y * W + X]);          // :433
            if (x == 0 || x == rw - 1 || y == 0 || y == rh - 1) {
                for (int k = 0; k < 8; k++) {                                                 // :455-474
                    const int tx = X + nbx[k], ty = Y + nby[k];
                    if (tx >= region.x && tx < region.x + rw && ty >= region.y && ty < region.y + rh) continue;
                    if (tx < 0 || tx >= W || ty < 0 || ty >= H) continue;
                    const les_plane& lps = labels[(size_t)Y * W + X];
                    const les_plane& lpt = labels[(size_t)ty * W + tx];
                    graph.add_tweights(s, term(lps, lpt, X, Y, k), term(label1, lpt, X, Y, k));
                }
            }
        }
    auto link = [&](int nb, int slot, int x0, int x1, int y1, int jdx, int jdy) {
        for (int y = 0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const int i = y * rw + x, j = (y + jdy) * rw + x + jdx;
                const float B = cost10[nb][(size_t)i], C = cost01[nb][(size_t)i], D = cost00[nb][(size_t)i];
                cap[(size_t)i * 4 + slot] += std::max(0.f, B + C - D);       // add_edge(i, j, max(0, B + C - D), 0)
                graph.add_tweights(i, C, 0);
                graph.add_tweights(j, D - C, 0);
            }
    };
    link(NB_GE, 0, 0, rw - 1, rh, +1, 0);                                   // :481-492
    link(NB_EG, 1, 0, rw, rh - 1, 0, +1);                                   // :498-509
    link(NB_LG, 2, 1, rw, rh - 1, -1, +1);                                  // :517-528
    link(NB_GG, 3, 0, rw - 1, rh - 1, +1, +1);                              // :534-545
    for (int s = 0; s < N; s++) {
        payload[(size_t)s * 5] = graph.tr[(size_t)s];
        for (int d = 0; d < 4; d++) payload[(size_t)s * 5 + 1 + d] = cap[(size_t)s * 4 + d];
    }
    if (flow_out) *flow_out = graph.flow;
}
