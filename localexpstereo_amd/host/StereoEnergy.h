// StereoEnergy.h -- host-side operator interface of the matching-cost path, mirroring the reference's
// class StereoEnergy (LES/StereoEnergy.h:42-627) for the members this path uses:
//   virtual ComputeUnaryPotential / ComputeUnaryPotentialWithoutCheck  (:625-626)
//   IsValiLabel (:560-610), createRandomLabel (:120-129), Reusable (:616-623), COST_FOR_INVALID (:45), params.
// The pairwise (smoothness) members used by the graph-cut fusion ("next" rows N1/N2 of SURVEY.md section 8(f)) are
// provided on the host: neighbors (:58), initSmoothnessCoeff (:131-163), computeSmoothnessTerm (:225-230),
// computeSmoothnessTermsExpansion (:398-453).
#pragma once

#include <array>

#include "les_types.h"

namespace les_host {

class StereoEnergy {
public:
    static constexpr int COST_FOR_INVALID = 1000000;                                  // LES/StereoEnergy.h:45

    // caller-owned per-cell scratch (LES/StereoEnergy.h:616-623).  The GPU operator needs none; the type is
    // kept so that call sites written for the reference compile unchanged.
    struct Reusable {
        std::vector<float> pIL;
        Rect filterRect;
    };

    enum { NB_LE = 0, NB_GE = 1, NB_EL = 2, NB_EG = 3, NB_LL = 4, NB_GL = 5, NB_LG = 6, NB_GG = 7 };   // LES/StereoEnergy.h:47-56
    std::array<Point, 8> neighbors{{{-1, 0}, {+1, 0}, {0, -1}, {0, +1}, {-1, -1}, {+1, -1}, {-1, +1}, {+1, +1}}};   // :99-110

    Parameters params;

    StereoEnergy(int width, int height, Parameters p, float MAX_DISPARITY, float MIN_DISPARITY = 0)
        : params(std::move(p)), width(width), height(height), MAX_DISPARITY(MAX_DISPARITY), MIN_DISPARITY(MIN_DISPARITY) {}
    virtual ~StereoEnergy() {}

    // ---- pairwise terms (host side of the graph cut) ------------------------------------------------------
    // images: H x W x 3 uint8 BGR of the two views (the reference keeps them as CV_32FC3, LES/StereoEnergy.h:96-97)
    void setImages(const uint8_t* imL, const uint8_t* imR)
    {
        const uint8_t* im[2] = {imL, imR};
        const size_t P = (size_t)width * height;
        for (int m = 0; m < 2; m++) {
            if (!im[m]) continue;
            I[m].assign(im[m], im[m] + P * 3);
            // initSmoothnessCoeff, LES/StereoEnergy.h:131-163: w = max(epsilon, exp(-|dI|_1 / omega)), 0 for pairs that
            // leave the image
            for (int k = 0; k < 8; k++) smoothnessCoeff[m][k].assign(P, 0.f);
            // (11 M exponentials per view at the Adirondack shape: 60 ms on one core inside the drivers' timed region)
#pragma omp parallel for collapse(2) schedule(static)
            for (int k = 0; k < 8; k++) {
                for (int y = 0; y < height; y++)
                    for (int x = 0; x < width; x++) {
                        const int xn = x + neighbors[k].x, yn = y + neighbors[k].y;
                        if (xn < 0 || xn >= width || yn < 0 || yn >= height) continue;
                        const float* a = &I[m][((size_t)y * width + x) * 3];
                        const float* b = &I[m][((size_t)yn * width + xn) * 3];
                        const float ad = std::fabs(b[0] - a[0]) + std::fabs(b[1] - a[1]) + std::fabs(b[2] - a[2]);
                        smoothnessCoeff[m][k][(size_t)y * width + x] = std::max(params.epsilon, std::exp(-ad / params.omega));
                    }
            }
        }
    }
    bool hasImages(int mode) const { return !I[mode].empty(); }

    // LES/StereoEnergy.h:225-230
    float computeSmoothnessTerm(const Plane& ls, const Plane& lt, Point ps, int neighborId, int mode = 0) const
    {
        const Point pt{ps.x + neighbors[neighborId].x, ps.y + neighbors[neighborId].y};
        return smoothnessCoeff[mode][neighborId][(size_t)ps.y * width + ps.x]
               * std::min(std::fabs(ls.GetZ((float)ps.x, (float)ps.y) - lt.GetZ((float)ps.x, (float)ps.y))
                          + std::fabs(ls.GetZ((float)pt.x, (float)pt.y) - lt.GetZ((float)pt.x, (float)pt.y)), params.th_smooth)
               * params.lambda;
    }

    // LES/StereoEnergy.h:398-453 for the forward neighbours GE, EG, LG, GG (onlyForward = true): per pixel `ee` of
    // region and neighbour `le` = ee + n:  cost00 (both keep their labels), cost01 (le takes label1), cost10 (ee takes
    // label1).  Outputs are region.height x region.width row-major arrays indexed by the 8-neighbour id; label0 outside
    // the image is the zero plane (the reference's label map has a 1-pixel zero margin) and its weight is 0 anyway.
    void computeSmoothnessTermsExpansion(const LabelMap& labeling0, const Plane& label1, const Rect& region,
                                         std::array<std::vector<float>, 8>& cost00, std::array<std::vector<float>, 8>& cost01,
                                         std::array<std::vector<float>, 8>& cost10, int mode = 0) const
    {
        const size_t n = (size_t)region.width * region.height;
        for (int k : {(int)NB_GE, (int)NB_EG, (int)NB_LG, (int)NB_GG}) {
            cost00[k].assign(n, 0.f); cost01[k].assign(n, 0.f); cost10[k].assign(n, 0.f);
            for (int y = 0; y < region.height; y++)
                for (int x = 0; x < region.width; x++) {
                    const size_t i = (size_t)y * region.width + x;
                    smoothnessTermsExpansionAt(labeling0, label1, region.x + x, region.y + y, k, cost00[k][i], cost01[k][i], cost10[k][i], mode);
                }
        }
    }

    // The three terms of computeSmoothnessTermsExpansion for one pixel `ee` = (ex, ey) and one forward neighbour k
    // (the one place this arithmetic lives: the array form above calls it; the expansion move builds its graph from it directly).
    // dot = channelDot / channelSum order of the reference.
    void smoothnessTermsExpansionAt(const LabelMap& labeling0, const Plane& label1, int ex, int ey, int k, float& c00, float& c01, float& c10,
                                    int mode = 0) const
    {
        auto dot = [](const Plane& l, float x, float y) { return ((l.a * x + l.b * y) + l.c * 1.0f) + l.v * 0.0f; };
        const int lx = ex + neighbors[k].x, ly = ey + neighbors[k].y;
        const bool inside = lx >= 0 && lx < width && ly >= 0 && ly < height;
        const Plane& l0_ee = labeling0.at(ey, ex);
        const Plane l0_le = inside ? labeling0.at(ly, lx) : Plane();
        const float fx = (float)ex, fy = (float)ey, gx = (float)lx, gy = (float)ly;
        const float d0_ee_at_ee = dot(l0_ee, fx, fy), d0_le_at_ee = dot(l0_le, fx, fy);
        const float d0_ee_at_le = dot(l0_ee, gx, gy), d0_le_at_le = dot(l0_le, gx, gy);
        const float d1_at_ee = dot(label1, fx, fy), d1_at_le = dot(label1, gx, gy);
        const float w = smoothnessCoeff[mode][k][(size_t)ey * width + ex];
        const float th = params.th_smooth;
        c00 = std::min(std::fabs(d0_ee_at_ee - d0_le_at_ee) + std::fabs(d0_ee_at_le - d0_le_at_le), th) * w * params.lambda;
        c01 = std::min(std::fabs(d0_ee_at_ee - d1_at_ee) + std::fabs(d0_ee_at_le - d1_at_le), th) * w * params.lambda;
        c10 = std::min(std::fabs(d1_at_ee - d0_le_at_ee) + std::fabs(d1_at_le - d0_le_at_le), th) * w * params.lambda;
    }

    // total smoothness energy of a labelling over forward pairs (StereoEnergy::computeSmoothnessCost, :165-203)
    double computeSmoothnessCost(const LabelMap& labeling, int mode = 0) const
    {
        double sum = 0;
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++)
                for (int k : {(int)NB_GE, (int)NB_EG, (int)NB_LG, (int)NB_GG}) {
                    const int xn = x + neighbors[k].x, yn = y + neighbors[k].y;
                    if (xn < 0 || xn >= width || yn < 0 || yn >= height) continue;
                    sum += computeSmoothnessTerm(labeling.at(y, x), labeling.at(yn, xn), Point{x, y}, k, mode);
                }
        return sum;
    }

    // costs: pointer to element (filterRect.y, filterRect.x) of a row-major float map with row_stride floats
    // per row, i.e. the view proposalCost(filterRect) of LES/FastGCStereo.h:49.  Only the sub-rect
    // targetRect - filterRect.tl() is written.
    virtual void ComputeUnaryPotentialWithoutCheck(const Rect& filterRect, const Rect& targetRect, float* costs, int row_stride,
                                                   const Plane& plane, Reusable& reusable, int mode = 0) const = 0;
    virtual void ComputeUnaryPotential(const Rect& filterRect, const Rect& targetRect, float* costs, int row_stride,
                                       const Plane& plane, Reusable& reusable, int mode = 0) const = 0;

    // "Avoid extreme labels", LES/StereoEnergy.h:560-574
    bool IsValiLabel(const Plane& label, Point pos) const
    {
        const float ds = label.GetZ((float)pos.x, (float)pos.y);
        const float a5 = label.a * 5, b5 = label.b * 5;
        float d;
        return (ds >= MIN_DISPARITY && ds <= MAX_DISPARITY
                && ((d = ds + a5 + b5) >= MIN_DISPARITY) && d <= MAX_DISPARITY
                && ((d = ds + a5 - b5) >= MIN_DISPARITY) && d <= MAX_DISPARITY
                && ((d = ds - a5 + b5) >= MIN_DISPARITY) && d <= MAX_DISPARITY
                && ((d = ds - a5 - b5) >= MIN_DISPARITY) && d <= MAX_DISPARITY);
    }

    // LES/StereoEnergy.h:120-129 with LES/Utilities.hpp:254-261 (MAX_VDISPARITY == 0)
    Plane createRandomLabel(Point s, RNG& rng) const
    {
        const double PI = 3.1415926535897932384626433832795;
        const float zs = rng.uniform(MIN_DISPARITY, MAX_DISPARITY);
        const double theta = rng.uniform(0.0, PI / 3), phi = rng.uniform(0.0, PI * 2.0);
        const double cosT = std::cos(theta), sinT = std::sin(theta), cosP = std::cos(phi), sinP = std::sin(phi);
        return Plane::CreatePlane((float)(sinT * cosP), (float)(sinT * sinP), (float)cosT, zs, (float)s.x, (float)s.y, 0.0f);
    }

    // StereoEnergy::computePatchWeight (LES/StereoEnergy.h:251-257): exp(-|I(s) - I(t)|_1 / omega) on the float BGR image
    float computePatchWeight(Point s, Point t, int mode = 0) const
    {
        const float* a = &I[mode][((size_t)s.y * width + s.x) * 3];
        const float* b = &I[mode][((size_t)t.y * width + t.x) * 3];
        const float absdiff = std::fabs(a[0] - b[0]) + std::fabs(a[1] - b[1]) + std::fabs(a[2] - b[2]);
        return std::exp(-absdiff / params.omega);
    }

    // StereoEnergy::computeDisparities (LES/StereoEnergy.h:269-272) and computeNormalMap (:274-289, the visualisation of the
    // plane normals: channels {nz, (1 - nx')/2 ... } exactly as the reference composes them: out[0] = nz = 1/sqrt(a^2+b^2+1),
    // out[1] = (-b nz + 1)/2, out[2] = (-a nz + 1)/2)
    std::vector<float> computeDisparities(const LabelMap& labeling) const
    {
        std::vector<float> d((size_t)width * height);
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) d[(size_t)y * width + x] = labeling.at(y, x).GetZ((float)x, (float)y);
        return d;
    }
    std::vector<float> computeNormalMap(const LabelMap& labeling) const
    {
        std::vector<float> n((size_t)width * height * 3);
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                const Plane& l = labeling.at(y, x);
                const float nz = 1.0f / std::sqrt(l.a * l.a + l.b * l.b + 1.0f);
                float* o = &n[((size_t)y * width + x) * 3];
                o[0] = nz; o[1] = (l.b * nz * -1.0f + 1.0f) / 2.0f; o[2] = (l.a * nz * -1.0f + 1.0f) / 2.0f;
            }
        return n;
    }
    // the reference's label maps carry a one-pixel margin (getRectWithoutMargin, LES/StereoEnergy.h:264-267); the maps here
    // have none, so the interior is the whole image
    Rect getRectWithoutMargin() const { return Rect(0, 0, width, height); }

    int getWidth() const { return width; }
    int getHeight() const { return height; }
    float maxDisparity() const { return MAX_DISPARITY; }
    float minDisparity() const { return MIN_DISPARITY; }

protected:
    const int width, height;
    const float MAX_DISPARITY, MIN_DISPARITY;
    std::vector<float> I[2];                                   // BGR as float, H x W x 3
    std::vector<float> smoothnessCoeff[2][8];                  // H x W each
};

}  // namespace les_host
