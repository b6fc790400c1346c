// StereoEnergy.h -- host-side operator interface of the matching-cost path, mirroring the reference's
// class StereoEnergy (LES/StereoEnergy.h:42-627) for the members this path uses:
//   virtual ComputeUnaryPotential / ComputeUnaryPotentialWithoutCheck  (:625-626)
//   IsValiLabel (:560-610), createRandomLabel (:120-129), Reusable (:616-623), COST_FOR_INVALID (:45), params.
// The pairwise (smoothness) members of the reference base class belong to the graph-cut side ("next" row
// N1 of SURVEY.md section 8(f)) and are not part of this path.
#pragma once

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

    Parameters params;

    StereoEnergy(int width, int height, Parameters p, float MAX_DISPARITY, float MIN_DISPARITY = 0)
        : params(std::move(p)), width(width), height(height), MAX_DISPARITY(MAX_DISPARITY), MIN_DISPARITY(MIN_DISPARITY) {}
    virtual ~StereoEnergy() {}

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

    int getWidth() const { return width; }
    int getHeight() const { return height; }
    float maxDisparity() const { return MAX_DISPARITY; }
    float minDisparity() const { return MIN_DISPARITY; }

protected:
    const int width, height;
    const float MAX_DISPARITY, MIN_DISPARITY;
};

}  // namespace les_host
