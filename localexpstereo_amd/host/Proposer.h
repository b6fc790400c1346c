// Proposer.h -- host-side label proposers with the reference's interface (LES/Proposer.h:6-31):
//   createInstance / startIterations(labeling, unitRegion, outerIter) / getNextProposal / isContinued.
// They read the live label map like the reference (LES/Proposer.h:64) and draw from an explicit
// cv::RNG-compatible generator (the reference uses the thread-local cv::theRNG()).
// The RANSAC proposer runs on the device in this framework (localexpstereo_amd/csrc/les_propose.h); the
// host loop uses the Expansion and Random proposers, which are trivially cheap.
#pragma once

#include "les_types.h"

namespace les_host {

class IProposer {
public:
    explicit IProposer(int K) : K(K) {}
    virtual ~IProposer() {}
    virtual IProposer* createInstance() = 0;
    virtual void startIterations(const LabelMap& labeling, Rect unitRegion, int outerIter, RNG* rng) = 0;
    virtual Plane getNextProposal() = 0;
    virtual bool isContinued() = 0;

protected:
    const int K;
    const LabelMap* labeling = nullptr;
    RNG* rng = nullptr;
    int iter = 0, outerIter = 0;
    Rect rect;
    Point selectRandomPixelInRect(Rect r)                                             // LES/Proposer.h:37-44
    {
        const int n = rng->uniform(0, r.height * r.width);
        return Point{r.x + n % r.width, r.y + n / r.width};
    }
};

class ExpansionProposer : public IProposer {                                          // LES/Proposer.h:34-80
public:
    explicit ExpansionProposer(int K) : IProposer(K) {}
    IProposer* createInstance() override { return new ExpansionProposer(K); }
    void startIterations(const LabelMap& l, Rect unitRegion, int outer, RNG* r) override
    {
        labeling = &l; rect = unitRegion; outerIter = outer; rng = r; iter = 0;
    }
    Plane getNextProposal() override
    {
        const Point p = selectRandomPixelInRect(rect);
        iter++;
        return labeling->at(p.y, p.x);
    }
    bool isContinued() override { return iter < K; }
};

class RandomProposer : public ExpansionProposer {                                     // LES/Proposer.h:84-153
public:
    RandomProposer(int K, float maxDisp, float minDisp = 0) : ExpansionProposer(K), MIN_DISPARITY(minDisp), MAX_DISPARITY(maxDisp) {}
    IProposer* createInstance() override { return new RandomProposer(K, MAX_DISPARITY, MIN_DISPARITY); }
    Plane getNextProposal() override
    {
        const double PI = 3.1415926535897932384626433832795;
        const Point s = selectRandomPixelInRect(rect);
        const Plane in = labeling->at(s.y, s.x);
        const int m = outerIter + iter;
        iter++;
        float zs = in.GetZ(float(s.x), float(s.y));
        const float dz = width(m);
        const float minz = std::max(MIN_DISPARITY, zs - dz), maxz = std::min(MAX_DISPARITY, zs + dz);
        zs = rng->uniform(minz, maxz);
        const float nr = (float)std::ldexp(1.0, -m);                                  // randomNmax (=1) * 0.5^m
        float n0[3];
        in.GetNormal(n0);
        const double theta = rng->uniform(0.0, PI), phi = rng->uniform(0.0, PI * 2.0);
        const double u[3] = {std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)};
        float nv[3];
        for (int c = 0; c < 3; c++) nv[c] = n0[c] + (float)u[c] * nr;
        const double inv = 1. / std::sqrt((double)nv[0] * nv[0] + (double)nv[1] * nv[1] + (double)nv[2] * nv[2]);
        for (int c = 0; c < 3; c++) nv[c] = (float)(nv[c] * inv);
        return Plane::CreatePlane(nv[0], nv[1], nv[2], zs, float(s.x), float(s.y), in.v);
    }
    bool isContinued() override { return iter < K && !(width(outerIter + iter) < 0.1); }   // early stop, :149-152

private:
    const float MIN_DISPARITY, MAX_DISPARITY;
    float width(int m) const { return (float)((double)(MAX_DISPARITY - MIN_DISPARITY) * std::ldexp(1.0, -(m + 1))); }
};

}  // namespace les_host
