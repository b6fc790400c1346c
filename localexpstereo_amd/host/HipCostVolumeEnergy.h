// HipCostVolumeEnergy.h -- drop-in replacement of class CostVolumeEnergy (LES/CostVolumeEnergy.h:6-184)
// whose ComputeUnaryPotential runs on an MI355X through the C ABI of include/localexp_hip.h.
//
// Same constructor shape (imL, imR, volL, volR, Parameters, MAX_DISPARITY, MIN_DISPARITY), same operator
// semantics, same threading contract: the operator is const and may be called concurrently from OpenMP
// threads, one cell per thread, each writing a disjoint rect of one shared map (LES/FastGCStereo.h:30-49).
// Error behaviour: like the reference the operator returns void and never throws; a device failure is
// reported on stderr and leaves COST_FOR_INVALID in the target rect (the reference's only failure
// signal is that sentinel, LES/CostVolumeEnergy.h:87-90).  The constructor throws std::runtime_error when
// no HIP device / library is available -- there is no CPU fallback.
#pragma once

#include <cstdio>
#include <mutex>
#include <stdexcept>

#include "StereoEnergy.h"
#include "localexp_hip.h"

namespace les_host {

class HipCostVolumeEnergy : public StereoEnergy {
public:
    // imL/imR: H x W x 3 uint8 BGR (cv::imread layout); volL/volR: float [D][H][W] (shared with the caller in
    // the reference, copied to HBM once here).
    HipCostVolumeEnergy(const uint8_t* imL, const uint8_t* imR, int width, int height, const float* volL, const float* volR,
                        int ndisp, Parameters p, float MAX_DISPARITY, float MIN_DISPARITY = 0, int device = 0)
        : StereoEnergy(width, height, std::move(p), MAX_DISPARITY, MIN_DISPARITY), ctx_(nullptr)
    {
        if (params.filterName != "GF") throw std::runtime_error("HipCostVolumeEnergy implements the default \"GF\" joint filter");
        les_hip_params hp;
        hp.H = height; hp.W = width; hp.D = ndisp;
        hp.windR = params.windR; hp.eps = params.filter_param1; hp.th_col = params.th_col;
        hp.max_disparity = MAX_DISPARITY; hp.min_disparity = MIN_DISPARITY;
        hp.device = device; hp.volumes_on_device = 0;
        if (les_hip_create(&ctx_, &hp, imL, imR, volL, volR) != LES_HIP_OK)
            throw std::runtime_error(std::string("les_hip_create: ") + les_hip_last_error());
        setImages(imL, imR);           // pairwise weights for the host graph cut
    }
    ~HipCostVolumeEnergy() override { les_hip_destroy(ctx_); }

protected:
    // image-based matching cost (HipNaiveStereoEnergy below)
    struct NaiveTag {};
    HipCostVolumeEnergy(NaiveTag, const uint8_t* imL, const uint8_t* imR, int width, int height, Parameters p, float MAX_DISPARITY,
                        float MIN_DISPARITY, int device)
        : StereoEnergy(width, height, std::move(p), MAX_DISPARITY, MIN_DISPARITY), ctx_(nullptr)
    {
        if (params.filterName != "GF") throw std::runtime_error("HipNaiveStereoEnergy implements the default \"GF\" joint filter");
        les_hip_params hp;
        hp.H = height; hp.W = width; hp.D = 1;
        hp.windR = params.windR; hp.eps = params.filter_param1; hp.th_col = params.th_col;
        hp.max_disparity = MAX_DISPARITY; hp.min_disparity = MIN_DISPARITY;
        hp.device = device; hp.volumes_on_device = 0;
        if (les_hip_create_naive(&ctx_, &hp, imL, imR, params.alpha, params.th_grad) != LES_HIP_OK)
            throw std::runtime_error(std::string("les_hip_create_naive: ") + les_hip_last_error());
        setImages(imL, imR);
    }

public:
    HipCostVolumeEnergy(const HipCostVolumeEnergy&) = delete;
    HipCostVolumeEnergy& operator=(const HipCostVolumeEnergy&) = delete;

    void ComputeUnaryPotentialWithoutCheck(const Rect& filterRect, const Rect& targetRect, float* costs, int row_stride,
                                           const Plane& plane, Reusable& reusable, int mode = 0) const override
    {
        call(filterRect, targetRect, costs, row_stride, plane, reusable, mode, 0);
    }
    void ComputeUnaryPotential(const Rect& filterRect, const Rect& targetRect, float* costs, int row_stride, const Plane& plane,
                               Reusable& reusable, int mode = 0) const override
    {
        call(filterRect, targetRect, costs, row_stride, plane, reusable, mode, 1);
    }

    // Batched form (one proposal index of one disjoint set): n calls into one H x W host map.
    void ComputeUnaryPotentialBatch(const std::vector<Rect>& filterRects, const std::vector<Rect>& targetRects,
                                    const std::vector<Plane>& planes, float* cost_map, int mode = 0, bool check = true) const
    {
        static_assert(sizeof(Rect) == sizeof(les_hip_rect) && sizeof(Plane) == sizeof(les_hip_plane), "ABI layout");
        std::lock_guard<std::mutex> lk(mu_);               // the batched form uses the context's own stream and scratch map
        if (les_hip_unary_batch(ctx_, mode, (int)filterRects.size(), reinterpret_cast<const les_hip_rect*>(filterRects.data()),
                                reinterpret_cast<const les_hip_rect*>(targetRects.data()),
                                reinterpret_cast<const les_hip_plane*>(planes.data()), cost_map, check ? 1 : 0) != LES_HIP_OK)
            fprintf(stderr, "HipCostVolumeEnergy: %s\n", les_hip_last_error());
    }

    les_hip_ctx* handle() const { return ctx_; }

private:
    void call(const Rect& fr, const Rect& tr, float* costs, int row_stride, const Plane& plane, Reusable&, int mode, int check) const
    {
        const les_hip_rect f{fr.x, fr.y, fr.width, fr.height}, t{tr.x, tr.y, tr.width, tr.height};
        const les_hip_plane p{plane.a, plane.b, plane.c, plane.v};
        // re-entrant: the library keeps one scratch (stream, device tile, pinned staging, job tables of the recent rect pairs) per
        // calling thread, so the OpenMP threads of the reference loop (LES/FastGCStereo.h:30-49) run their cells concurrently
        const int rc = les_hip_unary_one(ctx_, mode, &f, &t, &p, costs, row_stride, check);
        if (rc != LES_HIP_OK) {
            fprintf(stderr, "HipCostVolumeEnergy: %s\n", les_hip_last_error());
            for (int y = 0; y < tr.height; y++)
                for (int x = 0; x < tr.width; x++)
                    costs[(size_t)(tr.y - fr.y + y) * row_stride + (tr.x - fr.x + x)] = (float)COST_FOR_INVALID;
        }
    }

    les_hip_ctx* ctx_;
    mutable std::mutex mu_;
};

// Drop-in for NaiveStereoEnergy (LES/StereoEnergy.h:629-764), the energy of the MiddV2 configuration
// (LES/PMStereoBase.h:37, parameters LES/main.cpp:86-121): same operator, raw cost from the two images.
class HipNaiveStereoEnergy : public HipCostVolumeEnergy {
public:
    HipNaiveStereoEnergy(const uint8_t* imL, const uint8_t* imR, int width, int height, Parameters p, float MAX_DISPARITY,
                         float MIN_DISPARITY = 0, int device = 0)
        : HipCostVolumeEnergy(NaiveTag{}, imL, imR, width, height, std::move(p), MAX_DISPARITY, MIN_DISPARITY, device) {}
};

}  // namespace les_host
