// LayerManager.h -- grid cells of the local expansion moves (reference: LES/LayerManager.h:44-185).
// For a unit size u: unit cell u x u, shared (expansion) region 3u x 3u, filter region = shared +- windR,
// all clipped to the image; right/bottom remainders smaller than max(2, u/2) are merged into the last
// column/row; cells are grouped into 16 classes (i%4)*4 + (j%4) whose shared regions never overlap --
// the unit of parallelism on the GPU (one lock-step per class) and of multi-GPU sharding.
#pragma once

#include "les_types.h"

namespace les_host {

class LayerManager {
public:
    struct Layer {
        int heightBlocks = 0, widthBlocks = 0, regionUnitSize = 0;
        std::vector<Rect> unitRegions, sharedRegions, filterRegions;
        std::vector<std::vector<int>> disjointRegionSets;
    };
    std::vector<Layer> layers;

    LayerManager(int width, int height, int windowR) : width_(width), height_(height), windowR_(windowR) {}

    void addLayer(int unitRegionSize)
    {
        Layer L;
        const int u = unitRegionSize, W = width_, H = height_, R = windowR_;
        L.regionUnitSize = u;
        const int minsize = std::max(2, u / 2);
        const int frac_w = W % u, frac_h = H % u;
        const bool split_w = frac_w >= minsize, split_h = frac_h >= minsize;
        L.widthBlocks = W / u + (split_w ? 1 : 0);
        L.heightBlocks = H / u + (split_h ? 1 : 0);
        const Rect image(0, 0, W, H);
        const int n = L.widthBlocks * L.heightBlocks;
        L.unitRegions.resize(n); L.sharedRegions.resize(n); L.filterRegions.resize(n);
        // extents along one axis: cell index -> [begin, end) before clipping, with the merged remainder
        auto unit_span = [&](int idx, int blocks, bool split, int frac, int& b, int& e) {
            b = idx * u; e = b + u;
            if (!split && idx == blocks - 1) e += frac;             // last cell swallows the small remainder
        };
        auto shared_span = [&](int idx, int blocks, bool split, int frac, int& b, int& e) {
            b = (idx - 1) * u; e = b + 3 * u;
            if (!split && idx == blocks - 2) e += frac;             // neighbour of the enlarged last cell
        };
        for (int i = 0; i < L.heightBlocks; i++)
            for (int j = 0; j < L.widthBlocks; j++) {
                int ux0, ux1, uy0, uy1, sx0, sx1, sy0, sy1;
                unit_span(j, L.widthBlocks, split_w, frac_w, ux0, ux1);
                unit_span(i, L.heightBlocks, split_h, frac_h, uy0, uy1);
                shared_span(j, L.widthBlocks, split_w, frac_w, sx0, sx1);
                shared_span(i, L.heightBlocks, split_h, frac_h, sy0, sy1);
                const int r = i * L.widthBlocks + j;
                L.unitRegions[r] = Rect(ux0, uy0, ux1 - ux0, uy1 - uy0) & image;
                // the reference clips first and enlarges afterwards: an enlarged shared region keeps its clipped
                // origin and is NOT clipped again, the filter region is (LES/LayerManager.h:146-149,160-163)
                Rect s = Rect(sx0, sy0, 3 * u, 3 * u) & image;
                Rect f = Rect(sx0 - R, sy0 - R, 3 * u + 2 * R, 3 * u + 2 * R) & image;
                s.width += sx1 - sx0 - 3 * u; s.height += sy1 - sy0 - 3 * u;
                f.width += sx1 - sx0 - 3 * u; f.height += sy1 - sy0 - 3 * u;
                L.sharedRegions[r] = s;
                L.filterRegions[r] = f & image;
            }
        std::vector<std::vector<int>> sets(16);
        for (int i = 0; i < L.heightBlocks; i++)
            for (int j = 0; j < L.widthBlocks; j++) sets[(i % 4) * 4 + (j % 4)].push_back(i * L.widthBlocks + j);
        for (auto& s : sets)
            if (!s.empty()) L.disjointRegionSets.push_back(std::move(s));
        layers.push_back(std::move(L));
    }

private:
    int width_, height_, windowR_;
};

}  // namespace les_host
