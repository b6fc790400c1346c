// DemoScene.h -- small synthetic stereo scene (three slanted surfaces, colour guide that follows them, noisy
// truncated-absolute-difference cost volume) used by les_host_demo and the host self-tests.
#pragma once

#include <cmath>
#include <vector>

#include "les_types.h"

namespace les_host {

struct Scene {
    int W, H, D;
    std::vector<uint8_t> im;       // BGR
    std::vector<float> vol, gt;
};

inline Scene make_scene(int W, int H, int D)
{
    Scene s{W, H, D, std::vector<uint8_t>((size_t)W * H * 3), std::vector<float>((size_t)W * H * D), std::vector<float>((size_t)W * H)};
    RNG rng(4242);
    // three slanted surfaces separated by vertical / diagonal boundaries; guide colour follows the surface
    const Plane surf[3] = {Plane(0.02f, 0.01f, 0.25f * D), Plane(-0.03f, 0.0f, 0.6f * D), Plane(0.0f, -0.02f, 0.45f * D)};
    const int col[3][3] = {{200, 60, 40}, {40, 180, 70}, {60, 70, 210}};
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int k = x < W / 3 ? 0 : (x + y / 2 < (2 * W) / 3 ? 1 : 2);
            float d = surf[k].GetZ((float)x, (float)y);
            d = std::min(std::max(d, 1.0f), (float)D - 2.0f);
            s.gt[(size_t)y * W + x] = d;
            for (int c = 0; c < 3; c++) {
                int v = col[k][c] + (int)(rng.uniform(-12.0f, 12.0f)) + (int)(10.0 * std::sin(0.15 * x + 0.1 * y));
                s.im[((size_t)y * W + x) * 3 + c] = (uint8_t)std::min(255, std::max(0, v));
            }
            for (int dd = 0; dd < D; dd++) {
                const float e = std::fabs((float)dd - d);
                s.vol[((size_t)dd * H + y) * W + x] = std::min(1.0f, 0.12f * e) * 0.8f + rng.uniform(0.0f, 0.2f);
            }
        }
    return s;
}

inline double bad_pixels(const std::vector<float>& disp, const Scene& s, float thr)
{
    size_t bad = 0;
    for (size_t i = 0; i < disp.size(); i++) bad += std::fabs(disp[i] - s.gt[i]) > thr;
    return 100.0 * bad / disp.size();
}


}  // namespace les_host
