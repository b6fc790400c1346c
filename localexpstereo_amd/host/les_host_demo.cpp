// les_host_demo.cpp -- self-test of the C++ host side above the C ABI.
//
//   les_host_demo layers W H windR unit        print the LayerManager geometry (checked against the oracle
//                                              by tests/test_host_cpp.py; needs no GPU)
//   les_host_demo run [W H D iters]            synthetic piecewise-planar scene; runs the PatchMatch
//                                              iterations (a) through the drop-in operator from OpenMP
//                                              threads like the reference loop and (b) device-resident,
//                                              and checks that both reduce the energy and reach the
//                                              ground truth (needs an MI355X)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "PMStereo.h"

using namespace les_host;

static int cmd_layers(int argc, char** argv)
{
    if (argc < 6) return 2;
    const int W = atoi(argv[2]), H = atoi(argv[3]), windR = atoi(argv[4]), unit = atoi(argv[5]);
    LayerManager lm(W, H, windR);
    lm.addLayer(unit);
    const auto& L = lm.layers[0];
    printf("%d %d %zu\n", L.widthBlocks, L.heightBlocks, L.disjointRegionSets.size());
    for (size_t r = 0; r < L.unitRegions.size(); r++) {
        const Rect &u = L.unitRegions[r], &s = L.sharedRegions[r], &f = L.filterRegions[r];
        printf("%d %d %d %d  %d %d %d %d  %d %d %d %d\n", u.x, u.y, u.width, u.height, s.x, s.y, s.width, s.height, f.x, f.y, f.width, f.height);
    }
    for (const auto& set : L.disjointRegionSets) {
        for (int c : set) printf("%d ", c);
        printf("\n");
    }
    return 0;
}

struct Scene {
    int W, H, D;
    std::vector<uint8_t> im;       // BGR
    std::vector<float> vol, gt;
};

static Scene make_scene(int W, int H, int D)
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

static double bad_pixels(const std::vector<float>& disp, const Scene& s, float thr)
{
    size_t bad = 0;
    for (size_t i = 0; i < disp.size(); i++) bad += std::fabs(disp[i] - s.gt[i]) > thr;
    return 100.0 * bad / disp.size();
}

static int cmd_run(int argc, char** argv)
{
    const int W = argc > 2 ? atoi(argv[2]) : 240, H = argc > 3 ? atoi(argv[3]) : 160, D = argc > 4 ? atoi(argv[4]) : 32;
    const int iters = argc > 5 ? atoi(argv[5]) : 2;
    Scene s = make_scene(W, H, D);
    Parameters param(1.0f, 20, "GF", 1e-4f);                        // paramsGF, LES/main.cpp:73
    param.th_col = 0.5f;                                            // mc_threshold, LES/main.cpp:27,351
    const float maxdisp = (float)D - 1;

    auto build = [&](uint64_t seed) {
        auto st = std::make_unique<PMStereo>(W, H, param, maxdisp);
        st->setSeed(seed);
        st->setStereoEnergy(std::make_unique<HipCostVolumeEnergy>(s.im.data(), s.im.data(), W, H, s.vol.data(), s.vol.data(), D, param, maxdisp));
        // LES/main.cpp:391-397 layer set-up (RANSAC only exists on the device side)
        return st;
    };
    int fail = 0;
    // (a) drop-in operator called per cell from OpenMP threads (reference loop shape)
    {
        auto st = build(7);
        st->addLayer(std::max(2, int(W * 0.04)), {{LES_HIP_PROPOSE_EXPANSION, 1}, {LES_HIP_PROPOSE_RANDOM, 7}});
        st->addLayer(std::max(4, int(W * 0.12)), {{LES_HIP_PROPOSE_EXPANSION, 2}});
        st->initCurrentFast(0);
        double e_prev = st->totalCost(0);
        printf("drop-in   iter 0  E=%.1f  bad1.0=%.2f%%\n", e_prev, bad_pixels(st->computeDisparities(0), s, 1.0f));
        for (int it = 0; it < iters; it++) {
            for (size_t li = 0; li < st->layers().layers.size(); li++) st->localExpansionMovesForLayer((int)li, 0, it);
            const double e = st->totalCost(0);
            printf("drop-in   iter %d  E=%.1f  bad1.0=%.2f%%\n", it + 1, e, bad_pixels(st->computeDisparities(0), s, 1.0f));
            if (e > e_prev) { printf("FAIL: energy increased\n"); fail = 1; }
            e_prev = e;
        }
        if (bad_pixels(st->computeDisparities(0), s, 1.0f) > 25.0) { printf("FAIL: drop-in run did not converge\n"); fail = 1; }
    }
    // (b) device-resident lock-step iterations, with the RANSAC proposer
    {
        auto st = build(7);
        st->addLayer(std::max(2, int(W * 0.04)), {{LES_HIP_PROPOSE_EXPANSION, 1}, {LES_HIP_PROPOSE_RANSAC, 1}, {LES_HIP_PROPOSE_RANDOM, 7}});
        st->addLayer(std::max(4, int(W * 0.12)), {{LES_HIP_PROPOSE_EXPANSION, 2}, {LES_HIP_PROPOSE_RANSAC, 1}});
        double sec = 0;
        if (!st->runDevice(iters, {0}, &sec)) { printf("FAIL: runDevice\n"); return 1; }
        const double bad = bad_pixels(st->computeDisparities(0), s, 1.0f);
        printf("device    iter %d  E=%.1f  bad1.0=%.2f%%  (%.3f s)\n", iters, st->totalCost(0), bad, sec);
        if (bad > 15.0) { printf("FAIL: device run did not converge\n"); fail = 1; }
    }
    printf(fail ? "les_host_demo: FAILED\n" : "les_host_demo: OK\n");
    return fail;
}

int main(int argc, char** argv)
{
    if (argc >= 2 && !strcmp(argv[1], "layers")) return cmd_layers(argc, argv);
    if (argc >= 2 && !strcmp(argv[1], "run")) {
        try {
            return cmd_run(argc, argv);
        } catch (const std::exception& e) {
            printf("les_host_demo: %s\n", e.what());
            return 3;
        }
    }
    fprintf(stderr, "usage: les_host_demo layers W H windR unit | run [W H D iters]\n");
    return 2;
}
