// tiecheck.cpp -- do the Boykov-Kolmogorov cut and the BK-with-budget + push-relabel cut of a dumped cell differ, and if so, are both minimum
// cuts (equal capacity in double: the differing nodes are float ties)?   g++ -O2 -std=c++17 -I localexpstereo_amd/host -I include tools/cpp/tiecheck.cpp -o tools/cpp/tiecheck -lpthread
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ExpansionMove.h"
using namespace les_host;
static double cut_capacity(const std::vector<float>& pay, int w, int h, const std::vector<uint8_t>& src)
{
    double c = 0;
    const int dx[4] = {1, 0, -1, 1}, dy[4] = {0, 1, 1, 1};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* p = &pay[5 * ((size_t)y * w + x)];
            const bool s = src[(size_t)y * w + x] != 0;
            if (s && p[0] < 0) c += (double)-p[0];
            if (!s && p[0] > 0) c += (double)p[0];
            for (int k = 0; k < 4; k++) {
                const int xx = x + dx[k], yy = y + dy[k];
                if (xx < 0 || xx >= w || yy >= h) continue;
                if (s && !src[(size_t)yy * w + xx]) c += (double)p[1 + k];
            }
        }
    return c;
}
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb");
    int wh[2];
    if (!f || fread(wh, 4, 2, f) != 2) return 1;
    const int w = wh[0], h = wh[1];
    std::vector<float> pay((size_t)5 * w * h);
    if (fread(pay.data(), 4, pay.size(), f) != pay.size()) return 1;
    fclose(f);
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 0;
    if (seed) {                                    // perturb: a different float instance with the same structure
        srand(seed);
        for (size_t i = 0; i < pay.size(); i++) if (pay[i] != 0.f && std::fabs(pay[i]) < 1e5f) pay[i] *= 1.0f + 0.2f * ((float)rand() / RAND_MAX - 0.5f);
    }
    GridMaxFlow bk;
    bk.reset_for_load(w, h);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) bk.load_node(x, y, &pay[5 * ((size_t)y * w + x)]);
    const double fb = bk.maxflow();
    std::vector<uint8_t> a((size_t)w * h), b((size_t)w * h);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) a[(size_t)y * w + x] = bk.what_segment(x, y) == GridMaxFlow::SOURCE;
    const double ca = cut_capacity(pay, w, h, a);
    for (double budget : {0.5, 2.0, 6.0, 12.0}) {
        GridMaxFlow part;
        part.reset_for_load(w, h);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) part.load_node(x, y, &pay[5 * ((size_t)y * w + x)]);
        const double f1 = part.maxflow(1, budget);
        if (!part.exhausted()) { printf("  budget %.1f: BK finished\n", budget); continue; }
        GridPushRelabel pr;
        pr.reset_for_load(w, h);
        float rc8[8], tr;
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { part.residual(x, y, rc8, &tr); pr.load_residual(x, y, rc8, tr); }
        pr.set_base_flow(f1);
        const double f2 = pr.maxflow();
        size_t diff = 0;
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { b[(size_t)y * w + x] = pr.what_segment(x, y) == GridPushRelabel::SOURCE; diff += a[(size_t)y * w + x] != b[(size_t)y * w + x]; }
        const double cb = cut_capacity(pay, w, h, b);
        printf("  budget %.1f: %zu nodes differ; cut capacity BK %.6f hybrid %.6f (rel diff %.2e); flows %.6f %.6f\n", budget, diff, ca, cb, (cb - ca) / ca, fb, f2);
    }
    return 0;
}
