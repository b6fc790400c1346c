#include <chrono>
#include <cstdio>
#include "GridMaxFlow.h"
#include "les_types.h"
using namespace les_host;
using clk = std::chrono::steady_clock;
int main(int argc, char** argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 405, smooth = argc > 2 ? atoi(argv[2]) : 1;
    for (int bands : {1, 2, 4, 8}) {
        RNG rng(5);
        GridMaxFlow g(S, S);
        for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) {
            float s = rng.uniform(0.f, 0.5f), t = rng.uniform(0.f, 0.5f);
            if (smooth) { s = 0.2f + 0.1f * std::sin(0.05f * x) + rng.uniform(0.f, 0.02f); t = 0.2f + 0.1f * std::cos(0.04f * y) + rng.uniform(0.f, 0.02f); }
            g.add_tweights(x, y, s, t);
            const int dx[4] = {1, 0, -1, 1}, dy[4] = {0, 1, 1, 1}, dir[4] = {GridMaxFlow::E, GridMaxFlow::S, GridMaxFlow::SW, GridMaxFlow::SE};
            for (int k = 0; k < 4; k++) { int xx = x + dx[k], yy = y + dy[k]; if (xx < 0 || xx >= S || yy >= S) continue; g.add_edge(x, y, dir[k], rng.uniform(0.f, smooth ? 0.05f : 0.5f), 0.f); }
        }
        auto t0 = clk::now();
        double f = g.maxflow(bands);
        auto t1 = clk::now();
        printf("S=%d bands=%d: %.2f ms flow %.4f\n", S, bands, 1e3 * std::chrono::duration<double>(t1 - t0).count(), f);
    }
}
