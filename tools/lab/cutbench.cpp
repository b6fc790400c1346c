// cutbench.cpp -- times the host max-flow (host/GridMaxFlow.h) on dumped expansion-move graphs (device payload format: 5 floats per node).
//   g++ -O2 -std=c++17 -I localexpstereo_amd/host tools/cpp/cutbench.cpp -o tools/cpp/cutbench -lpthread && tools/cpp/cutbench cell.bin [bands] [reps]
// Input: int32 w, h, then 5 w h floats (LES_DUMP_GRAPHS=dir python tools/e2e_bench.py writes .npz lock-steps).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ExpansionMove.h"
using namespace les_host;
using clk = std::chrono::steady_clock;
int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: cutbench cell.bin [bands] [reps]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    int wh[2];
    if (fread(wh, 4, 2, f) != 2) return 1;
    const int w = wh[0], h = wh[1];
    std::vector<float> pay((size_t)5 * w * h);
    if (fread(pay.data(), 4, pay.size(), f) != pay.size()) return 1;
    fclose(f);
    const int bands = argc > 2 ? atoi(argv[2]) : 1, reps = argc > 3 ? atoi(argv[3]) : 3;
    std::vector<uint8_t> mask((size_t)w * h);
    double best = 1e30, flow = 0;
    for (int r = 0; r < reps; r++) {
        auto t0 = clk::now();
        flow = expansionMovePrebuilt(pay.data(), 0.0, Rect(0, 0, w, h), mask.data(), bands);
        best = std::min(best, std::chrono::duration<double>(clk::now() - t0).count());
    }
    size_t ch = 0;
    unsigned long long hsh = 1469598103934665603ull;
    for (uint8_t m : mask) { ch += m != 0; hsh = (hsh ^ (m != 0)) * 1099511628211ull; }
    printf("%s %dx%d bands %d: %.2f ms  flow %.6f  changed %.4f  mask hash %016llx\n", argv[1], w, h, bands, best * 1e3, flow, (double)ch / mask.size(), hsh);
    return 0;
}
