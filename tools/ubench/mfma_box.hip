// mfma_box.hip -- go / no-go measurement: horizontal 2R+1 box sums of int32 tiles on the matrix cores, EXACT (modulo 2^32).
//
// The march kernel's horizontal box sums are VALU + LDS work (les_march.h).  The one formulation DESIGN 6.2 had not measured: a 21-wide
// box sum of a 16 x 64 tile is a product with a banded 0/1 matrix, and with the operands split into int8 limbs
// v_mfma_i32_16x16x64_i8 accumulates it exactly in int32:
//     x = s0 + s1 2^8 + s2 2^16 + s3 2^24  with BALANCED digits s_j in [-128, 127]:   z = (x + 0x00808080) ^ 0x00808080, s_j = (int8) byte_j(z)
//     box(x) = sum_j 2^(8j) box(s_j)  (mod 2^32),   box(s_j) = S_j x Band   on the MFMA  (|box(s_j)| <= 21 * 128: no overflow)
// Per wave: a tile of 16 rows x 64 columns of int4 (the 4 stage-1 quantities) in LDS -> 16 rows x 44 valid output columns (R = 10).
//   VARIANT 0  "mfma": lane (row = l & 15, kgroup = l >> 4) reads its 16 columns (16 ds_read_b128), balances the digits (2 VALU per dword),
//               byte-transposes them into 4 quantities x 4 limbs of MFMA A operands (v_perm_b32, 2 per dword), 16 x 3 MFMAs against three
//               band matrices held in registers, recombines the limbs (3 v_lshl_add_u32 per output dword)
//   VARIANT 1  "slide": what role C of the march kernel would do on the same tile in its (row, 8-column segment) layout: window of the first
//               column (2R+1 reads, v_add3), then slide (2 reads + 8 adds per column) -- 56 lanes x 8 columns = 7 rows x 64 columns per pass;
//               run 16/7 times so that both variants are quoted per tile element
//   VARIANT 2  only the 48 MFMAs (operands ready): the matrix-core time alone
// Reported: cycles per tile per wave with 1 and with 3 waves per SIMD issuing (the kernel's occupancy), VALU instruction counts from the
// source, and an exactness check of VARIANT 0 against a CPU box sum modulo 2^32 on random int32 data.
//   hipcc --offload-arch=gfx950 -O3 mfma_box.hip -o mfma_box && ./mfma_box
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int R = 10, KS = 2 * R + 1, ROWS = 16, COLS = 64, PC = COLS + 8;       // row pitch in int4 elements (8 mod 16)
constexpr int NOUT = COLS - 2 * R;                                               // 44 valid output columns (10 .. 53)
#define ITER 200

__device__ __forceinline__ int perm(int hi, int lo, unsigned sel)
{
    return (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, sel);
}

// 4 x 4 byte transpose: in: 4 dwords (columns c..c+3 of one quantity), out[j] = bytes j of the four columns (limb j, 4 k-slots)
__device__ __forceinline__ void transpose4(const int a, const int b, const int c, const int d, int (&o)[4])
{
    // v_perm_b32 selects bytes from {src0 (bytes 7..4), src1 (bytes 3..0)}
    const int t0 = perm(b, a, 0x05010400);   // a0 b0 a1 b1
    const int t1 = perm(b, a, 0x07030602);   // a2 b2 a3 b3
    const int t2 = perm(d, c, 0x05010400);   // c0 d0 c1 d1
    const int t3 = perm(d, c, 0x07030602);   // c2 d2 c3 d3
    o[0] = perm(t2, t0, 0x05040100);         // a0 b0 c0 d0
    o[1] = perm(t2, t0, 0x07060302);         // a1 b1 c1 d1
    o[2] = perm(t3, t1, 0x05040100);
    o[3] = perm(t3, t1, 0x07060302);
}

template <int VARIANT>
__global__ void __launch_bounds__(768) box_kernel(const int* __restrict__ in, int* __restrict__ out, unsigned long long* __restrict__ cycles, int check)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // one tile per SIMD (waves w, w + 4, w + 8 of a workgroup land on the same SIMD and read the same tile: 12 tiles would not fit the LDS)
    v4i* T = reinterpret_cast<v4i*>(lds) + (size_t)(wave & 3) * ROWS * PC;
    if (wave < 4) {
        // fill the tile: in[workgroup * 4 + wave][row][col][4]
        for (int i = lane; i < ROWS * COLS; i += 64) {
            const int r = i / COLS, c = i % COLS;
            const int* p = in + (((size_t)(blockIdx.x * 4 + wave) * ROWS + r) * COLS + c) * 4;
            T[r * PC + c + c / 8] = v4i{p[0], p[1], p[2], p[3]};
        }
    }
    __syncthreads();
    v4i acc_out[3][4];                                                            // [band group][quantity]: 4 rows x 1 column per lane
#pragma unroll
    for (int t = 0; t < 3; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc_out[t][q] = v4i{0, 0, 0, 0};
    int sl[8][4];                                                                 // VARIANT 1 results
    const int m = lane & 15, g = lane >> 4;
    // band matrices: B operand of lane (n = l & 15, kgroup g): byte b = band(column 16 g + b, output column 10 + 16 t + n)
    v4i band[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int o = R + 16 * t + m;
        int w[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; b++) {
            const int c = 16 * g + b;
            if (c >= o - R && c <= o + R && o < COLS - R) w[b >> 2] |= 1 << (8 * (b & 3));
        }
        band[t] = v4i{w[0], w[1], w[2], w[3]};
    }
    v4i a[4][4];                                                                  // [quantity][limb]: 16 k-slots each
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
        asm volatile("" ::: "memory");                                            // the tile is re-read every iteration (no hoisting out of the loop)
        if (VARIANT == 0 || VARIANT == 2) {
            if (VARIANT == 2 && it > 0) {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int j = 0; j < 4; j++) asm volatile("" : "+v"(a[q][j]));  // operands opaque: the MFMAs stay in the loop
            }
            if (VARIANT == 0 || it == 0) {
                v4i x[16];
#pragma unroll
                for (int b = 0; b < 16; b++) { const int c = 16 * g + b; x[b] = T[m * PC + c + c / 8]; }
#pragma unroll
                for (int b = 0; b < 16; b++)                                      // balanced digits
#pragma unroll
                    for (int q = 0; q < 4; q++) x[b][q] = (x[b][q] + 0x00808080) ^ 0x00808080;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int c4 = 0; c4 < 4; c4++) {
                        int o4[4];
                        transpose4(x[4 * c4][q], x[4 * c4 + 1][q], x[4 * c4 + 2][q], x[4 * c4 + 3][q], o4);
#pragma unroll
                        for (int j = 0; j < 4; j++) a[q][j][c4] = o4[j];
                    }
            }
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v4i d[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) d[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[q][j], band[t], v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        int v = d[0][e];
                        v += d[1][e] << 8; v += d[2][e] << 16; v += d[3][e] << 24;   // (v_lshl_add_u32)
                        acc_out[t][q][e] += v;                                     // (accumulated over ITER so that nothing is optimised away)
                    }
                }
        } else {
            // (row, 8-column segment) layout of role C: lanes 0..55 = 7 rows x 8 segments
            const int rw = lane >> 3, sg = lane & 7, rr = rw < 7 ? rw : 0;
            const v4i* p1 = T + rr * PC + 8 * sg + sg;
            auto E = [](int d) { return d + (d >= 0 ? d / 8 : -((7 - d) / 8)); };
            v4i W = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < KS; u++) { const int d = u - R; const int e = 8 * sg + d; const v4i v = (e >= 0 && e < COLS) ? p1[E(d)] : v4i{0, 0, 0, 0}; W += v; }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j > 0) {
                    const int en = 8 * sg + R + j, eo = 8 * sg + j - 1 - R;
                    const v4i vn = (en < COLS) ? p1[E(R + j)] : v4i{0, 0, 0, 0}, vo = (eo >= 0) ? p1[E(j - 1 - R)] : v4i{0, 0, 0, 0};
                    W += vn - vo;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) sl[j][q] = (it == 0 ? 0 : sl[j][q]) + W[q];
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    if (VARIANT == 1) {
        int s = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) s += sl[j][0] + sl[j][1] + sl[j][2] + sl[j][3];
        if (s == 0x7fffffff) out[0] = s;
        return;
    }
    if (check) {
        // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + e; out[wave][row][output column][q], divided by ITER on the host
        for (int t = 0; t < 3; t++)
            for (int q = 0; q < 4; q++)
                for (int e = 0; e < 4; e++) {
                    const int o = 16 * t + m, row = 4 * g + e;
                    if (o < NOUT && wave < 4) out[(((size_t)(blockIdx.x * 4 + wave) * ROWS + row) * NOUT + o) * 4 + q] = acc_out[t][q][e];
                }
    } else if (acc_out[0][0][0] == 0x7fffffff) out[0] = 1;
}

// -> shader cycles per loop iteration per wave.  The counter s_memtime reads is calibrated against the kernel's wall-clock time (hipEvents)
// and the shader clock: ticks are reported as they are AND converted.
static double g_cycles_per_tick = 0.0;
template <int VARIANT>
double run(int waves_per_wg, const int* d_in, int* d_out, unsigned long long* d_cyc, int check)
{
    const int nwg = 256;
    const size_t lds = (size_t)4 * ROWS * PC * 16;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(box_kernel<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("attribute failed\n"); exit(2); }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(d_cyc, 0, (size_t)nwg * 12 * 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(box_kernel<VARIANT>, dim3(nwg), dim3(64 * waves_per_wg), lds, 0, d_in, d_out, d_cyc, check);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(2); }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c((size_t)nwg * waves_per_wg);
    hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (auto v : c) { s += (double)v; mx = std::max(mx, (double)v); }
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    // one round of workgroups (256 on 256 CUs): the longest wave's loop is (almost) the kernel -> ticks per second
    if (g_cycles_per_tick == 0.0 && mx > 0) {
        const double ticks_per_s = mx / (ms * 1e-3);
        g_cycles_per_tick = ticks_per_s > 1e9 ? 1.0 : (double)khz * 1e3 / 1e8;         // the counter runs at the shader clock, or at 100 MHz
        printf("counter: %.0f ticks in a %.3f ms kernel (%.2e ticks/s) -> %.1f shader cycles per tick (clock %d kHz)\n", mx, ms, ticks_per_s, g_cycles_per_tick, khz);
    }
    return s / c.size() / ITER * g_cycles_per_tick;
}

int main()
{
    const int maxw = 12, nwg = 256;
    const size_t n = (size_t)nwg * 4 * ROWS * COLS * 4;
    std::vector<int> h(n);
    uint64_t st = 88172645463325252ull;
    for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (int)(st >> 16); }
    int *d_in, *d_out;
    unsigned long long* d_cyc;
    hipMalloc(&d_in, n * 4);
    hipMalloc(&d_out, (size_t)nwg * maxw * ROWS * NOUT * 4 * 4);
    hipMalloc(&d_cyc, (size_t)nwg * maxw * 8);
    hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice);
    // ---- exactness of the limb formulation (four waves per workgroup, every wave a different random tile)
    run<0>(4, d_in, d_out, d_cyc, 1);
    std::vector<int> o((size_t)nwg * 4 * ROWS * NOUT * 4);
    hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int w = 0; w < nwg * 4; w++)
        for (int r = 0; r < ROWS; r++)
            for (int oc = 0; oc < NOUT; oc++)
                for (int q = 0; q < 4; q++) {
                    uint32_t ref = 0;
                    for (int c = oc; c <= oc + 2 * R; c++) ref += (uint32_t)h[(((size_t)w * ROWS + r) * COLS + c) * 4 + q];
                    ref *= (uint32_t)ITER;
                    bad += (uint32_t)o[(((size_t)w * ROWS + r) * NOUT + oc) * 4 + q] != ref;
                }
    printf("exactness: %zu of %zu box sums differ from the CPU sums modulo 2^32\n", bad, o.size());
    const double elems = (double)ROWS * NOUT;                       // valid outputs per tile (x 4 quantities)
    for (int wpw : {4, 12}) {                                        // 1 and 3 waves per SIMD
        const double c0 = run<0>(wpw, d_in, d_out, d_cyc, 0), c2 = run<2>(wpw, d_in, d_out, d_cyc, 0), c1 = run<1>(wpw, d_in, d_out, d_cyc, 0);
        const double slide_elems = 7.0 * 64;                         // outputs of one pass of the sliding variant (7 rows x 64 columns, 56 lanes)
        printf("%d wave(s) per SIMD: mfma formulation %.0f cycles per tile (%.2f per output element, 4 quantities), its MFMAs alone %.0f; "
               "sliding window %.0f cycles per 7 x 64 pass (%.2f per output element)\n",
               wpw / 4, c0, c0 / elems, c2, c1, c1 / slide_elems);
    }
    return bad != 0;
}
