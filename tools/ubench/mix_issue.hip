// mix_issue.hip -- how do waves of one SIMD share the instruction issue on gfx950?  One workgroup of 4 x NW waves per CU
// (wave w runs on SIMD w % 4; the waves w / 4 = 0, 1, 2 of a SIMD run different instruction streams).  Every wave times
// ITER x 64 instructions of its own stream with s_memtime; reported: cycles per instruction as seen by each stream.
//   hipcc --offload-arch=gfx950 -O3 mix_issue.hip -o mix_issue && ./mix_issue
// Streams: F = v_add_f32 (2-cycle VALU), D = v_cvt_f64_i32 (4-cycle VALU), S = s_add_u32 (SALU), X = VALU/SALU alternating,
//          P = v_add_u32_dpp, M = v_mad_i64_i32, C = v_cvt_f32_i32, W = ds_write_b128, R = ds_read_b128,
//          B = taken scalar branches (each followed by one v_add_f32), N = s_nop 0, - = idle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define ITER 1000
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

__global__ void __launch_bounds__(768) mix(unsigned long long* out, const int* kinds, float seed)
{
    extern __shared__ char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kind = kinds[wave >> 2];
    float a = seed, b = seed * 0.5f;
    double d = seed;
    int si = (int)seed;
    unsigned sacc = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (kind == 'F') {
        for (int i = 0; i < ITER; i++) asm volatile(REP64("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
    } else if (kind == 'D') {
        for (int i = 0; i < ITER; i++) asm volatile(REP64("v_cvt_f64_i32 %0, %1\n") : "+v"(d) : "v"(si));
    } else if (kind == 'P') {
        for (int i = 0; i < ITER; i++) asm volatile(REP64("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") : "+v"(si));
    } else if (kind == 'M') {
        long long acc = si;
        for (int i = 0; i < ITER; i++) asm volatile(REP64("v_mad_i64_i32 %0, s[10:11], %1, %1, %0\n") : "+v"(acc) : "v"(si) : "s10", "s11");
        d += (double)acc;
    } else if (kind == 'C') {
        for (int i = 0; i < ITER; i++) asm volatile(REP64("v_cvt_f32_i32 %0, %1\n") : "+v"(a) : "v"(si));
    } else if (kind == 'W') {
        const unsigned addr = threadIdx.x * 16;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v = {a, b, a, b};
        for (int i = 0; i < ITER; i++) asm volatile(REP64("ds_write_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v) : "memory");
    } else if (kind == 'R') {
        const unsigned addr = threadIdx.x * 16;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v;
        for (int i = 0; i < ITER; i++) { asm volatile(REP64("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(v) : "v"(addr) : "memory"); a += v.x; }
    } else if (kind == 'S') {
        for (int i = 0; i < ITER; i++) asm volatile(REP64("s_add_u32 %0, %0, 3\n") : "+s"(sacc) : : "scc");
    } else if (kind == 'X') {
        for (int i = 0; i < ITER; i++) asm volatile(REP8(REP8("v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 3\n")) : "+v"(a), "+s"(sacc) : "v"(b) : "scc");   // 128 instructions
    } else if (kind == 'N') {
        for (int i = 0; i < ITER; i++) asm volatile(REP64("s_nop 0\n"));
    } else if (kind == 'B') {
        for (int i = 0; i < ITER; i++) asm volatile(REP8(REP8("s_cmp_eq_u32 %1, %1\n s_cbranch_scc1 1f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n 1: v_add_f32 %0, %0, %2\n")) : "+v"(a), "+s"(sacc) : "v"(b) : "scc");   // 64 x (cmp, taken branch, add)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + wave] = t1 - t0;
    if (a + (float)d + (float)sacc == 123.456f) out[0] = 0;
}

int main(int argc, char** argv)
{
    const char* combos[] = {"F--", "D--", "S--", "X--", "N--", "B--", "FF-", "FFF", "FS-", "FFS", "FSS", "DS-", "DDS", "FD-", "FDS", "XX-", "XXX", "FX-", "FN-", "FFN", "BB-", "BBB", "FB-",
                            "P--", "PP-", "FP-", "DP-", "FFP", "M--", "MM-", "FM-", "DM-", "PM-", "C--", "CC-", "FC-", "DC-", "FFD", "FDD", "FFM", "FPD",
                            "W--", "WW-", "WWW", "FW-", "R--", "RR-", "RRR", "RW-", "FFW", "FFR"};
    unsigned long long* d_out;
    int* d_k;
    const int nb = 256;
    hipMalloc(&d_out, nb * 12 * 8);
    hipMalloc(&d_k, 3 * 4);
    std::vector<unsigned long long> h(nb * 12);
    for (const char* c : combos) {
        int k[3] = {c[0], c[1], c[2]};
        hipMemcpy(d_k, k, 12, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mix, dim3(nb), dim3(768), 100 * 1024, 0, d_out, d_k, 1.5f);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, nb * 12 * 8, hipMemcpyDeviceToHost);
        printf("%s :", c);
        for (int s = 0; s < 3; s++) {
            if (c[s] == '-') continue;
            double sum = 0;
            for (int b = 0; b < nb; b++) for (int w = 0; w < 4; w++) sum += (double)h[b * 12 + s * 4 + w];
            const double per = (c[s] == 'X') ? 128.0 : (c[s] == 'B' ? 192.0 : 64.0);
            // s_memtime counts at 100 MHz on this part; the shader clock is assumed 2.4 GHz in the conversion below, so only RATIOS between lines are meaningful
            printf("  %c %.2f", c[s], sum / (nb * 4) / (ITER * per) * 24.0);
        }
        printf("\n");
    }
    return 0;
}
