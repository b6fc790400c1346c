// les_post.h -- dual-view post-processing on the device ("next" row N4 of SURVEY.md section 8(f)):
// left-right consistency check, horizontal nearest-valid fill and colour-weighted median of the plane labels
// (reference: PMStereoBase::doConsistencyCheck / postProcess, LES/PMStereoBase.h:111-256; weights
// StereoEnergy::computePatchWeight, LES/StereoEnergy.h:251-257).  Labels are only ever copied, never re-estimated, so the
// result is bit-identical to the CPU order of operations:
//   * disparities  a*x + b*y + c  in float, un-fused (Plane::GetZ, LES/Plane.h:51-58)
//   * weights exp(-|dI|_1 / omega) come from a 766-entry table built on the host (|dI|_1 of 8-bit colours is an integer)
//   * the weighted median sorts (disparity at p, window scan index) -- the stable order -- and accumulates the weights
//     in double, sequentially, exactly like the reference's loops (LES/PMStereoBase.h:218-247).
#pragma once

#include "les_simt.h"

namespace les {

// disparity maps of the two label maps
__global__ void les_disparity_kernel(const float4* __restrict__ labels, float* __restrict__ disp, int H, int W)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float4 l = labels[(size_t)y * W + x];
    disp[(size_t)y * W + x] = (l.x * (float)x + l.y * (float)y) + l.z;
}

// doConsistencyCheck, LES/PMStereoBase.h:111-144: 255 = inconsistent, 128 = maps outside the other view, 0 = consistent
__global__ void les_lr_check_kernel(const float* __restrict__ disp_self, const float* __restrict__ disp_other, uint8_t* __restrict__ fail,
                                    int H, int W, float sign, float threshold)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float ds = disp_self[(size_t)y * W + x];
    const float v = ((float)x - ds * sign) + 0.5f;
    uint8_t f = 128;
    if (v > -1.0e9f && v < 1.0e9f) {                       // NaN / huge: outside (int conversion would be undefined)
        const int rx = (int)v;                             // truncation toward zero, as the reference's int(...)
        if (rx >= 0 && rx < W) {
            const float dsr = disp_other[(size_t)y * W + rx];
            f = (fabsf(dsr - ds) > threshold) ? 255 : 0;
        }
    }
    fail[(size_t)y * W + x] = f;
}

// fail > 0 -> 255, and its 3x3 dilation (cv::dilate default kernel; pixels outside the image do not contribute)
__global__ void les_fail_dilate_kernel(const uint8_t* __restrict__ fail, uint8_t* __restrict__ failb, uint8_t* __restrict__ fail2, int H, int W)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    uint8_t m = 0;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            const int xx = x + dx, yy = y + dy;
            if (xx >= 0 && xx < W && yy >= 0 && yy < H && fail[(size_t)yy * W + xx]) m = 255;
        }
    failb[(size_t)y * W + x] = fail[(size_t)y * W + x] ? 255 : 0;
    fail2[(size_t)y * W + x] = m;
}

// horizontal nearest-neighbour fill, LES/PMStereoBase.h:166-201.  The donors (first pixels left / right of p outside the
// dilated mask) are never failed pixels themselves, so in-place operation reads only unmodified labels.
__global__ void les_nn_fill_kernel(const uint8_t* __restrict__ failb, const uint8_t* __restrict__ fail2, float4* __restrict__ labels, int H, int W)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t row = (size_t)y * W;
    if (!failb[row + x]) return;
    int xl = x, xr = x;
    while (xl >= 0 && fail2[row + xl] == 255) xl--;
    while (xr < W && fail2[row + xr] == 255) xr++;
    const bool hl = xl >= 0, hr = xr < W;
    if (!hl && !hr) return;
    float4 out;
    if (!hl) out = labels[row + xr];
    else if (!hr) out = labels[row + xl];
    else {
        const float4 pl = labels[row + xl], pr = labels[row + xr];
        const float zl = (pl.x * (float)x + pl.y * (float)y) + pl.z, zr = (pr.x * (float)x + pr.y * (float)y) + pr.z;
        out = (zl < zr) ? pl : pr;
    }
    labels[row + x] = out;
}

// colour-weighted median of the labels over the (2 windR + 1)^2 window, LES/PMStereoBase.h:207-250.
// One workgroup per failed pixel; NMAX = power of two >= (2 windR + 1)^2.
__device__ __forceinline__ uint32_t float_order_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // monotone map float -> uint32 (-0 < +0; NaN at the ends)
}

template <int NMAX, int NT>
__global__ void __launch_bounds__(NT)
les_weighted_median_kernel(const uint8_t* __restrict__ failb, const float4* __restrict__ src, float4* __restrict__ dst,
                           const uint32_t* __restrict__ ipk, const float* __restrict__ wtab, int H, int W, int windR)
{
    __shared__ unsigned long long s_key[NMAX];
    __shared__ float s_w[NMAX];
    __shared__ float s_ws[NMAX];
    __shared__ int s_pick;
    const int x = blockIdx.x, y = blockIdx.y;
    if (!failb[(size_t)y * W + x]) return;                 // uniform per block
    const int tid = threadIdx.x;
    const int x0 = max(x - windR, 0), y0 = max(y - windR, 0), x1 = min(x + windR + 1, W), y1 = min(y + windR + 1, H);
    const int pw = x1 - x0, n = pw * (y1 - y0);
    const uint32_t ip = ipk[(size_t)y * W + x];
    const int pb = ip & 255, pg = (ip >> 8) & 255, pr = (ip >> 16) & 255;
    for (int i = tid; i < NMAX; i += NT) {
        unsigned long long key = ~0ull;                    // padding sorts last
        float w = 0.0f;
        if (i < n) {
            const int yy = y0 + i / pw, xx = x0 + i % pw;
            const size_t q = (size_t)yy * W + xx;
            const uint32_t iq = ipk[q];
            const int ad = abs(pb - (int)(iq & 255)) + abs(pg - (int)((iq >> 8) & 255)) + abs(pr - (int)((iq >> 16) & 255));
            w = wtab[ad];
            const float4 l = src[q];
            const float z = (l.x * (float)x + l.y * (float)y) + l.z;
            key = ((unsigned long long)float_order_key(z) << 32) | (unsigned)i;
        }
        s_key[i] = key;
        s_w[i] = w;
    }
    __syncthreads();
    // bitonic sort of the keys
    for (int k = 2; k <= NMAX; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NMAX; i += NT) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = s_key[i], b = s_key[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s_key[i] = b; s_key[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += NT) s_ws[i] = s_w[(int)(s_key[i] & 0xffffffffu)];
    __syncthreads();
    if (tid == 0) {
        double sumw = 0;
        for (int i = 0; i < n; i++) sumw += (double)s_w[i];              // window scan order
        const double center = sumw / 2.0;
        double cum = 0;
        int pick = -1;
        for (int j = 0; j < n; j++) {
            cum += (double)s_ws[j];
            if (cum > center) { pick = (int)(s_key[j] & 0xffffffffu); break; }
        }
        s_pick = pick;
    }
    __syncthreads();
    if (tid == 0 && s_pick >= 0) {
        const int i = s_pick;
        dst[(size_t)y * W + x] = src[(size_t)(y0 + i / pw) * W + (x0 + i % pw)];
    }
}

}  // namespace les
