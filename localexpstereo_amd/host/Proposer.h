// Proposer.h -- host-side label proposers with the reference's interface (LES/Proposer.h:6-31):
//   createInstance / startIterations(labeling, unitRegion, outerIter) / getNextProposal / isContinued.
// They read the live label map like the reference (LES/Proposer.h:64) and draw from an explicit
// cv::RNG-compatible generator (the reference uses the thread-local cv::theRNG()).
// All three proposers of the reference exist on the host (the drop-in loop PMStereo::run uses them with the reference's own
// proposer table) and on the device (csrc/les_propose.h, used by runDevice); the RANSAC proposer uses the same defined
// accumulation order as the device kernels, so the two give bit-identical planes and generator states.
#pragma once

#include <cmath>
#include <vector>

#include "les_types.h"

namespace les_host {

class IProposer {
public:
    explicit IProposer(int K) : K(K) {}
    virtual ~IProposer() {}
    virtual IProposer* createInstance() = 0;
    virtual void startIterations(const LabelMap& labeling, Rect unitRegion, int outerIter, RNG* rng) = 0;
    virtual Plane getNextProposal() = 0;
    virtual bool isContinued() = 0;

protected:
    const int K;
    const LabelMap* labeling = nullptr;
    RNG* rng = nullptr;
    int iter = 0, outerIter = 0;
    Rect rect;
    Point selectRandomPixelInRect(Rect r)                                             // LES/Proposer.h:37-44
    {
        const int n = rng->uniform(0, r.height * r.width);
        return Point{r.x + n % r.width, r.y + n / r.width};
    }
};

class ExpansionProposer : public IProposer {                                          // LES/Proposer.h:34-80
public:
    explicit ExpansionProposer(int K) : IProposer(K) {}
    IProposer* createInstance() override { return new ExpansionProposer(K); }
    void startIterations(const LabelMap& l, Rect unitRegion, int outer, RNG* r) override
    {
        labeling = &l; rect = unitRegion; outerIter = outer; rng = r; iter = 0;
    }
    Plane getNextProposal() override
    {
        const Point p = selectRandomPixelInRect(rect);
        iter++;
        return labeling->at(p.y, p.x);
    }
    bool isContinued() override { return iter < K; }
};

class RandomProposer : public ExpansionProposer {                                     // LES/Proposer.h:84-153
public:
    RandomProposer(int K, float maxDisp, float minDisp = 0) : ExpansionProposer(K), MIN_DISPARITY(minDisp), MAX_DISPARITY(maxDisp) {}
    IProposer* createInstance() override { return new RandomProposer(K, MAX_DISPARITY, MIN_DISPARITY); }
    Plane getNextProposal() override
    {
        const double PI = 3.1415926535897932384626433832795;
        const Point s = selectRandomPixelInRect(rect);
        const Plane in = labeling->at(s.y, s.x);
        const int m = outerIter + iter;
        iter++;
        float zs = in.GetZ(float(s.x), float(s.y));
        const float dz = width(m);
        const float minz = std::max(MIN_DISPARITY, zs - dz), maxz = std::min(MAX_DISPARITY, zs + dz);
        zs = rng->uniform(minz, maxz);
        const float nr = (float)std::ldexp(1.0, -m);                                  // randomNmax (=1) * 0.5^m
        float n0[3];
        in.GetNormal(n0);
        const double theta = rng->uniform(0.0, PI), phi = rng->uniform(0.0, PI * 2.0);
        const double u[3] = {std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)};
        float nv[3];
        for (int c = 0; c < 3; c++) nv[c] = n0[c] + (float)u[c] * nr;
        const double inv = 1. / std::sqrt((double)nv[0] * nv[0] + (double)nv[1] * nv[1] + (double)nv[2] * nv[2]);
        for (int c = 0; c < 3; c++) nv[c] = (float)(nv[c] * inv);
        return Plane::CreatePlane(nv[0], nv[1], nv[2], zs, float(s.x), float(s.y), in.v);
    }
    bool isContinued() override { return iter < K && !(width(outerIter + iter) < 0.1); }   // early stop, :149-152

private:
    const float MIN_DISPARITY, MAX_DISPARITY;
    float width(int m) const { return (float)((double)(MAX_DISPARITY - MIN_DISPARITY) * std::ldexp(1.0, -(m + 1))); }
};

// RansacProposer (LES/Proposer.h:155-312): startIterations snapshots the disparities of the unit region under the current
// labelling (:283-301); every proposal is one RANSACPlane run (:177-240) -- up to MAX_SAM three-point samples, inliers within
// `threshold` (1.0, :305), least-squares refit on the inliers among the FIRST no_i points (the reference's loop bound, :216),
// adaptive termination (:243-262).  cv::solve(DECOMP_SVD) of the m x 3 systems is the pseudo-inverse through the 3x3
// eigen-decomposition of A^T A in double; std::random_shuffle's first three entries are three distinct uniform draws from the
// proposer's generator.  Sums run in the order of csrc/les_propose.h (rows y = s mod 4 per partial sum s, combined as
// (p0 + p1) + (p2 + p3)): host and device agree bit for bit.
class RansacProposer : public IProposer {
public:
    explicit RansacProposer(int K, int maxSam = 500, float conf = 0.95f, float threshold = 1.0f)
        : IProposer(K), MAX_SAM(maxSam), conf(conf), threshold(threshold) {}
    IProposer* createInstance() override { return new RansacProposer(K, MAX_SAM, conf, threshold); }
    void startIterations(const LabelMap& l, Rect unitRegion, int outer, RNG* r) override
    {
        labeling = &l; rect = unitRegion; outerIter = outer; rng = r; iter = 0;
        disp.resize((size_t)rect.width * rect.height);
        for (int yy = 0; yy < rect.height; yy++)
            for (int xx = 0; xx < rect.width; xx++) {
                const float c0 = (float)xx + rect.x, c1 = (float)yy + rect.y;
                const Plane& v = l.at(yy + rect.y, xx + rect.x);
                disp[(size_t)yy * rect.width + xx] = v.a * c0 + v.b * c1 + v.c;           // :297
            }
    }
    Plane getNextProposal() override
    {
        iter++;
        const int len = rect.width * rect.height;
        int max_i = 3, max_sam = MAX_SAM, no_sam = 0, no_i_c = 0;                        // :180-185
        float result[3] = {0, 0, 0};
        while (no_sam < max_sam) {                                                       // :193
            no_sam++;
            int idx[3];                                                                  // first three entries of randperm(len), :163-174,196-201
            for (int i = 0; i < 3; i++) {
                bool again;
                do {
                    idx[i] = len > 0 ? rng->uniform(0, len) : 0;
                    again = false;
                    for (int q = 0; q < i; q++) if (idx[q] == idx[i] && len > i) again = true;
                } while (again);
            }
            double M[3][3] = {{0}}, rhs[3] = {0, 0, 0};
            for (int i = 0; i < 3; i++) {
                const int yy = idx[i] / rect.width, xx = idx[i] - yy * rect.width;
                const double c[3] = {(double)((float)xx + rect.x), (double)((float)yy + rect.y), 1.0};
                const double d = disp[(size_t)idx[i]];
                for (int a = 0; a < 3; a++) {
                    rhs[a] += c[a] * d;
                    for (int b = 0; b < 3; b++) M[a][b] += c[a] * c[b];
                }
            }
            float N[3];
            solveNormal3x3(M, rhs, N);                                                   // cv::solve(ranpts, div, N, DECOMP_SVD), :203
            const int no_i = countInliers(len, N);                                       // :204-206
            if (max_i < no_i) {                                                          // :208
                double t[9];
                refitSums(no_i, N, t);
                double A[3][3] = {{t[0], t[1], t[2]}, {t[1], t[3], t[4]}, {t[2], t[4], t[5]}};
                double r3[3] = {t[6], t[7], t[8]};
                float N2[3];
                solveNormal3x3(A, r3, N2);                                               // :224
                const int no = countInliers(len, N2);                                    // :225-227
                if (no > no_i_c) {                                                       // :229-236
                    result[0] = N2[0]; result[1] = N2[1]; result[2] = N2[2];
                    no_i_c = no;
                    max_i = no_i;
                    max_sam = std::min(max_sam, sampleCount(no, len, 3, conf));
                }
            }
        }
        return Plane{result[0], result[1], result[2], 0.0f};                             // :239
    }
    bool isContinued() override { return iter < K; }

    // LES/Proposer.h:243-262
    static int sampleCount(int ni, int ptNum, int pf, double conf)
    {
        double q = 1.0;
        for (double a = (ni - pf + 1), b = (ptNum - pf + 1); a <= ni; a += 1.0, b += 1.0) q *= (a / b);
        int cnt;
        if ((1.0 - q) < 1e-4) cnt = 1;
        else cnt = (int)(std::log(1.0 - conf) / std::log(1.0 - q));
        return cnt < 1 ? 1 : cnt;
    }

private:
    const int MAX_SAM;
    const float conf, threshold;
    std::vector<float> disp;

    bool inlier(int xx, int yy, int i, const float N[3]) const
    {
        const float x = (float)xx + rect.x, y = (float)yy + rect.y;
        const float dot = (float)(((double)x * N[0] + (double)y * N[1]) + (double)1.0f * N[2]);     // pts * N, :204
        return std::fabs(dot - disp[(size_t)i]) < threshold;
    }
    int countInliers(int upto, const float N[3]) const
    {
        int cnt = 0;
        for (int i = 0; i < upto; i++) cnt += inlier(i % rect.width, i / rect.width, i, N);
        return cnt;
    }
    // normal equations of the refit over the inliers among the first `upto` points: xx xy x yy y 1 | xd yd d.
    // The DEFINED accumulation order shared with the device kernel (csrc/les_propose.h: les_ransac_eval_kernel, one wave of 64 lanes per
    // candidate; round 6 -- rounds 1-5: four row phases): partial sum l = 16 (yy mod 4) + (xx mod 16) takes its points in increasing
    // (row, column) order, and the 64 partial sums are combined by the balanced tree of the butterfly l ^ 1, l ^ 2, ... l ^ 32.
    void refitSums(int upto, const float N[3], double t[9]) const
    {
        double acc[64][9] = {{0}};
        for (int yy = 0; yy < rect.height; yy++) {
            int i = yy * rect.width;
            if (i >= upto) break;
            for (int xx = 0; xx < rect.width && i < upto; xx++, i++) {
                if (!inlier(xx, yy, i, N)) continue;
                double* a = acc[16 * (yy & 3) + (xx & 15)];
                const double dx = (float)xx + rect.x, dy = (float)yy + rect.y, dd = disp[(size_t)i];
                a[0] += dx * dx; a[1] += dx * dy; a[2] += dx; a[3] += dy * dy; a[4] += dy; a[5] += 1.0;
                a[6] += dx * dd; a[7] += dy * dd; a[8] += dd;
            }
        }
        for (int m = 1; m < 64; m <<= 1)
            for (int l = 0; l < 64; l += 2 * m)
                for (int k = 0; k < 9; k++) acc[l][k] = acc[l][k] + acc[l + m][k];
        for (int k = 0; k < 9; k++) t[k] = acc[0][k];
    }
    // pseudo-inverse solve of the 3x3 normal equations by cyclic Jacobi sweeps in double (csrc/les_propose.h: solve_normal_3x3)
    static void solveNormal3x3(double M[3][3], const double rhs[3], float x[3])
    {
        double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int sweep = 0; sweep < 16; sweep++) {
            const double off = std::fabs(M[0][1]) + std::fabs(M[0][2]) + std::fabs(M[1][2]);
            if (off <= 1e-15 * (std::fabs(M[0][0]) + std::fabs(M[1][1]) + std::fabs(M[2][2]))) break;
            for (int p = 0; p < 2; p++)
                for (int q = p + 1; q < 3; q++) {
                    if (std::fabs(M[p][q]) < 1e-300) continue;
                    const double theta = (M[q][q] - M[p][p]) / (2 * M[p][q]);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                    const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                    for (int k = 0; k < 3; k++) {
                        const double mkp = M[k][p], mkq = M[k][q];
                        M[k][p] = c * mkp - s * mkq; M[k][q] = s * mkp + c * mkq;
                    }
                    for (int k = 0; k < 3; k++) {
                        const double mpk = M[p][k], mqk = M[q][k];
                        M[p][k] = c * mpk - s * mqk; M[q][k] = s * mpk + c * mqk;
                    }
                    for (int k = 0; k < 3; k++) {
                        const double vkp = V[k][p], vkq = V[k][q];
                        V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                    }
                }
        }
        double w[3], wsum = 0;
        for (int k = 0; k < 3; k++) { w[k] = std::sqrt(M[k][k] > 0.0 ? M[k][k] : 0.0); wsum += w[k]; }
        const double thr = wsum * 2 * 1.1920929e-07;
        double out[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++) {
            if (w[k] <= thr) continue;
            const double proj = (V[0][k] * rhs[0] + V[1][k] * rhs[1] + V[2][k] * rhs[2]) / (w[k] * w[k]);
            for (int r = 0; r < 3; r++) out[r] += V[r][k] * proj;
        }
        for (int r = 0; r < 3; r++) x[r] = (float)out[r];
    }
};

}  // namespace les_host
