/*
 * les_oracle.h -- CPU restatement ("oracle") of the LocalExpStereo matching-cost hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product path (localexpstereo_amd + liblocalexp_hip.so)
 * never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (t-taniai/LocalExpStereo) ships no tests, golden vectors or
 * fixtures for this path, and it cannot be built here (MSVC-only code; OpenCV 3.1 and BK maxflow are
 * un-vendored).  This file restates the reference's own source line by
 * line (citations are LES/<file>:<line> = LocalExpansionStereo/<file>) and restates, from
 * recollection, the published behaviour of the OpenCV 3.1 calls it makes (cv::boxFilter un-normalised
 * zero-padded window sums with double accumulation, cv::RNG multiply-with-carry generator,
 * convertTo/saturate semantics).  It is pinned only by known-answer properties derived from the
 * reference code (tests/test_oracle_*.py) and by an independent numpy restatement of the guided
 * filter formula.
 *
 * All functions are plain C ABI so tests can bind them with ctypes.
 */
#ifndef LES_ORACLE_H
#define LES_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int x, y, w, h; } les_rect;        /* cv::Rect {x,y,width,height} */
typedef struct { float a, b, c, v; } les_plane;     /* LES/Plane.h:4-9 */

#define LES_COST_FOR_INVALID 1000000.0f            /* LES/StereoEnergy.h:45 */

/* ---------------- cv::RNG restatement (OpenCV 3.1 core, recollection) ---------------- */
typedef struct { uint64_t state; } les_rng;
void     les_rng_seed(les_rng* r, uint64_t seed);
uint32_t les_rng_next(les_rng* r);
int      les_rng_uniform_int(les_rng* r, int a, int b);
float    les_rng_uniform_float(les_rng* r, float a, float b);
double   les_rng_uniform_double(les_rng* r, double a, double b);

/* ---------------- Plane (LES/Plane.h) ---------------- */
les_plane les_plane_create(float nx, float ny, float nz, float z, float x, float y, float v); /* :23-31 */
void      les_plane_normal(const les_plane* p, float n[3]);                                  /* :42-50 */
float     les_plane_z(const les_plane* p, float x, float y);                                 /* :51-54 */

/* ---------------- LayerManager (LES/LayerManager.h:44-185) ---------------- */
typedef struct les_layer les_layer;
les_layer* les_layer_create(int width, int height, int windR, int unitRegionSize);
void       les_layer_destroy(les_layer* L);
int        les_layer_height_blocks(const les_layer* L);
int        les_layer_width_blocks(const les_layer* L);
int        les_layer_num_cells(const les_layer* L);
void       les_layer_rects(const les_layer* L, les_rect* unit, les_rect* shared, les_rect* filter);
int        les_layer_num_sets(const les_layer* L);
int        les_layer_set_size(const les_layer* L, int set);
void       les_layer_set_cells(const les_layer* L, int set, int* cells);

/* ---------------- Energy context: CostVolumeEnergy + FastGuidedImageFilter<double> ---------------- */
typedef struct les_oracle les_oracle;

/* LES/CostVolumeEnergy.h:16-43 (filterName "GF": FastGuidedImageFilter<double>(im, windR/2, eps, 1/255)).
 * imL/imR: H x W x 3 uint8 (BGR interleaved, as cv::imread); volL/volR: float [D][H][W] (not copied,
 * must outlive the context; either may be NULL if that mode is never used).  use_float != 0 selects
 * the "GFfloat" variant (FastGuidedImageFilter<float>, LES/CostVolumeEnergy.h:33-37). */
les_oracle* les_oracle_create(const uint8_t* imL, const uint8_t* imR, int H, int W,
                              const float* volL, const float* volR, int D,
                              int windR, double eps, float th_col,
                              float max_disparity, float min_disparity, int use_float);
void les_oracle_destroy(les_oracle* o);

/* NaiveStereoEnergy (LES/StereoEnergy.h:629-764, MiddV2 mode / BASELINE config 1): photometric unary -- the
 * other view warped by the plane (bilinear, replicate border), truncated L1 colour + gradient -- followed by the
 * same guided filter.  The returned context works with every les_oracle_unary* / les_oracle_pm_* function.
 * cv::warpAffine's exact fixed-point interpolation cannot be pinned here: tolerance-level restatement. */
les_oracle* les_oracle_create_naive(const uint8_t* imL, const uint8_t* imR, int H, int W, int windR, double eps,
                                    float alpha, float th_col, float th_grad, float max_disparity, float min_disparity);

/* Guide statistics of view `mode` as 13 double planes of H*W:
 * order I_b,I_g,I_r (Ichannels[0..2]), mean_I[0..2], invrr,invrg,invrb,invgg,invgb,invbb, N
 * (LES/GuidedFilter.h:58-102).  "r,g,b" naming in the reference is cosmetic: channel 0 is B. */
void les_oracle_get_stats(const les_oracle* o, int mode, double* out13);

/* LES/CostVolumeEnergy.h:55-98 gather part only (interpolate == 1): raw truncated cost over filterRect,
 * written densely (Hf x Wf) to raw. */
void les_oracle_gather(const les_oracle* o, int mode, les_rect filterRect, les_plane plane, float* raw);

/* LES/GuidedFilter.h:301-326 + :248-266 (filter_raw :142-247): sub-region guided filter of a dense
 * Hf x Wf float image p over filterRect; dense output q (Hf x Wf). */
void les_oracle_filter_subregion(const les_oracle* o, int mode, les_rect filterRect, const float* p, float* q);

/* LES/StereoEnergy.h:577-610: validity mask (255/0) over rect pos, dense. */
void les_oracle_valid_mask(const les_oracle* o, les_rect pos, les_plane plane, uint8_t* mask);

/* LES/CostVolumeEnergy.h:55-174 / :176-183.  `costs` points at the element (filterRect.y, filterRect.x)
 * of a row-major float map with `stride` floats per row (== the view proposalCost(filterRect) of
 * LES/FastGCStereo.h:49).  Only costs(targetRect - filterRect.tl()) is written. */
void les_oracle_unary_nocheck(const les_oracle* o, int mode, les_rect filterRect, les_rect targetRect,
                              float* costs, int stride, les_plane plane);
void les_oracle_unary(const les_oracle* o, int mode, les_rect filterRect, les_rect targetRect,
                      float* costs, int stride, les_plane plane);

/* Batched form used by tests/bench: n independent calls of les_oracle_unary writing into one H x W
 * cost map (OpenMP over calls, as LES/FastGCStereo.h:30 parallelises over cells). */
void les_oracle_unary_batch(const les_oracle* o, int mode, int n, const les_rect* filterRects,
                            const les_rect* targetRects, const les_plane* planes, float* cost_map,
                            int check, int nthreads);

/* Whole-image aggregation of n hypothesis planes into out[n][H][W] (BASELINE.md H1/H2 workloads). */
void les_oracle_aggregate_planes(const les_oracle* o, int mode, int n, const les_plane* planes, float* out,
                                 int check, int nthreads);

/* PatchMatch-style winner-take-all update, LES/FastGCStereo.h:56-60:
 * mask = cur > prop (strict); cur <- prop, label <- plane under mask, over rect (maps are H x W). */
void les_oracle_wta_update(int W, les_rect rect, float* cur_cost, const float* prop_cost,
                           les_plane* labels, les_plane plane);

/* One disjoint set of the PatchMatch-style loop in the reference's own per-cell order (LES/FastGCStereo.h:30-64,
 * doGC == false) and initCurrentFast (LES/FastGCStereo.h:94-115).  kinds: 0 Expansion, 1 Random, 2 Ransac. */
void les_oracle_pm_set(const les_oracle* o, int mode, int n, const les_rect* units, const les_rect* shared,
                       const les_rect* filter, uint64_t* states, int nprop, const int* kinds, const int* Ks,
                       les_plane* labels, float* cur_cost, float* prop_cost, int iteration, int nthreads);
void les_oracle_pm_init(const les_oracle* o, int mode, int n, const les_rect* units, uint64_t* states,
                        les_plane* labels, float* cur_cost, int nthreads);

/* ---------------- label generation (LES/StereoEnergy.h:120-129, LES/Utilities.hpp:254-261,
 *                  LES/Proposer.h, LES/FastGCStereo.h:231-238) ---------------- */
void      les_random_unit_vector(les_rng* r, double thetaRange, double n[3]);
les_plane les_create_random_label(les_rng* r, float min_disp, float max_disp, int sx, int sy);
void      les_select_random_pixel(les_rng* r, les_rect rect, int* px, int* py);
/* ExpansionProposer::getNextProposal, LES/Proposer.h:69-75 (labels: H x W planes, row stride W) */
les_plane les_expansion_proposal(les_rng* r, const les_plane* labels, int W, les_rect unit);
/* RandomProposer, LES/Proposer.h:93-152; m = outerIter + iter */
float     les_random_perturbation_width(float min_disp, float max_disp, int m);
les_plane les_random_proposal(les_rng* r, const les_plane* labels, int W, les_rect unit, int m,
                              float min_disp, float max_disp);
int       les_random_is_continued(int iter, int K, int outerIter, float min_disp, float max_disp);
/* RansacProposer, LES/Proposer.h:163-311.  Uses its own shuffle stream (the reference uses
 * std::random_shuffle -> rand(), LES/Proposer.h:171) driven by `r`. */
les_plane les_ransac_proposal(les_rng* r, const les_plane* labels, int W, les_rect unit,
                              int max_sam, float conf, float threshold);
int       les_ransac_sample_count(int ni, int ptNum, int pf, double conf); /* :243-262 */
void      les_oracle_solve_mx3(const float* A, const float* b, int m, float* x); /* cv::solve(A, b, x, DECOMP_SVD), m x 3 (Proposer.h:203,224) */

/* ---------------- volume preparation (LES/main.cpp:146-199), "next" row N3 ---------------- */
void les_fill_out_of_view(float* vol, int D, int H, int W, int mode);
void les_convert_volume_l2r(const float* src, float* dst, int D, int H, int W);

/* PMStereoBase::doConsistencyCheck / postProcess (LES/PMStereoBase.h:111-256) */
void les_consistency_check(const float* dispL, const float* dispR, int H, int W, float dispThreshold, uint8_t* failL, uint8_t* failR);
void les_post_process(les_plane* labelsL, les_plane* labelsR, const uint8_t* imL, const uint8_t* imR, int H, int W, int windR,
                      float threshold, float omega);

/* ---------------- pairwise terms and graph construction of one expansion move ("next" row N1) ----------------
 * Restates, in the reference's own shape (whole-matrix passes, then the graph built in the reference's loop order):
 *   StereoEnergy::initSmoothnessCoeff           LES/StereoEnergy.h:131-163   (8 coefficient maps with a 1-pixel zero margin)
 *   StereoEnergy::computeSmoothnessTerm         LES/StereoEnergy.h:225-230
 *   StereoEnergy::computeSmoothnessTermsExpansion (onlyForward)  LES/StereoEnergy.h:398-453
 *   FastGCStereo::expansionMoveBK, graph part   LES/FastGCStereo.h:422-551
 * img: H x W x 3 BGR uint8 of the view; labels: H x W planes (the current labelling); cur / prop: H x W current and proposal
 * cost maps; region: the cell's shared region; label1: the proposal.  Output: 5 floats per node, row-major over the region:
 * { terminal residual (source - sink) after all add_tweights calls, capacity of the arcs i -> j towards GE, EG, LG, GG } and
 * the flow the t-links already routed.  The BK library is absent from /root/reference: Graph::add_tweights / add_edge are
 * restated from the published maxflow-v3 semantics [recollection], cv::reduce sums the 4 channels in order in float and
 * cv::exp is taken as expf [recollection]: this part of the oracle is unpinned like the rest. */
void les_oracle_expansion_graph(const uint8_t* img, int H, int W, const les_plane* labels, const float* cur, const float* prop,
                                les_rect region, les_plane label1, float lambda, float th_smooth, float omega, float epsilon,
                                float* payload, double* flow);

#ifdef __cplusplus
}
#endif
#endif
