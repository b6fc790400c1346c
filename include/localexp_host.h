/* localexp_host.h -- C ABI of the host-side graph-cut fusion (liblocalexp_host.so).
 *
 * "Next" rows N1/N2 of the hot-path scope: the local alpha-expansion that consumes the unary costs produced by
 * liblocalexp_hip.so.  It replaces, for callers that are not C++ (the Python driver, other FFI hosts):
 *   - StereoEnergy::initSmoothnessCoeff / computeSmoothnessTerm / computeSmoothnessTermsExpansion /
 *     computeSmoothnessCost (LES/StereoEnergy.h:131-230, 398-453)
 *   - FastGCStereo::expansionMoveBK (LES/FastGCStereo.h:411-597) including the Boykov-Kolmogorov max-flow library
 *     the reference links (maxflow/README.TXT; Graph<float,float,double>)
 *   - the doGC == true branch of FastGCStereo::localExpansionMovesForLayer_CPU for one lock-step of a disjoint set
 *     (LES/FastGCStereo.h:30-63): OpenMP over the cells, mask -> current cost / label update.
 * Pure host code (C++ inside, OpenMP); no GPU calls.  The label and cost maps live in the context so that a driver
 * can hand the device the updated labels without an extra copy (les_gc_labels returns the context's own buffer).
 * C++ callers use localexpstereo_amd/host/{MaxFlow,ExpansionMove,PMStereo}.h directly.
 */
#ifndef LOCALEXP_HOST_H
#define LOCALEXP_HOST_H

#include <stdint.h>

#include "localexp_hip.h"      /* les_hip_rect, les_hip_plane */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct les_gc_ctx les_gc_ctx;

/* imL / imR: H x W x 3 uint8 BGR (either may be NULL if its view is never used).  lambda, th_smooth, omega, epsilon:
 * Parameters::lambda / th_smooth / omega / epsilon (LES/StereoEnergy.h:13-40).  Returns 0 on success. */
int les_gc_create(les_gc_ctx** out, int H, int W, const uint8_t* imL, const uint8_t* imR, float lambda, float th_smooth, float omega,
                  float epsilon);
void les_gc_destroy(les_gc_ctx* ctx);
const char* les_gc_last_error(void);

/* The context's current solution of a view: labels H x W x 4 float (a, b, c, v), costs H x W float.  The pointers stay
 * valid for the life of the context; callers initialise them (e.g. from the PatchMatch iterations) and read them back. */
float* les_gc_labels(les_gc_ctx* ctx, int mode);
float* les_gc_costs(les_gc_ctx* ctx, int mode);

/* One lock-step: for every i < n fuse the proposal planes[i] into the current solution over regions[i] (the cell's
 * shared region) given its unary costs in proposal_cost (row-major H x W, only regions[i] is read).  Regions of one call
 * must come from one disjoint set (LES/LayerManager.h:168-172) -- they are cut concurrently on `nthreads` threads
 * (<= 0: one per cell, at most 24 -- larger teams were measured slower; OMP_WAIT_POLICY=passive helps).  check != 0 runs the reference's flow == energy self-check
 * (LES/FastGCStereo.h:561-594) and returns the largest relative gap in *max_gap (may be NULL). */
int les_gc_expansion_moves(les_gc_ctx* ctx, int mode, int n, const les_hip_rect* regions, const les_hip_plane* planes,
                           const float* proposal_cost, int nthreads, int check, double* max_gap);

/* The same lock-step on graphs whose capacities were computed on the device (include/localexp_hip.h:
 * les_hip_batch_expansion_graph): payload = 5 floats per node {terminal residual, caps E, S, SW, SE}, cell i at
 * 5 * offsets[i]; flow0[i] (may be NULL) = flow already routed by the t-links; flows (may be NULL) receives the flow
 * values.  Only the max-flow, the segment readout and the mask updates (LES/FastGCStereo.h:553-559, :61-62) run on
 * the host. */
int les_gc_expansion_moves_prebuilt(les_gc_ctx* ctx, int mode, int n, const les_hip_rect* regions, const les_hip_plane* planes,
                                    const float* proposal_cost, const float* payload, const long long* offsets, const double* flow0,
                                    int nthreads, double* flows);

/* Stateless form for device-resident solutions: max-flow + segment readout only.  masks: one byte per graph node in
 * payload order (255 = the node takes the proposal, LES/FastGCStereo.h:555-559); the caller applies them on the device
 * (les_hip_batch_apply_masks).  flows (may be NULL): flow through the n-links and residual t-links of every cell. */
int les_gc_solve_prebuilt(int n, const les_hip_rect* regions, const float* payload, const long long* offsets, int nthreads,
                          unsigned char* masks, double* flows);

/* The same cut continued from a RESIDUAL graph (the state in which the tiled device max-flow hands straggler cells over,
 * include/localexp_hip.h: les_hip_batch_solve_graphs_tiled; host/ResidualCut.h): rc8 = 8 residual capacities per node towards
 * E W S N SW NE SE NW, ex = one float per node (> 0 excess, < 0 remaining sink capacity), cell i at node offsets[i].  The
 * residual graph of any feasible preflow has the minimum cuts of the original graph, so the masks are those of
 * les_gc_solve_prebuilt on the payload the preflow started from (LES/FastGCStereo.h:553-559).  solver: 0 = search from the
 * excess nodes with the push-relabel continuation, 1 = push-relabel only.  flows (may be NULL): the flow routed HERE. */
int les_gc_solve_residual(int n, const les_hip_rect* regions, const float* rc8, const float* ex, const long long* offsets, int nthreads,
                          int solver, unsigned char* masks, double* flows);

/* Host construction of the same payload from the context's current solution (the code path of
 * les_gc_expansion_moves up to the max-flow): the parity reference of the device construction. */
int les_gc_build_graphs(les_gc_ctx* ctx, int mode, int n, const les_hip_rect* regions, const les_hip_plane* planes,
                        const float* proposal_cost, const long long* offsets, float* payload, double* flow0);

/* StereoEnergy::computeSmoothnessCost (LES/StereoEnergy.h:165-203) and the data term (sum of the current costs,
 * LES/Evaluator.h:119-121) of the context's current solution. */
double les_gc_smoothness_cost(les_gc_ctx* ctx, int mode);
double les_gc_data_cost(les_gc_ctx* ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif
