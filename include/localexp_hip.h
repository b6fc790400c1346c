/*
 * localexp_hip.h -- C ABI of liblocalexp_hip.so: the MI355X (gfx950) implementation of the
 * LocalExpStereo matching-cost hot path.
 *
 * Drop-in boundary: the reference's operator interface
 *     virtual void StereoEnergy::ComputeUnaryPotential(const cv::Rect& filterRect,
 *             const cv::Rect& targetRect, const cv::Mat& costs, const Plane& plane,
 *             Reusable& reusable, int mode) const            (LES/StereoEnergy.h:625-626)
 * as implemented by CostVolumeEnergy (LES/CostVolumeEnergy.h:16-183) with the default "GF" joint
 * filter (FastGuidedImageFilter<double>, LES/GuidedFilter.h:283-327).  A maintainer binds these
 * entry points from a StereoEnergy subclass installed through
 * PMStereoBase::setStereoEnergyCPU (LES/PMStereoBase.h:58-61); see INTEGRATION.md and
 * localexpstereo_amd/host/HipCostVolumeEnergy.h.
 *
 * Plain C: pointers and sizes only, int status returns (0 = OK), no exceptions cross the boundary.
 * Every function fails (non-zero) when no HIP device is available -- there is no CPU fallback.
 */
#ifndef LOCALEXP_HIP_H
#define LOCALEXP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct les_hip_ctx les_hip_ctx;       /* one energy object (both views), cf. CostVolumeEnergy   */
typedef struct les_hip_batch les_hip_batch;   /* prepared geometry of one lock-step (a disjoint set)    */

typedef struct { int x, y, w, h; } les_hip_rect;        /* cv::Rect                                       */
typedef struct { float a, b, c, v; } les_hip_plane;     /* struct Plane, LES/Plane.h:4-9                  */

enum {
    LES_HIP_OK = 0,
    LES_HIP_ERR_ARG = 1,          /* bad argument                                                      */
    LES_HIP_ERR_DEVICE = 2,       /* no usable HIP device / HIP runtime error (see les_hip_last_error) */
    LES_HIP_ERR_UNSUPPORTED = 3   /* e.g. a guided-filter radius no kernel was instantiated for        */
};

/* Constructor arguments of CostVolumeEnergy(imL, imR, volL, volR, Parameters, MAX_DISPARITY,
 * MIN_DISPARITY) -- LES/CostVolumeEnergy.h:16; Parameters fields LES/StereoEnergy.h:13-40. */
typedef struct {
    int H, W, D;                  /* image rows, cols; volume slices (ndisp)                           */
    int windR;                    /* Parameters::windR; guided-filter radius is windR/2 (:30)          */
    double eps;                   /* Parameters::filter_param1                                         */
    float th_col;                 /* Parameters::th_col (mc_threshold, LES/main.cpp:351)               */
    float max_disparity;          /* MAX_DISPARITY (ndisp-1, LES/main.cpp:341)                         */
    float min_disparity;          /* MIN_DISPARITY                                                     */
    int device;                   /* HIP device ordinal                                                */
    int volumes_on_device;        /* != 0: volL/volR are device pointers owned by the caller (shared, not
                                     copied -- like the ref-counted cv::Mat headers, :20-21).  Their
                                     contents must not change behind the context's back: the cost range
                                     of the fixed-point kernel and the tiled copy that steep planes
                                     gather from (les_hip_tiled_volume_bytes) are taken at creation --
                                     after refilling a volume in place call les_hip_refresh_volume    */
} les_hip_params;

/* replaces: CostVolumeEnergy::CostVolumeEnergy (LES/CostVolumeEnergy.h:16-43) including the two
 * FastGuidedImageFilter<double> constructions (global guide statistics, LES/GuidedFilter.h:58-102).
 * imL/imR: H x W x 3 uint8 BGR host images; volL/volR: float [D][H][W] (host, copied to HBM once;
 * or device when volumes_on_device).  Either view may be NULL if its mode is never used. */
int les_hip_create(les_hip_ctx** out, const les_hip_params* params, const uint8_t* imL, const uint8_t* imR,
                   const float* volL, const float* volR);

/* After the caller has overwritten the cost volume of view `mode` in place (volumes_on_device; same shape): re-derives what the context
 * keeps of it -- the cost range that fixes the fixed-point scales of the march kernel, the finite-ness test, the tiled copy that steep
 * planes gather from.  The guide statistics (LES/GuidedFilter.h:58-102) depend on the images only and stay.  No reference counterpart:
 * CostVolumeEnergy shares the caller's cv::Mat and derives nothing from it (LES/CostVolumeEnergy.h:20-21).  Synchronises the stream; must
 * not run concurrently with evaluations on the same context. */
int les_hip_refresh_volume(les_hip_ctx* ctx, int mode);

/* replaces: NaiveStereoEnergy::NaiveStereoEnergy (LES/StereoEnergy.h:638-689) -- the image-based matching cost of the
 * MiddV2 configuration (LES/main.cpp:86-121, PMStereoBase.h:37): no cost volume; the raw cost of a plane is the truncated
 * colour + x-gradient difference between this view and the other view warped by the plane (StereoEnergy.h:702-742), then
 * the same guided-filter aggregation and validity rule.  params->D and the volume fields are ignored, th_col is
 * Parameters::th_col (10 for MiddV2), alpha / th_grad are Parameters::alpha / th_grad.  Both images are required.
 * Every other entry point (unary_one / unary_batch / batch_* / wta) works on the returned context unchanged.
 * A prepared batch of such a context owns one raw-cost patch buffer per view (the sum of its filterRect areas in floats,
 * allocated on the view's first les_hip_batch_run): runs of the same batch and view must be stream-ordered. */
int les_hip_create_naive(les_hip_ctx** out, const les_hip_params* params, const uint8_t* imL, const uint8_t* imR,
                         float alpha, float th_grad);
void les_hip_destroy(les_hip_ctx* ctx);                 /* replaces: ~CostVolumeEnergy (:50-52)          */
const char* les_hip_last_error(void);                   /* thread-local description of the last failure  */

/* All launches go to this hipStream_t (NULL = the default stream).  Not in the reference. */
int les_hip_set_stream(les_hip_ctx* ctx, void* hip_stream);
/* bind != 0: from now on the launches the CALLING host thread makes on this context (and its les_hip_synchronize) go to
 * hip_stream instead; bind == 0: back to the context's stream.  For callers that advance the two views of one context from two
 * host threads (doDual: the views are independent until the post-processing, LES/FastGCStereo.h:172-185).
 * What two threads with their own streams may call concurrently on ONE context: everything that works on caller-owned or
 * batch-owned device memory -- les_hip_batch_run / _propose / _wta / _expansion_graph / _solve_graphs / _apply_masks with
 * planes_on_device != 0 and distinct batches, les_hip_unary_one[_scratch] (per-thread / per-scratch buffers), les_hip_synchronize.
 * NOT safe under per-thread streams, because they stage through one context-owned buffer: les_hip_batch_run with planes_on_device
 * == 0, les_hip_unary_batch and les_hip_wta_update -- serialise those in the caller (host/HipCostVolumeEnergy.h holds a mutex). */
int les_hip_set_thread_stream(les_hip_ctx* ctx, void* hip_stream, int bind);
int les_hip_synchronize(les_hip_ctx* ctx);

/* replaces: CostVolumeEnergy::ComputeUnaryPotential (check != 0, LES/CostVolumeEnergy.h:176-183) and
 * ::ComputeUnaryPotentialWithoutCheck (check == 0, :55-174) for ONE call.  `costs` is the HOST
 * pointer of the element (filterRect.y, filterRect.x) of a row-major float map with `row_stride`
 * floats per row, i.e. the view proposalCost(filterRect) of LES/FastGCStereo.h:49; only
 * costs(targetRect - filterRect.tl()) is written.  Synchronous. */
int les_hip_unary_one(les_hip_ctx* ctx, int mode, const les_hip_rect* filterRect, const les_hip_rect* targetRect,
                      const les_hip_plane* plane, float* costs, int row_stride, int check);

/* The re-entrant form of the same operator: `scratch` is the caller-owned per-cell scratch of the reference (struct Reusable,
 * LES/StereoEnergy.h:616-623: one per OpenMP thread / cell visit, LES/FastGCStereo.h:40).  It owns a HIP stream, a device tile
 * and pinned staging for the target rect, and the prepared job tables of the last 16 (filterRect, targetRect) pairs, so that
 * calls with distinct scratch objects may run concurrently from distinct host threads on one context (the method is `const`
 * and is called from an OpenMP team in the reference, LES/FastGCStereo.h:30-49) and nothing is allocated once a rect pair has
 * been seen.  les_hip_unary_one() itself uses one hidden scratch per calling thread, released with the context. */
typedef struct les_hip_scratch les_hip_scratch;
int les_hip_scratch_create(les_hip_ctx* ctx, les_hip_scratch** out);
void les_hip_scratch_destroy(les_hip_scratch* scratch);
int les_hip_unary_one_scratch(les_hip_ctx* ctx, les_hip_scratch* scratch, int mode, const les_hip_rect* filterRect,
                              const les_hip_rect* targetRect, const les_hip_plane* plane, float* costs, int row_stride, int check);

/* The same for n independent calls (one proposal index of one disjoint set of cells:
 * LES/FastGCStereo.h:30-49 run in lock-step) writing into one H x W map.  cost_map: HOST H*W floats;
 * only the target rects are written.  Synchronous. */
int les_hip_unary_batch(les_hip_ctx* ctx, int mode, int n, const les_hip_rect* filterRects,
                        const les_hip_rect* targetRects, const les_hip_plane* planes, float* cost_map, int check);

/* Prepared form for the hot loop: geometry is uploaded once, then reused for every proposal.
 * out_slabs == 0: outputs go into one H x W map (element (y,x) of call i at y*W+x);
 * out_slabs == k > 0: call i writes its target rect into slab i / k of a [ceil(n / k)][H][W] array.  k = 1: every call
 * its own slab (whole-image aggregation of many hypothesis planes); k = cells of a disjoint set,
 * n = k * slots: several proposal slots of the set evaluated in ONE launch, slot s into map s (the calls of a slot are
 * consecutive) -- the launch then fills the GPU where a single lock-step of a coarse layer cannot. */
int les_hip_batch_create(les_hip_ctx* ctx, int n, const les_hip_rect* filterRects, const les_hip_rect* targetRects,
                         int out_slabs, les_hip_batch** out);
void les_hip_batch_destroy(les_hip_batch* b);
int les_hip_batch_num_jobs(const les_hip_batch* b);     /* workgroups one run launches                   */
/* Diagnostic: which kernel les_hip_batch_run launches for this batch and view: 1 = the fixed-point march kernel
 * (csrc/les_march.h; needs a finite volume with th_col - min <= 8 th_col, a guided-filter radius of 2 .. 10 and every target at least 2 x radius away from
 * filterRect borders that are not image borders -- the geometry of every LayerManager cell), 0 = the fp64 strip kernel
 * (csrc/les_kernels.h; any input), -1 = bad argument.  Both implement LES/CostVolumeEnergy.h:55-183. */
int les_hip_batch_kernel_kind(const les_hip_ctx* ctx, const les_hip_batch* b, int mode);
/* planes: n labels, HOST (planes_on_device == 0) or DEVICE memory; out: DEVICE memory.  Asynchronous
 * on the context's stream. */
int les_hip_batch_run(les_hip_ctx* ctx, const les_hip_batch* b, int mode, const les_hip_plane* planes,
                      int planes_on_device, float* out_dev, int check);

/* ---- hypothesis generation for the cells of a prepared batch (one proposal per cell per call) ----
 * replaces: IProposer::startIterations/getNextProposal as driven by LES/FastGCStereo.h:41-48, for
 * ExpansionProposer (LES/Proposer.h:62-79), RandomProposer (:120-152, `m` = outerIter + iter) and
 * RansacProposer (:163-311, MAX_SAM 500, conf 0.95, threshold 1.0), and the label part of
 * FastGCStereo::initCurrentFast (LES/FastGCStereo.h:105-109: createRandomLabel + fill of the unit region).
 * unitRects: the cells' unit regions (n rects, host).  labels_dev: H*W planes; rng_dev: n uint64
 * generator states (cv::RNG-compatible multiply-with-carry; one per cell, advanced in place);
 * planes_dev: n output labels.  Asynchronous. */
enum { LES_HIP_PROPOSE_EXPANSION = 0, LES_HIP_PROPOSE_RANDOM = 1, LES_HIP_PROPOSE_RANSAC = 2, LES_HIP_PROPOSE_INIT = 3 };
int les_hip_batch_set_units(les_hip_ctx* ctx, les_hip_batch* b, const les_hip_rect* unitRects);
int les_hip_batch_propose(les_hip_ctx* ctx, const les_hip_batch* b, int kind, int m, les_hip_plane* labels_dev,
                          uint64_t* rng_dev, les_hip_plane* planes_dev);
/* winner-take-all update over the batch's target rects with one device-resident plane per cell */
int les_hip_batch_wta(les_hip_ctx* ctx, const les_hip_batch* b, const les_hip_plane* planes_dev, float* cur_cost_dev,
                      const float* prop_cost_dev, les_hip_plane* labels_dev);

/* replaces: the winner-take-all update of the PatchMatch iterations, LES/FastGCStereo.h:56-60
 * (mask = cur > prop; cur <- prop, label <- plane under mask) for n shared regions, on DEVICE maps:
 * cur_cost/prop_cost H*W floats, labels H*W planes (row stride W).  Asynchronous. */
int les_hip_wta_update(les_hip_ctx* ctx, int n, const les_hip_rect* rects, const les_hip_plane* planes,
                       int planes_on_device, float* cur_cost_dev, const float* prop_cost_dev,
                       les_hip_plane* labels_dev);

/* ---- volume preparation on the device ("next" row N3 of the scope table) ----
 * replaces: fillOutOfView (LES/main.cpp:146-176) and convertVolumeL2R (LES/main.cpp:178-199), margin 0, on a
 * DEVICE float [D][H][W] volume (e.g. before handing it to les_hip_create with volumes_on_device).
 * mode 0 = left view, 1 = right view.  Asynchronous on hip_stream (NULL = default stream) of `device`. */
int les_hip_fill_out_of_view(float* vol_dev, int D, int H, int W, int mode, int device, void* hip_stream);
int les_hip_convert_volume_l2r(const float* src_dev, float* dst_dev, int D, int H, int W, int device, void* hip_stream);

/* replaces (on the device): the pairwise side of FastGCStereo::expansionMoveBK's graph construction
 * (LES/FastGCStereo.h:425-551) with StereoEnergy::initSmoothnessCoeff / computeSmoothnessTerm /
 * computeSmoothnessTermsExpansion (LES/StereoEnergy.h:131-163, 225-230, 398-453): for every cell i of the batch (its
 * target rect = the cell's shared region, proposal d_planes[i]) the ready-made capacities of the cell's 8-connected
 * grid graph.  d_payload: 5 floats per node {source-minus-sink terminal residual, arc capacities towards E, S, SW, SE}
 * (reverse capacities are 0), cell i at 5 * offsets[i] floats, nodes row-major over its rect; total
 * les_hip_batch_graph_nodes() nodes.  The values are bit-identical to the host construction
 * (localexpstereo_amd/host/ExpansionMove.h).  d_labels / d_cur / d_prop: current label map, current and proposal cost
 * maps (H x W, device).  flow0_host (n doubles, may be NULL): flow already routed by the t-links per cell. */
long long les_hip_batch_graph_nodes(const les_hip_batch* batch);
int les_hip_batch_graph_offsets(const les_hip_batch* batch, long long* offsets /* n */);
int les_hip_batch_expansion_graph(les_hip_ctx* ctx, const les_hip_batch* batch, int mode, const les_hip_plane* d_planes,
                                  const les_hip_plane* d_labels, const float* d_cur, const float* d_prop, float lambda, float th_smooth,
                                  float omega, float epsilon, float* d_payload, double* flow0_host);

/* replaces (on the device, for cells of at most LES_HIP_MAXFLOW_MAX_NODES nodes): the max-flow and the segment read-out of
 * FastGCStereo::expansionMoveBK -- graph.maxflow(); graph.what_segment(i) == SOURCE (LES/FastGCStereo.h:553-559) -- on the
 * payload of les_hip_batch_expansion_graph, for all cells of the batch, one workgroup per cell with the whole graph on chip
 * (synchronous push-relabel; the cut is the canonical one of the reference's solver: SINK side = nodes that can still reach the
 * sink).  Two kernels: when every cell of the batch has at most 2048 nodes, (w + 2) * (h + 2) <= 2304 and h <= 70 -- the finest
 * layer's cells do -- csrc/les_maxflow_cell.h (residuals in registers, two barriers per iteration); otherwise csrc/les_maxflow.h
 * (residuals in LDS).  Same cut from both up to nodes on exact ties that float rounding moves; les_hip_batch_graph_solver_kind says
 * which one a call would launch (0 = les_maxflow_cell.h, 1 / 2 = les_maxflow.h with 1024 / 512 threads, -1 = a cell above the limit);
 * LES_HIP_MAXFLOW_CELL_KERNEL=0 in the environment forces les_maxflow.h.  d_masks: one byte per graph node (255 = the node takes the proposal), the input of les_hip_batch_apply_masks;
 * d_status: n ints (0 = solved, 1 = iteration limit reached: cut that cell with the host solver instead); d_flows: n doubles or
 * NULL (flow through the n-links; add flow0 of les_hip_batch_expansion_graph for the value of the cut).
 * les_hip_batch_max_cell_nodes: the largest w * h of the batch's target rects (callers check it against the limit). */
#define LES_HIP_MAXFLOW_MAX_NODES 2304
long long les_hip_batch_max_cell_nodes(const les_hip_batch* batch);
int les_hip_batch_graph_solver_kind(const les_hip_batch* batch);
int les_hip_batch_solve_graphs(les_hip_ctx* ctx, const les_hip_batch* batch, const float* d_payload, unsigned char* d_masks, int* d_status,
                               double* d_flows);
/* The same with a running count: *d_unsolved_total (a device int the caller zeroed) += 1 for every cell that hits the iteration limit.  A caller
 * that enqueues the lock-steps of a whole disjoint set without synchronising reads this ONE word at the end of the set instead of n status words
 * per lock-step (and repeats the set the slow way in the -- so far unobserved -- case that it is not zero). */
int les_hip_batch_solve_graphs_counted(les_hip_ctx* ctx, const les_hip_batch* batch, const float* d_payload, unsigned char* d_masks, int* d_status,
                                       double* d_flows, int* d_unsolved_total);

/* The same replacement (LES/FastGCStereo.h:553-559 on the graph of :411-551) for cells of ANY size -- the coarse layers, whose cells
 * (129 x 129 ... 404 x 387 nodes at the Adirondack shape) do not fit a workgroup's LDS: the graphs stay in device memory, a cell is
 * cut into tiles of <= 1920 nodes, one workgroup per tile, and the lock-step is a sequence of launches in which every cell moves
 * through exact relabelling / discharge phases on its own (region-parallel push-relabel, csrc/les_maxflow_tiled.h).  Same payload,
 * masks, status (0 = solved, non-zero = launch limit reached: cut that cell with the host solver) and flows as
 * les_hip_batch_solve_graphs; same cut (SINK side = the nodes that can still reach the sink).  Bit-reproducible from run to run (heights relabelled from a snapshot, flow
 * values summed as 64-bit integers; round 6).
 * d_workspace: caller-owned device scratch of at least les_hip_batch_tiled_workspace_bytes(batch) bytes, 256-byte aligned, not
 * shared between host threads that call concurrently (109 bytes per graph node + 64 per cell).  The call synchronises the calling
 * thread's stream (between groups of launches it looks at a host-mapped "cells done" word the kernel adds to: no copy); launches_out
 * (or NULL): launches enqueued; unsolved_out (or NULL): cells that hit the launch limit (their d_status is non-zero) -- callers that
 * only need "all solved?" read it instead of copying d_status back. */
long long les_hip_batch_tiled_workspace_bytes(const les_hip_batch* batch);
int les_hip_batch_solve_graphs_tiled(les_hip_ctx* ctx, const les_hip_batch* batch, const float* d_payload, unsigned char* d_masks, int* d_status,
                                     double* d_flows, void* d_workspace, long long workspace_bytes, int* launches_out, int* unsolved_out);
/* Hand-over (round 6): a lock-step lasts as long as its slowest cell, and the launches are at their worst on the tail of a hard cell (a few
 * hundred small excesses, one cell's tiles on a 256-CU chip).  After 28 launches, once at most 8 cells of at most 140 000 nodes in total are
 * still open AND a whole group of 16 launches went by without a cell finishing, the open cells' residual graphs (8 residual capacities + the
 * excess per node, 36 bytes) go to host-mapped memory and the host cores -- idle during device cuts -- finish each with FIFO push-relabel from
 * the remaining excess nodes (host/ResidualCut.h; one thread per cell).  Larger open sets (the coarsest layer's 150 000-node cells) only after
 * 220 launches, everything that is still open after 300.  The residual graph of a feasible preflow has the minimum cuts of the graph it came
 * from, so masks, status and flows mean what they mean without it.  LES_HIP_MAXFLOW_HANDOVER=0 switches it off.
 * The _stats form reports what happened (the plain form = the _stats form without the report). */
typedef struct les_hip_tiled_stats {
    int launches;            /* launches of les_maxflow_tiled_kernel enqueued */
    int unsolved;            /* cells that hit the launch limit (d_status non-zero) */
    int handed_cells;        /* cells finished by the host cores from their residual graphs */
    long long handed_nodes;  /* ... and their graph nodes (36 bytes each crossed PCIe, 1 byte came back) */
    double host_ms;          /* wall-clock of the host cores' part */
} les_hip_tiled_stats;
int les_hip_batch_solve_graphs_tiled_stats(les_hip_ctx* ctx, const les_hip_batch* batch, const float* d_payload, unsigned char* d_masks, int* d_status,
                                           double* d_flows, void* d_workspace, long long workspace_bytes, les_hip_tiled_stats* stats);

/* replaces: the mask updates after a graph cut -- subProposalCost.copyTo(subCurrentCost, updateMask);
 * subCurrentLabeling.setTo(label, updateMask) (LES/FastGCStereo.h:61-62) -- for all cells of the batch.  d_masks: one
 * byte per graph node in the payload order of les_hip_batch_expansion_graph (non-zero = the node takes the proposal). */
int les_hip_batch_apply_masks(les_hip_ctx* ctx, const les_hip_batch* batch, const les_hip_plane* d_planes, const unsigned char* d_masks,
                              float* d_cur, const float* d_prop, les_hip_plane* d_labels);

/* replaces: PMStereoBase::doConsistencyCheck (LES/PMStereoBase.h:111-144) -- left-right check of the disparities of two
 * device label maps (H x W planes each): fail = 255 where |d_other(x -/+ d) - d| > threshold, 128 where the pixel maps
 * outside the other view, 0 otherwise.  d_failL / d_failR: H x W bytes on the device. */
int les_hip_consistency_check(les_hip_ctx* ctx, const les_hip_plane* d_labelsL, const les_hip_plane* d_labelsR, float threshold,
                              unsigned char* d_failL, unsigned char* d_failR);

/* replaces: PMStereoBase::postProcess (LES/PMStereoBase.h:146-256), called by FastGCStereo::run for two-view runs
 * (LES/FastGCStereo.h:199-203, threshold 1.5): consistency check, horizontal fill of the failed pixels from the nearest
 * consistent neighbours (smaller disparity wins), then the colour-weighted median of the labels over the
 * (2 windR + 1)^2 window with weights exp(-|dI|_1 / omega) (StereoEnergy::computePatchWeight, LES/StereoEnergy.h:251-257).
 * Both device label maps are updated in place.  Needs both views' images in the context; windR <= 31. */
int les_hip_post_process(les_hip_ctx* ctx, les_hip_plane* d_labelsL, les_hip_plane* d_labelsR, float threshold, float omega);

/* ---- multi-GPU: publishing the tiles a rank updated (not in the reference, which is single-process; SURVEY 8(e): the cells of a
 * disjoint set -- LES/LayerManager.h:168-172 -- are split over the ranks, replicas of the label / cost maps are kept coherent by one
 * all-gather per set of the label (16 B/px) and cost (4 B/px) tiles of the shared regions, LES/FastGCStereo.h:22-72).
 * An exchange object is the plan for one (layer, set): rects = the target rects of ALL ranks' cells in rank order, first[r] .. first[r+1]
 * those of rank r (first has world + 1 entries).  Slot of a rank in the gathered buffer = les_hip_exchange_slot_floats floats:
 * [float4 labels x lmax][float costs x lmax] in rect order, row-major inside a rect.
 *   les_hip_exchange_pack    this rank's tiles (device maps) -> its slot (device buffer of slot_floats floats)
 *   les_hip_exchange_unpack  the gathered world x slot_floats buffer -> the other ranks' tiles into this rank's device maps
 *   les_hip_exchange_tiles   pack -> ncclAllGather on the ncclComm_t the HOST created (RCCL, resolved at run time with dlopen:
 *                            single-GPU users need no librccl) -> unpack, all enqueued on the calling thread's stream of the
 *                            context, no host synchronisation.  world == 1 with a null communicator is a no-op.
 * pack / unpack are exposed so that a host with another transport (torch.distributed in the tests) can move the slot itself. */
typedef struct les_hip_exchange les_hip_exchange;
int les_hip_exchange_create(les_hip_ctx* ctx, int rank, int world, int n_rects, const les_hip_rect* rects, const int* first, les_hip_exchange** out);
void les_hip_exchange_destroy(les_hip_exchange* x);
long long les_hip_exchange_slot_floats(const les_hip_exchange* x);
int les_hip_exchange_pack(les_hip_ctx* ctx, const les_hip_exchange* x, const les_hip_plane* d_labels, const float* d_cost, float* d_slot);
int les_hip_exchange_unpack(les_hip_ctx* ctx, const les_hip_exchange* x, const float* d_gathered, les_hip_plane* d_labels, float* d_cost);
int les_hip_exchange_tiles(les_hip_ctx* ctx, les_hip_exchange* x, void* nccl_comm, les_hip_plane* d_labels, float* d_cost);

/* diagnostics: dword-per-lane streaming copy of n floats (device pointers), the known-byte-count pattern used to
 * calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for this library's access width (profiles/). */
int les_hip_calib_copy(const float* d_src, float* d_dst, size_t n, int device, void* stream);
/* the same with 16 bytes per lane (n a multiple of 4, 16-byte aligned pointers): the streaming-copy ceiling that bench.py measures
 * in its own run and reports as roofline.peak_achievable. */
int les_hip_calib_copy_wide(const float* d_src, float* d_dst, size_t n, int device, void* stream);

/* Device memory helpers for callers without a HIP toolchain (host C++ adapter, ctypes). */
int les_hip_malloc(les_hip_ctx* ctx, void** dev_ptr, size_t bytes);
int les_hip_free(les_hip_ctx* ctx, void* dev_ptr);
int les_hip_memcpy_h2d(les_hip_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int les_hip_memcpy_d2h(les_hip_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int les_hip_memset(les_hip_ctx* ctx, void* dst_dev, int value, size_t bytes);

/* Test/diagnostic: guide statistics of view `mode` as the kernels consume them:
 * out[(y*W+x)*12 + k*4 + {0,1,2,3}] = {mean_I_k - 1/2, inv[k][0], inv[k][1], inv[k][2]} (float32). */
int les_hip_get_stats(les_hip_ctx* ctx, int mode, float* out_host);
/* Strip geometry the build was compiled with for radius R (0 if unsupported): output columns per
 * workgroup. */
int les_hip_strip_width(int R);
/* Bytes of the TILED copy of view `mode`'s cost volume ([H][ceil(W/8)][D][8] floats: 8 columns x all slices contiguous), which the
 * march kernel's gather reads for planes that are steep along x (their two taps per pixel then lie within a few contiguous 32-byte
 * pieces instead of one 128-byte line per slice of [D][H][W]).  0 when the context holds none: LES_HIP_TILED=0, the image-based energy,
 * a view on the strip kernel, a copy of 2^30 floats or more, or an allocation that failed -- every plane then gathers from [D][H][W]
 * (identical costs).  A second resident copy of the volume is the price: this many bytes per view.
 * (no reference counterpart; LES/CostVolumeEnergy.h:73-96 reads its cv::Mat volume in place) */
size_t les_hip_tiled_volume_bytes(les_hip_ctx* ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif
