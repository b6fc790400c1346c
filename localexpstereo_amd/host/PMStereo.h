// PMStereo.h -- host loop of the PatchMatch-style iterations of the local expansion framework
// (reference: PMStereoBase state LES/PMStereoBase.h:14-61, FastGCStereo::addLayer / initCurrentFast /
// localExpansionMovesForLayer_CPU / run with doGC == false, LES/FastGCStereo.h:22-72,88-169).
//
// Two equivalent drivers over one energy operator:
//   run()        the reference's loop shape: OpenMP over the cells of a disjoint set, each thread calling
//                stereoEnergy->ComputeUnaryPotential for its cell (the drop-in path: any StereoEnergy works)
//   runDevice()  the same iteration with label map, cost maps, generator states and proposers resident on
//                the GPU: per proposal index one lock-step  propose -> unary -> winner-take-all  for all
//                cells of the set, no host synchronisation inside an iteration.
// With doGC == true (the reference's main iterations) the winner-take-all update is replaced by a local
// alpha-expansion per cell (ExpansionMove.h over MaxFlow.h; "next" rows N1/N2).  In runDevice() the proposals and
// the unary costs still come from the GPU; the graph cuts of the cells of a set run on the host cores (OpenMP) and
// the updated label map goes back to the device after every lock-step.
#pragma once
#include <algorithm>
#include <stdexcept>

#include <omp.h>

#include <chrono>
#include <functional>
#include <memory>

#include "ExpansionMove.h"
#include "HipCostVolumeEnergy.h"
#include "LayerManager.h"
#include "Proposer.h"

namespace les_host {

struct ProposerSpec { int kind; int K; };   // kind: LES_HIP_PROPOSE_EXPANSION / _RANDOM / _RANSAC

class PMStereo {
public:
    PMStereo(int width, int height, Parameters params, float maxDisparity, float minDisparity = 0)
        : width(width), height(height), params(params), MAX_DISPARITY(maxDisparity), MIN_DISPARITY(minDisparity),
          layermng(width, height, params.windR)
    {
        for (int m = 0; m < 2; m++) {
            currentLabeling_[m] = LabelMap(height, width);
            currentCost_[m] = CostMap(height, width, 0.f);
        }
    }

    // replaces PMStereoBase::setStereoEnergyCPU (LES/PMStereoBase.h:58-61)
    void setStereoEnergy(std::unique_ptr<StereoEnergy> energy) { stereoEnergy = std::move(energy); }
    const StereoEnergy& getEnergyInstance() const { return *stereoEnergy; }

    // replaces FastGCStereo::addLayer (LES/FastGCStereo.h:88-92)
    void addLayer(int unitRegionSize, std::vector<ProposerSpec> proposers)
    {
        layermng.addLayer(unitRegionSize);
        layerProposers.push_back(std::move(proposers));
        // one generator per cell: proposals of a cell are a pure function of (labels, rect, state)
        const size_t n = layermng.layers.back().unitRegions.size();
        std::vector<uint64_t> st(n);
        for (size_t i = 0; i < n; i++) st[i] = splitmix(seed_ + 0x9E3779B97F4A7C15ULL * (layermng.layers.size() * 1000003ULL + i));
        rngStates.push_back(std::move(st));
    }
    void setSeed(uint64_t s) { seed_ = s; }

    // ---------------------------------------------------------------------------------------------
    // Drop-in driver: LES/FastGCStereo.h:94-115 (init) and :22-72 (moves, doGC == false)
    // ---------------------------------------------------------------------------------------------
    void initCurrentFast(int mode)
    {
        CostMap& cost = currentCost_[mode];
        LabelMap& lab = currentLabeling_[mode];
        std::fill(cost.data.begin(), cost.data.end(), 0.f);
        const auto& layer = layermng.layers[0];
        const Rect image(0, 0, width, height);
        const int R = params.windR;
#pragma omp parallel for schedule(dynamic, 8)
        for (int j = 0; j < (int)layer.unitRegions.size(); j++) {
            const Rect unit = layer.unitRegions[j];
            RNG rng(rngStates[0][j]);
            const int n = rng.uniform(0, unit.height * unit.width);
            const Point pnt{unit.x + n % unit.width, unit.y + n / unit.width};
            const Plane label = stereoEnergy->createRandomLabel(pnt, rng);
            for (int y = 0; y < unit.height; y++)
                for (int x = 0; x < unit.width; x++) lab.at(unit.y + y, unit.x + x) = label;
            const Rect filterRegion = Rect(unit.x - R, unit.y - R, unit.width + 2 * R, unit.height + 2 * R) & image;
            StereoEnergy::Reusable tmp;
            stereoEnergy->ComputeUnaryPotential(filterRegion, unit, cost.view(filterRegion), width, label, tmp, mode);
            rngStates[0][j] = rng.state;
        }
    }

    // fuse `label` into the current solution over `sharedRegion` given its unary costs in proposalCost
    // (LES/FastGCStereo.h:52-63)
    void fuseProposal(const Plane& label, const Rect& sharedRegion, CostView proposalCost, int mode, bool doGC)
    {
        CostMap& currentCost = currentCost_[mode];
        LabelMap& currentLabeling = currentLabeling_[mode];
        if (!doGC) {
            for (int y = sharedRegion.y; y < sharedRegion.y + sharedRegion.height; y++)
                for (int x = sharedRegion.x; x < sharedRegion.x + sharedRegion.width; x++)
                    if (currentCost.at(y, x) > proposalCost.at(y, x)) {
                        currentCost.at(y, x) = proposalCost.at(y, x);
                        currentLabeling.at(y, x) = label;
                    }
            return;
        }
        std::vector<uint8_t> mask;
        const double flow = expansionMove(*stereoEnergy, currentLabeling, currentCost, proposalCost, label, sharedRegion, mask, mode);
        if (checkFlowEnergy) {
            const double e = fusedEnergy(*stereoEnergy, currentLabeling, currentCost, proposalCost, label, sharedRegion, mask, mode);
            const double gap = std::fabs(flow - e) / std::max(1.0, std::fabs(e));
#pragma omp critical(les_gap)
            { maxFlowEnergyGap = std::max(maxFlowEnergyGap, gap); numMoves++; }
        }
        applyMask(label, sharedRegion, proposalCost, mask, mode);
    }
    // subProposalCost.copyTo(subCurrentCost, updateMask); subCurrentLabeling.setTo(label, updateMask)  (LES/FastGCStereo.h:61-62)
    void applyMask(const Plane& label, const Rect& sharedRegion, CostView proposalCost, const std::vector<uint8_t>& mask, int mode)
    {
        CostMap& currentCost = currentCost_[mode];
        LabelMap& currentLabeling = currentLabeling_[mode];
        for (int y = 0; y < sharedRegion.height; y++)
            for (int x = 0; x < sharedRegion.width; x++)
                if (mask[(size_t)y * sharedRegion.width + x]) {
                    currentCost.at(sharedRegion.y + y, sharedRegion.x + x) = proposalCost.at(sharedRegion.y + y, sharedRegion.x + x);
                    currentLabeling.at(sharedRegion.y + y, sharedRegion.x + x) = label;
                }
    }

    void localExpansionMovesForLayer(int li, int mode, int iteration, bool doGC = false)
    {
        const auto& layer = layermng.layers[li];
        CostMap& currentCost = currentCost_[mode];
        LabelMap& currentLabeling = currentLabeling_[mode];
        CostMap proposalCost(height, width);
        for (const auto& set : layer.disjointRegionSets) {
#pragma omp parallel for schedule(dynamic, 4)
            for (int n = 0; n < (int)set.size(); n++) {
                const int r = set[n];
                const Rect& sharedRegion = layer.sharedRegions[r];
                const Rect& unitRegion = layer.unitRegions[r];
                RNG rng(rngStates[li][r]);
                StereoEnergy::Reusable reusable;
                for (const ProposerSpec& spec : layerProposers[li]) {
                    std::unique_ptr<IProposer> prop(makeHostProposer(spec));
                    prop->startIterations(currentLabeling, unitRegion, iteration, &rng);
                    while (prop->isContinued()) {
                        const Plane label = prop->getNextProposal();
                        stereoEnergy->ComputeUnaryPotential(layer.filterRegions[r], sharedRegion, proposalCost.view(layer.filterRegions[r]),
                                                            width, label, reusable, mode);
                        fuseProposal(label, sharedRegion, proposalCost, mode, doGC);
                    }
                }
                rngStates[li][r] = rng.state;
            }
        }
    }

    // FastGCStereo::run (LES/FastGCStereo.h:133-199): pmInit winner-take-all iterations, then maxIteration
    // graph-cut iterations (the iteration counter restarts, so the random search widths do too)
    void run(int pmInit, const std::vector<int>& viewModes = {0}, int maxIteration = 0)
    {
        for (int mode : viewModes) initCurrentFast(mode);
        for (int iteration = 0; iteration < pmInit; iteration++)
            for (int mode : viewModes)
                for (int li = 0; li < (int)layermng.layers.size(); li++) localExpansionMovesForLayer(li, mode, iteration, false);
        for (int iteration = 0; iteration < maxIteration; iteration++)
            for (int mode : viewModes)
                for (int li = 0; li < (int)layermng.layers.size(); li++) localExpansionMovesForLayer(li, mode, iteration, true);
    }

    // ---------------------------------------------------------------------------------------------
    // Device-resident driver (needs a HipCostVolumeEnergy)
    // ---------------------------------------------------------------------------------------------
    bool runDevice(int pmInit, const std::vector<int>& viewModes = {0}, double* seconds = nullptr, int maxIteration = 0)
    {
        auto* hip = dynamic_cast<HipCostVolumeEnergy*>(stereoEnergy.get());
        if (!hip) return false;
        les_hip_ctx* ctx = hip->handle();
        const size_t P = (size_t)width * height;
        bool ok = true;
        auto chk = [&](int rc) { if (rc != LES_HIP_OK) { if (ok) fprintf(stderr, "PMStereo::runDevice: %s\n", les_hip_last_error()); ok = false; } };
        // prepared geometry: one batch per (layer, disjoint set) + the init batch (unit +- windR -> unit)
        // Several ranks (one process per GPU): the cells of every disjoint set are dealt out in contiguous bands, rank r owns
        // cells [n r / world, n (r + 1) / world) of the set; volume, statistics and maps are replicated, and after a set's lock-steps the tiles
        // (target rects) of all ranks are published with one all-gather (les_hip_exchange_*).  Only with device-built graphs: the solution
        // then lives on the device and the host keeps no state that would have to follow the exchange.
        if (world > 1 && maxIteration > 0 && (!deviceGraph || checkFlowEnergy)) { fprintf(stderr, "PMStereo::runDevice: several ranks need deviceGraph and no self-check\n"); return false; }
        if (world > 1 && viewModes.size() > 1) { fprintf(stderr, "PMStereo::runDevice: with several ranks give each view its own group of ranks (one view per call)\n"); return false; }
        struct SetBatch { les_hip_batch* b = nullptr; uint64_t* rng = nullptr; les_hip_plane* planes = nullptr; int n = 0; std::vector<int> cells; les_hip_exchange* x = nullptr; };
        std::vector<std::vector<SetBatch>> batches(layermng.layers.size());
        // all = the indices of the set's cells into the layer's rect arrays; tr / fr / un / st are indexed by cell
        auto make = [&](const std::vector<int>& all, const std::vector<Rect>& frs, const std::vector<Rect>& trs, const std::vector<Rect>& uns, const std::vector<uint64_t>& sts) {
            SetBatch sb;
            const int n = (int)all.size();
            std::vector<int> first((size_t)world + 1);
            for (int r = 0; r <= world; r++) first[(size_t)r] = (int)((long long)n * r / world);
            std::vector<Rect> fr, tr, un;
            std::vector<uint64_t> st;
            for (int i = first[(size_t)rank]; i < first[(size_t)rank + 1]; i++) {
                const int c = all[(size_t)i];
                sb.cells.push_back(c); fr.push_back(frs[(size_t)c]); tr.push_back(trs[(size_t)c]); un.push_back(uns[(size_t)c]); st.push_back(sts[(size_t)c]);
            }
            sb.n = (int)sb.cells.size();
            chk(les_hip_batch_create(ctx, sb.n, reinterpret_cast<const les_hip_rect*>(fr.data()), reinterpret_cast<const les_hip_rect*>(tr.data()), 0, &sb.b));
            if (sb.b && sb.n > 0) chk(les_hip_batch_set_units(ctx, sb.b, reinterpret_cast<const les_hip_rect*>(un.data())));
            chk(les_hip_malloc(ctx, (void**)&sb.rng, sizeof(uint64_t) * std::max(1, sb.n)));
            chk(les_hip_malloc(ctx, (void**)&sb.planes, sizeof(les_hip_plane) * std::max(1, sb.n)));
            if (ok && sb.n > 0) chk(les_hip_memcpy_h2d(ctx, sb.rng, st.data(), sizeof(uint64_t) * sb.n));
            if (world > 1) {
                std::vector<Rect> every;
                for (int c : all) every.push_back(trs[(size_t)c]);
                chk(les_hip_exchange_create(ctx, rank, world, n, reinterpret_cast<const les_hip_rect*>(every.data()), first.data(), &sb.x));
            }
            return sb;
        };
        // publish the tiles of the set this rank updated: on the host's RCCL communicator, or through gatherFn (another transport / tests)
        float *d_xsend = nullptr, *d_xrecv = nullptr;
        long long xcap = 0;
        auto exchange = [&](SetBatch& sb, les_hip_plane* d_lab, float* d_cost) {
            if (world == 1 || !sb.x || !ok) return;
            if (!gatherFn) { chk(les_hip_exchange_tiles(ctx, sb.x, ncclComm, d_lab, d_cost)); return; }
            const long long slot = les_hip_exchange_slot_floats(sb.x);
            if (slot == 0) return;
            if (slot > xcap) {
                if (d_xsend) les_hip_free(ctx, d_xsend);
                if (d_xrecv) les_hip_free(ctx, d_xrecv);
                d_xsend = d_xrecv = nullptr;
                xcap = slot;
                chk(les_hip_malloc(ctx, (void**)&d_xsend, sizeof(float) * (size_t)slot));
                chk(les_hip_malloc(ctx, (void**)&d_xrecv, sizeof(float) * (size_t)slot * (size_t)world));
            }
            chk(les_hip_exchange_pack(ctx, sb.x, d_lab, d_cost, d_xsend));
            chk(les_hip_synchronize(ctx));
            if (ok) gatherFn(rank, d_xsend, d_xrecv, slot);
            chk(les_hip_exchange_unpack(ctx, sb.x, d_xrecv, d_lab, d_cost));
        };
        const Rect image(0, 0, width, height);
        for (size_t li = 0; li < layermng.layers.size() && ok; li++) {
            const auto& L = layermng.layers[li];
            for (const auto& set : L.disjointRegionSets) batches[li].push_back(make(set, L.filterRegions, L.sharedRegions, L.unitRegions, rngStates[li]));
        }
        SetBatch init;
        {
            const auto& L = layermng.layers[0];
            std::vector<Rect> fr;
            std::vector<int> all(L.unitRegions.size());
            const int R = params.windR;
            for (size_t i = 0; i < L.unitRegions.size(); i++) { const Rect& u = L.unitRegions[i]; all[i] = (int)i; fr.push_back(Rect(u.x - R, u.y - R, u.width + 2 * R, u.height + 2 * R) & image); }
            init = make(all, fr, L.unitRegions, L.unitRegions, rngStates[0]);
        }
        les_hip_plane* d_labels = nullptr;
        float *d_cur = nullptr, *d_prop = nullptr;
        chk(les_hip_malloc(ctx, (void**)&d_labels, P * sizeof(les_hip_plane)));
        chk(les_hip_malloc(ctx, (void**)&d_cur, P * sizeof(float)));
        chk(les_hip_malloc(ctx, (void**)&d_prop, P * sizeof(float)));
        const auto t0 = std::chrono::steady_clock::now();
        for (int mode : viewModes) {
            if (!ok) break;
            // initCurrentFast on the device: random label per layer-0 cell, cost of its unit region
            chk(les_hip_memset(ctx, d_labels, 0, P * sizeof(les_hip_plane)));
            chk(les_hip_memset(ctx, d_cur, 0, P * sizeof(float)));
            if (init.n > 0) {
                chk(les_hip_batch_propose(ctx, init.b, LES_HIP_PROPOSE_INIT, 0, d_labels, init.rng, init.planes));
                chk(les_hip_batch_run(ctx, init.b, mode, init.planes, 1, d_cur, 1));
            }
            exchange(init, d_labels, d_cur);
            for (int iteration = 0; iteration < pmInit && ok; iteration++)
                for (size_t li = 0; li < batches.size(); li++)
                    for (SetBatch& sb : batches[li]) {
                        for (const ProposerSpec& spec : layerProposers[li])
                            for (int it = 0; it < spec.K && sb.n > 0; it++) {
                                const int m = iteration + it;
                                if (spec.kind == LES_HIP_PROPOSE_RANDOM && randomWidth(m) < 0.1) break;      // LES/Proposer.h:149-152
                                chk(les_hip_batch_propose(ctx, sb.b, spec.kind, m, d_labels, sb.rng, sb.planes));
                                chk(les_hip_batch_run(ctx, sb.b, mode, sb.planes, 1, d_prop, 1));
                                chk(les_hip_batch_wta(ctx, sb.b, sb.planes, d_cur, d_prop, d_labels));
                            }
                        exchange(sb, d_labels, d_cur);
                    }
            chk(les_hip_synchronize(ctx));
            if (ok) {
                chk(les_hip_memcpy_d2h(ctx, currentLabeling_[mode].data.data(), d_labels, P * sizeof(les_hip_plane)));
                chk(les_hip_memcpy_d2h(ctx, currentCost_[mode].data.data(), d_cur, P * sizeof(float)));
            }
            // graph-cut iterations: GPU proposes and evaluates the unary costs of all cells of a set, the host cores
            // cut the cells' graphs, the fused label map returns to the device for the next proposals
            CostMap proposalCost(height, width);
            std::vector<les_hip_plane> hplanes;
            std::vector<float> payload;
            std::vector<uint8_t> masks;
            std::vector<long long> goff;
            float* d_payload = nullptr;
            unsigned char* d_masks = nullptr;
            int* d_status = nullptr;
            std::vector<int> status;
            long long payload_cap = 0, status_cap = 0, tiled_cap = 0;
            void* d_tiled = nullptr;                 // scratch of the tiled device max-flow (this thread's: one per runDevice call)
            // deviceGraph: the solution stays on the GPU; the host receives the ready-made graphs and returns one mask byte
            // per node.  Otherwise (or with the self-check on): host-resident solution and host graph construction.
            const bool devGraph = deviceGraph && !checkFlowEnergy;
            for (int iteration = 0; iteration < maxIteration && ok; iteration++)
                for (size_t li = 0; li < batches.size(); li++)
                    for (SetBatch& sb : batches[li]) {
                        for (const ProposerSpec& spec : layerProposers[li])
                            for (int it = 0; it < spec.K && ok && sb.n > 0; it++) {
                                const int m = iteration + it;
                                if (spec.kind == LES_HIP_PROPOSE_RANDOM && randomWidth(m) < 0.1) break;
                                const auto tA = std::chrono::steady_clock::now();
                                chk(les_hip_batch_propose(ctx, sb.b, spec.kind, m, d_labels, sb.rng, sb.planes));
                                chk(les_hip_batch_run(ctx, sb.b, mode, sb.planes, 1, d_prop, 1));
                                const auto& L = layermng.layers[li];
                                // team of the host cuts: one thread per cell, at most 24, never more than the CPUs the process is granted
                                const int nthreads = std::max(1, std::min(sb.n, hostThreads > 0 ? hostThreads : std::min({24, omp_get_max_threads(), std::max(1, cpuBudget())})));
                                std::chrono::steady_clock::time_point tB, tC;
                                if (devGraph) {
                                    const long long nodes = les_hip_batch_graph_nodes(sb.b);
                                    if (nodes > payload_cap) {
                                        if (d_payload) les_hip_free(ctx, d_payload);
                                        if (d_masks) les_hip_free(ctx, d_masks);
                                        d_payload = nullptr; d_masks = nullptr;
                                        payload_cap = nodes;
                                        chk(les_hip_malloc(ctx, (void**)&d_payload, (size_t)nodes * 5 * sizeof(float)));
                                        chk(les_hip_malloc(ctx, (void**)&d_masks, (size_t)nodes));
                                    }
                                    chk(les_hip_batch_expansion_graph(ctx, sb.b, mode, sb.planes, d_labels, d_cur, d_prop, params.lambda, params.th_smooth,
                                                                      params.omega, params.epsilon, d_payload, nullptr));
                                    // cells that fit a workgroup's LDS (the finest layer) are also CUT on the device: neither their graphs
                                    // nor their masks cross PCIe (LES/FastGCStereo.h:553-559 -> les_hip_batch_solve_graphs)
                                    // ... and since round 5 the larger cells (the coarse layers) as well: their graphs stay in device memory and
                                    // are cut by the tiled solver, one workgroup per tile (les_hip_batch_solve_graphs_tiled)
                                    bool cutOnDevice = false;
                                    const bool fitsLds = les_hip_batch_max_cell_nodes(sb.b) <= LES_HIP_MAXFLOW_MAX_NODES;
                                    if (deviceCuts && ok && (fitsLds || deviceCutsCoarse)) {
                                        if (sb.n > status_cap) {
                                            if (d_status) les_hip_free(ctx, d_status);
                                            d_status = nullptr;
                                            status_cap = sb.n;
                                            chk(les_hip_malloc(ctx, (void**)&d_status, sizeof(int) * (size_t)sb.n));
                                        }
                                        status.assign((size_t)sb.n, 1);
                                        if (fitsLds) {
                                            chk(les_hip_batch_solve_graphs(ctx, sb.b, d_payload, d_masks, d_status, nullptr));
                                        } else {
                                            const long long need = les_hip_batch_tiled_workspace_bytes(sb.b);
                                            if (need > tiled_cap) {
                                                if (d_tiled) les_hip_free(ctx, d_tiled);
                                                d_tiled = nullptr;
                                                tiled_cap = need;
                                                chk(les_hip_malloc(ctx, &d_tiled, (size_t)need));
                                            }
                                            int launches = 0, unsolved = 0;
                                            if (ok) chk(les_hip_batch_solve_graphs_tiled(ctx, sb.b, d_payload, d_masks, d_status, nullptr, d_tiled, tiled_cap, &launches, &unsolved));
                                            gcTiledLaunches += launches;
                                            status.assign((size_t)sb.n, unsolved ? 1 : 0);       // (reported through a host-mapped word: no copy of d_status)
                                        }
                                        if (ok && fitsLds) chk(les_hip_memcpy_d2h(ctx, status.data(), d_status, sizeof(int) * (size_t)sb.n));
                                        cutOnDevice = ok && std::all_of(status.begin(), status.end(), [](int v) { return v == 0; });
                                    }
                                    if (cutOnDevice) {
                                        tB = tC = std::chrono::steady_clock::now();
                                        chk(les_hip_batch_apply_masks(ctx, sb.b, sb.planes, d_masks, d_cur, d_prop, d_labels));
                                        gcCellsCutOnDevice += sb.n;
                                    } else {
                                    payload.resize((size_t)nodes * 5);
                                    masks.resize((size_t)nodes);
                                    goff.resize((size_t)sb.n);
                                    chk(les_hip_batch_graph_offsets(sb.b, goff.data()));
                                    if (ok) chk(les_hip_memcpy_d2h(ctx, payload.data(), d_payload, (size_t)nodes * 5 * sizeof(float)));
                                    if (!ok) break;
                                    tB = std::chrono::steady_clock::now();
                                    // large cells (the coarser layers) split their max-flow over row bands on helper threads, like
                                    // les_gc_solve_prebuilt does for non-C++ hosts (ExpansionMove.h: bandsFor / tuneBandSpin)
                                    tuneBandSpin(sb.n, nthreads, [&](int n) { return L.sharedRegions[sb.cells[n]]; });
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
                                    for (int n = 0; n < sb.n; n++) {
                                        const Rect& region = L.sharedRegions[sb.cells[n]];
                                        expansionMovePrebuilt(payload.data() + 5 * goff[n], 0.0, region, masks.data() + goff[n], bandsFor(region, sb.n));
                                    }
                                    tC = std::chrono::steady_clock::now();
                                    chk(les_hip_memcpy_h2d(ctx, d_masks, masks.data(), (size_t)nodes));
                                    chk(les_hip_batch_apply_masks(ctx, sb.b, sb.planes, d_masks, d_cur, d_prop, d_labels));
                                    }
                                } else {
                                    hplanes.resize(sb.n);
                                    chk(les_hip_memcpy_d2h(ctx, hplanes.data(), sb.planes, sizeof(les_hip_plane) * sb.n));
                                    chk(les_hip_memcpy_d2h(ctx, proposalCost.data.data(), d_prop, P * sizeof(float)));
                                    if (!ok) break;
                                    tB = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
                                    for (int n = 0; n < sb.n; n++) {
                                        const les_hip_plane& hp = hplanes[n];
                                        fuseProposal(Plane(hp.a, hp.b, hp.c, hp.v), L.sharedRegions[sb.cells[n]], proposalCost, mode, true);
                                    }
                                    tC = std::chrono::steady_clock::now();
                                    chk(les_hip_memcpy_h2d(ctx, d_labels, currentLabeling_[mode].data.data(), P * sizeof(les_hip_plane)));
                                }
                                const auto tD = std::chrono::steady_clock::now();
                                gcSeconds[0] += std::chrono::duration<double>(tB - tA).count();
                                gcSeconds[1] += std::chrono::duration<double>(tC - tB).count();
                                gcSeconds[2] += std::chrono::duration<double>(tD - tC).count();
                                gcLockSteps++;
                            }
                        exchange(sb, d_labels, d_cur);
                    }
            if (devGraph && maxIteration > 0 && ok) {
                chk(les_hip_synchronize(ctx));
                chk(les_hip_memcpy_d2h(ctx, currentLabeling_[mode].data.data(), d_labels, P * sizeof(les_hip_plane)));
                chk(les_hip_memcpy_d2h(ctx, currentCost_[mode].data.data(), d_cur, P * sizeof(float)));
            }
            if (d_payload) les_hip_free(ctx, d_payload);
            if (d_masks) les_hip_free(ctx, d_masks);
            if (d_status) les_hip_free(ctx, d_status);
            if (d_tiled) les_hip_free(ctx, d_tiled);
        }
        // two-view runs end with the left-right post-processing (LES/FastGCStereo.h:199-203)
        if (ok && viewModes.size() == 2) ok = postProcess(1.5f);
        if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (auto& lb : batches)
            for (SetBatch& sb : lb) { les_hip_batch_destroy(sb.b); les_hip_free(ctx, sb.rng); les_hip_free(ctx, sb.planes); les_hip_exchange_destroy(sb.x); }
        les_hip_batch_destroy(init.b); les_hip_free(ctx, init.rng); les_hip_free(ctx, init.planes); les_hip_exchange_destroy(init.x);
        if (d_xsend) les_hip_free(ctx, d_xsend);
        if (d_xrecv) les_hip_free(ctx, d_xrecv);
        les_hip_free(ctx, d_labels); les_hip_free(ctx, d_cur); les_hip_free(ctx, d_prop);
        return ok;
    }

    // PMStereoBase::postProcess (LES/PMStereoBase.h:146-256) on the device: consistency check, fill, weighted median of both
    // views' label maps.  rawLabeling0 keeps the left labelling before it (the reference's `rawlabeling`).
    bool postProcess(float threshold = 1.5f)
    {
        auto* hip = dynamic_cast<HipCostVolumeEnergy*>(stereoEnergy.get());
        if (!hip) return false;
        les_hip_ctx* ctx = hip->handle();
        const size_t bytes = (size_t)width * height * sizeof(les_hip_plane);
        rawLabeling0 = currentLabeling_[0];
        les_hip_plane* d[2] = {nullptr, nullptr};
        bool ok = true;
        for (int m = 0; m < 2 && ok; m++)
            ok = les_hip_malloc(ctx, (void**)&d[m], bytes) == LES_HIP_OK && les_hip_memcpy_h2d(ctx, d[m], currentLabeling_[m].data.data(), bytes) == LES_HIP_OK;
        if (ok) ok = les_hip_post_process(ctx, d[0], d[1], threshold, params.omega) == LES_HIP_OK;
        for (int m = 0; m < 2 && ok; m++) ok = les_hip_memcpy_d2h(ctx, currentLabeling_[m].data.data(), d[m], bytes) == LES_HIP_OK;
        if (!ok) fprintf(stderr, "PMStereo::postProcess: %s\n", les_hip_last_error());
        for (int m = 0; m < 2; m++) les_hip_free(ctx, d[m]);
        return ok;
    }
    LabelMap rawLabeling0;

    // disparity of the current labelling (StereoEnergy::computeDisparities, LES/StereoEnergy.h:269-272)
    std::vector<float> computeDisparities(int mode) const
    {
        std::vector<float> d((size_t)width * height);
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) d[(size_t)y * width + x] = currentLabeling_[mode].at(y, x).GetZ((float)x, (float)y);
        return d;
    }
    double totalCost(int mode) const
    {
        double s = 0;
        for (float v : currentCost_[mode].data) s += v;
        return s;
    }

    // data + smoothness energy of the current solution (what the graph-cut iterations minimise)
    double totalEnergy(int mode) const { return totalCost(mode) + stereoEnergy->computeSmoothnessCost(currentLabeling_[mode], mode); }

    LabelMap currentLabeling_[2];
    CostMap currentCost_[2];
    LayerManager& layers() { return layermng; }
    bool checkFlowEnergy = false;       // the reference's disabled self-check (LES/FastGCStereo.h:561-594)
    double maxFlowEnergyGap = 0;        // max |flow - energy| / max(1, |energy|) over the checked moves
    long numMoves = 0;
    double gcSeconds[3] = {0, 0, 0};    // runDevice graph-cut lock-steps: GPU propose+unary+D2H / host cuts / H2D labels
    long gcLockSteps = 0;
    bool deviceGraph = true;            // runDevice: pairwise terms / graph capacities of the moves computed on the GPU (N1)
    bool deviceCuts = true;             // runDevice: cells of at most LES_HIP_MAXFLOW_MAX_NODES nodes are cut on the GPU as well
    bool deviceCutsCoarse = true;       // ... and the larger cells too, by the tiled solver (false: the coarse layers' cuts stay on the host cores, rounds 2-4)
    long long gcTiledLaunches = 0;      // launches of the tiled solver in the last runDevice
    long gcCellsCutOnDevice = 0;
    int hostThreads = 0;                // threads of the host graph cuts in runDevice (0: at most 24 and one per cell -- larger teams are slower)
    // runDevice on several GPUs (one process -- or, in the self-test, one host thread -- per GPU): this rank's place among the ranks that share
    // the view's cells, and the transport of the per-set tile exchange: the host's ncclComm_t (RCCL; les_hip_exchange_tiles enqueues pack,
    // ncclAllGather and unpack on the context's stream) or, when set, gatherFn(rank, d_send, d_recv, slot_floats), which must leave every
    // rank's slot in d_recv[r * slot_floats ..] (another transport; les_host_demo's two-ranks-on-one-GPU self-test).
    int rank = 0, world = 1;
    void* ncclComm = nullptr;
    std::function<void(int, const float*, float*, long long)> gatherFn;

private:
    static uint64_t splitmix(uint64_t x)
    {
        x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL; x ^= x >> 31;
        return x ? x : 0xffffffffULL;
    }
    float randomWidth(int m) const { return (float)((double)(MAX_DISPARITY - MIN_DISPARITY) * std::ldexp(1.0, -(m + 1))); }
    IProposer* makeHostProposer(const ProposerSpec& s) const
    {
        if (s.kind == LES_HIP_PROPOSE_EXPANSION) return new ExpansionProposer(s.K);
        if (s.kind == LES_HIP_PROPOSE_RANDOM) return new RandomProposer(s.K, MAX_DISPARITY, MIN_DISPARITY);
        if (s.kind == LES_HIP_PROPOSE_RANSAC) return new RansacProposer(s.K);                        // MAX_SAM 500, conf 0.95, LES/Proposer.h:265
        throw std::runtime_error("PMStereo: unknown proposer kind in the layer table");
    }

    const int width, height;
    const Parameters params;
    const float MAX_DISPARITY, MIN_DISPARITY;
    LayerManager layermng;
    std::vector<std::vector<ProposerSpec>> layerProposers;
    std::vector<std::vector<uint64_t>> rngStates;
    std::unique_ptr<StereoEnergy> stereoEnergy;
    uint64_t seed_ = 1234;
};

}  // namespace les_host
