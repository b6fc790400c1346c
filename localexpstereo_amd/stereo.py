"""Python driver of the whole local expansion loop around the GPU path (mirror of FastGCStereo / PMStereoBase,
LES/FastGCStereo.h:88-227, LES/PMStereoBase.h) and of the two data-set front ends MidV2 / MidV3 (LES/main.cpp:270-420).

What runs where: proposals, unary costs, winner-take-all updates, post-processing -> MI355X through the C ABI
(liblocalexp_hip.so); graph cuts of the main iterations -> host cores (liblocalexp_host.so); file formats and the
Evaluator -> numpy (io.py).  One process per GPU: pass rank/world to shard the cells of every disjoint set.
"""
import os
import time

import numpy as np
import torch

from . import api, gc, io, pm

PARAMS_GF = dict(lambda_=1.0, windR=20, eps=1e-4, alpha=0.9, omega=10.0, th_grad=2.0, th_col=10.0, th_smooth=1.0, epsilon=0.01)   # paramsGF, LES/main.cpp:73


class FastGCStereo:
    def __init__(self, energy, imL, imR, params, device="cuda", rank=0, world=1, seed=1, host_threads=0, device_cuts=None):
        self.e, self.imL, self.imR, self.p = energy, imL, imR, dict(PARAMS_GF, **params)
        self.device, self.rank, self.world, self.seed = device, rank, world, seed
        self.units, self.table = [], []
        self.evaluator = None
        self.log = []
        self.check_flow_energy = False
        # two-view runs: graph-cut iterations of the two views in parallel host threads.  Pays when the host cuts dominate
        # (1436 x 992: 14.2 -> 11.3 s); on small images the shared stream's synchronisations cost more (cones: 2.3 -> 3.1 s)
        # (never with several ranks: the per-set all-gathers of the two views would be issued from two threads in an order
        # that differs between ranks).  joint_views: the alternative -- both views advance in lock-step and ONE host team cuts
        # the cells of both (pm.PMRunner.gc_iteration_joint); measured 12.4 s at 1436 x 992 because the right view's cuts are the
        # slow ones (7-8 s of the 10 s of cuts) and then sit on the critical path of every lock-step, so it is not the default.
        self.concurrent_views = world == 1 and int(np.asarray(imL).shape[0]) * int(np.asarray(imL).shape[1]) >= 500_000
        self.joint_views = False
        if os.environ.get("LES_VIEWS"):                 # tooling: "joint" | "concurrent" | "serial" | "concurrent-swapped"
            v = os.environ["LES_VIEWS"]
            self.joint_views = v == "joint"
            self.concurrent_views = v.startswith("concurrent")
            self._swap_view_threads = v == "concurrent-swapped"
        self.host_threads = host_threads         # threads of the host max-flows (0: library default = at most 16)
        # which cells are CUT on the GPU: None / "all" = every layer when there is a GPU (cells that fit a workgroup's LDS by les_maxflow_kernel, larger
        # ones by the tiled solver, csrc/les_maxflow_tiled.h), "fine" = only the cells that fit the LDS (rounds 2-4), "none" / False = all cuts on the host
        self.device_cuts = device_cuts
        self._view_groups = None                 # two-view runs on several ranks: one process group per view, created once, destroyed by close()
        self.bytes_exchanged, self.all_gathers = 0, 0       # of the last run(): payload received by this rank in the per-set tile all-gathers, and their number

    def close(self):
        """Releases the per-view process groups of multi-rank two-view runs (the energy context belongs to the caller)."""
        if self._view_groups is not None:
            import torch.distributed as dist
            for gr in self._view_groups:
                try:
                    dist.destroy_process_group(gr)
                except Exception:
                    pass
            self._view_groups = None

    def addLayer(self, unit_region_size, proposers):
        """proposers: list of (kind, K) with kind in api.PROPOSE_EXPANSION / _RANDOM / _RANSAC (LES/FastGCStereo.h:88-92)."""
        self.units.append(int(unit_region_size))
        self.table.append(list(proposers))

    def setEvaluator(self, evaluator, precision=-1.0):
        self.evaluator, self.precision = evaluator, precision

    def _evaluate(self, index, mode, runner, g, t0):
        """Evaluator::evaluate (LES/Evaluator.h:113-187).  Like the reference's evaluator, it stops the run's clock while it
        works (stop() / start() around the body, :115-116,183-184): `time` excludes evaluation."""
        if mode != 0:
            return
        runner._sync()
        te = time.perf_counter()
        try:
            self._evaluate_body(index, mode, runner, g, t0)
        finally:
            self.eval_seconds += time.perf_counter() - te

    def _evaluate_body(self, index, mode, runner, g, t0):
        disp = runner.disparities().cpu().numpy()
        if g is not None and getattr(runner, "gc", None) is g:
            runner.sync_gc_state()
            dc, sc = g.data_cost(mode), g.smoothness_cost(mode)
        else:
            dc, sc = float(runner.cur.sum(dtype=torch.float64)), float("nan")
        row = dict(index=index, time=time.perf_counter() - t0 - self.eval_seconds, energy=dc + (0.0 if sc != sc else sc), data=dc, smooth=sc)
        if self.evaluator is not None:
            d = disp
            if self.precision > 0:                                         # Evaluator::quantize, LES/Evaluator.h:106-111
                d = (np.rint(disp / np.float32(self.precision)) * np.float32(self.precision)).astype(np.float32)
            row["all"], row["nonocc"] = self.evaluator.evaluate(d)
        self.log.append(row)

    def run(self, maxIteration, viewModes=(0,), pmInit=0, labeling=None):
        """FastGCStereo::run (LES/FastGCStereo.h:133-227).  Returns (labeling, rawlabeling) of the left view as
        H x W x 4 float arrays (the raw one is the labelling before the two-view post-processing).  `labeling`: optional
        start labelling (the reference's `labeling` argument; every view starts from it, as in the reference)."""
        t0 = time.perf_counter()
        self.eval_seconds = 0.0
        # Several ranks and two views: the views are independent until the post-processing (LES/FastGCStereo.h:172-185), so the ranks are
        # split into one group per view -- the first ceil(world / 2) ranks advance the left view, the others the right one -- and each
        # group shards the cells of ITS view.  A coarse layer has only 4-6 cells per disjoint set (SURVEY 8 geometry table): spread
        # over world / 2 ranks instead of world, and the two views no longer take turns.  The groups meet once, before the
        # post-processing: one broadcast per view of its final label map.
        all_views = tuple(viewModes)
        view_group, view_rank, view_world, view_root = None, self.rank, self.world, {}
        if self.world > 1 and len(all_views) == 2:
            import torch.distributed as dist
            n0 = (self.world + 1) // 2
            if self._view_groups is None:                      # (every rank creates both groups; once per object, not per run)
                self._view_groups = [dist.new_group(list(range(0, n0))), dist.new_group(list(range(n0, self.world)))]
            groups = self._view_groups
            mine = 0 if self.rank < n0 else 1
            view_group, view_rank, view_world = groups[mine], self.rank - (0 if mine == 0 else n0), (n0 if mine == 0 else self.world - n0)
            view_root = {all_views[0]: 0, all_views[1]: n0}
            viewModes = (all_views[mine],)
        runners = {m: pm.PMRunner(self.e, self.units, self.table, seed=self.seed + 7919 * m, rank=view_rank, world=view_world,
                                  device=self.device, mode=m, group=view_group) for m in viewModes}
        g = gc.GraphCut(self.imL, self.imR, lambda_=self.p["lambda_"], th_smooth=self.p["th_smooth"], omega=self.p["omega"],
                        epsilon=self.p["epsilon"]) if maxIteration > 0 else None
        for m in viewModes:
            if labeling is None:
                runners[m].init_labels()
            else:
                runners[m].init_from_labels(labeling)          # warm start from a given labelling (LES/FastGCStereo.h:116-130)
            self._evaluate(0, m, runners[m], None, t0)
        # the reference starts its clock HERE -- START_TIMER after initCurrentFast and the first evaluation (LES/FastGCStereo.h:135-141),
        # with the layers (addLayer, LES/main.cpp:395-397) and the energy built before run() -- `seconds` below keeps counting from the top
        # of this function; `seconds_reference_clock` is the same run on the reference's clock
        self.init_seconds = time.perf_counter() - t0 - self.eval_seconds
        for it in range(pmInit):
            for m in viewModes:
                runners[m].iteration(it)
                self._evaluate(it + 1, m, runners[m], None, t0)
        self.gc_max_gap, self.gc_seconds = 0.0, {}
        if maxIteration > 0:
            for m in viewModes:
                if self.device_cuts is not None:
                    runners[m].device_cuts = self.device_cuts
                runners[m].begin_gc(g, mode=m)
            main_device = torch.cuda.current_device() if torch.device(self.device).type == "cuda" else 0

            def one_view(m, it, nthreads=None):
                dev = torch.device(self.device)
                if dev.type == "cuda":
                    torch.cuda.set_device(dev.index if dev.index is not None else main_device)   # current device is per host thread
                runners[m].gc_iteration(it, check=self.check_flow_energy, nthreads=self.host_threads if nthreads is None else nthreads)
            for it in range(maxIteration):
                if len(viewModes) == 2 and self.joint_views and self.world == 1 and not self.check_flow_energy:
                    # the two views are independent until the post-processing (LES/FastGCStereo.h:172-185)
                    pm.PMRunner.gc_iteration_joint([runners[m] for m in viewModes], it, nthreads=self.host_threads)
                elif len(viewModes) == 2 and self.concurrent_views:
                    # their graph-cut iterations run in two host threads (the C calls release the GIL) sharing the GPU stream
                    import threading
                    errors = []

                    # two teams cut at the same time: 16 threads each measured best on the 2 x 64-core host (1436 x 992, two views:
                    # 12 threads 10.3 s, 16 threads 10.2 s, 24 threads 10.8 s, 32 threads 11.1 s, 64 threads 11.7 s)
                    per_view = self.host_threads if self.host_threads > 0 else 16

                    on_gpu = torch.device(self.device).type == "cuda"

                    def guarded(m):
                        try:
                            if on_gpu:
                                # each view advances on its own stream: its synchronisations (one per lock-step) then wait for its
                                # own kernels only, and the two views' kernels overlap on the GPU
                                dev = torch.device(self.device)
                                torch.cuda.set_device(dev.index if dev.index is not None else main_device)
                                side = torch.cuda.Stream()
                                side.wait_stream(torch.cuda.default_stream())
                                with torch.cuda.stream(side):
                                    self.e.set_thread_stream(side.cuda_stream)
                                    try:
                                        one_view(m, it, per_view)
                                        side.synchronize()
                                    finally:
                                        self.e.set_thread_stream(0, bind=False)
                            else:
                                one_view(m, it, per_view)
                        except BaseException as ex:          # re-raised in the caller's thread below
                            errors.append(ex)
                    ths = [threading.Thread(target=guarded, args=(m,)) for m in (reversed(viewModes) if getattr(self, "_swap_view_threads", False) else viewModes)]
                    for th in ths:
                        th.start()
                    for th in ths:
                        th.join()
                    if errors:
                        raise errors[0]
                else:
                    for m in viewModes:
                        one_view(m, it)
                for m in viewModes:
                    self._evaluate(it + 1 + pmInit, m, runners[m], g, t0)
            for m in viewModes:
                self.gc_max_gap = max(self.gc_max_gap, runners[m].gc_max_gap)
                for k, v in runners[m].gc_seconds.items():
                    self.gc_seconds[k] = self.gc_seconds.get(k, 0.0) + v
        if view_root:
            # the view groups meet: every rank receives both final label maps (16 B/px each) from the first rank of each group
            import torch.distributed as dist
            H_, W_ = self.e.H, self.e.W
            other = {m: torch.zeros((H_, W_, 4), dtype=torch.float32, device=torch.device(self.device)) for m in all_views if m not in runners}
            final = {m: (runners[m].labels if m in runners else other[m]) for m in all_views}
            for m in all_views:
                if final[m].is_cuda and dist.get_backend() == "gloo":          # (one-GPU functional tests of the multi-rank path: gloo moves host memory)
                    h = final[m].cpu()
                    dist.broadcast(h, src=view_root[m])
                    final[m].copy_(h)
                else:
                    dist.broadcast(final[m], src=view_root[m])
        else:
            final = {m: runners[m].labels for m in all_views}
        raw = final[0].cpu().numpy().copy() if 0 in final else None
        if len(all_views) == 2:
            self.e.post_process(final[0].data_ptr(), final[1].data_ptr(), 1.5, self.p["omega"])     # (left, right) whatever the order of viewModes; LES/FastGCStereo.h:202
            if 0 in runners:
                self._evaluate(maxIteration + 1 + pmInit, 0, runners[0], None, t0)
            # (the rows of the log belong to the ranks of the left view's group)
        lab = final[0].cpu().numpy().copy() if 0 in final else None
        self.seconds = time.perf_counter() - t0 - self.eval_seconds          # evaluation excluded (as in the reference), set-up and initialisation included (unlike it)
        self.seconds_reference_clock = self.seconds - self.init_seconds
        self.bytes_exchanged = sum(r.bytes_exchanged for r in runners.values())
        self.all_gathers = sum(r.exchanges for r in runners.values())
        self.exchange_seconds = sum(r.exchange_seconds() for r in runners.values())     # device time inside pack -> all-gather -> unpack on this rank
        # per view and layer: how long the lock-steps cut by the tiled solver took (p50 / p90 / max ms, launches)
        self.tiled_lockstep_stats = {}
        for m, r in runners.items():
            for li, rows in getattr(r, "tiled_lockstep_ms", {}).items():
                a = np.array(rows, np.float64)
                self.tiled_lockstep_stats[f"view{m}_layer{li}"] = dict(locksteps=len(a), ms_p50=round(float(np.percentile(a[:, 0], 50)), 2), ms_p90=round(float(np.percentile(a[:, 0], 90)), 2),
                                                                       ms_max=round(float(a[:, 0].max()), 2), ms_sum=round(float(a[:, 0].sum()), 1), launches_p50=int(np.percentile(a[:, 1], 50)),
                                                                       launches_max=int(a[:, 1].max()))
        for r in runners.values():
            r.close()
        if g is not None:
            g.close()
        return lab, raw


def disparities(labeling):
    H, W = labeling.shape[:2]
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    return labeling[..., 0] * xs + labeling[..., 1] * ys + labeling[..., 2]


def _layers(st, sizes):
    e, r, p = api.PROPOSE_EXPANSION, api.PROPOSE_RANSAC, api.PROPOSE_RANDOM
    st.addLayer(sizes[0], [(e, 1), (r, 1), (p, 7)])                    # LES/main.cpp:300-306 / :391-397
    st.addLayer(sizes[1], [(e, 2), (r, 1)])
    st.addLayer(sizes[2], [(e, 2), (r, 1)])


def MidV2(data, iterations=5, pmIterations=2, doDual=False, smooth_weight=1.0, filterRadious=20, device="cuda", seed=1, lib=None, **kw):
    """MidV2 (LES/main.cpp:270-328) on a data dict of io.load_data: image-based matching cost, layers 5/15/25, error
    threshold 0.5, disparities quantised to the ground-truth precision before evaluation."""
    maxdisp = float(data["ndisp"] - 1)
    e = api.HipCostVolumeEnergy.naive(data["imL"], data["imR"], windR=filterRadious, eps=PARAMS_GF["eps"], alpha=PARAMS_GF["alpha"],
                                      th_col=PARAMS_GF["th_col"], th_grad=PARAMS_GF["th_grad"], max_disp=maxdisp,
                                      device=torch.device(device).index or 0, lib=lib)
    st = FastGCStereo(e, data["imL"], data["imR"], dict(lambda_=smooth_weight, windR=filterRadious), device=device, seed=seed, **kw)
    st.setEvaluator(io.Evaluator(data["dispGT"], data["nonocc"], 0.5), precision=data.get("gt_prec", -1.0))
    _layers(st, (5, 15, 25))
    lab, raw = st.run(iterations, (0, 1) if doDual else (0,), pmIterations)
    st.close()
    e.close()
    return st, lab, raw


def MidV3(data, volL, volR, iterations=5, pmIterations=2, doDual=False, smooth_weight=0.5, mc_threshold=0.5, filterRadious=20,
          error_threshold=1.0, device="cuda", seed=1, lib=None, **kw):
    """MidV3 (LES/main.cpp:330-420): cost-volume energy (volumes ingested on the device), layers 1 % / 3 % / 9 % of the
    image width.  volL / volR: host arrays / memmaps [ndisp][H][W] (volR None: synthesised from the left one)."""
    maxdisp = float(data["ndisp"] - 1)
    tl, tr = io.ingest_volumes(volL, volR, device=device, lib=lib)
    D, H, W = tl.shape
    e = api.HipCostVolumeEnergy(data["imL"], data["imR"], tl.data_ptr(), tr.data_ptr(), windR=filterRadious, eps=PARAMS_GF["eps"],
                                th_col=mc_threshold, max_disp=maxdisp, device=torch.device(device).index or 0, volumes_on_device=True,
                                shape=(D, H, W), lib=lib)
    st = FastGCStereo(e, data["imL"], data["imR"], dict(lambda_=smooth_weight, windR=filterRadious, th_col=mc_threshold), device=device, seed=seed, **kw)
    st.setEvaluator(io.Evaluator(data["dispGT"], data["nonocc"], error_threshold), precision=-1.0)
    _layers(st, (int(W * 0.01), int(W * 0.03), int(W * 0.09)))
    lab, raw = st.run(iterations, (0, 1) if doDual else (0,), pmIterations)
    st.close()
    e.close()
    del tl, tr
    return st, lab, raw
