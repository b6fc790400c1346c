"""Device-resident PatchMatch iterations of the local expansion loop, optionally sharded over the GPUs
of one node.

Reference loop: FastGCStereo::run / initCurrentFast / localExpansionMovesForLayer_CPU with doGC == false
(LES/FastGCStereo.h:22-72, 94-115, 133-169).  Cells of one disjoint set are independent
(LES/LayerManager.h:168-172), so every rank owns a contiguous band of the cells of each set, runs their
lock-steps (propose -> unary -> winner-take-all, all on the device through the C ABI) and then one
all-gather over RCCL/xGMI publishes the updated label/cost tiles of the set to every replica.  The
volume, the guide statistics and the label/cost maps are replicated; nothing is ever reduced.

torch is plumbing here: device buffers and torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" for
the CPU tests that run against the simulator build of the same C ABI).
"""
import numpy as np
import torch

from . import api


def layer_geometry(W, H, windR, unit):
    """LayerManager::addLayer geometry (LES/LayerManager.h:44-185) as numpy rect arrays + disjoint sets."""
    minsize = max(2, unit // 2)
    frac_w, frac_h = W % unit, H % unit
    split_w, split_h = frac_w >= minsize, frac_h >= minsize
    wb, hb = W // unit + int(split_w), H // unit + int(split_h)

    # (vectorised over the cells: the per-cell Python loop was 30 ms of a run's set-up at the Adirondack shape)
    def clip(x0, y0, x1, y1):
        x0, y0, x1, y1 = np.maximum(x0, 0), np.maximum(y0, 0), np.minimum(x1, W), np.minimum(y1, H)
        ok = (x1 > x0) & (y1 > y0)
        z = np.zeros_like(x0)
        return np.where(ok, x0, z), np.where(ok, y0, z), np.where(ok, x1 - x0, z), np.where(ok, y1 - y0, z)

    i, j = np.meshgrid(np.arange(hb, dtype=np.int64), np.arange(wb, dtype=np.int64), indexing="ij")
    i, j = i.reshape(-1), j.reshape(-1)
    ux1 = (j + 1) * unit + np.where((not split_w) & (j == wb - 1), frac_w, 0)
    uy1 = (i + 1) * unit + np.where((not split_h) & (i == hb - 1), frac_h, 0)
    units = np.stack(clip(j * unit, i * unit, ux1, uy1), axis=1)
    ex = np.where((not split_w) & (j == wb - 2), frac_w, 0)
    ey = np.where((not split_h) & (i == hb - 2), frac_h, 0)
    sx, sy, sw, sh = clip((j - 1) * unit, (i - 1) * unit, (j + 2) * unit, (i + 2) * unit)
    shared = np.stack([sx, sy, sw + ex, sh + ey], axis=1)
    fx, fy, fw, fh = clip((j - 1) * unit - windR, (i - 1) * unit - windR, (j + 2) * unit + windR, (i + 2) * unit + windR)
    filt = np.stack(clip(fx, fy, fx + fw + ex, fy + fh + ey), axis=1)
    phase = (i % 4) * 4 + (j % 4)
    cell = i * wb + j
    sets = [cell[phase == s] for s in range(16)]
    to = lambda a: np.array(a, np.int32).reshape(-1, 4).view(api.RECT_DT).reshape(-1)
    return to(units), to(shared), to(filt), [np.asarray(s, np.int64) for s in sets if len(s)]


def seeds_for(n, seed):
    """Non-zero 64-bit cv::RNG states, one per cell."""
    x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(31)
    return np.where(x == 0, np.uint64(0xFFFFFFFF), x).astype(np.uint64)


class _Shard:
    """One (layer, disjoint set) on one rank: prepared batch of the rank's own cells + exchange indices."""

    def __init__(self, runner, units, shared, filt, cells, seeds, target_is_unit=False):
        e, dev = runner.e, runner.device
        world, rank = runner.world, runner.rank
        bounds = np.linspace(0, len(cells), world + 1).astype(int)          # contiguous bands of cells
        self.own = cells[bounds[rank]:bounds[rank + 1]]
        tgt = units if target_is_unit else shared
        self.n = len(self.own)
        self.regions = np.ascontiguousarray(tgt[self.own])
        self.batch_filter = np.ascontiguousarray(filt[self.own])
        self.batch = api.Batch(e, filt[self.own], tgt[self.own])
        self.batch.set_units(units[self.own])
        self.rng = torch.from_numpy(seeds[self.own].view(np.int64).copy()).to(dev)
        self.planes = torch.zeros((max(1, self.n), 4), dtype=torch.float32, device=dev)
        self.graph_off = self.batch.graph_offsets()
        self.graph_nodes = self.batch.graph_nodes()
        self.payload = None                      # device / pinned host buffers of the graph capacities, allocated on first use
        # the plan of the set's tile exchange (C ABI: les_hip_exchange_*): every rank's target rects, this rank's slot layout
        self.xchg = api.Exchange(e, rank, [tgt[cells[bounds[r]:bounds[r + 1]]] for r in range(world)]) if world > 1 else None


class PMRunner:
    def __init__(self, energy, layer_units, proposer_table, seed=1, rank=0, world=1, device="cuda", mode=0, group=None):
        """rank / world: this process's place among the ranks that share THIS view's cells; group: their torch.distributed process group
        (None = the default group).  Two-view runs on several ranks give each view its own group (stereo.FastGCStereo.run)."""
        self.e, self.rank, self.world, self.mode, self.group = energy, rank, world, mode, group
        self.device = torch.device(device)
        if self.device.type == "cuda":
            # the torch ops of this class (exchange index_copy_, label / mask copies) run on torch's current stream: bind the
            # library's launches to the same stream so that their order is the program order under any stream context
            energy.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.H, self.W = energy.H, energy.W
        self.table = proposer_table
        self.maxd, self.mind = float(energy.max_disp), float(energy.params.min_disparity)
        windR = energy.params.windR
        self.labels = torch.zeros((self.H, self.W, 4), dtype=torch.float32, device=self.device)
        self.cur = torch.zeros((self.H, self.W), dtype=torch.float32, device=self.device)
        self.prop = torch.zeros((self.H, self.W), dtype=torch.float32, device=self.device)
        self.shards = []
        for li, unit in enumerate(layer_units):
            units, shared, filt, sets = layer_geometry(self.W, self.H, windR, unit)
            seeds = seeds_for(len(units), seed + 1000 * li)
            self.shards.append([_Shard(self, units, shared, filt, cells, seeds) for cells in sets])
            if li == 0:
                x0 = np.maximum(units["x"] - windR, 0); y0 = np.maximum(units["y"] - windR, 0)
                x1 = np.minimum(units["x"] + units["w"] + windR, self.W); y1 = np.minimum(units["y"] + units["h"] + windR, self.H)
                fr = np.stack([x0, y0, x1 - x0, y1 - y0], 1).astype(np.int32).view(api.RECT_DT).reshape(-1)
                self.init = _Shard(self, units, shared, fr, np.arange(len(units)), seeds_for(len(units), seed + 777), target_is_unit=True)
        self.bytes_exchanged = 0
        self.exchanges = 0                       # all-gathers issued
        self._xevents = []

    # -- exchange: one all-gather of the updated tiles of a set (labels 16 B/px + cost 4 B/px).  Pack and unpack are kernels of the
    # library on the runner's stream; with the nccl backend the collective is enqueued on the same stream by torch, so nothing here
    # waits on the host (gloo works on host tensors: the simulator's "device" memory is host memory).
    def _exchange(self, sh):
        if self.world == 1:
            return
        import torch.distributed as dist
        x = sh.xchg
        if x.slot_floats == 0:
            return
        if getattr(self, "_xbuf", None) is None or self._xbuf[0].numel() < x.slot_floats:
            n = max(s_.xchg.slot_floats for layer in self.shards for s_ in layer) if self.shards else x.slot_floats
            n = max(n, x.slot_floats, self.init.xchg.slot_floats if getattr(self, "init", None) is not None and self.init.xchg else 0)
            self._xbuf = (torch.zeros(n, dtype=torch.float32, device=self.device), torch.zeros(n * self.world, dtype=torch.float32, device=self.device))
        send, recv = self._xbuf[0][: x.slot_floats], self._xbuf[1][: x.slot_floats * self.world]
        ev = None
        if self.device.type == "cuda":           # device time of pack -> all-gather -> unpack on this rank's stream (summed by exchange_seconds())
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(self.device))
        x.pack(self.labels.data_ptr(), self.cur.data_ptr(), send.data_ptr())
        if self.device.type != "cuda":
            self._sync()
        if send.is_cuda and dist.get_backend(self.group) == "gloo":
            # functional tests of the multi-rank path on a box with ONE GPU (bench.py: LES_BENCH_BACKEND=gloo): gloo moves host memory
            send_h, recv_h = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(recv_h, send_h, group=self.group)
            recv.copy_(recv_h)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        self.bytes_exchanged += recv.numel() * 4
        self.exchanges += 1
        x.unpack(recv.data_ptr(), self.labels.data_ptr(), self.cur.data_ptr())
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(self.device))
            self._xevents.append(ev)

    def exchange_seconds(self):
        """Device seconds this rank's stream spent in the tile exchanges so far (pack -> all-gather -> unpack, events around each): what separates the
        collective's share from the cuts' in a multi-GPU run.  Synchronises."""
        if self.device.type != "cuda" or not self._xevents:
            return 0.0
        torch.cuda.synchronize(self.device)
        return sum(a.elapsed_time(b) for a, b in self._xevents) * 1e-3

    def _sync(self):
        self.e.synchronize()

    def init_labels(self):
        """initCurrentFast (LES/FastGCStereo.h:94-115): random label per layer-0 cell + its unit-region cost."""
        sh = self.init
        if sh.n:
            sh.batch.propose(api.PROPOSE_INIT, self.labels.data_ptr(), sh.rng.data_ptr(), sh.planes.data_ptr())
            sh.batch.run(sh.planes.data_ptr(), self.cur.data_ptr(), mode=self.mode, check=True, planes_on_device=True)
        self._sync()
        self._exchange(sh)

    def init_from_labels(self, labels, rows_per_launch=64):
        """The warm-start branch of initCurrentFast (LES/FastGCStereo.h:116-130, "very slow" on the CPU): start from a
        given H x W x 4 label map; the current cost of every pixel is the unary cost of its own label, evaluated with
        a 1 x 1 target and the filter region pixel +- windR -- one job per pixel, `rows_per_launch` image rows per launch."""
        lab = torch.as_tensor(np.ascontiguousarray(labels, np.float32)).to(self.device)
        assert tuple(lab.shape) == (self.H, self.W, 4)
        self.labels.copy_(lab)
        R, W, H = self.e.params.windR, self.W, self.H
        xs = np.arange(W, dtype=np.int32)
        x0, x1 = np.maximum(xs - R, 0), np.minimum(xs + R + 1, W)
        for ya in range(0, H, rows_per_launch):
            yb = min(H, ya + rows_per_launch)
            ys = np.arange(ya, yb, dtype=np.int32)
            y0, y1 = np.maximum(ys - R, 0), np.minimum(ys + R + 1, H)
            fr = np.stack([np.broadcast_to(x0, (yb - ya, W)), np.broadcast_to(y0[:, None], (yb - ya, W)),
                           np.broadcast_to(x1 - x0, (yb - ya, W)), np.broadcast_to((y1 - y0)[:, None], (yb - ya, W))], -1).reshape(-1, 4)
            tr = np.stack([np.broadcast_to(xs, (yb - ya, W)), np.broadcast_to(ys[:, None], (yb - ya, W)),
                           np.ones((yb - ya, W), np.int32), np.ones((yb - ya, W), np.int32)], -1).reshape(-1, 4)
            b = api.Batch(self.e, np.ascontiguousarray(fr, np.int32), np.ascontiguousarray(tr, np.int32))
            b.run(self.labels[ya:yb].data_ptr(), self.cur.data_ptr(), mode=self.mode, check=True, planes_on_device=True)
            self._sync()
            b.destroy()
        if self.world > 1:
            pass        # every rank evaluates the whole map here (replicated state, nothing to exchange)

    def iteration(self, iteration):
        """One PatchMatch iteration over all layers (LES/FastGCStereo.h:143-157 with doGC == false)."""
        for li, layer in enumerate(self.shards):
            for sh in layer:
                if sh.n:
                    for kind, K in self.table[li]:
                        for it in range(K):
                            m = iteration + it
                            if kind == api.PROPOSE_RANDOM and (self.maxd - self.mind) * 0.5 ** (m + 1) < 0.1:      # LES/Proposer.h:149-152
                                break
                            sh.batch.propose(kind, self.labels.data_ptr(), sh.rng.data_ptr(), sh.planes.data_ptr(), m=m)
                            sh.batch.run(sh.planes.data_ptr(), self.prop.data_ptr(), mode=self.mode, check=True, planes_on_device=True)
                            sh.batch.wta(sh.planes.data_ptr(), self.cur.data_ptr(), self.prop.data_ptr(), self.labels.data_ptr())
                if self.world == 1 or self.device.type != "cuda":
                    self._sync()                     # (bounds the launch queue; with several ranks on GPUs the collective orders the stream itself)
                self._exchange(sh)

    # -- graph-cut iterations (LES/FastGCStereo.h:171-185: the main loop, doGC == true) ----------------------------
    # Proposals and unary costs come from the GPU exactly as in iteration(); the winner-take-all update is replaced by
    # the local expansion moves of the rank's own cells on the host cores (gc.GraphCut), and the fused labels go
    # back to the device for the next proposals.  Cross-rank coherence is the same per-set all-gather.
    def begin_gc(self, graph_cut, mode=None, device_graph=True):
        """device_graph (default): the solution stays on the GPU -- pairwise terms / graph capacities of every move are
        computed there (les_hip_batch_expansion_graph), the host receives only the graphs, runs the max-flows and
        returns one mask byte per node, which the GPU applies (les_hip_batch_apply_masks).  False (or check=True in
        gc_iteration) = the reference's shape: host-resident solution, host graph construction."""
        self.gc = graph_cut
        self.device_graph = device_graph
        # device_cuts: cells small enough for a workgroup's LDS (the finest layer) are also CUT on the GPU (les_hip_batch_solve_graphs),
        # so neither their graphs nor their masks cross PCIe and the host cores only see the larger cells.  On by default on a GPU
        # (the simulator build used by the CPU tests would spend minutes in it).
        # "all" (default on a GPU): the larger cells as well, by the tiled solver (les_hip_batch_solve_graphs_tiled: graphs resident in device
        # memory, a workgroup per tile); "fine": only the cells that fit the LDS (rounds 2-4); "none": every cut on the host.
        dc = getattr(self, "device_cuts", None)
        if dc is None:
            dc = "all" if self.device.type == "cuda" else "none"     # (the simulator build used by the CPU tests would spend minutes in it)
        elif dc is True:
            dc = "all"
        elif dc is False:
            dc = "none"
        if dc not in ("all", "fine", "none"):
            raise ValueError(f"device_cuts: {dc!r} (all | fine | none)")
        self.device_cuts = dc
        self._gc_mode = self.mode if mode is None else mode
        self.sync_gc_state()
        pin = (lambda t: t.pin_memory()) if self.device.type == "cuda" else (lambda t: t)
        self._prop_host = pin(torch.empty((self.H, self.W), dtype=torch.float32))
        self.gc_max_gap = 0.0
        self.tiled_lockstep_ms = {}              # layer -> [(ms, launches)] of every lock-step the tiled solver cut (per view: a runner is a view)
        self.gc_seconds = {"device": 0.0, "host_cuts": 0.0, "h2d": 0.0}
        self.gc_seconds.update({f"host_cuts_layer{li}": 0.0 for li in range(len(self.shards))})

    def sync_gc_state(self):
        """Copy the device solution into the host graph-cut context (for its energy queries / the host-construction path)."""
        m = self._gc_mode
        self._sync()
        self.gc.labels[m][...] = self.labels.cpu().numpy()
        self.gc.costs[m][...] = self.cur.cpu().numpy()

    def _gc_buffers(self, sh):
        """Views of the runner-wide graph / mask staging buffers (one device + one pinned host allocation, sized for the
        largest lock-step) cut to this shard's node count."""
        if getattr(self, "_gc_payload", None) is None:
            pin = (lambda t: t.pin_memory()) if self.device.type == "cuda" else (lambda t: t)
            n = max([1] + [s.graph_nodes for layer in self.shards for s in layer])
            self._gc_payload = torch.empty(n * 5, dtype=torch.float32, device=self.device)
            self._gc_payload_host = pin(torch.empty(n * 5, dtype=torch.float32))
            self._gc_masks = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._gc_masks_host = pin(torch.zeros(n, dtype=torch.uint8))
        if sh.payload is None:
            n = max(1, sh.graph_nodes)
            sh.payload, sh.payload_host = self._gc_payload[: n * 5], self._gc_payload_host[: n * 5]
            sh.masks, sh.masks_host = self._gc_masks[:n], self._gc_masks_host[:n]

    def _dump_tiled(self, sh, m, iteration, li, ms, launches):
        """Tooling (tools/tiled_cut_replay.py): LES_DUMP_TILED=dir LES_DUMP_EVERY=n [LES_DUMP_VIEW=v LES_DUMP_MAX=k] writes the graphs of every
        n-th lock-step the tiled solver cut (regions, node offsets, the 5-float payload) with its wall time and launch count."""
        import os
        d = os.environ.get("LES_DUMP_TILED")
        if not d or os.environ.get("LES_DUMP_VIEW", str(m)) != str(m):
            return
        self._dump_tiled_count = getattr(self, "_dump_tiled_count", 0) + 1
        every = int(os.environ.get("LES_DUMP_EVERY", "50"))
        if self._dump_tiled_count % every or getattr(self, "_dump_tiled_done", 0) >= int(os.environ.get("LES_DUMP_MAX", "6")):
            return
        self._dump_tiled_done = getattr(self, "_dump_tiled_done", 0) + 1
        nn = int(sh.graph_off[-1] + int(sh.regions[-1]["w"]) * int(sh.regions[-1]["h"]))
        np.savez_compressed(os.path.join(d, f"tiled_view{m}_it{iteration}_layer{li}_{self._dump_tiled_count}.npz"), regions=sh.regions, offsets=sh.graph_off,
                            payload=sh.payload[: nn * 5].cpu().numpy(), ms=ms, launches=launches, cells=sh.n)

    def _gc_set_without_round_trips(self, sh, li, iteration):
        """All proposals of one disjoint set of the FINEST layer (cells that fit a workgroup's LDS: propose -> unary costs -> graph -> cut -> apply, nine
        times) enqueued without a single host round trip; the cuts count the cells that hit their iteration limit in ONE device word, read once at the
        end of the set (rounds 2-5 read a status word per lock-step: 720 synchronisations per view).  In the -- so far unobserved -- case that the word
        is not zero the set is rolled back (labels, costs, generator states were saved in device memory: 30 MB, microseconds) and the caller repeats it
        lock-step by lock-step with the host fall-back.  -> True: done."""
        import os
        import time
        if self.device.type != "cuda" and not getattr(self, "speculative_sets_on_cpu", False):
            return False
        if self.device_cuts not in ("all", "fine") or sh.batch.max_cell_nodes > api.Batch.MAXFLOW_MAX_NODES:
            return False
        if os.environ.get("LES_DUMP_GRAPHS") or os.environ.get("LES_GC_PER_LOCKSTEP_CHECK"):
            return False
        t0 = time.perf_counter()
        if getattr(self, "_gc_snap", None) is None:
            self._gc_snap = (torch.empty_like(self.labels), torch.empty_like(self.cur))
            self._gc_fail = torch.zeros(1, dtype=torch.int32, device=self.device)
            nmax = max([1] + [s_.n for layer_ in self.shards for s_ in layer_])
            if getattr(self, "_gc_status", None) is None:
                self._gc_status = torch.zeros(nmax, dtype=torch.int32, device=self.device)
        self._gc_buffers(sh)
        self._gc_snap[0].copy_(self.labels)
        self._gc_snap[1].copy_(self.cur)
        rng0 = sh.rng.clone()
        self._gc_fail.zero_()
        st = self._gc_status[: sh.n]
        p, m = self.gc.params, self.mode
        locksteps = 0
        for kind, K in self.table[li]:
            for it in range(K):
                mm = iteration + it
                if kind == api.PROPOSE_RANDOM and (self.maxd - self.mind) * 0.5 ** (mm + 1) < 0.1:
                    break
                sh.batch.propose(kind, self.labels.data_ptr(), sh.rng.data_ptr(), sh.planes.data_ptr(), m=mm)
                sh.batch.run(sh.planes.data_ptr(), self.prop.data_ptr(), mode=m, check=True, planes_on_device=True)
                sh.batch.expansion_graph(sh.planes.data_ptr(), self.labels.data_ptr(), self.cur.data_ptr(), self.prop.data_ptr(), sh.payload.data_ptr(), mode=m,
                                         lambda_=p["lambda_"], th_smooth=p["th_smooth"], omega=p["omega"], epsilon=p["epsilon"])
                sh.batch.solve_graphs(sh.payload.data_ptr(), sh.masks.data_ptr(), st.data_ptr(), unsolved_total_dev=self._gc_fail.data_ptr())
                sh.batch.apply_masks(sh.planes.data_ptr(), sh.masks.data_ptr(), self.cur.data_ptr(), self.prop.data_ptr(), self.labels.data_ptr())
                locksteps += 1
        failed = int(self._gc_fail.item())                 # the set's only synchronisation
        self.gc_seconds["device"] += time.perf_counter() - t0
        if failed:
            self.labels.copy_(self._gc_snap[0])
            self.cur.copy_(self._gc_snap[1])
            sh.rng.copy_(rng0)
            self.gc_seconds["sets_rolled_back"] = self.gc_seconds.get("sets_rolled_back", 0) + 1
            return False
        self.gc_seconds["cells_cut_on_device"] = self.gc_seconds.get("cells_cut_on_device", 0) + sh.n * locksteps
        self.gc_seconds["sets_without_round_trips"] = self.gc_seconds.get("sets_without_round_trips", 0) + 1
        return True

    def gc_iteration(self, iteration, check=False, nthreads=0):
        import os
        import time
        from . import gc as lgc
        gc, m = self.gc, self.mode
        host_path = check or not self.device_graph
        if host_path:
            self.sync_gc_state()
            lab_host = torch.from_numpy(gc.labels[m])
            cur_host = torch.from_numpy(gc.costs[m])
        for li, layer in enumerate(self.shards):
            for sh in layer:
                if sh.n and not host_path and self._gc_set_without_round_trips(sh, li, iteration):
                    if self.world > 1:
                        self._exchange(sh)
                    continue
                if sh.n:
                    for kind, K in self.table[li]:
                        for it in range(K):
                            mm = iteration + it
                            if kind == api.PROPOSE_RANDOM and (self.maxd - self.mind) * 0.5 ** (mm + 1) < 0.1:
                                break
                            t0 = time.perf_counter()
                            sh.batch.propose(kind, self.labels.data_ptr(), sh.rng.data_ptr(), sh.planes.data_ptr(), m=mm)
                            sh.batch.run(sh.planes.data_ptr(), self.prop.data_ptr(), mode=m, check=True, planes_on_device=True)
                            if host_path:
                                self._sync()
                                self._prop_host.copy_(self.prop)
                                planes = sh.planes[: sh.n].cpu().numpy()
                                t1 = time.perf_counter()
                                gap = gc.expansion_moves(sh.regions, planes, self._prop_host.numpy(), mode=m, nthreads=nthreads, check=check)
                                self.gc_max_gap = max(self.gc_max_gap, gap)
                                t2 = time.perf_counter()
                                self.labels.copy_(lab_host)
                                self.cur.copy_(cur_host)
                            else:
                                self._gc_buffers(sh)
                                p = gc.params
                                sh.batch.expansion_graph(sh.planes.data_ptr(), self.labels.data_ptr(), self.cur.data_ptr(), self.prop.data_ptr(),
                                                         sh.payload.data_ptr(), mode=m, lambda_=p["lambda_"], th_smooth=p["th_smooth"], omega=p["omega"],
                                                         epsilon=p["epsilon"])
                                on_dev = False
                                small = sh.batch.max_cell_nodes <= api.Batch.MAXFLOW_MAX_NODES
                                if sh.n and (self.device_cuts == "all" or (self.device_cuts == "fine" and small)):
                                    if getattr(self, "_gc_status", None) is None:
                                        nmax = max([1] + [s_.n for layer_ in self.shards for s_ in layer_])
                                        self._gc_status = torch.zeros(nmax, dtype=torch.int32, device=self.device)
                                    st = self._gc_status[: sh.n]
                                    if small:
                                        sh.batch.solve_graphs(sh.payload.data_ptr(), sh.masks.data_ptr(), st.data_ptr())
                                    else:
                                        if getattr(self, "_gc_tiled_ws", None) is None:      # one scratch per runner (= per view and host thread)
                                            nb = max(s_.batch.tiled_workspace_bytes() for layer_ in self.shards for s_ in layer_
                                                     if s_.n and s_.batch.max_cell_nodes > api.Batch.MAXFLOW_MAX_NODES)
                                            self._gc_tiled_ws = torch.empty(nb + 256, dtype=torch.uint8, device=self.device)
                                        ws = self._gc_tiled_ws
                                        wp = (ws.data_ptr() + 255) & ~255
                                        tl0 = time.perf_counter()
                                        nl = sh.batch.solve_graphs_tiled(sh.payload.data_ptr(), sh.masks.data_ptr(), st.data_ptr(), wp, ws.numel() - (wp - ws.data_ptr()))
                                        self.gc_seconds["tiled_launches"] = self.gc_seconds.get("tiled_launches", 0) + nl
                                        self.gc_seconds["tiled_locksteps"] = self.gc_seconds.get("tiled_locksteps", 0) + 1
                                        ts_ = sh.batch.tiled_stats                       # cells the host cores finished from their residual graphs (hand-over)
                                        self.gc_seconds["tiled_handed_cells"] = self.gc_seconds.get("tiled_handed_cells", 0) + ts_["handed_cells"]
                                        self.gc_seconds["tiled_handed_locksteps"] = self.gc_seconds.get("tiled_handed_locksteps", 0) + (1 if ts_["handed_cells"] else 0)
                                        self.gc_seconds["tiled_handed_host_seconds"] = self.gc_seconds.get("tiled_handed_host_seconds", 0.0) + 1e-3 * ts_["host_ms"]
                                        self.gc_seconds[f"tiled_seconds_layer{li}"] = self.gc_seconds.get(f"tiled_seconds_layer{li}", 0.0) + time.perf_counter() - tl0
                                        self.tiled_lockstep_ms.setdefault(li, []).append((1e3 * (time.perf_counter() - tl0), nl))      # (wall of the solve call, launches enqueued)
                                        self._dump_tiled(sh, m, iteration, li, 1e3 * (time.perf_counter() - tl0), nl)
                                    # (the only synchronisation of the lock-step; the tiled solver reports "cells that gave up" through a host-mapped word: no copy)
                                    on_dev = (sh.batch.tiled_unsolved == 0) if not small else not bool(st.any().item())
                                    if on_dev:
                                        self.gc_seconds["cells_cut_on_device"] = self.gc_seconds.get("cells_cut_on_device", 0) + sh.n
                                if on_dev:
                                    t1 = t2 = time.perf_counter()
                                else:
                                    self._sync()
                                    sh.payload_host.copy_(sh.payload)
                                    # only the cells the device gave up on are cut again (their status word is non-zero); the masks of the others stay
                                    failed = None
                                    if sh.n and self.device_cuts in ("all", "fine") and getattr(self, "_gc_status", None) is not None and (self.device_cuts == "all" or small):
                                        bad = np.nonzero(self._gc_status[: sh.n].cpu().numpy())[0]
                                        if 0 < len(bad) < sh.n:
                                            failed = bad
                                    t1 = time.perf_counter()
                                    if failed is None:
                                        lgc.solve_prebuilt(sh.regions, sh.payload_host.numpy(), sh.graph_off, sh.masks_host.numpy(), nthreads=nthreads)
                                    else:
                                        sh.masks_host.copy_(sh.masks)
                                        lgc.solve_prebuilt(np.ascontiguousarray(sh.regions[failed]), sh.payload_host.numpy(), np.ascontiguousarray(sh.graph_off[failed]),
                                                           sh.masks_host.numpy(), nthreads=nthreads)
                                        self.gc_seconds["cells_recut_on_host"] = self.gc_seconds.get("cells_recut_on_host", 0) + len(failed)
                                    t2 = time.perf_counter()
                                dump = os.environ.get("LES_DUMP_GRAPHS")           # tooling: timing log of the lock-steps + the graphs of the slowest one of the coarsest layer
                                if dump:
                                    with open(os.path.join(dump, f"cutlog_view{m}.txt"), "a") as f:
                                        f.write(f"{iteration} {li} {kind} {it} {sh.n} {t2 - t1:.6f}\n")
                                every = int(os.environ.get("LES_DUMP_EVERY", "0")) if dump else 0   # tooling: a sample of ordinary lock-steps (first two cells of every N-th one that was cut on the host)
                                if every and not on_dev and os.environ.get("LES_DUMP_VIEW", str(m)) == str(m):
                                    self._dump_count = getattr(self, "_dump_count", 0) + 1
                                    if self._dump_count % every == 0:
                                        k2 = sh.n if os.environ.get("LES_DUMP_FULL") else min(2, sh.n)
                                        nn = int(sh.graph_off[k2 - 1] + int(sh.regions[k2 - 1]["w"]) * int(sh.regions[k2 - 1]["h"]))
                                        np.savez_compressed(os.path.join(dump, f"sample_view{m}_it{iteration}_layer{li}_{self._dump_count}.npz"), regions=sh.regions[:k2],
                                                            offsets=sh.graph_off[:k2], payload=sh.payload_host.numpy()[: nn * 5].copy(), seconds=t2 - t1, cells=sh.n)
                                if dump and li == len(self.shards) - 1 and iteration >= 1 and t2 - t1 > getattr(self, "_dump_worst", 0.012):
                                    self._dump_worst = t2 - t1
                                    nn = int(sh.graph_off[-1] + int(sh.regions[-1]["w"]) * int(sh.regions[-1]["h"]))
                                    np.savez_compressed(os.path.join(dump, f"graphs_view{m}_layer{li}.npz"), regions=sh.regions, offsets=sh.graph_off,
                                                        payload=sh.payload_host.numpy()[: nn * 5].copy(), seconds=t2 - t1)
                                if not on_dev:
                                    sh.masks.copy_(sh.masks_host)
                                sh.batch.apply_masks(sh.planes.data_ptr(), sh.masks.data_ptr(), self.cur.data_ptr(), self.prop.data_ptr(), self.labels.data_ptr())
                            t3 = time.perf_counter()
                            self.gc_seconds["device"] += t1 - t0
                            self.gc_seconds["host_cuts"] += t2 - t1
                            self.gc_seconds[f"host_cuts_layer{li}"] += t2 - t1
                            self.gc_seconds["h2d"] += t3 - t2
                if self.world > 1:
                    self._exchange(sh)
                    if host_path:
                        lab_host.copy_(self.labels)
                        cur_host.copy_(self.cur)
        self._sync()

    @staticmethod
    def gc_iteration_joint(runners, iteration, nthreads=0):
        """Graph-cut iteration of SEVERAL views in lock-step (two-view runs, LES/FastGCStereo.h:172-185: the views are independent
        until the post-processing).  Every lock-step evaluates the proposals of all views on the GPU, moves ONE payload to the host,
        cuts the cells of all views in ONE OpenMP team (twice the cells per fork/join, and for the coarsest layer twice the
        otherwise scarce parallelism) and applies the masks per view.  Same results as gc_iteration per view: the cuts of
        different views touch disjoint state.  Single rank, device-built graphs."""
        import time
        from . import gc as lgc
        r0 = runners[0]
        assert all(r.world == 1 and r.device_graph for r in runners)
        p = r0.gc.params
        if getattr(r0, "_joint", None) is None:
            pin = (lambda t: t.pin_memory()) if r0.device.type == "cuda" else (lambda t: t)
            n = max([1] + [sum(r.shards[li][si].graph_nodes for r in runners) for li in range(len(r0.shards)) for si in range(len(r0.shards[li]))])
            r0._joint = dict(payload=torch.empty(n * 5, dtype=torch.float32, device=r0.device), payload_host=pin(torch.empty(n * 5, dtype=torch.float32)),
                             masks=torch.empty(n, dtype=torch.uint8, device=r0.device), masks_host=pin(torch.zeros(n, dtype=torch.uint8)), batches={})
        J = r0._joint
        for li in range(len(r0.shards)):
            for si in range(len(r0.shards[li])):
                shs = [r.shards[li][si] for r in runners]
                if not any(sh.n for sh in shs):
                    continue
                base = np.cumsum([0] + [sh.graph_nodes for sh in shs])
                regions = np.concatenate([sh.regions for sh in shs])
                offsets = np.concatenate([sh.graph_off + int(base[v]) for v, sh in enumerate(shs)]).astype(np.int64)
                total = int(base[-1])
                for kind, K in r0.table[li]:
                    for it in range(K):
                        mm = iteration + it
                        if kind == api.PROPOSE_RANDOM and (r0.maxd - r0.mind) * 0.5 ** (mm + 1) < 0.1:
                            break
                        t0 = time.perf_counter()
                        for v, (r, sh) in enumerate(zip(runners, shs)):
                            if not sh.n:
                                continue
                            sh.batch.propose(kind, r.labels.data_ptr(), sh.rng.data_ptr(), sh.planes.data_ptr(), m=mm)
                            sh.batch.run(sh.planes.data_ptr(), r.prop.data_ptr(), mode=r.mode, check=True, planes_on_device=True)
                            sh.batch.expansion_graph(sh.planes.data_ptr(), r.labels.data_ptr(), r.cur.data_ptr(), r.prop.data_ptr(),
                                                     J["payload"].data_ptr() + 20 * int(base[v]), mode=r.mode, lambda_=p["lambda_"], th_smooth=p["th_smooth"],
                                                     omega=p["omega"], epsilon=p["epsilon"])
                        # device cuts: ONE solve over the cells of all views -- a joint batch (the views' target rects one after the other) has
                        # exactly the node offsets of the concatenated payload, so the single-batch entry points serve it unchanged
                        on_dev = False
                        ncell = sum(sh.n for sh in shs)
                        small = max([0] + [sh.batch.max_cell_nodes for sh in shs if sh.n]) <= api.Batch.MAXFLOW_MAX_NODES
                        if ncell and (r0.device_cuts == "all" or (r0.device_cuts == "fine" and small)):
                            jb = J["batches"].get((li, si))
                            if jb is None:
                                fr = np.concatenate([sh.batch_filter for sh in shs])
                                jb = J["batches"][(li, si)] = api.Batch(r0.e, fr, regions)
                                assert jb.graph_nodes() == total and np.array_equal(jb.graph_offsets(), offsets)
                            if J.get("status") is None or J["status"].numel() < ncell:
                                J["status"] = torch.zeros(max(ncell, 1024), dtype=torch.int32, device=r0.device)
                            st = J["status"][:ncell]
                            if small:
                                jb.solve_graphs(J["payload"].data_ptr(), J["masks"].data_ptr(), st.data_ptr())
                            else:
                                nb = jb.tiled_workspace_bytes()
                                if J.get("ws") is None or J["ws"].numel() < nb + 256:
                                    J["ws"] = torch.empty(nb + 256, dtype=torch.uint8, device=r0.device)
                                wp = (J["ws"].data_ptr() + 255) & ~255
                                nl = jb.solve_graphs_tiled(J["payload"].data_ptr(), J["masks"].data_ptr(), st.data_ptr(), wp, J["ws"].numel() - (wp - J["ws"].data_ptr()))
                                r0.gc_seconds["tiled_launches"] = r0.gc_seconds.get("tiled_launches", 0) + nl
                                r0.gc_seconds["tiled_locksteps"] = r0.gc_seconds.get("tiled_locksteps", 0) + 1
                            on_dev = (jb.tiled_unsolved == 0) if not small else not bool(st.any().item())
                            if on_dev:
                                r0.gc_seconds["cells_cut_on_device"] = r0.gc_seconds.get("cells_cut_on_device", 0) + ncell
                        if on_dev:
                            t1 = t2 = time.perf_counter()
                        else:
                            r0._sync()
                            J["payload_host"][: total * 5].copy_(J["payload"][: total * 5])
                            t1 = time.perf_counter()
                            lgc.solve_prebuilt(regions, J["payload_host"].numpy()[: total * 5], offsets, J["masks_host"].numpy()[:total], nthreads=nthreads)
                            t2 = time.perf_counter()
                            J["masks"][:total].copy_(J["masks_host"][:total])
                        for v, (r, sh) in enumerate(zip(runners, shs)):
                            if sh.n:
                                sh.batch.apply_masks(sh.planes.data_ptr(), J["masks"].data_ptr() + int(base[v]), r.cur.data_ptr(), r.prop.data_ptr(), r.labels.data_ptr())
                        t3 = time.perf_counter()
                        r0.gc_seconds["device"] += t1 - t0
                        r0.gc_seconds["host_cuts"] += t2 - t1
                        r0.gc_seconds[f"host_cuts_layer{li}"] += t2 - t1
                        r0.gc_seconds["h2d"] += t3 - t2
        r0._sync()

    def run(self, pm_iterations, iterations=0, graph_cut=None):
        """FastGCStereo::run for one view (LES/FastGCStereo.h:133-199): init, pmInit winner-take-all iterations, then
        `iterations` graph-cut iterations (their counter restarts at 0)."""
        self.init_labels()
        for it in range(pm_iterations):
            self.iteration(it)
        if iterations > 0:
            self.begin_gc(graph_cut)
            for it in range(iterations):
                self.gc_iteration(it)
            self.sync_gc_state()
        return self.labels, self.cur

    def disparities(self):
        ys, xs = torch.meshgrid(torch.arange(self.H, device=self.device, dtype=torch.float32),
                                torch.arange(self.W, device=self.device, dtype=torch.float32), indexing="ij")
        return self.labels[..., 0] * xs + self.labels[..., 1] * ys + self.labels[..., 2]

    def close(self):
        if getattr(self, "_joint", None) is not None:
            for jb in self._joint["batches"].values():
                jb.destroy()
            self._joint = None
        for sh in [s_ for layer in self.shards for s_ in layer] + [self.init]:
            sh.batch.destroy()
            if sh.xchg is not None:
                sh.xchg.destroy()
