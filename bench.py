#!/usr/bin/env python
"""bench.py -- headline benchmark of the matching-cost hot path (BASELINE.json metric).

Workload at N = 1 (BASELINE.json configs[2], the configuration the metric is quoted on; BASELINE.md "H1"):
whole-image guided-filter cost aggregation of 256 fronto-parallel hypothesis planes (c = k) against a
synthetic 1500 x 1000 x 256 float32 U[0,1) cost volume, windR = 20 (guided-filter radius 10),
eps = 1e-4, th_col = 0.5.  One "step" = one pass over all 256 hypotheses = W*H*D = 384 M cost
evaluations (gather the volume taps -> truncate -> colour guided filter), inputs resident in HBM.

Multi-GPU (--gpus N > 1, launched by torch.distributed.run, one rank per GPU): the shard of BASELINE.json configs[4] --
hypotheses (disparity slices) are independent, so every rank aggregates 64 slices of a 3000 x 2000 image (= 384 M evaluations,
the evaluation count of the 1-GPU workload) with no data-path collective and replicated guide statistics; at N = 8 the ranks
together hold exactly the 3000 x 2000 x 512 volume of configs[4].  Per-GPU work is fixed for every N: "weak" scaling (the
default; `--scaling strong` splits the 512 slices of configs[4] as 512 / N per rank instead -- the two coincide at N = 8 -- and the
record says which one ran).  N = 1 stays on configs[2] (the configuration the metric is quoted on) and also reports the per-rank
shape of the N > 1 runs as the sub-record `n8_rank_shape`, so that a 1 -> 8 curve has a like-for-like baseline.  At N > 1 the line
also carries `e2e_sharded`: BASELINE configs[3] -- the two-view optimiser at the Adirondack-H shape with the views split over two
rank groups, the cells of every disjoint set sharded inside a group and one RCCL all-gather of the updated tiles per set -- so
that the first multi-GPU run measures the path that HAS a collective, not only the collective-free one.

At N = 1 the JSON line also carries the H2 (slanted planes, two volume taps) and H3 (LayerManager cell batches, the
optimiser's geometry) measurements of the same build as sub-records (`h2`, `h3`), the copy ceiling measured in the same run
(`roofline.peak_achievable`) and, with --e2e 1 (default), the end-to-end wall-clock of the MidV3 loop at the Adirondack-H shape
(`e2e`: one view and two views + post-processing; north_star's third target).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import math
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # before any OpenMP runtime starts (as tools/e2e_bench.py does): the e2e sub-record's two cut teams
                                                         # sleep between lock-steps instead of spinning against each other

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GUIDE_GBS = 6290.0  # float4-copy ceiling quoted by the guide (79 % of peak); the run measures its own, see copy_ceiling()


def kernel_source_hash():
    """sha1 over the sources that determine the unary-cost kernels and their launches (the kernel headers + les_hip_march_tables.inc: the instantiations +
    les_hip_march.inc: job tables, launches, per-view set-up; the rest of the C ABI -- context, proposers, cuts, exchange -- does not touch them):
    profiles/traffic.json is only quoted when it was collected for this very code."""
    h = hashlib.sha1()
    for f in ("les_march.h", "les_kernels.h", "les_simt.h", "les_hip_march_tables.inc", "les_hip_march.inc"):
        h.update(open(os.path.join(ROOT, "localexpstereo_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="h1", choices=["h1", "h2", "h3", "h3b"])
    ap.add_argument("--height", type=int, default=0, help="default: 1000 at N = 1 (configs[2]), 2000 at N > 1 (configs[4])")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--ndisp", type=int, default=0, help="slices per rank; default 256 at N = 1, 64 at N > 1 (weak) or 512 / N (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = 64 slices of 3000 x 2000 per rank whatever N (default); strong = the 512 slices of configs[4] split as 512 / N per rank")
    ap.add_argument("--e2e-sharded", type=int, default=1, help="N > 1: 1 = add the sharded two-view end-to-end record (configs[3]); 0: skip it")
    ap.add_argument("--e2e", type=int, default=1, help="1: add the end-to-end sub-record (N = 1, ~15 s); 0: skip it")
    ap.add_argument("--sub-steps", type=int, default=20, help="steps of the H2 / H3 sub-records at N = 1 (0: skip them)")
    ap.add_argument("--cpu-planes", type=int, default=-1, help="planes of the CPU-baseline sample (-1: auto, 0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from localexpstereo_amd import api, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # LES_BENCH_BACKEND=gloo LES_BENCH_ONE_DEVICE=1: functional test of the N > 1 code path on a box with one GPU (all ranks on
    # cuda:0, rendezvous and reductions over gloo); never the measured configuration
    backend = os.environ.get("LES_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("LES_BENCH_ONE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    multi = world > 1
    H = args.height or (2000 if multi else 1000)
    W = args.width or (3000 if multi else 1500)
    D = args.ndisp or ((64 if args.scaling == "weak" else max(1, 512 // world)) if multi else 256)      # slices of this rank
    P = H * W
    # ---- synthetic inputs (seeded); the volume shard is generated directly in HBM
    guide = synth.make_guide(H, W, 1234)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    vol = torch.rand((D, H, W), device=dev, dtype=torch.float32, generator=gen)
    e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1,
                                device=dev_index, volumes_on_device=True, shape=(D, H, W))
    stream = torch.cuda.current_stream(dev)
    e.set_stream(stream.cuda_stream)
    out = torch.empty((D, H, W), device=dev, dtype=torch.float32)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    def make_workload(name):
        """-> (step function, evaluations per step of this rank, algorithmic bytes per evaluation, representative batch,
        description, planes or None)"""
        if name in ("h3", "h3b"):
            # H3: the optimiser's geometry -- LayerManager cells of units 1 % / 3 % / 9 % of the width, one
            # random plane per cell and proposal slot (9 / 3 / 3 per cell), one launch per disjoint set and slot; the contract
            # number counts filter-domain pixels x hypotheses
            from localexpstereo_amd import pm
            rng = np.random.default_rng(7 + rank)
            batches, evals = [], 0
            for unit, slots in zip((int(W * 0.01), int(W * 0.03), int(W * 0.09)), (9, 3, 3)):
                units_, shared, filt, sets = pm.layer_geometry(W, H, 20, unit)
                for cells in sets:
                    # h3b: the proposal slots of a set in ONE launch, slot s into cost map s (out_slabs = cells per slot) -- the same evaluations
                    # in 48 launches instead of 240.  NOT what the optimiser does (its proposals depend on the previous fusion, LES/FastGCStereo.h:
                    # 30-64): it shows what the kernel delivers once a launch fills the GPU.
                    b = (api.Batch(e, np.tile(filt[cells], slots), np.tile(shared[cells], slots), out_slabs=len(cells)) if name == "h3b"
                         else api.Batch(e, filt[cells], shared[cells]))
                    pl = np.zeros((slots, len(cells), 4), np.float32)
                    pl[..., 0] = rng.uniform(-0.05, 0.05, pl.shape[:2]); pl[..., 1] = rng.uniform(-0.05, 0.05, pl.shape[:2])
                    cx, cy = shared[cells]["x"] + shared[cells]["w"] / 2, shared[cells]["y"] + shared[cells]["h"] / 2
                    pl[..., 2] = rng.uniform(0.2, 0.8, pl.shape[:2]) * (D - 1) - pl[..., 0] * cx - pl[..., 1] * cy
                    batches.append((b, [torch.from_numpy(pl.reshape(-1, 4)).to(dev)] if name == "h3b" else [torch.from_numpy(pl[k]).to(dev) for k in range(slots)]))
                    evals += slots * int(sum(int(f["w"]) * int(f["h"]) for f in filt[cells]))
            nl = sum(len(p) for _, p in batches)

            def step():
                for b, pls in batches:
                    for p in pls:
                        b.run(p.data_ptr(), out.data_ptr(), mode=0, check=True, planes_on_device=True)
            desc = (f"H3: LayerManager cells (units 1/3/9 % of W={W}), 9/3/3 random planes per cell, one launch per disjoint set and "
                    f"slot ({nl} launches per step), filter-domain evaluations counted; volume {W}x{H}x{D} f32 U[0,1)")
            if name == "h3b":
                desc = (f"H3 with the proposal slots of a set batched: the same cells, planes and evaluations as H3, one launch per disjoint set with all its "
                        f"9/3/3 slots, slot s into cost map s ({nl} launches per step instead of 240).  An upper bound for the kernel on this geometry, not "
                        f"the optimiser's schedule: its proposals depend on the previous fusion (LES/FastGCStereo.h:30-64)")
            return step, float(evals), 12.0, batches[0][0], desc, None
        if name == "h1":
            planes = synth.fronto_planes(D)
            bpe = 8.0             # 4 B raw cost read + 4 B aggregated cost written
        else:
            planes = synth.slanted_planes(D, H, W, D - 1, seed=7 + rank)
            bpe = 12.0            # two volume taps + write
        d_planes = torch.from_numpy(planes).to(dev)
        full = [(0, 0, W, H)] * D
        batch = api.Batch(e, full, full, out_slabs=True)

        def step():
            batch.run(d_planes.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)
        desc = (f"{name.upper()}: {D} {'fronto-parallel' if name == 'h1' else 'slanted'} planes x {W}x{H} image, volume {W}x{H}x{D} f32 "
                f"U[0,1) per GPU, windR=20 (GF radius 10), eps=1e-4, th_col=0.5")
        return step, float(P) * D, bpe, batch, desc, planes

    def measure(step, steps, warmup):
        for _ in range(warmup):
            step()
        barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in evs:
            a.record(stream)
            step()
            b.record(stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        barrier()
        elapsed = torch.tensor([t1 - t0], device=red_dev, dtype=torch.float64)
        kern_ms = torch.tensor([sum(a.elapsed_time(b) for a, b in evs) / max(1, steps)], device=red_dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
            dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
        return float(elapsed.item()), float(kern_ms.item())

    def copy_ceiling():
        """GB/s (read + written) of a 16-byte-per-lane device copy of 2 x 1 GiB, best of 5 launches after a warm-up: the streaming
        ceiling of THIS box, measured in this run."""
        n = 1 << 28
        a = torch.empty(n, device=dev, dtype=torch.float32).normal_()
        b = torch.empty(n, device=dev, dtype=torch.float32)
        best = 0.0
        for it in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = e.L.les_hip_calib_copy_wide(a.data_ptr(), b.data_ptr(), n, dev_index, stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            if rc != 0:
                return None
            if it > 0:
                best = max(best, 2.0 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del a, b
        return best

    step, evals_rank, bytes_per_eval, batch, desc, planes = make_workload(args.workload)
    elapsed, kern_ms = measure(step, args.steps, args.warmup)
    kind = batch.kernel_kind(0)
    kernel_name = ("les_march_kernel<R=10> (gather + guided filter fused: exact integer box sums, wave-specialised pipeline)" if kind == 1
                   else "les_strip_kernel<R=10> (gather + guided filter fused, fp64 running sums)")

    evals_per_step = evals_rank * world
    value = evals_per_step * args.steps / elapsed / 1e6
    alg_bytes = evals_rank * bytes_per_eval + float(P) * 48.0        # per step and rank (H1 / H2: one launch)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    ceiling = copy_ceiling()

    # HBM bytes per launch from the rocprofv3 PMC passes of the same command (FETCH_SIZE + WRITE_SIZE, separate passes,
    # tools/collect_profiles.sh): cannot be collected live inside this process, so the committed measurement is quoted --
    # only when it was taken on exactly these kernel sources and this shape.
    traffic, traffic_source, co_bounds = None, None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            t = json.load(open(tf)).get(args.workload)
            if t and t.get("kernel_source_sha1") == kernel_source_hash() and [H, W, D] == list(t.get("shape", ())):
                traffic = t["bytes_per_launch"]
                traffic_source = t.get("source")
                co_bounds = t.get("co_bounds")
            elif t:
                traffic_source = (f"profiles/traffic.json was collected on kernel sources {t.get('kernel_source_sha1')} / shape {t.get('shape')}, "
                                  f"not on this run's ({kernel_source_hash()} / {[H, W, D]}): not quoted")
        except Exception as ex:
            traffic, traffic_source = None, f"profiles/traffic.json unreadable: {ex!r}"

    # What actually bounds the kernel: it is frozen at this formulation, and a reader of the line alone should see why `frac` ends where
    # it does.  From the committed counters: VALU wave-instructions per launch x 64 lanes / evaluations = lane-operations per evaluation; the issue floor
    # is those instructions at the measured issue costs with every wait hidden; the kernel runs at kernel_ms / floor of it.
    bound_actual = None
    if co_bounds and co_bounds.get("valu_insts_per_launch") and args.workload in ("h1", "h2"):
        lane_ops = co_bounds["valu_insts_per_launch"] * 64.0 / evals_rank
        floor_ms = co_bounds.get("valu_issue_floor_ms")
        bound_actual = {
            "bound_actual": "valu issue",
            "lane_ops_per_eval": round(lane_ops, 1),
            "valu_issue_floor_ms": floor_ms,
            "frac_of_issue_floor": round(floor_ms / kern_ms, 4) if floor_ms else None,
            "hbm_frac_at_issue_floor": round(alg_bytes / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if floor_ms else None,
            "note": ("the HBM roofline is the wrong ceiling for this formulation: at %.0f VALU lane-operations per evaluation (exact integer box sums: 8 box filters = "
                     "4 vertical running sums + 4 prefix-sum passes, the 3x3 algebra, fixed-point conversions) the 1024 SIMDs need valu_issue_floor_ms with every "
                     "wait hidden, i.e. `frac` could reach hbm_frac_at_issue_floor at most; >= 0.50 needs <= 60 lane-operations per evaluation (LES/GuidedFilter.h:142-247 "
                     "has 8 box filters + 25 multiply-adds per pixel: no such formulation is known to us; the i8-MFMA box sums measured slower, profiles/round3_mfma_box.log)" % lane_ops),
        }

    result = {
        "metric": "Mcost-evals/s (pixels x hypotheses / s), guided-filter cost aggregation, 1500x1000x256 vol",
        "value": round(value, 2),
        "unit": "Mcost-evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling if multi else "weak",
        "vs_baseline": None,
        "dtype": "i32 fixed-point box sums (exact; i64 stage-2 window sums), f32 3x3 algebra and combination" if kind == 1 else "f64 sums / f32 algebra",
        "data": "synthetic",
        "config": {
            "workload": desc + (f"; the per-rank shard of BASELINE configs[4] ({D} slices of {W}x{H} per rank, {D * world} in total; --scaling {args.scaling})" if multi else "; BASELINE configs[2]"),
            "evals_per_step_per_gpu": int(evals_rank),
            "sharding": "hypotheses (disparity slices) split across ranks, no data-path collective",
            "workgroups_per_launch": batch.num_jobs,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "peak_achievable": round(ceiling, 1) if ceiling else None,       # 16-byte-per-lane copy measured in THIS run (les_hip_calib_copy_wide)
            "frac_of_achievable": round(achieved / ceiling, 5) if ceiling else None,
            "peak_achievable_guide": HBM_ACHIEVABLE_GUIDE_GBS,
            "traffic": traffic,
            # where `traffic` and `co_bounds` come from: rocprofv3 --pmc passes of this very command (tools/collect_profiles.sh), quoted from the
            # committed file only when it was taken on exactly these kernel sources and this shape (counters cannot be read inside this process)
            "traffic_source": traffic_source,
            # what else bounds the kernel besides HBM -- instruction issue (the VALU instructions per launch at the measured issue
            # costs of the two instruction classes, with every wait hidden), the share of instructions outside the dual-issue class, how busy
            # the LDS pipe is, and the occupancy the 155 KB of LDS per workgroup leave
            "co_bounds": co_bounds,
            "bound_actual": bound_actual["bound_actual"] if bound_actual else None,
            "lane_ops_per_eval": bound_actual["lane_ops_per_eval"] if bound_actual else None,
            "frac_of_issue_floor": bound_actual["frac_of_issue_floor"] if bound_actual else None,
            "hbm_frac_at_issue_floor": bound_actual["hbm_frac_at_issue_floor"] if bound_actual else None,
            "bound_note": bound_actual["note"] if bound_actual else None,
            "kernel": kernel_name,
            "kernel_ms": round(kern_ms, 4),
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_source_sha1": kernel_source_hash(),
        },
    }

    # ---- sub-records: the other two workloads of SURVEY.md 8(d) on the same context
    if world == 1 and args.sub_steps > 0 and args.workload == "h1":
        sub_steps_by_name = {}
        for name in ("h2", "h3", "h3b"):
            st, ev, bpe, bt, ds, _ = make_workload(name)
            sub_steps_by_name[name] = st
            el, km = measure(st, args.sub_steps, 2)
            ab = ev * bpe + float(P) * 48.0
            result["h3_batched" if name == "h3b" else name] = {
                "workload": ds,
                "ms_per_step": round(el / args.sub_steps * 1e3, 4),
                "value": round(ev * args.sub_steps / el / 1e6, 2),
                "unit": "Mcost-evals/s",
                "steps": args.sub_steps,
                "algorithmic_GBps": round(ab / (km * 1e-3) / 1e9, 2),
                "frac": round(ab / (km * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "kernel": "march" if bt.kernel_kind(0) == 1 else "strip",
                "workgroups_first_launch": bt.num_jobs,
            }
            # HBM bytes per step from the committed counter passes of THIS workload (tools/pmc_workload.sh + tools/traffic_merge.py), quoted only for these sources and shape
            try:
                tw = json.load(open(tf)).get(name) if os.path.exists(tf) else None
                if tw and tw.get("kernel_source_sha1") == kernel_source_hash() and [H, W, D] == list(tw.get("shape", ())):
                    result[name]["traffic"] = tw["bytes_per_step"]
                    result[name]["traffic_over_algorithmic"] = round(tw["bytes_per_step"] / ab, 3)
                    result[name]["traffic_source"] = tw.get("source")
            except Exception:
                pass
            if name == "h3":
                # How much of what a finest-layer launch pays for is used: a cell's filter tile is 85 of the 128 lanes of its
                # job slot, 85 rows are 12.1 of the 16 ticks of 7 rows the march takes (3 of them pipeline fill), and the set's workgroups (two cells
                # each) are one round on fewer than the 256 CUs.  The product bounds any re-packing of these dependent launches.
                u0 = int(W * 0.01)
                fw = 3 * u0 + 40
                wgs = bt.num_jobs
                result["h3"]["useful_lane_row_share"] = round((fw / 128.0) * ((fw / 7.0) / (math.ceil(fw / 7.0) + 3)) * min(1.0, wgs / 256.0), 3)
                result["h3"]["useful_lane_row_share_note"] = (f"finest layer (81 % of H3's evaluations): filter tile {fw} of 128 lanes x {fw / 7.0:.1f} of {math.ceil(fw / 7.0) + 3} ticks x "
                                                              f"{wgs} of 256 CUs per dependent launch; the optimiser's schedule leaves no second launch to fill the rest")
        step()                        # the H1 result buffer is compared with the oracle below
        torch.cuda.synchronize(dev)
        saved = out[: min(D, 256)].clone() if args.cpu_planes != 0 else None

        # ---- the per-rank shape of the N > 1 runs (64 slices of 3000 x 2000) on this one GPU: baseline of a future scaling curve
        try:
            H8, W8, D8 = 2000, 3000, 64
            del batch
            vol8 = torch.rand((D8, H8, W8), device=dev, dtype=torch.float32, generator=gen)
            e8 = api.HipCostVolumeEnergy(synth.make_guide(H8, W8, 1234), None, vol8.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D8 - 1,
                                         device=dev_index, volumes_on_device=True, shape=(D8, H8, W8))
            e8.set_stream(stream.cuda_stream)
            out8 = out.view(-1)[: D8 * H8 * W8].view(D8, H8, W8)          # same byte count as the headline output
            pl8 = torch.from_numpy(synth.fronto_planes(D8)).to(dev)
            b8 = api.Batch(e8, [(0, 0, W8, H8)] * D8, [(0, 0, W8, H8)] * D8, out_slabs=True)
            el8, km8 = measure(lambda: b8.run(pl8.data_ptr(), out8.data_ptr(), mode=0, check=False, planes_on_device=True), args.sub_steps, 2)
            ev8 = float(H8) * W8 * D8
            result["n8_rank_shape"] = {
                "workload": f"H1 on the shard one rank holds at N = 8: {D8} fronto-parallel planes x {W8}x{H8} (configs[4] / 8)",
                "ms_per_step": round(el8 / args.sub_steps * 1e3, 4), "value": round(ev8 * args.sub_steps / el8 / 1e6, 2), "unit": "Mcost-evals/s",
                "steps": args.sub_steps, "kernel": "march" if b8.kernel_kind(0) == 1 else "strip", "workgroups_per_launch": b8.num_jobs,
                "algorithmic_GBps": round((ev8 * 8.0 + float(H8) * W8 * 48.0) / (km8 * 1e-3) / 1e9, 2),
            }
            # the slanted-plane pass (H2) on the same shard: 64 planes with slopes in +-1/2, centre disparity inside the shard's range -- steep planes
            # take their taps from the tiled copy of the shard (1.5 GB more per rank)
            pl8s = torch.from_numpy(synth.slanted_planes(D8, H8, W8, D8 - 1, seed=7)).to(dev)
            el8s, km8s = measure(lambda: b8.run(pl8s.data_ptr(), out8.data_ptr(), mode=0, check=False, planes_on_device=True), args.sub_steps, 2)
            result["n8_rank_shape_h2"] = {
                "workload": f"H2 on the shard one rank holds at N = 8: {D8} slanted planes (slopes +-1/2) x {W8}x{H8}",
                "ms_per_step": round(el8s / args.sub_steps * 1e3, 4), "value": round(ev8 * args.sub_steps / el8s / 1e6, 2), "unit": "Mcost-evals/s",
                "steps": args.sub_steps, "algorithmic_GBps": round((ev8 * 12.0 + float(H8) * W8 * 48.0) / (km8s * 1e-3) / 1e9, 2),
                "tiled_copy_bytes": e8.tiled_volume_bytes(0),
            }
            b8.destroy(); e8.close()
            del b8, e8, vol8, pl8s
            # ---- configs[4] itself on ONE GPU (3000 x 2000 x 512: 12.3 GB, 3.07e9 floats -- beyond 32-bit element offsets): 64 slanted planes whose
            # disparity stays inside the range over the image ("tame": what the short gather needs at this size), taps from the tiled copy (12.3 GB more; round 5:
            # the kernel's descriptor starts at the first row a job gathers, so the copy may have any size) against taps from [D][H][W]
            try:
                free_b, _ = torch.cuda.mem_get_info()
                D4 = 512
                if free_b > 60 * 2**30:
                    vol4 = torch.empty((D4, H8, W8), device=dev, dtype=torch.float32)
                    for d0 in range(0, D4, 64):
                        vol4[d0:d0 + 64].uniform_(0.0, 1.0, generator=gen)
                    rng4 = np.random.default_rng(11)
                    pl4 = np.zeros((64, 4), np.float32)
                    pl4[:, 0] = rng4.uniform(-0.06, 0.06, 64) * np.where(np.arange(64) % 2, 1.0, 0.3)     # |a| up to 0.06: 180 slices across the image
                    pl4[:, 1] = rng4.uniform(-0.03, 0.03, 64)
                    pl4[:, 2] = 256.0 - pl4[:, 0] * (W8 / 2) - pl4[:, 1] * (H8 / 2) + rng4.uniform(-20, 20, 64)
                    rec4 = {"workload": f"configs[4] on one GPU: 64 slanted planes (|a| <= 0.06, disparity inside [0, 511] over the image) x {W8}x{H8}, volume {W8}x{H8}x{D4} (12.3 GB)"}
                    full4 = [(0, 0, W8, H8)] * 64
                    d_pl4 = torch.from_numpy(pl4).to(dev)
                    outs4 = {}
                    for tag, env in (("tiled", None), ("planar", "0")):
                        if env is None:
                            os.environ.pop("LES_HIP_TILED", None)
                        else:
                            os.environ["LES_HIP_TILED"] = env
                        e4 = api.HipCostVolumeEnergy(synth.make_guide(H8, W8, 1234), None, vol4.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D4 - 1,
                                                     device=dev_index, volumes_on_device=True, shape=(D4, H8, W8))
                        e4.set_stream(stream.cuda_stream)
                        b4 = api.Batch(e4, full4, full4, out_slabs=True)
                        el4, km4 = measure(lambda: b4.run(d_pl4.data_ptr(), out8.data_ptr(), mode=0, check=False, planes_on_device=True), max(2, args.sub_steps // 4), 1)
                        rec4[tag] = {"ms_per_step": round(km4, 4), "algorithmic_GBps": round((64.0 * H8 * W8 * 12.0 + float(H8) * W8 * 48.0) / (km4 * 1e-3) / 1e9, 2),
                                     "tiled_copy_bytes": e4.tiled_volume_bytes(0), "kernel": "march" if b4.kernel_kind(0) == 1 else "strip"}
                        outs4[tag] = out8[:8].clone()
                        b4.destroy(); e4.close()
                    os.environ.pop("LES_HIP_TILED", None)
                    rec4["tiled_equals_planar_bit_for_bit"] = bool(torch.equal(outs4["tiled"], outs4["planar"]))
                    result["config4_h2_one_gpu"] = rec4
                    del vol4, outs4
                    torch.cuda.empty_cache()
            except Exception as ex:
                result["config4_h2_one_gpu"] = {"error": repr(ex)}
        except Exception as ex:                      # never lose the headline line to a sub-record
            result["n8_rank_shape"] = {"error": repr(ex)}
        if saved is not None:
            out[: saved.shape[0]].copy_(saved)
            del saved

        # ---- end to end (north_star: Adirondack-H wall-clock < 10 s): the MidV3 loop of tools/e2e_bench.py on a synthetic pair of the Adirondack-H shape
        if args.e2e:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import e2e_bench
                rec = {}
                # two scenes: "objects" (nine small objects over a background plane: the easy one for the cuts) and "three_surfaces" (the C++ host demo's
                # scene, host/DemoScene.h: three large slanted surfaces, proposals flip most of a coarse cell at once -- the hard one), one and two views each
                for scene in ("objects", "three_surfaces"):
                    rec[scene] = {}
                    for dual in (0, 1):
                        r = e2e_bench.run(dual=dual, quiet=True, scene=scene)
                        rec[scene]["dual" if dual else "single"] = {k: r[k] for k in ("seconds_total_including_ingest", "seconds_optimiser", "seconds_reference_clock", "gc_seconds", "tiled_locksteps", "shape", "iterations", "pm_iterations") if k in r}
                rec["single"], rec["dual"] = rec["objects"]["single"], rec["objects"]["dual"]           # (the keys of rounds 3-4: the "objects" scene)
                rec["all_under_10_s"] = all(rec[sc][v]["seconds_optimiser"] < 10.0 for sc in ("objects", "three_surfaces") for v in ("single", "dual"))
                from localexpstereo_amd.gc import cpu_budget
                rec["host_cpus"] = cpu_budget()
                rec["note"] = ("synthetic pairs at the Adirondack-H shape (the data set is not in the container); MidV3 defaults: layers 14/43/129, 2 PatchMatch + 5 graph-cut iterations; "
                               "every cut on the GPU since round 5 (finest layer: one workgroup per cell in LDS; coarse layers: the tiled solver, gc_seconds.tiled_*); "
                               "seconds_optimiser counts from the top of run() (layers, job tables, host energy context, label initialisation included), seconds_reference_clock from where the "
                               "reference starts its timer (after initCurrentFast, LES/FastGCStereo.h:141); both exclude the evaluator")
                result["e2e"] = rec
            except Exception as ex:
                result["e2e"] = {"error": repr(ex)}

    # ---- N > 1: BASELINE configs[3], the path WITH a collective -- the two-view optimiser at the Adirondack-H shape, views split over two rank
    # groups, the cells of every disjoint set sharded inside a group, one all-gather of the updated tiles per set (RCCL over xGMI)
    if multi and args.e2e_sharded and args.workload == "h1":
        rec = None
        # A collective that one rank never reaches would hang the whole job and lose the headline line with it (this leg has never run on more
        # than one GPU).  A watchdog bounds it: after LES_BENCH_E2E_TIMEOUT seconds (default 240; the one-GPU two-view run takes 6) rank 0 prints
        # the line with an error in place of the sub-record and every rank leaves.
        import threading
        leg_done = threading.Event()

        def watchdog():
            if leg_done.wait(timeout=float(os.environ.get("LES_BENCH_E2E_TIMEOUT", "240"))):
                return
            if rank == 0:
                result["e2e_sharded"] = {"error": "timeout: the sharded end-to-end leg did not finish (a rank failed or a collective hung); headline measured before it"}
                print(json.dumps(result), flush=True)
                os._exit(0)                                      # (the headline line is out: the driver's parse succeeds, the error is in the record)
            os._exit(3)                                          # every other rank: a hung collective is a failure, not a clean exit
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            # the headline's context and buffers make room (1.5 GB volume + its tiled copy + output): the batch and the context are released
            # explicitly -- `step` (a closure) still refers to them, and a live context must not outlive the volume it points into
            torch.cuda.synchronize(dev)
            batch.destroy()
            e.close()
            del step, batch, out, vol, e
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import e2e_bench
            small = os.environ.get("LES_BENCH_ONE_DEVICE") == "1"          # the one-GPU functional test runs a small scene
            shape = dict(width=360, height=240, ndisp=32, iterations=1, pm_iterations=1) if small else {}
            barrier()
            t0 = time.perf_counter()
            mine = e2e_bench.run_sharded(rank, world, f"cuda:{dev_index}", **shape)
            barrier()
            wall = torch.tensor([time.perf_counter() - t0], device=red_dev, dtype=torch.float64)
            dist.all_reduce(wall, op=dist.ReduceOp.MAX)
            allrec = [None] * world
            dist.all_gather_object(allrec, mine)
            rec = {
                "workload": ("BASELINE configs[3] on a synthetic pair at the Adirondack-H shape (1436x992x256; the data set is not in the container): MidV3 defaults, 2 PatchMatch + 5 "
                             "graph-cut iterations, doDual = 1; ranks split into one group per view, the cells of every disjoint set sharded inside a group, one all-gather of the "
                             "updated tiles (labels 16 B/px + cost 4 B/px) per set, one broadcast per view of its final label map, post-processing replicated") if not small else
                            "functional test shape 360x240x32, 1 PatchMatch + 1 graph-cut iteration",
                "seconds": round(float(wall.item()), 3),                                   # scene ingest + optimiser + post-processing, max over ranks
                "seconds_optimiser_max": max(r["seconds_optimiser"] for r in allrec),
                "bytes_exchanged_per_rank": [r["bytes_exchanged"] for r in allrec],
                "all_gathers_per_rank": [r["all_gathers"] for r in allrec],
                "exchange_seconds_per_rank": [r.get("exchange_seconds") for r in allrec],     # device time in pack -> all-gather -> unpack (events around each exchange)
                "tiled_cut_seconds_per_rank": [round(sum(v for k, v in r.get("gc_seconds", {}).items() if k.startswith("tiled_seconds")), 3) for r in allrec],
                "host_cut_seconds_per_rank": [r["host_cut_seconds"] for r in allrec],
                "bad_all_last": next((r["bad_all_last"] for r in allrec if r["bad_all_last"] is not None), None),
                "host_cpus_shared_by_all_ranks": os.cpu_count(),
            }
        except Exception as ex:                      # never lose the headline line to a sub-record
            rec = {"error": repr(ex)}
        result["e2e_sharded"] = rec
        if rank == 0 and "error" in rec:
            # (the other ranks may be waiting in a collective this rank left: print now, the watchdogs end them)
            print(json.dumps(result), flush=True)
            os._exit(0)
        leg_done.set()

    # ---- CPU baseline: the oracle (CPU restatement, double guided filter like the reference default),
    # rank 0 at N = 1 only, on a bounded sample of the same workload: the first `ns` hypotheses.
    if rank == 0 and world == 1 and args.cpu_planes != 0 and planes is not None:
        from oracle import oracle as om
        # threads = the CPUs this process may keep busy: a cgroup CPU-time quota counts (the MI355X box shows 256 hardware threads
        # and grants 16 CPUs of time per period; 256 threads then spend most of every period throttled: 111 instead of 522 Mevals/s)
        from localexpstereo_amd.gc import cpu_budget
        cores = cpu_budget()
        ns = args.cpu_planes if args.cpu_planes > 0 else D - 1          # (almost) the whole workload: ~12 CPU-seconds on 16 cores
        ns = min(ns, D - 1)
        vol_host = vol.cpu().numpy() if args.sub_steps > 0 else vol[: ns + 1].cpu().numpy()      # (the H2 / H3 samples below gather from every slice)
        o = om.Oracle(guide, None, vol_host, None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1)
        o.aggregate_planes(planes[: min(ns, cores)], nthreads=cores)                 # warm the thread scratch
        c0 = time.perf_counter()
        ref = o.aggregate_planes(planes[:ns], nthreads=cores)
        c1 = time.perf_counter()
        s0 = time.perf_counter()
        o.aggregate_planes(planes[:1], nthreads=1)
        s1 = time.perf_counter()
        got = out[:ns].cpu().numpy()
        err = float(np.max(np.abs(got.astype(np.float64) - ref)))
        # north_star states the tolerance as RELATIVE (1e-4); the kernel's error is absolute (a fixed-point step that scales with th_col - vmin), so the
        # relative error is largest on the smallest costs: reported here for the costs below 1 % of th_col (none on a U[0,1) volume after a 441-pixel average:
        # the count says so) and, as the floor of the claim, the smallest cost above which every evaluation of the sample is within 1e-4 relative
        small = ref < 0.01 * 0.5
        rel_all = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref.astype(np.float64)), 1e-30)
        bad = rel_all > 1e-4
        rel_floor = float(ref[bad].max()) if bad.any() else 0.0
        rel_small = float(rel_all[small].max()) if small.any() else None
        n_small = int(small.sum())
        del rel_all, bad, small
        result["cpu_baseline"] = {
            "value": round(ns * P / (c1 - c0) / 1e6, 2),
            "unit": "Mcost-evals/s",
            "cores": cores,
            "kind": "port",
            "sample": f"first {ns} of the {D} hypotheses of the same workload ({ns * P / 1e6:.0f} M evals, {c1 - c0:.2f} s wall = {(c1 - c0) * cores:.0f} core-seconds, "
                      f"OpenMP over hypotheses, double-precision guided filter as the reference default)",
            "single_thread_value": round(P / (s1 - s0) / 1e6, 2),
            "gpu_vs_oracle_max_abs_err_on_sample": err,
            "gpu_vs_oracle_max_rel_err_on_costs_below_1pct_of_th": rel_small,
            "costs_below_1pct_of_th_in_sample": n_small,
            "smallest_cost_with_rel_err_above_1e-4": rel_floor,          # 0.0: every evaluation of the sample is within 1e-4 relative
        }
        del ref, got
        # ---- the same for the other two workloads of SURVEY 8(d) (bounded samples: a few CPU-seconds each)
        if args.sub_steps > 0 and args.workload == "h1":
            try:
                n2 = D                                                      # the whole H2 workload: ~1 s on 16 cores
                pl2 = synth.slanted_planes(D, H, W, D - 1, seed=7 + rank)[:n2]
                c0 = time.perf_counter()
                ref2 = o.aggregate_planes(pl2, nthreads=cores)
                c1 = time.perf_counter()
                result["cpu_baseline"]["h2"] = {"value": round(n2 * P / (c1 - c0) / 1e6, 2), "unit": "Mcost-evals/s", "cores": cores,
                                                "sample": f"the {n2} slanted planes of H2 ({n2 * P / 1e6:.0f} M evals, {c1 - c0:.2f} s, OpenMP over hypotheses)"}
                # the GPU's H2 pass against these slabs: the only place where the tiled-copy gather (steep planes, role A's KIND 5) meets the oracle at configs[2]'s size
                sub_steps_by_name["h2"]()
                torch.cuda.synchronize(dev)
                got2 = out[:n2].cpu().numpy()
                d2 = np.abs(got2.astype(np.float64) - ref2)
                result["h2"]["gpu_vs_oracle_max_abs_err"] = float(d2.max())
                result["h2"]["gpu_vs_oracle_max_rel_err"] = float((d2 / np.maximum(np.abs(ref2), 1e-30)).max())
                result["h2"]["oracle_slabs_compared"] = int(n2)
                result["h2"]["steep_planes_in_sample"] = int((np.abs(pl2[:, 0]) >= 0.05).sum())
                del ref2, got2, d2
                from localexpstereo_amd import pm
                rng3 = np.random.default_rng(7 + rank)
                ev3, t3, nl3, err3, px3 = 0, 0.0, 0, 0.0, 0
                for unit, slots in zip((int(W * 0.01), int(W * 0.03), int(W * 0.09)), (9, 3, 3)):
                    units_, shared, filt, sets = pm.layer_geometry(W, H, 20, unit)
                    for cells in sets:                                      # every disjoint set of each layer, one of its proposal slots
                        pl = np.zeros((len(cells), 4), np.float32)
                        pl[:, 0] = rng3.uniform(-0.05, 0.05, len(cells)); pl[:, 1] = rng3.uniform(-0.05, 0.05, len(cells))
                        cx, cy = shared[cells]["x"] + shared[cells]["w"] / 2, shared[cells]["y"] + shared[cells]["h"] / 2
                        pl[:, 2] = rng3.uniform(0.2, 0.8, len(cells)) * (D - 1) - pl[:, 0] * cx - pl[:, 1] * cy
                        c0 = time.perf_counter()
                        ref3 = o.unary_batch(filt[cells], shared[cells], pl, mode=0, check=True, nthreads=cores)
                        t3 += time.perf_counter() - c0
                        got3 = e.unary_batch(filt[cells], shared[cells], pl, mode=0, check=True)      # the same lock-step on the GPU (march kernel, cell geometry)
                        w3 = ~np.isnan(ref3)
                        assert np.array_equal(w3, ~np.isnan(got3)), "H3: the written pixels differ from the oracle's"
                        assert np.array_equal(ref3[w3] == 1e6, got3[w3] == 1e6), "H3: invalid-label sentinels differ from the oracle's"
                        v3 = w3 & (ref3 != 1e6)
                        err3 = max(err3, float(np.abs(got3[v3].astype(np.float64) - ref3[v3]).max())) if v3.any() else err3
                        px3 += int(v3.sum())
                        ev3 += int(sum(int(f["w"]) * int(f["h"]) for f in filt[cells]))
                        nl3 += 1
                result["cpu_baseline"]["h3"] = {"value": round(ev3 / t3 / 1e6, 2), "unit": "Mcost-evals/s (filter-domain)", "cores": cores,
                                                "sample": f"{nl3} of the 240 lock-steps of H3: every disjoint set of each layer, one proposal slot, one plane per cell ({ev3 / 1e6:.0f} M "
                                                          f"filter-domain evals, {t3:.2f} s, OpenMP over cells as the reference does)"}
                result["h3"]["gpu_vs_oracle_max_abs_err"] = err3
                result["h3"]["oracle_locksteps_compared"] = nl3
                result["h3"]["consumed_pixels_compared"] = px3
            except Exception as ex:                  # never lose the headline line to a sub-record
                result["cpu_baseline"]["h2_h3_error"] = repr(ex)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
