#!/usr/bin/env python
"""bench.py -- headline benchmark of the matching-cost hot path (BASELINE.json metric).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; BASELINE.md "H1"):
whole-image guided-filter cost aggregation of 256 fronto-parallel hypothesis planes (c = k) against a
synthetic 1500 x 1000 x 256 float32 U[0,1) cost volume, windR = 20 (guided-filter radius 10),
eps = 1e-4, th_col = 0.5.  One "step" = one pass over all 256 hypotheses = W*H*D = 384 M cost
evaluations (gather 2 volume taps -> truncate -> colour guided filter), inputs resident in HBM.

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU): hypotheses are
independent, so they are sharded across ranks with no data-path collective ("weak" scaling: every
rank aggregates its own 256 hypotheses of a N*256-slice volume; guide statistics are replicated).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6.3 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="h1", choices=["h1", "h2", "h3"])
    ap.add_argument("--height", type=int, default=1000)
    ap.add_argument("--width", type=int, default=1500)
    ap.add_argument("--ndisp", type=int, default=256)
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    H, W, D = args.height, args.width, args.ndisp
    P = H * W
    # ---- synthetic inputs (seeded); the volume is generated directly in HBM
    guide = synth.make_guide(H, W, 1234)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    vol = torch.rand((D, H, W), device=dev, dtype=torch.float32, generator=gen)
    if args.workload == "h1":
        planes = synth.fronto_planes(D)
        bytes_per_eval = 8.0          # 4 B raw cost read + 4 B aggregated cost written
    else:
        planes = synth.slanted_planes(D, H, W, D - 1, seed=7 + rank)
        bytes_per_eval = 12.0         # two volume taps + write
    d_planes = torch.from_numpy(planes).to(dev)
    out = torch.empty((D, H, W) if args.workload != "h3" else (1, H, W), device=dev, dtype=torch.float32)

    e = api.HipCostVolumeEnergy(guide, None, vol.data_ptr(), None, windR=20, eps=1e-4, th_col=0.5, max_disp=D - 1,
                                device=local_rank, volumes_on_device=True, shape=(D, H, W))
    stream = torch.cuda.current_stream(dev)
    e.set_stream(stream.cuda_stream)
    if args.workload == "h3":
        # H3: the optimiser's geometry -- LayerManager cells of units 1 % / 3 % / 9 % of the width, one
        # random plane per cell and proposal slot (9 / 3 / 3 per cell), one launch per disjoint set and slot; the contract
        # number counts filter-domain pixels x hypotheses
        from localexpstereo_amd import pm
        rng = np.random.default_rng(7 + rank)
        batches, evals = [], 0
        for unit, slots in zip((int(W * 0.01), int(W * 0.03), int(W * 0.09)), (9, 3, 3)):
            units_, shared, filt, sets = pm.layer_geometry(W, H, 20, unit)
            for cells in sets:
                b = api.Batch(e, filt[cells], shared[cells])
                pl = np.zeros((slots, len(cells), 4), np.float32)
                pl[..., 0] = rng.uniform(-0.05, 0.05, pl.shape[:2]); pl[..., 1] = rng.uniform(-0.05, 0.05, pl.shape[:2])
                cx, cy = shared[cells]["x"] + shared[cells]["w"] / 2, shared[cells]["y"] + shared[cells]["h"] / 2
                pl[..., 2] = rng.uniform(0.2, 0.8, pl.shape[:2]) * (D - 1) - pl[..., 0] * cx - pl[..., 1] * cy
                batches.append((b, [torch.from_numpy(pl[k]).to(dev) for k in range(slots)]))
                evals += slots * int(sum(int(f["w"]) * int(f["h"]) for f in filt[cells]))
        batch = batches[0][0]
        evals_per_step_h3 = evals

        def step():
            for b, pls in batches:
                for p in pls:
                    b.run(p.data_ptr(), out.data_ptr(), mode=0, check=True, planes_on_device=True)
    else:
        full = [(0, 0, W, H)] * D
        batch = api.Batch(e, full, full, out_slabs=True)

        def step():
            batch.run(d_planes.data_ptr(), out.data_ptr(), mode=0, check=False, planes_on_device=True)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    barrier()
    elapsed = torch.tensor([t1 - t0], device=dev, dtype=torch.float64)
    kern_ms = torch.tensor([sum(a.elapsed_time(b) for a, b in evs) / max(1, args.steps)], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    kern_ms = float(kern_ms.item())

    # measured device-copy ceiling on this box (quote both denominators): dword-per-lane copy of the volume
    # shard through the library's calibration kernel, 2 x bytes moved / time
    import ctypes as C
    copy_gbs = None
    try:
        nn = int(vol.numel())
        tmp = torch.empty_like(vol)
        ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L = api.load()
        L.les_hip_calib_copy(C.c_void_p(vol.data_ptr()), C.c_void_p(tmp.data_ptr()), C.c_size_t(nn), local_rank, C.c_void_p(stream.cuda_stream))
        ca.record(stream)
        for _ in range(3):
            L.les_hip_calib_copy(C.c_void_p(vol.data_ptr()), C.c_void_p(tmp.data_ptr()), C.c_size_t(nn), local_rank, C.c_void_p(stream.cuda_stream))
        cb.record(stream)
        torch.cuda.synchronize(dev)
        copy_gbs = 3 * 2.0 * nn * 4 / (ca.elapsed_time(cb) * 1e-3) / 1e9
        del tmp
    except Exception:
        copy_gbs = None

    evals_rank = float(evals_per_step_h3) if args.workload == "h3" else float(P) * D
    evals_per_step = evals_rank * world
    value = evals_per_step * args.steps / elapsed / 1e6
    alg_bytes = evals_rank * bytes_per_eval + float(P) * 48.0        # per step and rank (H1 / H2: one launch)
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    # HBM bytes per launch from the rocprofv3 PMC passes of the same command (FETCH_SIZE + WRITE_SIZE, separate
    # passes; see profiles/): cannot be collected live inside this process, so the committed measurement is quoted.
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            t = json.load(open(tf)).get(args.workload)
            if t and t.get("strip_width") == e.strip_width() and (H, W, D) == tuple(t.get("shape", ())):
                traffic = t["bytes_per_launch"]
        except Exception:
            traffic = None

    result = {
        "metric": "Mcost-evals/s (pixels x hypotheses / s), guided-filter cost aggregation, 1500x1000x256 vol",
        "value": round(value, 2),
        "unit": "Mcost-evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"{args.workload.upper()}: {D} {'fronto-parallel' if args.workload == 'h1' else 'slanted'} planes x "
                         f"{W}x{H} image, volume {W}x{H}x{D} f32 U[0,1) per GPU, windR=20 (GF radius 10), eps=1e-4, th_col=0.5")
                        if args.workload != "h3" else
                        (f"H3: LayerManager cells (units 1/3/9 % of W={W}), 9/3/3 random planes per cell, one launch per disjoint set and "
                         f"slot ({len(batches) and sum(len(p) for _, p in batches)} launches per step), filter-domain evaluations counted; "
                         f"volume {W}x{H}x{D} f32 U[0,1)"),
            "evals_per_step_per_gpu": int(evals_rank),
            "sharding": "hypotheses (disparity slices) split across ranks, no data-path collective",
            "strip_width": e.strip_width(),
            "workgroups_per_launch": batch.num_jobs,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            # second denominator: device copy measured on this box in this run, one dword per lane like the
            # strip kernel's accesses (a float4 copy reaches ~6.3 TB/s on MI355X, MI355X_MICROARCH.md)
            "measured_dword_copy": None if copy_gbs is None else round(copy_gbs, 1),
            "frac_of_measured_dword_copy": None if not copy_gbs else round(achieved / copy_gbs, 5),
            "traffic": traffic,
            "kernel": "les_strip_kernel<R=10> (gather + guided filter fused; strip width %d)" % e.strip_width(),
            "kernel_ms": round(kern_ms, 4),
            "algorithmic_bytes_per_launch": alg_bytes,
        },
    }

    # ---- CPU baseline: the oracle (CPU restatement, double guided filter like the reference default),
    # rank 0 at N = 1 only, on a bounded sample of the same workload: the first `ns` hypotheses.
    if rank == 0 and world == 1 and args.cpu_planes != 0 and args.workload != "h3":
        from oracle import oracle as om
        cores = os.cpu_count() or 1
        ns = args.cpu_planes if args.cpu_planes > 0 else max(cores, min(D - 1, 4 * cores, 96))
        ns = min(ns, D - 1)
        vol_host = vol[: ns + 1].cpu().numpy()
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
        result["cpu_baseline"] = {
            "value": round(ns * P / (c1 - c0) / 1e6, 2),
            "unit": "Mcost-evals/s",
            "cores": cores,
            "kind": "port",
            "sample": f"first {ns} of the {D} hypotheses of the same workload ({ns * P / 1e6:.0f} M evals, {c1 - c0:.1f} s, "
                      f"OpenMP over hypotheses, double-precision guided filter as the reference default)",
            "single_thread_value": round(P / (s1 - s0) / 1e6, 2),
            "gpu_vs_oracle_max_abs_err_on_sample": err,
        }
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
