#!/usr/bin/env python
"""Replays dumped lock-steps of the coarse layers (LES_DUMP_GRAPHS=dir LES_DUMP_EVERY=n LES_DUMP_FULL=1 python tools/e2e_bench.py ->
sample_*.npz / graphs_view*_layer2.npz: regions, node offsets, payload) through the tiled device max-flow
(les_hip_batch_solve_graphs_tiled) and through the host solver (les_gc_solve_prebuilt): masks compared node for node, launches and
milliseconds per lock-step for both.

  python tools/tiled_cut_replay.py gpurun_out/ts1/*.npz [--sim] [--cells N] [--reps 5] [--threads 16]

--sim: the CPU simulator build of the same kernel sources (build container: no GPU); timings are then meaningless."""
import argparse
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402
from localexpstereo_amd import api, build, gc as lgc, synth      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--sim", action="store_true")
    ap.add_argument("--cells", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    lib = build.build_sim() if args.sim else None
    worst = 0
    for f in args.files:
        z = np.load(f)
        reg, off, pay = z["regions"], z["offsets"].astype(np.int64), z["payload"]
        n = len(reg) if not args.cells else min(args.cells, len(reg))
        reg, off = reg[:n].copy(), off[:n].copy()
        W = int(max(r["x"] + r["w"] for r in reg)); H = int(max(r["y"] + r["h"] for r in reg))
        worst = max(worst, W * H)
        e = api.HipCostVolumeEnergy(synth.make_guide(H, W, 1), synth.make_guide(H, W, 2), np.zeros((2, H, W), np.float32), np.zeros((2, H, W), np.float32),
                                    windR=20, eps=1e-4, th_col=0.5, lib=lib)
        trs = np.ascontiguousarray(reg).view(api.RECT_DT).reshape(-1)
        batch = api.Batch(e, trs, trs)
        boff, nn = batch.graph_offsets(), batch.graph_nodes()
        p = np.zeros((nn, 5), np.float32)
        for i in range(n):
            k = int(reg[i]["w"]) * int(reg[i]["h"])
            p[boff[i]: boff[i] + k] = pay[off[i] * 5:(off[i] + k) * 5].reshape(k, 5)
        p = np.ascontiguousarray(p.reshape(-1))
        dp, dm, ds = api.DeviceBuffer(e, nn * 20), api.DeviceBuffer(e, nn), api.DeviceBuffer(e, 4 * n)
        ws = api.DeviceBuffer(e, batch.tiled_workspace_bytes())
        dp.upload(p)
        ms, launches = [], 0
        for _ in range(1 if args.sim else args.reps + 1):
            e.synchronize()
            t0 = time.perf_counter()
            launches = batch.solve_graphs_tiled(dp.ptr, dm.ptr, ds.ptr, ws.ptr, ws.nbytes)
            e.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
        st = ds.download((n,), np.int32)
        dev = dm.download((nn,), np.uint8) != 0
        host = np.zeros(nn, np.uint8)
        hms = []
        for _ in range(args.reps + 1):
            t0 = time.perf_counter()
            lgc.solve_prebuilt(trs, p, boff, host, nthreads=args.threads)
            hms.append(1e3 * (time.perf_counter() - t0))
        diff = int((dev != (host != 0)).sum())
        print(f"{os.path.basename(f)}: {n} cells, {nn} nodes, status {int(st.any())}, launches <= {launches}, device {min(ms):.2f} ms, host({args.threads} threads) {min(hms[1:]):.2f} ms "
              f"(recorded {(1e3 * float(z['seconds'])) if 'seconds' in z else float(z['ms']):.1f}), {int(dev.sum())} nodes switch, {diff} differ from the host cut; hand-over: {batch.tiled_stats['handed_cells']} cells / {batch.tiled_stats['handed_nodes']} nodes, host {batch.tiled_stats['host_ms']:.2f} ms", flush=True)
        for b_ in (dp, dm, ds, ws):
            b_.free()
        batch.destroy()
        e.close()


if __name__ == "__main__":
    main()
