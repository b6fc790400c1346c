#!/usr/bin/env python
"""Where the time of a launch of les_maxflow_tiled_kernel goes: a lab build (-DLES_MT_PROBE: bash tools/build_variant.sh mtprobe -DLES_MT_PROBE) stamps the
100-MHz clock at eight points of every tile's launch; this script replays one dumped lock-step through it and prints, per launch, the phase and the
median / maximum over the tiles of every segment (microseconds):
  entry->ctl | ->LDS init | ->state loaded (heights, excess, inbox probe) | ->residuals loaded | ->iterations done | ->written back + verdict barrier | ->cell verdict
and the span first entry -> last exit of the launch and the gap to the next launch.

  LES_HIP_LIB=localexpstereo_amd/csrc/libles_mtprobe.so python tools/lab/mt_probe.py tools/_samples/r6/tiled_view0_it1_layer1_110.npz [--launches 40]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402
from localexpstereo_amd import api, synth           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--launches", type=int, default=48)
    args = ap.parse_args()
    os.environ["LES_HIP_MAXFLOW_HANDOVER"] = "0"
    z = np.load(args.file)
    reg, off, pay = z["regions"], z["offsets"].astype(np.int64), z["payload"]
    n = len(reg)
    W = int(max(r["x"] + r["w"] for r in reg)); H = int(max(r["y"] + r["h"] for r in reg))
    e = api.HipCostVolumeEnergy(synth.make_guide(H, W, 1), synth.make_guide(H, W, 2), np.zeros((2, H, W), np.float32), np.zeros((2, H, W), np.float32), windR=20, eps=1e-4, th_col=0.5)
    trs = np.ascontiguousarray(reg).view(api.RECT_DT).reshape(-1)
    batch = api.Batch(e, trs, trs)
    boff, nn = batch.graph_offsets(), batch.graph_nodes()
    p = np.zeros((nn, 5), np.float32)
    for i in range(n):
        k = int(reg[i]["w"]) * int(reg[i]["h"])
        p[boff[i]: boff[i] + k] = pay[off[i] * 5:(off[i] + k) * 5].reshape(k, 5)
    dp, dm, ds = api.DeviceBuffer(e, nn * 20), api.DeviceBuffer(e, nn), api.DeviceBuffer(e, 4 * n)
    ws = api.DeviceBuffer(e, batch.tiled_workspace_bytes())
    dp.upload(np.ascontiguousarray(p.reshape(-1)))
    batch.solve_graphs_tiled(dp.ptr, dm.ptr, ds.ptr, ws.ptr, ws.nbytes)           # warm-up (tile table, code object)
    e.synchronize()
    ntiles = 4096                                                                   # upper bound of the grid; the kernel indexes with gridDim.x
    probe = api.DeviceBuffer(e, args.launches * ntiles * 128)
    probe.upload(np.zeros(args.launches * ntiles * 16, np.uint64))
    lib = e.L
    fn = lib.les_hip_debug_mt_probe
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]; fn.restype = ctypes.c_int
    assert fn(probe.ptr, args.launches) == 0
    launches = batch.solve_graphs_tiled(dp.ptr, dm.ptr, ds.ptr, ws.ptr, ws.nbytes)
    e.synchronize()
    raw = probe.download((args.launches * ntiles * 16,), np.uint64)
    g = int(raw[9])                                                                  # gridDim.x, as the kernel saw it
    t = raw[: args.launches * g * 16].reshape(args.launches, g, 16).astype(np.int64)
    names = ["RELABEL0", "RELABEL", "DISCHARGE", "FINAL"]
    print(f"{os.path.basename(args.file)}: {n} cells, {g} tiles, {launches} launches; microseconds, median / max over the tiles that ran the step")
    prev_end = None
    for l in range(args.launches):
        rows = t[l][t[l][:, 0] != 0]
        if len(rows) == 0:
            break
        ph = rows[:, 8]
        end = rows[:, 7]
        st = rows[:, :7].copy()
        seg = []
        pts = [st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4], st[:, 5], st[:, 6], end]
        lab = ["ctl", "lds", "state", "resid", "iter", "wback", "verdict"]
        for a in range(7):
            ok = (pts[a] != 0) & (pts[a + 1] != 0)
            # a stamp that was skipped (idle tile: no residual load, no iterations) is carried forward
            if a + 1 < 7:
                pts[a + 1] = np.where(pts[a + 1] == 0, pts[a], pts[a + 1])
            d = (pts[a + 1] - pts[a])[ok] / 100.0
            seg.append(f"{lab[a]} {np.median(d):5.1f}/{d.max():5.1f}" if len(d) else f"{lab[a]}   -  /  -  ")
        span = (end.max() - st[:, 0].min()) / 100.0
        spread = (st[:, 0].max() - st[:, 0].min()) / 100.0
        tile = (end - st[:, 0]) / 100.0
        gap = "" if prev_end is None else f" gap {(st[:, 0].min() - prev_end) / 100.0:5.1f}"
        prev_end = end.max()
        kinds = ",".join(f"{names[k] if k < 4 else k}x{int((ph == k).sum())}" for k in sorted(set(ph.tolist())))
        print(f"launch {l:3d} [{kinds}] " + " | ".join(seg) + f" | tile {np.median(tile):5.1f}/{tile.max():5.1f} starts within {spread:5.1f} | span {span:6.1f}{gap}")


if __name__ == "__main__":
    main()
