#!/usr/bin/env python
"""Per-cell launch / round counts of the tiled device max-flow on dumped lock-steps (LES_DUMP_TILED, pm.py): which cells are the stragglers of a
lock-step and when the others finished.  Reads the per-cell control words back from the workspace after the solve.

  python tools/tiled_cell_stats.py tools/_samples/r6/*.npz [--sim]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402
from localexpstereo_amd import api, build, synth      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--sim", action="store_true")
    args = ap.parse_args()
    lib = build.build_sim() if args.sim else None
    for f in args.files:
        z = np.load(f)
        reg, off, pay = z["regions"], z["offsets"].astype(np.int64), z["payload"]
        n = len(reg)
        W = int(max(r["x"] + r["w"] for r in reg)); H = int(max(r["y"] + r["h"] for r in reg))
        e = api.HipCostVolumeEnergy(synth.make_guide(H, W, 1), synth.make_guide(H, W, 2), np.zeros((2, H, W), np.float32), np.zeros((2, H, W), np.float32),
                                    windR=20, eps=1e-4, th_col=0.5, lib=lib)
        trs = np.ascontiguousarray(reg).view(api.RECT_DT).reshape(-1)
        batch = api.Batch(e, trs, trs)
        nn = batch.graph_nodes()
        assert np.array_equal(batch.graph_offsets(), off)
        dp, dm, ds = api.DeviceBuffer(e, nn * 20), api.DeviceBuffer(e, nn), api.DeviceBuffer(e, 4 * n)
        ws = api.DeviceBuffer(e, batch.tiled_workspace_bytes())
        dp.upload(np.ascontiguousarray(pay[: nn * 5]))
        launches = batch.solve_graphs_tiled(dp.ptr, dm.ptr, ds.ptr, ws.ptr, ws.nbytes)
        e.synchronize()
        raw = ws.download((256 + 64 * n,), np.uint8)
        ctl = raw[256:].view(np.int32).reshape(n, 16)
        ln, rounds = ctl[:, 6], ctl[:, 7]
        order = np.argsort(ln)
        print(f"{os.path.basename(f)}: {n} cells, {nn} nodes, enqueued {launches}; per-cell launches sorted: {ln[order].tolist()}")
        print(f"    rounds: {rounds[order].tolist()}")
        for b_ in (dp, dm, ds, ws):
            b_.free()
        batch.destroy()
        e.close()


if __name__ == "__main__":
    main()
