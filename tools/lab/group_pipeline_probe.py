#!/usr/bin/env python
"""Would a per-cell-group pipeline beat the lock-step barrier of the coarse layers?  (round 6 experiment)

A disjoint set's cells are independent, but the driver cuts proposal k of ALL cells before proposal k+1 of any: a lock-step lasts as long as its slowest cell,
and the stragglers of consecutive proposals are not the same cells.  This probe takes dumped lock-steps (LES_DUMP_TILED) as stand-ins for the consecutive
proposals of one set and cuts them (a) lock-step by lock-step, all cells per call, (b) in G groups of cells, every group on its own host thread and stream
cutting ITS cells of dump 1, then of dump 2, ... without waiting for the other groups.  -> wall-clock of both and the per-group times.

  python tools/lab/group_pipeline_probe.py a.npz b.npz c.npz [--groups 8] [--cells 40]"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402
import torch                                          # noqa: E402
from localexpstereo_amd import api, synth             # noqa: E402


class Job:
    def __init__(self, e, reg, pay):
        trs = np.ascontiguousarray(reg).view(api.RECT_DT).reshape(-1)
        self.batch = api.Batch(e, trs, trs)
        nn, n = self.batch.graph_nodes(), len(reg)
        self.dp, self.dm, self.ds = api.DeviceBuffer(e, nn * 20), api.DeviceBuffer(e, nn), api.DeviceBuffer(e, 4 * n)
        self.ws = api.DeviceBuffer(e, self.batch.tiled_workspace_bytes())
        self.dp.upload(np.ascontiguousarray(pay))
        self.nn = nn

    def solve(self):
        return self.batch.solve_graphs_tiled(self.dp.ptr, self.dm.ptr, self.ds.ptr, self.ws.ptr, self.ws.nbytes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--groups", type=int, default=8)
    ap.add_argument("--cells", type=int, default=40)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dumps = []
    W = H = 0
    for f in args.files:
        z = np.load(f)
        reg, off, pay = z["regions"][: args.cells].copy(), z["offsets"].astype(np.int64), z["payload"]
        W = max(W, int(max(r["x"] + r["w"] for r in reg))); H = max(H, int(max(r["y"] + r["h"] for r in reg)))
        dumps.append((reg, off, pay))
    e = api.HipCostVolumeEnergy(synth.make_guide(H, W, 1), synth.make_guide(H, W, 2), np.zeros((2, H, W), np.float32), np.zeros((2, H, W), np.float32), windR=20, eps=1e-4, th_col=0.5)

    def cells_payload(reg, off, pay, idx):
        parts = [pay[off[i] * 5:(off[i] + int(reg[i]["w"]) * int(reg[i]["h"])) * 5] for i in idx]
        return reg[idx], np.concatenate(parts)

    G = args.groups
    whole = [Job(e, *cells_payload(reg, off, pay, np.arange(len(reg)))) for (reg, off, pay) in dumps]
    groups = [[Job(e, *cells_payload(reg, off, pay, np.arange(g, len(reg), G))) for (reg, off, pay) in dumps] for g in range(G)]
    for _ in range(args.reps):
        e.synchronize()
        t0 = time.perf_counter()
        per = []
        for j in whole:
            t1 = time.perf_counter()
            j.solve()
            per.append(1e3 * (time.perf_counter() - t1))
        lock = 1e3 * (time.perf_counter() - t0)
        gt = [0.0] * G

        def work(g):
            torch.cuda.set_device(0)
            side = torch.cuda.Stream()
            e.set_thread_stream(side.cuda_stream)
            try:
                t1 = time.perf_counter()
                for j in groups[g]:
                    j.solve()
                gt[g] = 1e3 * (time.perf_counter() - t1)
            finally:
                e.set_thread_stream(0, bind=False)
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        grouped = 1e3 * (time.perf_counter() - t0)
        print(f"lock-steps {lock:7.2f} ms ({', '.join('%.2f' % p for p in per)});  {G} groups {grouped:7.2f} ms (per group: {', '.join('%.1f' % x for x in gt)})", flush=True)


if __name__ == "__main__":
    main()
