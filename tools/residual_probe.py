#!/usr/bin/env python
"""What the tiled device max-flow hands over (LES_HIP_MAXFLOW_HANDOVER_DUMP=file): per straggler cell the excess that is left, and how long the host
finishers need for it (search trees with budgets / push-relabel), one thread per cell.

  python tools/residual_probe.py /tmp/res.bin"""
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402
from localexpstereo_amd import api, gc as lgc        # noqa: E402


def main():
    raw = open(sys.argv[1], "rb").read()
    handed, hn = struct.unpack_from("<iq", raw, 0)
    o = 12
    cells = []
    for q in range(handed):
        w, h, hoff = struct.unpack_from("<iiq", raw, o)
        o += 16
        cells.append((w, h, hoff))
    rc8 = np.frombuffer(raw, np.float32, hn * 8, o).reshape(hn, 8).copy()
    ex = np.frombuffer(raw, np.float32, hn, o + hn * 32).copy()
    for (w, h, hoff) in cells:
        n = w * h
        r, e = rc8[hoff: hoff + n], ex[hoff: hoff + n]
        pos = e[e > 0]
        print(f"cell {w}x{h}: excess nodes {len(pos)} (sum {pos.sum():.4f}, median {np.median(pos) if len(pos) else 0:.2e}), sink nodes {(e < 0).sum()} (capacity {-e[e < 0].sum():.2f}), "
              f"residual arcs {(r > 0).sum()} of {8 * n}; arcs below 1e-2: {((r > 0) & (r < 1e-2)).sum()}")
        reg = api._rects(np.array([(0, 0, w, h)], np.int32))
        for solver, ops in ((0, "12"), (0, "0"), (0, "3"), (0, "50"), (1, "12")):
            os.environ["LES_GC_RESIDUAL_BK_OPS_PER_NODE"] = ops
            m, f = np.zeros(n, np.uint8), np.zeros(1)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                lgc.solve_residual(reg, np.ascontiguousarray(r.reshape(-1)), np.ascontiguousarray(e), np.array([0], np.int64), m, nthreads=1, solver=solver, flows_out=f)
                ts.append(1e3 * (time.perf_counter() - t0))
            print(f"    solver {solver} budget {ops:>3}: {min(ts):7.2f} ms, flow routed {f[0]:.5f}, source side {int((m != 0).sum())}")


if __name__ == "__main__":
    main()
