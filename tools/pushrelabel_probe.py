"""How many synchronous push-relabel iterations does a layer-0 sized expansion graph need?  (Design probe for a workgroup-per-cell
max-flow in LDS)  Emulates, vectorised in numpy, the data-parallel scheme such a kernel would run: per
iteration every active node pushes to the sink, then along each of the 8 grid directions in turn (height-admissible arcs only),
then relabels; every G iterations a global relabelling (residual distances to the sink).  The cut read out at the end (nodes that
can still reach the sink) is compared with the host Boykov-Kolmogorov solver.  Input: 42 x 42 crops of dumped layer-2 graphs
(LES_DUMP_GRAPHS, pm.py), arcs leaving the crop dropped."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from localexpstereo_amd import gc as lgc, api

DIRS = [(0, 1), (0, -1), (1, 0), (-1, 0), (1, -1), (-1, 1), (1, 1), (-1, -1)]      # (dy, dx): E W S N SW NE SE NW; sister = k ^ 1


def shift(a, dy, dx, fill):
    h, w = a.shape
    o = np.full_like(a, fill)
    ys, yd = slice(max(dy, 0), h + min(dy, 0)), slice(max(-dy, 0), h + min(-dy, 0))
    xs, xd = slice(max(dx, 0), w + min(dx, 0)), slice(max(-dx, 0), w + min(-dx, 0))
    o[yd, xd] = a[ys, xs]
    return o


def global_relabel(r, tcap, BIG):
    h, w = tcap.shape
    d = np.where(tcap > 0, 1, BIG).astype(np.int64)
    sweeps = 0
    while True:
        sweeps += 1
        best = d.copy()
        for k, (dy, dx) in enumerate(DIRS):
            nd = shift(d, dy, dx, BIG)                     # distance of the neighbour in direction k
            cand = np.where(r[k] > 0, nd + 1, BIG)
            best = np.minimum(best, cand)
        if np.array_equal(best, d):
            return np.minimum(d, BIG), sweeps
        d = best


def push_relabel(p5, G=16, max_it=20000):
    h, w, _ = p5.shape
    N = h * w
    BIG = N + 2
    r = np.zeros((8, h, w), np.float32)
    r[0], r[2], r[4], r[6] = p5[..., 1], p5[..., 2], p5[..., 3], p5[..., 4]
    r[0][:, -1] = 0; r[2][-1, :] = 0; r[4][-1, :] = 0; r[4][:, 0] = 0; r[6][-1, :] = 0; r[6][:, -1] = 0
    tr = p5[..., 0].astype(np.float32)
    e = np.maximum(tr, 0).astype(np.float32)
    tcap = np.maximum(-tr, 0).astype(np.float32)
    hgt, sw = global_relabel(r, tcap, BIG)
    its, sweeps = 0, sw
    while its < max_it:
        active = (e > 0) & (hgt < BIG)
        if not active.any():
            break
        its += 1
        # push to the sink (height 0): admissible when the node's height is 1
        m = active & (hgt == 1) & (tcap > 0)
        d = np.where(m, np.minimum(e, tcap), 0).astype(np.float32)
        e -= d; tcap -= d
        for k, (dy, dx) in enumerate(DIRS):
            nh = shift(hgt, dy, dx, BIG)
            m = (e > 0) & (hgt < BIG) & (r[k] > 0) & (hgt == nh + 1)
            d = np.where(m, np.minimum(e, r[k]), 0).astype(np.float32)
            e -= d; r[k] -= d
            got = shift(d, -dy, -dx, 0)                      # what the neighbour in direction k^1 sent to me
            e += got; r[k ^ 1] += got
        # relabel active nodes without an admissible arc
        act = (e > 0) & (hgt < BIG)
        best = np.where(tcap > 0, 1, BIG).astype(np.int64)
        for k, (dy, dx) in enumerate(DIRS):
            nh = shift(hgt, dy, dx, BIG)
            best = np.minimum(best, np.where(r[k] > 0, nh + 1, BIG))
        hgt = np.where(act & (best > hgt), np.minimum(best, BIG), hgt)
        if its % G == 0:
            hgt2, sw = global_relabel(r, tcap, BIG)
            sweeps += sw
            hgt = np.maximum(hgt, hgt2)
    d, sw = global_relabel(r, tcap, BIG)
    return (d >= BIG), its, sweeps + sw                      # True = cannot reach the sink = SOURCE side = takes the proposal


def main():
    rng = np.random.default_rng(0)
    res = []
    for v in (0, 1):
        path = f"gpurun_out/graphs_view{v}_layer2.npz"
        if not os.path.exists(path):
            continue
        d = np.load(path)
        reg, off, pay = d["regions"], d["offsets"], d["payload"]
        for trial in range(12):
            i = int(rng.integers(0, len(reg)))
            w, h = int(reg[i]["w"]), int(reg[i]["h"])
            p = pay[off[i] * 5:(off[i] + w * h) * 5].reshape(h, w, 5)
            S = 42
            y0, x0 = int(rng.integers(0, h - S)), int(rng.integers(0, w - S))
            c = p[y0:y0 + S, x0:x0 + S].copy()
            c[:, -1, 1] = 0; c[-1, :, 2] = 0; c[-1, :, 3] = 0; c[:, 0, 3] = 0; c[-1, :, 4] = 0; c[:, -1, 4] = 0
            mask, its, sweeps = push_relabel(c)
            r1 = np.zeros(1, dtype=api.RECT_DT); r1["w"] = S; r1["h"] = S
            ref = np.zeros(S * S, np.uint8)
            lgc.solve_prebuilt(r1, np.ascontiguousarray(c.reshape(-1)), np.zeros(1, np.int64), ref, nthreads=1)
            diff = int(((ref != 0) != mask.reshape(-1)).sum())
            res.append((v, its, sweeps, diff, int(mask.sum())))
            print(f"view {v} crop {trial}: {its} push/relabel iterations, {sweeps} global-relabel sweeps, {int(mask.sum())} nodes change, {diff} differ from the host cut")
    a = np.array([r[1] for r in res]); s = np.array([r[2] for r in res])
    print(f"iterations: median {np.median(a):.0f}, mean {a.mean():.0f}, max {a.max()}; relabel sweeps: median {np.median(s):.0f}, max {s.max()}")


if __name__ == "__main__":
    main()
