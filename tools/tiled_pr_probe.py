#!/usr/bin/env python
"""Design probe for the multi-workgroup device max-flow of the coarse layers (cells beyond a workgroup's LDS): a numpy model of
*tiled* synchronous push-relabel -- the cell's graph lives in global memory, every workgroup owns a tile of TSY x TSX nodes, a
launch ("sweep") lets every tile run up to K synchronous push / relabel iterations in LDS with the heights of the one-node
halo frozen at their values from the start of the sweep; what a tile pushes across its border goes to an outbox that the owner of
the receiving node applies at the start of the next sweep.  Global relabelling = the same tiling: every tile relaxes residual
distances to its local fixed point, launches repeat until nothing changes anywhere.

Prints, per dumped lock-step (LES_DUMP_GRAPHS ... LES_DUMP_FULL=1): sweeps, the sum over sweeps of the slowest tile's inner
iterations (what a lock-step of launches lasts), and the mask difference to the host solver.

  python tools/tiled_pr_probe.py gpurun_out/dump7/*.npz [--tile 32 32 --k 32 --cells 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

DIRS = [(0, 1), (0, -1), (1, 0), (-1, 0), (1, -1), (-1, 1), (1, 1), (-1, -1)]      # (dy, dx): E W S N SW NE SE NW; sister = k ^ 1


def shift(a, dy, dx, fill):
    h, w = a.shape
    o = np.full_like(a, fill)
    ys, yd = slice(max(dy, 0), h + min(dy, 0)), slice(max(-dy, 0), h + min(-dy, 0))
    xs, xd = slice(max(dx, 0), w + min(dx, 0)), slice(max(-dx, 0), w + min(-dx, 0))
    o[yd, xd] = a[ys, xs]
    return o


class Cell:
    def __init__(self, p5, tsy, tsx):
        h, w, _ = p5.shape
        self.h, self.w = h, w
        self.BIG = h * w + 2
        r = np.zeros((8, h, w), np.float32)
        r[0], r[2], r[4], r[6] = p5[..., 1], p5[..., 2], p5[..., 3], p5[..., 4]
        r[0][:, -1] = 0; r[2][-1, :] = 0; r[4][-1, :] = 0; r[4][:, 0] = 0; r[6][-1, :] = 0; r[6][:, -1] = 0
        self.r = r
        self.ex = p5[..., 0].astype(np.float32).copy()          # > 0 excess, < 0 remaining sink capacity
        ty, tx = np.arange(h) // tsy, np.arange(w) // tsx
        self.tile = ty[:, None] * ((w + tsx - 1) // tsx) + tx[None, :]
        self.ntiles = int(self.tile.max()) + 1
        # cross[k]: the neighbour in direction k exists and lies in another tile
        self.cross = []
        for k, (dy, dx) in enumerate(DIRS):
            nt = shift(self.tile, dy, dx, -1)
            self.cross.append((nt >= 0) & (nt != self.tile))
        self.hgt = np.zeros((h, w), np.int64)

    def tile_max(self, per_node):
        """max over the nodes of every tile -> array per tile"""
        out = np.zeros(self.ntiles, np.int64)
        np.maximum.at(out, self.tile.reshape(-1), per_node.reshape(-1))
        return out

    def global_relabel(self):
        """tiled relaxation from scratch.  Returns (launches, sum over launches of the slowest tile's local sweeps)."""
        BIG = self.BIG
        d = np.where(self.ex < 0, 1, BIG).astype(np.int64)
        launches, cost = 0, 0
        while True:
            launches += 1
            frozen = d.copy()
            local_sweeps = np.zeros(self.ntiles, np.int64)
            changed_any = False
            s = 0
            while True:
                s += 1
                best = d.copy()
                for k, (dy, dx) in enumerate(DIRS):
                    nd_live = shift(d, dy, dx, BIG)
                    nd_frozen = shift(frozen, dy, dx, BIG)
                    nd = np.where(self.cross[k], nd_frozen, nd_live)
                    best = np.minimum(best, np.where(self.r[k] > 0, nd + 1, BIG))
                ch = best < d
                if not ch.any():
                    break
                changed_any = True
                tch = self.tile_max(ch.astype(np.int64))
                local_sweeps = np.where(tch > 0, s, local_sweeps)
                d = best
            cost += int(local_sweeps.max()) + 1
            if not changed_any:
                break
        return np.minimum(d, BIG), launches, cost

    def discharge_sweep(self, K, region_relabel=True):
        """one launch: apply nothing (the caller merged the outbox), K inner iterations per tile with frozen halo heights.
        Returns (slowest tile's inner iterations, number of tiles that did anything)."""
        BIG = self.BIG
        r, ex = self.r, self.ex
        frozen = self.hgt.copy()
        hgt = self.hgt
        inner_cost = 0
        if region_relabel:
            # local Bellman-Ford given the frozen halo heights (raise only): valid lower bounds stay valid
            d = np.where(ex < 0, 1, BIG).astype(np.int64)
            s = 0
            while True:
                s += 1
                best = d.copy()
                for k, (dy, dx) in enumerate(DIRS):
                    nd = np.where(self.cross[k], shift(frozen, dy, dx, BIG), shift(d, dy, dx, BIG))
                    best = np.minimum(best, np.where(r[k] > 0, nd + 1, BIG))
                if not (best < d).any():
                    break
                d = best
            hgt = np.maximum(hgt, np.minimum(d, BIG))
            inner_cost += s // 4                                    # (a relaxation sweep costs about a quarter of a push iteration)
        out = [np.zeros_like(ex) for _ in range(8)]
        its_tile = np.zeros(self.ntiles, np.int64)
        for it in range(1, K + 1):
            active = (ex > 0) & (hgt < BIG)
            if not active.any():
                break
            ta = self.tile_max(active.astype(np.int64))
            its_tile = np.where(ta > 0, it, its_tile)
            for k, (dy, dx) in enumerate(DIRS):
                nh = np.where(self.cross[k], shift(frozen, dy, dx, BIG), shift(hgt, dy, dx, BIG))
                m = (ex > 0) & (hgt < BIG) & (r[k] > 0) & (hgt == nh + 1)
                dlt = np.where(m, np.minimum(ex, r[k]), 0).astype(np.float32)
                ex -= dlt; r[k] -= dlt
                inside = np.where(self.cross[k], 0, dlt).astype(np.float32)
                out[k] += np.where(self.cross[k], dlt, 0).astype(np.float32)
                got = shift(inside, -dy, -dx, 0)
                ex += got; r[k ^ 1] += got
            act = (ex > 0) & (hgt < BIG)
            best = np.full(hgt.shape, BIG, np.int64)
            for k, (dy, dx) in enumerate(DIRS):
                nh = np.where(self.cross[k], shift(frozen, dy, dx, BIG), shift(hgt, dy, dx, BIG))
                best = np.minimum(best, np.where(r[k] > 0, nh + 1, BIG))
            hgt = np.where(act & (best > hgt), np.minimum(best, BIG), hgt)
        # merge the outboxes (what the next launch does first)
        for k, (dy, dx) in enumerate(DIRS):
            got = shift(out[k], -dy, -dx, 0)
            ex += got; r[k ^ 1] += got
        self.hgt = hgt
        self.ex = ex
        return inner_cost + int(its_tile.max()), int((its_tile > 0).sum())


def solve(p5, tsy, tsx, K, S, region_relabel, verbose=False, K2=None, S2=None):
    c = Cell(p5, tsy, tsx)
    launches, cost = 0, 0
    rounds = 0
    while True:
        rounds += 1
        d, l, cst = c.global_relabel()
        launches += l; cost += cst // 4
        c.hgt = d if rounds == 1 else np.maximum(c.hgt, d)
        c.hgt = d                                                     # exact distances are valid: take them
        act = (c.ex > 0) & (c.hgt < c.BIG)
        if verbose:
            print(f"    round {rounds}: relabel launches {l} (cost {cst}), active {int(act.sum())}, excess {float(c.ex[act].sum()):.4f}")
        if not act.any():
            break
        Kr, Sr = (K, S) if rounds == 1 or K2 is None else (K2, S2)
        for s in range(Sr):
            cs, nt = c.discharge_sweep(Kr, region_relabel)
            launches += 1; cost += cs
            act = (c.ex > 0) & (c.hgt < c.BIG)
            if verbose:
                print(f"      sweep: slowest tile {cs} its, {nt} busy tiles, active {int(act.sum())}")
            if not act.any():
                break
        if rounds > 200:
            print("    NOT CONVERGED")
            break
    return (c.hgt >= c.BIG), launches, cost, rounds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--tile", type=int, nargs=2, default=[32, 64])
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--s", type=int, default=8, help="discharge sweeps between global relabellings")
    ap.add_argument("--k2", type=int, default=None, help="inner iterations per sweep after the first round")
    ap.add_argument("--s2", type=int, default=None)
    ap.add_argument("--cells", type=int, default=0)
    ap.add_argument("--no-region-relabel", action="store_true")
    ap.add_argument("-v", action="store_true")
    args = ap.parse_args()
    from localexpstereo_amd import gc as lgc, api
    for f in args.files:
        z = np.load(f)
        reg, off, pay = z["regions"], z["offsets"], z["payload"]
        n = len(reg) if not args.cells else min(args.cells, len(reg))
        worst = (0, 0)
        for i in range(n):
            w, h = int(reg[i]["w"]), int(reg[i]["h"])
            p = pay[off[i] * 5:(off[i] + w * h) * 5].reshape(h, w, 5).copy()
            t0 = time.perf_counter()
            mask, launches, cost, rounds = solve(p, args.tile[0], args.tile[1], args.k, args.s, not args.no_region_relabel, args.v, args.k2, args.s2 or args.s)
            t1 = time.perf_counter()
            r1 = np.zeros(1, dtype=api.RECT_DT); r1["w"] = w; r1["h"] = h
            ref = np.zeros(w * h, np.uint8)
            t2 = time.perf_counter()
            lgc.solve_prebuilt(r1, np.ascontiguousarray(p.reshape(-1)), np.zeros(1, np.int64), ref, nthreads=1)
            t3 = time.perf_counter()
            diff = int(((ref != 0) != mask.reshape(-1)).sum())
            print(f"  {os.path.basename(f)} cell {i} {w}x{h}: {rounds} rounds, {launches} launches, cost {cost} inner its, {int(mask.sum())} change, "
                  f"{diff} differ; host {1e3 * (t3 - t2):.1f} ms (model {t1 - t0:.1f} s)", flush=True)
            worst = max(worst, (cost, launches))
        print(f"{os.path.basename(f)}: slowest cell cost {worst[0]} inner iterations, {worst[1]} launches; host lock-step {1e3 * float(z['seconds']):.1f} ms")


if __name__ == "__main__":
    main()
