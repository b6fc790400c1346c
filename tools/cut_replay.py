#!/usr/bin/env python
"""Replays sampled host-cut lock-steps (LES_DUMP_GRAPHS=dir LES_DUMP_EVERY=n python tools/e2e_bench.py -> sample_*.npz, first two cells of a
lock-step) through les_gc_solve_prebuilt the way the optimiser calls it: the two cells are repeated up to the lock-step's real cell count,
calls are separated by an idle gap (the device phase of a lock-step, during which the thread pools go to sleep), alone and with a second
host thread doing the same (two views).  Prints ms per lock-step.

  python tools/cut_replay.py tools/_samples/*.npz [--threads 16 --reps 30 --gap-ms 1.0]
"""
import argparse
import os
import sys
import threading
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                   # noqa: E402
from localexpstereo_amd import api, gc as lgc        # noqa: E402


def lockstep(path):
    z = np.load(path)
    reg2, off2, pay2, cells = z["regions"], z["offsets"], z["payload"], int(z["cells"])
    regs, pays = [], []
    for i in range(cells):
        k = i % len(reg2)
        w, h = int(reg2[k]["w"]), int(reg2[k]["h"])
        regs.append((0, 0, w, h))
        pays.append(pay2[off2[k] * 5:(off2[k] + w * h) * 5])
    reg = np.array(regs, np.int32).view(api.RECT_DT).reshape(-1)
    sizes = np.array([r[2] * r[3] for r in regs], np.int64)
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    pay = np.concatenate(pays).astype(np.float32)
    return reg, off, pay, np.zeros(int(sizes.sum()), np.uint8), float(z["seconds"])


def loop(ls, reps, nt, gap, out):
    reg, off, pay, masks, _ = ls
    lgc.solve_prebuilt(reg, pay, off, masks, nthreads=nt)
    tot = 0.0
    for _ in range(reps):
        if gap > 0:
            time.sleep(gap)
        t = time.perf_counter()
        lgc.solve_prebuilt(reg, pay, off, masks, nthreads=nt)
        tot += time.perf_counter() - t
    out.append(tot / reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--gap-ms", type=float, default=1.0)
    args = ap.parse_args()
    for f in args.files:
        ls = lockstep(f)
        ls2 = tuple(a.copy() if isinstance(a, np.ndarray) else a for a in ls)
        o = []
        loop(ls, args.reps, args.threads, args.gap_ms * 1e-3, o)
        alone = o[0]
        o = []
        ths = [threading.Thread(target=loop, args=(x, args.reps, args.threads, args.gap_ms * 1e-3, o)) for x in (ls, ls2)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        print(f"{os.path.basename(f)}: {len(ls[0])} cells of {ls[0][0]['w']}x{ls[0][0]['h']}, in the run {ls[4] * 1e3:.2f} ms; replay alone {alone * 1e3:.2f} ms, two host threads {o[0] * 1e3:.2f} / {o[1] * 1e3:.2f} ms "
              f"(threads {args.threads}, gap {args.gap_ms} ms, prepush {os.environ.get('LES_GC_PREPUSH', '1')}, OMP_WAIT_POLICY {os.environ.get('OMP_WAIT_POLICY')})", flush=True)


if __name__ == "__main__":
    main()
