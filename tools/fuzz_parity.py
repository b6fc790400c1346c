#!/usr/bin/env python
"""Randomised parity sweep: random image sizes, radii, (filterRect, targetRect) pairs and planes through the C ABI
(`les_hip_unary_batch`, check on/off, both views) against the oracle.  Also random label maps through the device
post-processing and the expansion-graph construction.  Exits non-zero on the first mismatch.

  python tools/fuzz_parity.py [--seconds 120] [--seed 0] [--lib PATH]     (default library: the HIP build)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from localexpstereo_amd import api, synth          # noqa: E402
from oracle import oracle as om                     # noqa: E402
from tests import parity_cases as pc                # noqa: E402

RADII = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15]


def one_case(rng, lib, stats):
    # half of the configurations are shaped for the fixed-point march kernel (radius 10, targets at least windR away from
    # filterRect borders that are not image borders -- LayerManager-like cells -- or whole-image slabs), the others roam freely
    march_shaped = rng.random() < 0.5
    R = int(rng.choice([10, 10, 10, 10, 7, 4, 5, 6, 8, 9])) if march_shaped else int(rng.choice(RADII))      # radii with a march-kernel instantiation
    windR = 2 * R + int(rng.integers(0, 2))                       # windR / 2 == R
    H, W = (int(rng.integers(30, 260)), int(rng.integers(30, 520))) if march_shaped else (int(rng.integers(8, 150)), int(rng.integers(8, 200)))
    D = int(rng.integers(2, 24))
    mind = float(rng.choice([0.0, 0.0, -3.0]))
    maxd = float(D - 1 + mind)
    imL, imR = synth.make_guide(H, W, int(rng.integers(1 << 30))), synth.make_guide(H, W, int(rng.integers(1 << 30)))
    if rng.random() < 0.2:
        imL[:] = imL[0, 0]                                          # constant guide: Sigma = eps I
    volL, volR = synth.make_volume(D, H, W, int(rng.integers(1 << 30))), synth.make_volume(D, H, W, int(rng.integers(1 << 30)))
    if rng.random() < 0.3:                                          # other cost ranges: negative costs, costs far above the threshold
        sc, sh = float(rng.choice([1.0, 3.0, 0.2])), float(rng.choice([0.0, -0.4, 0.3]))
        volL, volR = (volL * sc + sh).astype(np.float32), (volR * sc + sh).astype(np.float32)
    eps = float(rng.choice([1e-4, 1e-2, 1e-6]))
    th = float(rng.choice([0.5, 0.12, 2.0]))
    o = om.Oracle(imL, imR, volL, volR, windR=windR, eps=eps, th_col=th, max_disp=maxd, min_disp=mind)
    e = api.HipCostVolumeEnergy(imL, imR, volL, volR, windR=windR, eps=eps, th_col=th, max_disp=maxd, min_disp=mind, lib=lib)
    n = int(rng.integers(1, 12))
    frs, trs = np.zeros(n, api.RECT_DT), np.zeros(n, api.RECT_DT)
    occupied = np.zeros((H, W), bool)
    k = 0
    for _ in range(n * 4):
        if march_shaped:
            tw, th_ = int(rng.integers(1, min(W, 300) + 1)), int(rng.integers(1, min(H, 120) + 1))
            tx, ty = int(rng.integers(0, W - tw + 1)), int(rng.integers(0, H - th_ + 1))
            m = [windR + int(rng.integers(0, 6)) for _ in range(4)]                       # margins >= windR, clipped at the image like LayerManager.h:129-132
            fx, fy = max(0, tx - m[0]), max(0, ty - m[1])
            fw, fh = min(W, tx + tw + m[2]) - fx, min(H, ty + th_ + m[3]) - fy
            if rng.random() < 0.15:
                fx, fy, fw, fh, tx, ty, tw, th_ = 0, 0, W, H, 0, 0, W, H                # whole-image slab
        else:
            fw, fh = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1))
            fx, fy = int(rng.integers(0, W - fw + 1)), int(rng.integers(0, H - fh + 1))
            tw, th_ = int(rng.integers(1, fw + 1)), int(rng.integers(1, fh + 1))
            tx, ty = fx + int(rng.integers(0, fw - tw + 1)), fy + int(rng.integers(0, fh - th_ + 1))
        if occupied[ty:ty + th_, tx:tx + tw].any():
            continue                                                 # targets of one batch are disjoint (one disjoint set)
        occupied[ty:ty + th_, tx:tx + tw] = True
        frs[k], trs[k] = (fx, fy, fw, fh), (tx, ty, tw, th_)
        k += 1
        if k == n:
            break
    frs, trs = frs[:k], trs[:k]
    planes = pc.random_planes(k, D, H, W, int(rng.integers(1 << 30)), slant=float(rng.choice([0.0, 0.05, 0.5])))
    planes[:, 2] += mind
    if rng.random() < 0.15:
        planes[0, :3] = rng.choice([np.nan, np.inf, -np.inf, 1e30])
    bq = api.Batch(e, frs, trs)
    stats["march"] = stats.get("march", 0) + int(bq.kernel_kind(0) == 1)
    bq.destroy()
    for mode in (0, 1):
        for check in (True, False):
            ref = o.unary_batch(frs, trs, planes, mode=mode, check=check)
            got = e.unary_batch(frs, trs, planes, mode=mode, check=check)
            # absolute tolerance scaled to the cost range (th_col)
            m = ~np.isnan(ref)
            assert np.array_equal(np.isnan(got), np.isnan(ref)), "written set"
            assert np.array_equal(got[m] == np.float32(1e6), ref[m] == np.float32(1e6)), "sentinels"
            v = m & (ref != np.float32(1e6))
            if v.any():
                err = np.abs(got[v].astype(np.float64) - ref[v])
                tol = 1e-4 * np.abs(ref[v]) + 2e-6 * max(1.0, th)
                if not np.all(err <= tol):
                    raise AssertionError(f"max err {err.max():.3e} (R={R} {W}x{H}x{D} eps={eps} th={th} mode={mode} check={check})")
                stats["max_err"] = max(stats["max_err"], float(err.max() / max(1.0, th)))
    stats["calls"] += 4 * k
    # post-processing on random piecewise label maps (both views): labels must come out bit-identical
    if rng.random() < 0.35 and windR <= 31:
        def labels():
            lab = np.zeros((H, W, 4), np.float32)
            lab[..., 2] = rng.uniform(0, D)
            for _ in range(int(rng.integers(1, 8))):
                x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
                w, h = int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1))
                lab[y0:y0 + h, x0:x0 + w] = (rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-2, D + 2), 0.0)
            noisy = rng.random((H, W)) < 0.03
            lab[noisy, 2] += rng.uniform(-9, 9, int(noisy.sum())).astype(np.float32)
            return lab
        LL, LR = labels(), labels()
        thr = float(rng.choice([1.0, 1.5]))
        ref = om.post_process(LL, LR, imL, imR, windR=windR, threshold=thr, omega=10.0)
        got = e.post_process_host(LL, LR, threshold=thr, omega=10.0)
        for g_, r_ in zip(got, ref):
            if not np.array_equal(g_.view(np.uint32), r_.view(np.uint32)):
                raise AssertionError(f"post-processed labels differ ({W}x{H} windR={windR} thr={thr})")
        stats["post"] += 1
    # graph capacities of the expansion moves: device construction vs host construction, bit for bit
    if rng.random() < 0.35:
        from localexpstereo_amd import gc as lgc
        lab = np.zeros((H, W, 4), np.float32)
        lab[..., 0] = rng.uniform(-0.05, 0.05); lab[..., 1] = rng.uniform(-0.05, 0.05); lab[..., 2] = rng.uniform(0, D)
        for _ in range(int(rng.integers(1, 10))):
            x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
            w, h = int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1))
            lab[y0:y0 + h, x0:x0 + w] = (rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-2, D + 2), 0.0)
        cur = rng.uniform(0, th, (H, W)).astype(np.float32)
        prop = rng.uniform(0, th, (H, W)).astype(np.float32)
        lam, ths = float(rng.choice([0.5, 1.0, 20.0])), float(rng.choice([1.0, 0.3]))
        mode = int(rng.integers(0, 2))
        g = lgc.GraphCut(imL, imR, lambda_=lam, th_smooth=ths, omega=10.0, epsilon=0.01)
        g.labels[mode][...] = lab
        g.costs[mode][...] = cur
        batch = api.Batch(e, frs, trs)
        off, nn = batch.graph_offsets(), batch.graph_nodes()
        ref_payload, _ = g.build_graphs(trs, planes, prop, off, mode=mode)
        bufs = [api.DeviceBuffer(e, max(1, k) * 16), api.DeviceBuffer(e, H * W * 16), api.DeviceBuffer(e, H * W * 4), api.DeviceBuffer(e, H * W * 4),
                api.DeviceBuffer(e, max(1, nn) * 20)]
        bufs[0].upload(api._planes(planes).view(np.float32)); bufs[1].upload(lab); bufs[2].upload(cur); bufs[3].upload(prop)
        batch.expansion_graph(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, mode=mode, lambda_=lam, th_smooth=ths, omega=10.0, epsilon=0.01)
        e.synchronize()
        got = bufs[4].download((nn * 5,), np.float32)
        finite = np.isfinite(ref_payload)
        if not (np.array_equal(got.view(np.uint32)[finite], ref_payload.view(np.uint32)[finite]) and np.array_equal(np.isnan(got), np.isnan(ref_payload))):
            raise AssertionError(f"expansion-graph payload differs ({W}x{H}, {k} cells, mode {mode}, lambda {lam})")
        # the cuts themselves: device solver (cells that fit a workgroup's LDS) against the host solver on the same payload.
        # (NaN planes give NaN capacities: such graphs have no defined cut and are skipped, as are overlapping rects -- the
        # payload offsets require disjoint node ranges, which the random rect pairs satisfy by construction)
        if k > 0 and nn > 0 and batch.max_cell_nodes <= api.Batch.MAXFLOW_MAX_NODES and np.isfinite(got).all():
            dm, ds, df = api.DeviceBuffer(e, nn), api.DeviceBuffer(e, 4 * k), api.DeviceBuffer(e, 8 * k)
            batch.solve_graphs(bufs[4].ptr, dm.ptr, ds.ptr, df.ptr)
            e.synchronize()
            if ds.download((k,), np.int32).any():
                raise AssertionError(f"device max-flow hit its iteration limit ({W}x{H}, {k} cells)")
            dev_m = dm.download((nn,), np.uint8)
            host_m, host_f = np.zeros(nn, np.uint8), np.zeros(k, np.float64)
            lgc.solve_prebuilt(trs, got, off, host_m, flows_out=host_f)
            dev_f = df.download((k,), np.float64)
            sizes = np.array([max(0, int(t["w"])) * max(0, int(t["h"])) for t in trs], np.int64)
            tsum = np.array([np.abs(got.reshape(-1, 5)[int(o):int(o) + int(n_), 0]).astype(np.float64).sum() for o, n_ in zip(off, sizes)])
            if not (np.abs(dev_f - host_f) <= 1e-6 * tsum + 1e-5 * np.abs(host_f) + 1e-5).all():
                raise AssertionError(f"device / host flow values differ by {np.abs(dev_f - host_f).max():.3e} ({W}x{H}, {k} cells, lambda {lam})")
            ndiff = int(((dev_m != 0) != (host_m != 0)).sum())
            if ndiff > max(2, 2e-5 * nn):
                raise AssertionError(f"device cut differs from the host cut in {ndiff} of {nn} nodes ({W}x{H}, {k} cells, lambda {lam})")
            stats["cuts"] = stats.get("cuts", 0) + 1
            stats["cuts_cell_kernel"] = stats.get("cuts_cell_kernel", 0) + (1 if batch.graph_solver_kind == 0 else 0)      # csrc/les_maxflow_cell.h (else les_maxflow.h)
            stats["cut_nodes"] = stats.get("cut_nodes", 0) + nn
            stats["cut_diff"] = stats.get("cut_diff", 0) + ndiff
            for b_ in (dm, ds, df):
                b_.free()
        # ... and the tiled solver (cells of any size: csrc/les_maxflow_tiled.h) on every lock-step with a defined cut, against the host solver
        if k > 0 and nn > 0 and np.isfinite(got).all():
            dm2, ds2, ws = api.DeviceBuffer(e, nn), api.DeviceBuffer(e, 4 * k), api.DeviceBuffer(e, batch.tiled_workspace_bytes())
            ws.fill(0xA5)
            batch.solve_graphs_tiled(bufs[4].ptr, dm2.ptr, ds2.ptr, ws.ptr, ws.nbytes)
            e.synchronize()
            if batch.tiled_unsolved or ds2.download((k,), np.int32).any():
                raise AssertionError(f"tiled device max-flow hit its launch limit ({W}x{H}, {k} cells)")
            dev2 = dm2.download((nn,), np.uint8)
            host2 = np.zeros(nn, np.uint8)
            lgc.solve_prebuilt(trs, got, off, host2)
            nd2 = int(((dev2 != 0) != (host2 != 0)).sum())
            if nd2 > max(2, 2e-5 * nn):
                raise AssertionError(f"tiled device cut differs from the host cut in {nd2} of {nn} nodes ({W}x{H}, {k} cells, lambda {lam})")
            stats["tiled_cuts"] = stats.get("tiled_cuts", 0) + 1
            stats["tiled_nodes"] = stats.get("tiled_nodes", 0) + nn
            stats["tiled_diff"] = stats.get("tiled_diff", 0) + nd2
            stats["tiled_multi"] = stats.get("tiled_multi", 0) + (1 if batch.max_cell_nodes > 1920 else 0)
            for b_ in (dm2, ds2, ws):
                b_.free()
        batch.destroy()
        for b in bufs:
            b.free()
        g.close()
        stats["graphs"] = stats.get("graphs", 0) + 1
    e.close()
    # image-based matching cost (NaiveStereoEnergy) on the same rects
    if rng.random() < 0.3:
        on = om.Oracle.naive(imL, imR, maxd if mind == 0 else float(D - 1), windR=windR, eps=eps)
        en = api.HipCostVolumeEnergy.naive(imL, imR, windR=windR, eps=eps, max_disp=maxd if mind == 0 else float(D - 1), lib=lib)
        pl = planes.copy()
        pl[:, 2] -= mind
        for mode in (0, 1):
            ref = on.unary_batch(frs, trs, pl, mode=mode, check=True)
            got = en.unary_batch(frs, trs, pl, mode=mode, check=True)
            pc.compare_maps(got, ref, tight=False)
        bt = api.Batch(en, frs, trs)
        stats["naive_march"] = stats.get("naive_march", 0) + (1 if bt.kernel_kind(0) == 1 else 0)
        bt.destroy()
        en.close()
        stats["naive"] = stats.get("naive", 0) + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    stats = dict(calls=0, max_err=0.0, post=0)
    t0 = time.time()
    cases = 0
    while time.time() - t0 < args.seconds:
        state = rng.bit_generator.state
        try:
            one_case(rng, args.lib, stats)
        except Exception:
            print("FAILED case", cases, "rng state:", state["state"])
            raise
        cases += 1
    print(f"fuzz OK: {cases} configurations, {stats['calls']} operator calls, {stats['post']} post-processing runs, "
          f"{stats.get('graphs', 0)} expansion-graph lock-steps ({stats.get('cuts', 0)} of them also cut on the device, {stats.get('cuts_cell_kernel', 0)} by les_maxflow_cell.h and the rest by les_maxflow.h: {stats.get('cut_diff', 0)} of {stats.get('cut_nodes', 0)} nodes differ from the host cut; {stats.get('tiled_cuts', 0)} cut by the tiled solver, {stats.get('tiled_multi', 0)} of them with cells of several tiles: {stats.get('tiled_diff', 0)} of {stats.get('tiled_nodes', 0)} nodes differ), {stats.get('naive', 0)} image-based energies ({stats.get('naive_march', 0)} of them on the march kernel), {stats.get('march', 0)} configurations on the march kernel, "
          f"max abs err / max(1, th_col) = {stats['max_err']:.2e}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
