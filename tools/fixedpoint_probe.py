"""Numerical probe of the fixed-point formulation of the guided-filter aggregation.

Emulates in numpy, with exact integer arithmetic where the kernel uses integers and float32 where it uses float32,
what les_march_kernel computes, and compares it with the CPU oracle (double, the reference's default "GF").
TEST / DESIGN TOOLING: it imports the oracle, never the product package.

  python tools/fixedpoint_probe.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as om  # noqa: E402
from localexpstereo_amd import synth  # noqa: E402

PB = 22          # bits of the fixed-point cost
SH = 9           # right shift of the vertical sums of I'*p before the horizontal pass


def box_v(a, R, axis):
    """zero-padded 2R+1 window sum along axis, exact for integer arrays (object/int64)"""
    n = a.shape[axis]
    pad = [(0, 0)] * a.ndim
    pad[axis] = (R + 1, R)
    c = np.cumsum(np.pad(a, pad), axis=axis)
    hi = np.take(c, np.arange(2 * R + 1, 2 * R + 1 + n), axis=axis)
    lo = np.take(c, np.arange(0, n), axis=axis)
    return hi - lo


def emulate(o, fr, p, mode, R, eps, th, vmin, scale_margin=1.5):
    """p: float32 raw truncated cost over the filter rect fr=(x,y,w,h); returns q float32 over fr"""
    x0, y0, w, h = fr
    st = o.stats(mode)                                    # 13 x H x W double: I(3), mean(3), inv(6), N
    I = st[0:3, y0:y0 + h, x0:x0 + w]
    mean = st[3:6, y0:y0 + h, x0:x0 + w]
    inv6 = st[6:12, y0:y0 + h, x0:x0 + w]
    Iu8 = np.rint(I * 255).astype(np.int64)
    Iq = Iu8 - 128
    rng = np.float32(max(th - vmin, 1e-30))
    sp = np.float32((2 ** PB - 1)) / rng
    pint = np.floor((p.astype(np.float32) - np.float32(vmin)) * sp + np.float32(0.5)).astype(np.int64)
    pint = np.clip(pint, 0, 2 ** PB - 1)
    # pass 1: vertical sums (exact), quantised
    sV = box_v(pint, R, 0)
    tV = [box_v(Iq[c] * pint, R, 0) for c in range(3)]
    tVq = [(t + (1 << (SH - 1))) >> SH for t in tV]
    # pass 2: horizontal box (exact int32, modular in the kernel)
    s = box_v(sV, R, 1)
    t = [box_v(tq, R, 1) for tq in tVq]
    assert s.max() < 2 ** 31 and max(np.abs(tc).max() for tc in t) < 2 ** 31
    # statistics in kernel format
    mu = mean * 255.0 - 128.0
    M = np.rint(mu * 2.0 ** 23).astype(np.int64)
    assert np.abs(M).max() < 2 ** 31
    inv = inv6.astype(np.float32)
    idx = [[0, 1, 2], [1, 3, 4], [2, 4, 5]]
    ones = np.ones((h, w), np.int64)
    N = box_v(box_v(ones, R, 0), R, 1)
    rn = (1.0 / N)
    d = [(t[c] - ((M[c] * s + (1 << 31)) >> 32)).astype(np.float32) for c in range(3)]     # hi dword of (t<<32) - M s
    u_p = float(rng) / (2 ** PB - 1)
    kap = np.float32(2.0 ** SH * u_p / 255.0)
    rnf = rn.astype(np.float32)
    a = []
    for c in range(3):
        acc = inv[idx[c][0]] * d[0]
        acc = acc + inv[idx[c][1]] * d[1]
        acc = acc + inv[idx[c][2]] * d[2]
        a.append((acc * (kap * rnf)).astype(np.float32))
    mp = (s.astype(np.float32) * (np.float32(u_p) * rnf)).astype(np.float32)
    muf = (mu / 255.0).astype(np.float32)
    b = mp
    for c in range(3):
        b = (b - a[c] * muf[c]).astype(np.float32)
    # quantisation for stage 2
    A_max = 0.433 * float(rng) / np.sqrt(eps)
    bound = max(A_max, float(rng) + A_max * 0.87) * scale_margin
    scale = np.float32(2.0 ** 30 / (441.0 * bound)) if R == 10 else np.float32(2.0 ** 30 / ((2 * R + 1) ** 2 * bound))
    aq = [np.rint(ac * scale).astype(np.int64) for ac in a]
    bq = np.rint(b * scale).astype(np.int64)
    worst = max(max(np.abs(x).max() for x in aq), np.abs(bq).max())
    A = [box_v(box_v(x, R, 0), R, 1) for x in aq]
    B = box_v(box_v(bq, R, 0), R, 1)
    assert max(max(np.abs(x).max() for x in A), np.abs(B).max()) < 2 ** 31, "stage-2 overflow"
    qi = B * 255 + A[0] * Iq[0] + A[1] * Iq[1] + A[2] * Iq[2]
    q = qi.astype(np.float64) * (1.0 / (255.0 * float(scale))) * rn + float(vmin)
    return q.astype(np.float32), worst * 441 / 2.0 ** 31


def run_case(name, im, vol, R, eps, th, rects, seed=0):
    H, W = im.shape[:2]
    o = om.Oracle(im, im, vol, vol, windR=2 * R, eps=eps, th_col=th)
    rng = np.random.default_rng(seed)
    vmin = float(vol.min())
    worst_abs, worst_rel, fill = 0.0, 0.0, 0.0
    for fr in rects:
        x0, y0, w, h = fr
        k = int(rng.integers(0, vol.shape[0]))
        p = np.minimum(vol[k, y0:y0 + h, x0:x0 + w], np.float32(th)).astype(np.float32)
        ref = o.filter_subregion(fr, p, 0)
        got, f = emulate(o, fr, p, 0, R, eps, th, vmin)
        # only pixels at least 2R from a clip border that is not the image border are consumed by the optimiser;
        # report both the whole rect and that interior
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        worst_abs = max(worst_abs, err.max())
        worst_rel = max(worst_rel, (err / np.maximum(np.abs(ref), 0.05 * th)).max())
        fill = max(fill, f)
    print(f"{name:44s} R={R} eps={eps:g} th={th:g}: max abs err {worst_abs:.3e}  max rel err (floor 5% th) {worst_rel:.3e}  int32 fill {fill:.3f}")
    return worst_abs


if __name__ == "__main__":
    H, W, D = 200, 260, 6
    im = synth.make_guide(H, W, 1234)
    vol = synth.make_volume(D, H, W, 42)
    whole = [(0, 0, W, H)]
    for eps in (1e-4, 1e-6, 1e-2):
        run_case("synthetic guide, whole image", im, vol, 10, eps, 0.5, whole)
    # image-border cells and interior cells of the layer geometry
    cells = [(0, 0, 85, 85), (60, 40, 85, 85), (W - 85, H - 85, 85, 85), (0, 100, 55, 85), (100, 0, 130, 60)]
    run_case("synthetic guide, cell rects", im, vol, 10, 1e-4, 0.5, cells)
    # flat guide (degenerate covariance), constant zones
    flat = np.full((H, W, 3), 77, np.uint8)
    run_case("constant guide", flat, vol, 10, 1e-4, 0.5, whole)
    steps = im.copy(); steps[:, : W // 2] = 20; steps[:, W // 2:] = 230
    run_case("two flat zones with a step edge", steps, vol, 10, 1e-4, 0.5, whole)
    run_case("two flat zones with a step edge", steps, vol, 10, 1e-6, 0.5, whole)
    # cost ranges: MiddV2-like th_col 10 on a [0, 20) volume; negative costs
    run_case("volume in [0,20), th 10", im, vol * 20, 10, 1e-4, 10.0, whole)
    run_case("volume in [-1,1), th 0.5", im, vol * 2 - 1, 10, 1e-4, 0.5, whole)
    # natural image crop
    try:
        from PIL import Image
        g = np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cones", "imL.png")).convert("RGB"))[:, :, ::-1]
        g = np.ascontiguousarray(g[:H, :W])
        for eps in (1e-4, 1e-6):
            run_case("cones crop", g, vol, 10, eps, 0.5, whole)
        run_case("cones crop, smaller radius", g, vol, 4, 1e-4, 0.5, whole)
    except Exception as e:  # pragma: no cover
        print("cones crop skipped:", e)
