"""Numerical probe of the fixed-point formulation of the guided-filter aggregation (DESIGN.md "Numerics").

Emulates in numpy -- exact integer arithmetic where les_march_kernel uses integers, float32 (one rounding per fma) where it uses
float32 -- what the kernel computes, with the same scales the host derives (les_hip.hip: build_march_view), and compares it with the
CPU oracle (double, the reference's default "GF").  The variants are the design alternatives weighed in round 4:

  r3       round-3 arithmetic: 22-bit uncentred cost, 64-bit vertical sums >> 9, fp64 stage-2 vertical sums and final combination
  r4       round-4 arithmetic: centred PB-bit cost in int32, vertical sums >> 5, M with 24 fraction bits, stage-2 horizontal sums
           rounded to 2^-S2 per row, int32 vertical sums, fp32 combination
  r4-late  the same, but the stage-2 sums are accumulated exactly (64-bit) and rounded once at the end

TEST / DESIGN TOOLING: it imports the oracle, never the product package's native code.

  python tools/fixedpoint_probe.py            the bound proofs + the small cases
  python tools/fixedpoint_probe.py --full     + the full-size linearity property of tests/test_gpu_parity.py (1500 x 1000, th_col = 1)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as om  # noqa: E402
from localexpstereo_amd import synth  # noqa: E402

f32 = np.float32


def fma32(a, b, c):
    """float32 fma: the product of two float32 is exact in float64; one rounding to float32 (the float64 sum is exact or far below half a float32 ulp off)."""
    return (a.astype(np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def box(a, R, axis):
    """zero-padded 2R+1 window sum along axis, exact for int64 arrays"""
    n = a.shape[axis]
    pad = [(0, 0)] * a.ndim
    pad[axis] = (R + 1, R)
    c = np.cumsum(np.pad(a, pad), axis=axis)
    return np.take(c, np.arange(2 * R + 1, 2 * R + 1 + n), axis=axis) - np.take(c, np.arange(0, n), axis=axis)


def march_pb(R):
    return 20 if (2 * R + 1) ** 2 < 512 else 19


def prove_bounds():
    """The integer ranges the kernel relies on (les_march.h: MarchCfg static_asserts), for every radius it is instantiated for."""
    SH, MB, SL, S2 = 5, 24, 3, 4
    for R in range(1, 12):
        K, PB = 2 * R + 1, march_pb(R)
        pmax = 1 << (PB - 1)                                  # |count| <= 2^(PB-1)
        assert pmax < (1 << 23) and 128 < (1 << 23)           # both operands of v_mul_i32_i24 are 24-bit
        assert K * 128 * pmax < (1 << 31), R                  # vertical sums of Iq * count
        assert K * pmax < (1 << 31)
        assert K * ((K * 128 * pmax + 16) >> SH) < (1 << 31), R   # horizontal sums of the shifted vertical sums
        assert (K * K * pmax) << SL < (1 << 31), R            # s << SL
        assert 128 << MB <= (1 << 31)                         # |M_c| <= 2^31 (the kernel clamps +2^31 to 2^31 - 1)
        assert K * (1 << (30 - S2)) < (1 << 31), R            # vertical sums of the rounded stage-2 box sums (|h| < 2^30 by the scale)
    print("integer ranges hold for R = 1..11 (PB = 20 up to R = 10, 19 for R = 11)")


class Setup:
    """per-(oracle, mode) constants, as build_march_view derives them"""

    def __init__(self, o, vol, th, R, mode=0):
        st = o.stats(mode)                                   # 13 x H x W double: I(3), mean(3), inv(6), N
        self.I = st[0:3]
        self.mean = st[3:6]
        self.inv = st[6:12].astype(f32)
        self.Iq = np.rint(self.I * 255).astype(np.int64) - 128
        self.R, self.th = R, float(th)
        vmin = min(float(vol.min()), 0.5 * float(th))
        self.vmin = vmin
        self.range = float(th) - vmin
        dmax = float(max(self.inv[0].max(), self.inv[3].max(), self.inv[5].max()))
        K = 2 * R + 1
        Ba = 0.5 * self.range * np.sqrt(dmax)
        Bb = self.range + 1.5 * Ba
        self.scale = 2.0 ** 30 / (K * Bb * 1.25)


def emulate(S, fr, p, variant="r4", PB=None, SH=5, S2=4):
    """p: float32 raw truncated cost over the filter rect fr = (x, y, w, h); returns q float32 over fr"""
    x0, y0, w, h = fr
    R = S.R
    sl = (slice(None), slice(y0, y0 + h), slice(x0, x0 + w))
    Iq, mean, inv = S.Iq[sl], S.mean[sl], S.inv[sl]
    ones = np.ones((h, w), np.int64)
    nx, ny = box(ones, R, 1), box(ones, R, 0)
    rnx, rny = (1.0 / nx).astype(f32), (1.0 / ny).astype(f32)
    mu_u8 = mean * 255.0 - 128.0
    idx = [[0, 1, 2], [1, 3, 4], [2, 4, 5]]
    scale = S.scale
    if variant == "r3":
        PB, SH = 22, 9
        sp = f32((2 ** PB - 1) / S.range)
        up = S.range / (2 ** PB - 1)
        pint = np.floor((p.astype(np.float64) * np.float64(sp) + np.float64(f32(0.5 - S.vmin * float(sp)))).astype(f32)).astype(np.int64)
        off = S.vmin
        sV = box(pint, R, 0)
        tV = [(box(Iq[c] * pint, R, 0) + (1 << (SH - 1))) >> SH for c in range(3)]
        s = box(sV, R, 1)
        t = [box(x, R, 1) for x in tV]
        M = np.rint(mu_u8 * 2.0 ** 23).astype(np.int64)
        d = [(t[c] - ((M[c] * s + (1 << 31)) >> 32)).astype(f32) for c in range(3)]
        muf = (mu_u8 / 255.0).astype(f32)
    else:
        PB = PB or march_pb(R)
        sp = f32((2 ** PB - 1) / S.range)
        up = 1.0 / float(sp)
        c0 = np.rint(-S.vmin * float(sp)) - 2 ** (PB - 1)
        pint = np.rint(p.astype(np.float64) * np.float64(sp) + (12582912.0 + c0)).astype(np.int64) - 12582912
        assert np.abs(pint).max() <= 2 ** (PB - 1)
        off = float(f32(-c0 * up))
        sV = box(pint, R, 0)
        tVx = [box(Iq[c] * pint, R, 0) for c in range(3)]
        assert max(np.abs(x).max() for x in tVx) < 2 ** 31
        tV = [(x + (1 << (SH - 1))) >> SH for x in tVx]
        s = box(sV, R, 1)
        t = [box(x, R, 1) for x in tV]
        assert np.abs(s << 3).max() < 2 ** 31 and max(np.abs(x).max() for x in t) < 2 ** 31
        MBITS = 32 - SH - 3
        M = np.clip(np.rint(-mu_u8 * 2.0 ** MBITS), -2 ** 31, 2 ** 31 - 1).astype(np.int64)      # the record holds MINUS the centred mean
        d = [(t[c] + ((M[c] * (s << 3) + (1 << 31)) >> 32)).astype(f32) for c in range(3)]
        Mf = M.astype(f32)
        kmu = f32(1.0 / (2.0 ** MBITS * 255.0))
    kapS, upS = f32((1 << SH) * up / 255.0 * scale), f32(up * scale)
    ka = ((kapS * rnx) * rny).astype(f32)
    a = []
    for c in range(3):
        acc = (inv[idx[c][0]] * d[0]).astype(f32)
        acc = fma32(inv[idx[c][1]], d[1], acc)
        acc = fma32(inv[idx[c][2]], d[2], acc)
        a.append((acc * ka).astype(f32))
    mp = (s.astype(f32) * ((upS * rnx) * rny).astype(f32)).astype(f32)
    if variant == "r3":
        b = mp
        for c in range(3):
            b = fma32(-a[c], muf[c], b)
    else:
        tmu = (a[0] * Mf[0]).astype(f32)
        tmu = fma32(a[1], Mf[1], tmu)
        tmu = fma32(a[2], Mf[2], tmu)
        b = fma32(kmu * np.ones_like(tmu), tmu, mp)
    q4 = [np.floor(x.astype(np.float64) + 0.5).astype(np.int64) for x in (a[0], a[1], a[2], b)]       # v_cvt_rpi_i32_f32
    hq = [box(x, R, 1) for x in q4]
    assert max(np.abs(x).max() for x in hq) < 2 ** 30, "stage-2 overflow"
    if variant == "r3":
        A = [box(x, R, 0) for x in hq]
        qi = A[3] * 255 + A[0] * Iq[0] + A[1] * Iq[1] + A[2] * Iq[2]
        c_lane = (1.0 / (255.0 * scale)) / nx
        return fma32((qi.astype(np.float64) * c_lane).astype(f32), rny, f32(off))
    if variant == "r4":
        A = [box((x + (1 << (S2 - 1))) >> S2, R, 0) for x in hq]
    else:                                                   # r4-late: exact 64-bit vertical sums, one rounding at the end
        A = [(box(x, R, 0) + (1 << (S2 - 1))) >> S2 for x in hq]
    assert max(np.abs(x).max() for x in A) < 2 ** 31
    Af = [x.astype(f32) for x in A]
    If = Iq.astype(f32)
    acc = (Af[3] * f32(255.0)).astype(f32)
    for c in range(3):
        acc = fma32(Af[c], If[c], acc)
    c_lane = (f32((1 << S2) / (255.0 * scale)) * rnx).astype(f32)
    return fma32((acc * c_lane).astype(f32), rny, f32(off))


def run_case(name, im, vol, R, eps, th, rects, variants=("r3", "r4", "r4-late"), seed=0):
    o = om.Oracle(im, im, vol, vol, windR=2 * R, eps=eps, th_col=th)
    S = Setup(o, vol, th, R)
    rng = np.random.default_rng(seed)
    worst = {v: 0.0 for v in variants}
    for fr in rects:
        x0, y0, w, h = fr
        k = int(rng.integers(0, vol.shape[0]))
        p = np.minimum(vol[k, y0:y0 + h, x0:x0 + w], f32(th)).astype(f32)
        ref = o.filter_subregion(fr, p, 0).astype(np.float64)
        for v in variants:
            got = emulate(S, fr, p, v)
            worst[v] = max(worst[v], float(np.abs(got.astype(np.float64) - ref).max()))
    print(f"{name:40s} R={R:2d} eps={eps:g} th={th:g}: max abs err  " + "  ".join(f"{v} {worst[v]:.2e}" for v in variants))
    return worst


def full_size_linearity(variants=("r3", "r4", "r4-late")):
    """tests/test_gpu_parity.py::test_full_size_linearity_in_cost: q(c = 4.5) against (q(4) + q(5)) / 2, bound 5e-7"""
    H, W, D, th = 1000, 1500, 8, 1.0
    im = synth.make_guide(H, W, 1234)
    vol = synth.make_volume(D, H, W, 42)
    o = om.Oracle(im, im, vol, vol, windR=20, eps=1e-4, th_col=th)
    S = Setup(o, vol, th, 10)
    fr = (0, 0, W, H)
    p4, p5 = np.minimum(vol[4], f32(th)), np.minimum(vol[5], f32(th))
    p45 = np.minimum((f32(0.5) * vol[4] + f32(0.5) * vol[5]).astype(f32), f32(th))
    refs = [o.filter_subregion(fr, p, 0).astype(np.float64) for p in (p4, p5, p45)]
    for v in variants:
        q = [emulate(S, fr, p, v) for p in (p4, p5, p45)]
        lin = float(np.abs(q[2] - f32(0.5) * (q[0] + q[1])).max())
        err = max(float(np.abs(q[i].astype(np.float64) - refs[i]).max()) for i in range(3))
        print(f"full-size linearity, {v:8s}: max |q(4.5) - (q(4) + q(5)) / 2| = {lin:.3e}   max abs err vs the double oracle = {err:.3e}")


if __name__ == "__main__":
    prove_bounds()
    H, W, D = 200, 260, 6
    im = synth.make_guide(H, W, 1234)
    vol = synth.make_volume(D, H, W, 42)
    whole = [(0, 0, W, H)]
    for eps in (1e-4, 1e-6, 1e-2):
        run_case("synthetic guide, whole image", im, vol, 10, eps, 0.5, whole)
    cells = [(0, 0, 85, 85), (60, 40, 85, 85), (W - 85, H - 85, 85, 85), (0, 100, 55, 85), (100, 0, 130, 60)]
    run_case("synthetic guide, cell rects", im, vol, 10, 1e-4, 0.5, cells)
    flat = np.full((H, W, 3), 77, np.uint8)
    run_case("constant guide", flat, vol, 10, 1e-4, 0.5, whole)
    steps = im.copy(); steps[:, : W // 2] = 20; steps[:, W // 2:] = 230
    run_case("two flat zones with a step edge", steps, vol, 10, 1e-4, 0.5, whole)
    black = im.copy(); black[:, : W // 2] = 0
    run_case("half black image, costs at the threshold", black, np.full_like(vol, 0.75), 10, 1e-4, 0.5, whole)
    run_case("volume in [0,20), th 10", im, vol * 20, 10, 1e-4, 10.0, whole)
    run_case("volume in [-1,1), th 0.5", im, vol * 2 - 1, 10, 1e-4, 0.5, whole)
    run_case("synthetic guide, radius 7", im, vol, 7, 1e-4, 0.5, whole)
    run_case("synthetic guide, radius 4", im, vol, 4, 1e-4, 0.5, whole)
    try:
        from PIL import Image
        g = np.asarray(Image.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cones", "imL.png")).convert("RGB"))[:, :, ::-1]
        g = np.ascontiguousarray(g[:H, :W])
        for eps in (1e-4, 1e-6):
            run_case("cones crop", g, vol, 10, eps, 0.5, whole)
    except Exception as e:  # pragma: no cover
        print("cones crop skipped:", e)
    if "--full" in sys.argv:
        full_size_linearity()
