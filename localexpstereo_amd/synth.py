"""Seeded synthetic inputs for the matching-cost path.

Guide images are smooth + noise (low-pass filtered uniform noise plus sigma~8 gaussian noise) so the
3x3 colour covariance of the guided filter is not degenerate; volumes are iid U[0,1) float32
[D][H][W].  Pure numpy; used by tests and bench.py on both the CPU and the GPU legs.
"""
import numpy as np


def _box_blur(a, r):
    """Separable mean filter with edge replication (only used to synthesise smooth guides)."""
    for axis in (0, 1):
        pad = [(0, 0)] * a.ndim
        pad[axis] = (r + 1, r)
        c = np.cumsum(np.pad(a, pad, mode="edge"), axis=axis, dtype=np.float64)
        n = a.shape[axis]
        hi = np.take(c, np.arange(2 * r + 1, 2 * r + 1 + n), axis=axis)
        lo = np.take(c, np.arange(0, n), axis=axis)
        a = (hi - lo) / (2 * r + 1)
    return a


def make_guide(H, W, seed=1234):
    """H x W x 3 uint8 BGR guide image: smooth structure + sigma=8 noise."""
    rng = np.random.default_rng(seed)
    base = rng.uniform(0, 255, size=(H // 8 + 3, W // 8 + 3, 3))
    base = _box_blur(base, 1)
    up = np.repeat(np.repeat(base, 8, axis=0), 8, axis=1)[:H, :W]
    up = _box_blur(up, 6)
    # stretch contrast so that edges and flat zones both exist
    up = (up - up.mean()) * 2.2 + 128.0
    img = up + rng.normal(0, 8.0, size=(H, W, 3))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_volume(D, H, W, seed=42, chunk=16):
    """float32 [D][H][W] iid U[0,1) (seed 42)."""
    rng = np.random.default_rng(seed)
    vol = np.empty((D, H, W), np.float32)
    for d0 in range(0, D, chunk):
        d1 = min(D, d0 + chunk)
        vol[d0:d1] = rng.random((d1 - d0, H, W), dtype=np.float32)
    return vol


def fronto_planes(D):
    """H1: D fronto-parallel planes a=b=0, c=k."""
    p = np.zeros((D, 4), np.float32)
    p[:, 2] = np.arange(D, dtype=np.float32)
    return p


def slanted_planes(n, H, W, max_disp, seed=7):
    """H2: a,b ~ U(-.5,.5), c such that the disparity at the image centre ~ U(0, max_disp)."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    zc = rng.uniform(0, max_disp, n).astype(np.float32)
    c = (zc - a * np.float32(W / 2) - b * np.float32(H / 2)).astype(np.float32)
    p = np.zeros((n, 4), np.float32)
    p[:, 0], p[:, 1], p[:, 2] = a, b, c
    return p


# ---- synthetic stereo scene with ground truth (end-to-end runs at data-set shapes that are not in the container)
def make_scene(H, W, D, seed=3):
    """Piecewise-planar ground-truth disparity, textured left image, right image = left forward-warped by the ground truth
    (z-buffered, holes filled horizontally).  Returns (imL, imR, gt)."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    # piecewise planar disparity: background plane + a few slanted quadrilateral "objects"
    gt = 0.08 * D + 0.00004 * D * xs + 0.00006 * D * ys
    for _ in range(9):
        cx, cy = rng.uniform(0.1, 0.9) * W, rng.uniform(0.1, 0.9) * H
        rw, rh = rng.uniform(0.06, 0.2) * W, rng.uniform(0.08, 0.25) * H
        m = (np.abs(xs - cx) < rw) & (np.abs(ys - cy) < rh)
        a, b = rng.uniform(-0.03, 0.03), rng.uniform(-0.03, 0.03)
        z = rng.uniform(0.25, 0.8) * D + a * (xs - cx) + b * (ys - cy)
        gt = np.where(m & (z > gt), z, gt)
    gt = np.clip(gt, 1, D - 2).astype(np.float32)
    # texture: multi-scale noise, colour
    tex = np.zeros((H, W, 3), np.float32)
    for s in (2, 5, 13, 37):
        n = rng.uniform(0, 1, (H // s + 2, W // s + 2, 3)).astype(np.float32)
        tex += np.kron(n, np.ones((s, s, 1), np.float32))[:H, :W] / 4
    imL = np.clip(tex * 255, 0, 255).astype(np.uint8)
    # right view by forward warping (nearest, z-buffered by processing small disparities first), holes filled from the left
    imR = np.zeros_like(imL)
    filled = np.zeros((H, W), bool)
    order = np.argsort(gt, axis=None)
    yy, xx = np.unravel_index(order, gt.shape)
    xr = np.rint(xx - gt[yy, xx]).astype(int)
    ok = (xr >= 0) & (xr < W)
    imR[yy[ok], xr[ok]] = imL[yy[ok], xx[ok]]
    filled[yy[ok], xr[ok]] = True
    for y in range(H):                                    # horizontal hole filling
        row = filled[y]
        if not row.all():
            idx = np.where(row, np.arange(W), -1)
            np.maximum.accumulate(idx, out=idx)
            idx[idx < 0] = np.argmax(row)
            imR[y] = imR[y, idx]
    return imL, imR, gt


def ad_volume(imL, imR, D, device):
    """vol[d][y][x] = min(1, mean_c |L(y,x,c) - R(y,x-d,c)| / 64): a stand-in for the MC-CNN matching cost in [0,1]."""
    import torch
    L = torch.from_numpy(imL).to(device).float()
    R = torch.from_numpy(imR).to(device).float()
    H, W = L.shape[:2]
    vol = torch.empty((D, H, W), device=device, dtype=torch.float32)
    for d in range(D):
        Rs = torch.roll(R, shifts=d, dims=1)
        vol[d] = ((L - Rs).abs().mean(dim=2) / 64.0).clamp_(max=1.0)
    return vol




def make_scene_three_surfaces(H, W, D, seed=4242):
    """The scene of the C++ host demo (localexpstereo_amd/host/DemoScene.h: make_scene) for the Python driver: three large slanted
    surfaces separated by a vertical and a diagonal boundary, a guide image whose colour follows the surface, and a noisy truncated
    absolute-difference volume built from the ground truth -- `min(1, 0.12 |d - gt|) * 0.8 + U[0, 0.2)`.  Proposals flip most of a
    coarse cell at once on it, which is what makes its expansion moves hard for a max-flow code.  Same
    construction as the C++ one with numpy's generator instead of cv::RNG.  Returns (imL, imR, gt, volL): the right image is the left
    one warped by the ground truth as in make_scene; volL is float32 [D][H][W]."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W]
    surf = [(0.02, 0.01, 0.25 * D), (-0.03, 0.0, 0.6 * D), (0.0, -0.02, 0.45 * D)]
    col = np.array([[200, 60, 40], [40, 180, 70], [60, 70, 210]], np.int64)
    k = np.where(xs < W // 3, 0, np.where(xs + ys // 2 < (2 * W) // 3, 1, 2))
    gt = np.zeros((H, W), np.float32)
    for i, (a, b, c) in enumerate(surf):
        z = (np.float32(a) * xs.astype(np.float32) + np.float32(b) * ys.astype(np.float32)) + np.float32(c)
        gt = np.where(k == i, z, gt)
    gt = np.clip(gt, 1.0, D - 2.0).astype(np.float32)
    wave = (10.0 * np.sin(0.15 * xs + 0.1 * ys)).astype(np.int64)                     # (int) of a double: truncation towards zero
    noise = np.trunc(rng.uniform(-12.0, 12.0, (H, W, 3))).astype(np.int64)
    imL = np.clip(col[k] + noise + wave[..., None], 0, 255).astype(np.uint8)
    imR = np.zeros_like(imL)
    filled = np.zeros((H, W), bool)
    order = np.argsort(gt, axis=None)
    yy, xx = np.unravel_index(order, gt.shape)
    xr = np.rint(xx - gt[yy, xx]).astype(int)
    ok = (xr >= 0) & (xr < W)
    imR[yy[ok], xr[ok]] = imL[yy[ok], xx[ok]]
    filled[yy[ok], xr[ok]] = True
    for y in range(H):
        row = filled[y]
        if not row.all():
            idx = np.where(row, np.arange(W), -1)
            np.maximum.accumulate(idx, out=idx)
            idx[idx < 0] = np.argmax(row)
            imR[y] = imR[y, idx]
    vol = np.empty((D, H, W), np.float32)
    for d in range(D):
        e = np.abs(np.float32(d) - gt)
        vol[d] = np.minimum(np.float32(1.0), np.float32(0.12) * e) * np.float32(0.8) + rng.random((H, W), dtype=np.float32) * np.float32(0.2)
    return imL, imR, gt, vol
