"""Shared helpers for the test-suite (numpy only)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cones_crop():
    z = np.load(os.path.join(GOLDEN, "cones_crop.npz"))
    return z["imL"], z["imR"]


def box_sum(a, R):
    """Un-normalised (2R+1)^2 window sum with zero padding, float64, via padded cumsum."""
    a = np.asarray(a, np.float64)
    H, W = a.shape
    p = np.pad(a, ((R + 1, R), (R + 1, R)))
    c = p.cumsum(0).cumsum(1)
    k = 2 * R + 1
    return c[k:k + H, k:k + W] - c[0:H, k:k + W] - c[k:k + H, 0:W] + c[0:H, 0:W]


def guided_filter_numpy(img_u8, p, R, eps):
    """Independent float64 restatement of the colour guided filter (He et al. Eq. 14-16) with the
    reference's border convention (zero-padded sums divided by the true window pixel count N)."""
    I = img_u8.astype(np.float64) * (1.0 / 255)
    p = np.asarray(p, np.float64)
    N = box_sum(np.ones(p.shape), R)
    mI = [box_sum(I[..., c], R) / N for c in range(3)]
    mp = box_sum(p, R) / N
    cov = [box_sum(I[..., c] * p, R) / N - mI[c] * mp for c in range(3)]
    S = np.empty(p.shape + (3, 3))
    for a in range(3):
        for b in range(3):
            S[..., a, b] = box_sum(I[..., a] * I[..., b], R) / N - mI[a] * mI[b] + (eps if a == b else 0.0)
    a_vec = np.linalg.solve(S, np.stack(cov, -1)[..., None])[..., 0]
    b = mp - sum(a_vec[..., c] * mI[c] for c in range(3))
    q = (box_sum(b, R) + sum(box_sum(a_vec[..., c], R) * I[..., c] for c in range(3))) / N
    return q


def relerr(x, ref, floor):
    x = np.asarray(x, np.float64)
    ref = np.asarray(ref, np.float64)
    return np.abs(x - ref) / np.maximum(np.abs(ref), floor)
