"""Generates the committed data fixtures under tests/golden/ (run once, in the build container).

  cones_crop.npz : 120 x 96 BGR uint8 crops of the reference's bundled Middlebury-V2 "cones" pair
                   (data/MiddV2/cones/imL.png, imR.png -- dataset images, not source code), used as
                   a natural-image guide for the guided-filter parity tests.
  golden_unary.npz: oracle outputs (float32) for a handful of (filterRect, targetRect, plane, mode)
                   calls on a seeded synthetic 16 x 96 x 120 volume + the cones crop.  Produced by
                   oracle/libles_oracle.so (the CPU restatement: parity is unpinned against the real
                   reference, see oracle/les_oracle.h); it pins the oracle against regressions and
                   gives the GPU tests a fixture that does not need the oracle at run time.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def main():
    from PIL import Image
    ref = "/root/reference/data/MiddV2/cones"
    out = {}
    for name in ("imL", "imR"):
        rgb = np.asarray(Image.open(os.path.join(ref, name + ".png")).convert("RGB"))
        bgr = rgb[:, :, ::-1]
        out[name] = np.ascontiguousarray(bgr[150:246, 200:320])  # 96 rows x 120 cols
    # wider right-view crop (the disparity search range of the crop) and the ground-truth crop, for the
    # end-to-end quality test: groundtruth.png / scale (data/MiddV2/cones/info.txt: "4 59"), 0 = unknown
    rgbR = np.asarray(Image.open(os.path.join(ref, "imR.png")).convert("RGB"))[:, :, ::-1]
    out["imR_wide"] = np.ascontiguousarray(rgbR[150:246, 200 - 64:320])          # 96 rows x 184 cols, col 64 == x 200
    gt = np.asarray(Image.open(os.path.join(ref, "groundtruth.png"))).astype(np.float32) / 4.0
    out["gt"] = np.ascontiguousarray(gt[150:246, 200:320])
    np.savez_compressed(os.path.join(HERE, "cones_crop.npz"), **out)

    from oracle import oracle as om
    from localexpstereo_amd import synth
    H, W, D = 96, 120, 16
    volL = synth.make_volume(D, H, W, seed=42)
    volR = synth.make_volume(D, H, W, seed=43)
    o = om.Oracle(out["imL"], out["imR"], volL, volR, windR=20, eps=1e-4, th_col=0.5)
    calls = [
        # (mode, filterRect, targetRect, plane)
        (0, (0, 0, 62, 62), (0, 0, 42, 42), (0.0, 0.0, 3.0, 0.0)),
        (0, (19, 22, 82, 74), (39, 42, 42, 34), (0.05, -0.03, 4.25, 0.0)),
        (1, (58, 34, 62, 62), (78, 54, 42, 42), (-0.11, 0.07, 9.5, 0.0)),
        (0, (0, 0, 120, 96), (0, 0, 120, 96), (0.01, 0.02, 2.125, 0.0)),
        (1, (30, 0, 90, 60), (50, 0, 50, 40), (0.3, 0.2, -20.0, 0.0)),     # mostly invalid / clamped low
        (0, (0, 36, 70, 60), (0, 56, 50, 40), (0.0, 0.0, 15.0, 0.0)),      # d >= MAXD clamp
        (0, (40, 40, 41, 41), (60, 60, 1, 1), (0.02, 0.01, 5.0, 0.0)),     # 1x1 target (warm start form)
    ]
    gold = {"volL_seed": 42, "volR_seed": 43, "n": len(calls)}
    for i, (mode, fr, tr, pl) in enumerate(calls):
        cm = o.unary(fr, tr, pl, mode=mode, check=True)
        x, y, w, h = tr
        gold[f"mode{i}"] = mode
        gold[f"fr{i}"] = np.array(fr, np.int32)
        gold[f"tr{i}"] = np.array(tr, np.int32)
        gold[f"plane{i}"] = np.array(pl, np.float32)
        gold[f"out{i}"] = cm[y:y + h, x:x + w].copy()
    np.savez_compressed(os.path.join(HERE, "golden_unary.npz"), **gold)
    print("wrote fixtures")


if __name__ == "__main__":
    main()
