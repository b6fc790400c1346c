#!/usr/bin/env python
"""Writes the synthetic Adirondack-shape scene of tools/e2e_bench.py to a directory as raw files, with the left cost volume as the
device ingest leaves it (fillOutOfView applied), so that the C++ host driver can run the same data:

  python tools/dump_scene.py --out /tmp/scene && localexpstereo_amd/host/les_host_demo scene /tmp/scene 5 2

Files: meta.txt ("W H D"), imL.bgr / imR.bgr (uint8 H x W x 3), volL.f32 (float32 [D][H][W]), gt.f32 (float32 H x W).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                               # noqa: E402
from localexpstereo_amd import io as lio                          # noqa: E402
from localexpstereo_amd.synth import ad_volume, make_scene        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--width", type=int, default=1436)
    ap.add_argument("--height", type=int, default=992)
    ap.add_argument("--ndisp", type=int, default=256)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    H, W, D = a.height, a.width, a.ndisp
    imL, imR, gt = make_scene(H, W, D)
    volL = ad_volume(imL, imR, D, "cuda").cpu().numpy()
    tl, tr = lio.ingest_volumes(volL, None, device="cuda")
    np.ascontiguousarray(imL, np.uint8).tofile(os.path.join(a.out, "imL.bgr"))
    np.ascontiguousarray(imR, np.uint8).tofile(os.path.join(a.out, "imR.bgr"))
    tl.cpu().numpy().astype(np.float32).tofile(os.path.join(a.out, "volL.f32"))
    np.ascontiguousarray(gt, np.float32).tofile(os.path.join(a.out, "gt.f32"))
    open(os.path.join(a.out, "meta.txt"), "w").write(f"{W} {H} {D}\n")
    print("scene written to", a.out, (W, H, D))


if __name__ == "__main__":
    main()
