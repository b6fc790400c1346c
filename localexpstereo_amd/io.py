"""Data formats on either side of the hot path ("next" rows N3/N4 of SURVEY.md section 8(f)).

* matching-cost volumes: headerless little-endian float32 `[ndisp][H][W]` files `im0.acrt` / `im1.acrt`
  (cvutils::io::loadMatBinary(..., readHeader=false), LES/Utilities.hpp:171-201; call sites LES/main.cpp:353-368) and
  their preparation on the device: fillOutOfView (LES/main.cpp:146-176), convertVolumeL2R when im1.acrt is absent
  (LES/main.cpp:178-199, 360-364)
* PFM disparity maps (cvutils::io::read_pfm_file / save_pfm_file, LES/Utilities.hpp:20-137): rows stored bottom-up,
  the writer always emits scale -1/255 (little endian)
* data-set folders (loadData, LES/main.cpp:201-268; Calib, LES/main.cpp:84-144)
* the Evaluator's bad-pixel rates (LES/Evaluator.h:77-81, 133-140)

Host-side file handling is numpy; the volume preparation runs on the GPU through the C ABI.
"""
import os
import re

import numpy as np

from . import api


# ------------------------------------------------------------------------------------------------
# PFM
# ------------------------------------------------------------------------------------------------
def write_pfm(path, image):
    """save_pfm_file (LES/Utilities.hpp:84-137): 'Pf' (1 channel) or 'PF' (3), '%d %d', '%lf' of -1/255, rows bottom-up."""
    a = np.asarray(image, np.float32)
    if a.ndim == 2:
        tag, ch = "Pf", 1
    elif a.ndim == 3 and a.shape[2] == 3:
        tag, ch = "PF", 3
    else:
        raise ValueError("PFM images have 1 or 3 channels")
    h, w = a.shape[:2]
    with open(path, "wb") as f:
        f.write(("%s\n%d %d\n%f\n" % (tag, w, h, -1.0 / 255.0)).encode("ascii"))
        f.write(np.ascontiguousarray(a[::-1]).astype("<f4").tobytes())


def read_pfm(path):
    """read_pfm_file (LES/Utilities.hpp:20-82): negative scale = little endian; the pixel block is the last
    w*h*channels floats of the file (the reference seeks from the end); rows are stored bottom-up."""
    with open(path, "rb") as f:
        data = f.read()
    m = re.match(rb"^(P[fF])\s+(\d+)\s+(\d+)\s+([-+]?(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][-+]?[0-9]+)?)\s", data)    # %lf of the reference: 1e-05, -3.9e-03, ...
    if not m:
        raise ValueError(f"{path}: not a 1/3 channel PFM file")
    ch = 1 if m.group(1) == b"Pf" else 3
    w, h, scale = int(m.group(2)), int(m.group(3)), float(m.group(4))
    n = w * h * ch
    if len(data) < n * 4:
        raise ValueError(f"{path}: expected {n} floats")
    a = np.frombuffer(data[len(data) - n * 4:], "<f4" if scale < 0 else ">f4").astype(np.float32)
    a = a.reshape(h, w, ch)[::-1]
    return np.ascontiguousarray(a[..., 0] if ch == 1 else a)


# ------------------------------------------------------------------------------------------------
# cost volumes
# ------------------------------------------------------------------------------------------------
def load_cost_volume(path, ndisp, H, W, mmap=True):
    """Raw float32 [ndisp][H][W] (LES/main.cpp:353-357).  Returns None when the file does not exist."""
    if not os.path.exists(path):
        return None
    n = int(ndisp) * int(H) * int(W)
    if os.path.getsize(path) < n * 4:
        raise ValueError(f"{path}: {os.path.getsize(path)} bytes, expected {n * 4} for a {ndisp}x{H}x{W} float32 volume")
    if mmap:
        return np.memmap(path, dtype="<f4", mode="r", shape=(int(ndisp), int(H), int(W)))
    return np.fromfile(path, dtype="<f4", count=n).reshape(int(ndisp), int(H), int(W))


def save_cost_volume(path, vol):
    np.ascontiguousarray(vol, "<f4").tofile(path)


def ingest_volumes(volL, volR=None, device="cuda", lib=None):
    """The MiddV3 volume preparation of LES/main.cpp:353-368 on the device.  volL / volR: host arrays (or memmaps) of
    shape [D][H][W]; volR None => synthesised from the left volume (convertVolumeL2R) before the fills, exactly in
    the reference's order.  Returns two torch device tensors."""
    import torch
    dev = torch.device(device)
    idx = dev.index or 0 if dev.type == "cuda" else 0
    def upload(v):
        # never alias the caller's (possibly read-only, memory-mapped) array: the fills work in place
        if dev.type == "cuda":
            a = np.ascontiguousarray(v, np.float32)
            if not a.flags.writeable:                     # read-only memmap: stage through a writable (page-cache backed) copy
                a = np.array(a, np.float32, copy=True)
            return torch.from_numpy(a).to(dev)
        return torch.from_numpy(np.array(v, np.float32, copy=True))

    tl = upload(volL)
    D, H, W = tl.shape
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    api.fill_out_of_view(tl.data_ptr(), D, H, W, 0, device=idx, stream=stream, lib=lib)           # LES/main.cpp:358
    if volR is not None:
        tr = upload(volR)
        assert tuple(tr.shape) == (D, H, W)
    else:
        tr = torch.empty_like(tl)
        api.convert_volume_l2r(tl.data_ptr(), tr.data_ptr(), D, H, W, device=idx, stream=stream, lib=lib)   # LES/main.cpp:363
    api.fill_out_of_view(tr.data_ptr(), D, H, W, 1, device=idx, stream=stream, lib=lib)           # LES/main.cpp:365
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return tl, tr


# ------------------------------------------------------------------------------------------------
# data-set folders
# ------------------------------------------------------------------------------------------------
def read_calib(path):
    """Calib (LES/main.cpp:84-144): 'key = value' lines of the Middlebury v3 calib.txt."""
    out = {}
    with open(path) as f:
        for line in f:
            if "=" not in line:
                continue
            k, v = (t.strip() for t in line.split("=", 1))
            if k in ("cam0", "cam1"):
                out[k] = np.array([[float(t) for t in row.split()] for row in v.strip("[]").split(";")], np.float32)
            else:
                try:
                    out[k] = int(v)
                except ValueError:
                    out[k] = float(v)
    return out


def _imread_bgr(path, gray=False):
    from PIL import Image           # only the data-set loader needs an image decoder
    if not os.path.exists(path):
        return None
    im = Image.open(path)
    if gray:
        return np.asarray(im.convert("L"))
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def load_data(input_dir, ndisp=0):
    """loadData (LES/main.cpp:201-268).  Returns dict(imL, imR [BGR uint8], dispGT float32 (inf where unknown for the
    PNG ground truth), nonocc bool, ndisp, gt_prec)."""
    d = input_dir if input_dir.endswith(os.sep) else input_dir + os.sep
    gt_prec = -1.0
    info = d + "info.txt"
    if os.path.exists(info):
        toks = open(info).read().split()
        gt_scale, nd = int(toks[0]), int(toks[1])
        gt_prec = 1.0 / gt_scale
        if ndisp <= 0:
            ndisp = nd
    elif os.path.exists(d + "calib.txt"):
        c = read_calib(d + "calib.txt")
        if ndisp <= 0:
            ndisp = int(c.get("ndisp", 0))
    if ndisp <= 0:
        raise ValueError("ndisp is not specified")
    imL, imR = _imread_bgr(d + "imL.png"), _imread_bgr(d + "imR.png")
    if imL is None or imR is None:
        imL, imR = _imread_bgr(d + "im0.png"), _imread_bgr(d + "im1.png")
    if imL is None or imR is None:
        raise FileNotFoundError(f"image pairs (im0.png, im1.png) or (imL.png, imR.png) not found in {d}")
    gt = _imread_bgr(d + "groundtruth.png", gray=True)
    if gt is not None:
        gtf = gt.astype(np.float32)
        if gt_prec > 0:
            gtf = gtf * np.float32(gt_prec)
        gtf[gt == 0] = np.inf
        gt = gtf
    elif os.path.exists(d + "disp0GT.pfm"):
        gt = read_pfm(d + "disp0GT.pfm")
    else:
        gt = np.zeros(imL.shape[:2], np.float32)
    nonocc = _imread_bgr(d + "nonocc.png", gray=True)
    if nonocc is None:
        nonocc = _imread_bgr(d + "mask0nocc.png", gray=True)
    nonocc = (nonocc == 255) if nonocc is not None else np.ones(imL.shape[:2], bool)
    return dict(imL=imL, imR=imR, dispGT=gt, nonocc=nonocc, ndisp=int(ndisp), gt_prec=gt_prec)


# ------------------------------------------------------------------------------------------------
# Evaluator
# ------------------------------------------------------------------------------------------------
class Evaluator:
    """Bad-pixel rates of LES/Evaluator.h: valid = gt > 0 and finite (:77); a pixel is good when |d - gt| <= threshold
    (:133); all = 100 (1 - good&valid / valid), nonocc = 100 (1 - good&nonocc / |nonocc|) (:137-140)."""

    def __init__(self, dispGT, nonocc_mask, error_threshold=0.5):
        self.gt = np.asarray(dispGT, np.float32)
        self.nonocc = np.asarray(nonocc_mask, bool)
        self.valid = (self.gt > 0) & np.isfinite(self.gt)
        self.threshold = float(error_threshold)

    def evaluate(self, disp):
        with np.errstate(invalid="ignore"):
            good = np.abs(np.asarray(disp, np.float32) - self.gt) <= np.float32(self.threshold)
        nv, nn = int(self.valid.sum()), int(self.nonocc.sum())
        all_ = 100.0 * (1.0 - (good & self.valid).sum() / nv) if nv else float("nan")
        non_ = 100.0 * (1.0 - (good & self.nonocc).sum() / nn) if nn else float("nan")
        return float(all_), float(non_)
