"""Data formats either side of the path (localexpstereo_amd/io.py): PFM, raw cost volumes, data-set folders, Evaluator."""
import os

import numpy as np
import pytest

from localexpstereo_amd import io as lio

CONES = "/root/reference/data/MiddV2/cones"


def test_pfm_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 60, (7, 11)).astype(np.float32)
    p = str(tmp_path / "d.pfm")
    lio.write_pfm(p, a)
    raw = open(p, "rb").read()
    assert raw.startswith(b"Pf\n11 7\n-0.003922\n")                       # '%lf' of -1/255 (LES/Utilities.hpp:99)
    body = np.frombuffer(raw[-7 * 11 * 4:], "<f4").reshape(7, 11)
    assert np.array_equal(body[0], a[-1]) and np.array_equal(body[-1], a[0])   # rows bottom-up
    assert np.array_equal(lio.read_pfm(p), a)
    c = rng.uniform(0, 1, (5, 4, 3)).astype(np.float32)
    lio.write_pfm(p, c)
    assert open(p, "rb").read().startswith(b"PF\n4 5\n")
    assert np.array_equal(lio.read_pfm(p), c)
    # big-endian file with positive scale (official Middlebury ground truth uses either)
    with open(p, "wb") as f:
        f.write(b"Pf\n11 7\n1.0\n")
        f.write(a[::-1].astype(">f4").tobytes())
    assert np.array_equal(lio.read_pfm(p), a)
    with open(p, "wb") as f:
        f.write(b"P6\n1 1\n255\n")
    with pytest.raises(ValueError):
        lio.read_pfm(p)


def test_cost_volume_files(tmp_path):
    rng = np.random.default_rng(1)
    v = rng.uniform(0, 1, (5, 6, 7)).astype(np.float32)
    p = str(tmp_path / "im0.acrt")
    lio.save_cost_volume(p, v)
    assert os.path.getsize(p) == v.size * 4                                 # headerless
    for mm in (True, False):
        assert np.array_equal(np.asarray(lio.load_cost_volume(p, 5, 6, 7, mmap=mm)), v)
    assert lio.load_cost_volume(str(tmp_path / "im1.acrt"), 5, 6, 7) is None    # absent right volume (LES/main.cpp:361)
    with pytest.raises(ValueError):
        lio.load_cost_volume(p, 6, 6, 7)


def test_evaluator_counts():
    gt = np.array([[0, 1, 2, np.inf], [4, 5, 6, 7]], np.float32)
    nonocc = np.array([[1, 1, 0, 0], [1, 1, 1, 0]], bool)
    d = np.array([[9, 1.4, 2.6, 3], [4.5, 5.51, np.nan, 7]], np.float32)
    ev = lio.Evaluator(gt, nonocc, 0.5)
    # valid: 6 pixels (gt 1,2,4,5,6,7); good&valid: 1.4, 4.5 (<= 0.5 inclusive), 7 -> 3
    all_, non_ = ev.evaluate(d)
    assert all_ == pytest.approx(100 * (1 - 3 / 6))
    # nonocc: 5 pixels incl. the gt==0 one (the reference does not intersect with valid): good: 1.4, 4.5 -> 2
    assert non_ == pytest.approx(100 * (1 - 2 / 5))


def test_calib_parser(tmp_path):
    p = tmp_path / "calib.txt"
    p.write_text("cam0=[1 0 2; 0 1 3; 0 0 1]\ncam1=[1 0 4; 0 1 3; 0 0 1]\ndoffs=2.5\nbaseline=176.2\nwidth=1436\nheight=992\nndisp=145\n"
                 "isint=0\nvmin=36\nvmax=218\ndyavg=0.408\ndymax=1.923\n")
    c = lio.read_calib(str(p))
    assert c["ndisp"] == 145 and c["width"] == 1436 and c["doffs"] == 2.5 and c["cam1"][0, 2] == 4


@pytest.mark.skipif(not os.path.isdir(CONES), reason="reference data not present")
def test_load_data_cones():
    pytest.importorskip("PIL")
    d = lio.load_data(CONES)
    assert d["ndisp"] == 59 and d["gt_prec"] == 0.25 and d["imL"].shape == (375, 450, 3)
    assert lio.load_data(CONES, ndisp=64)["ndisp"] == 64                        # the command line wins (LES/main.cpp:275)
    gt = d["dispGT"]
    assert np.isinf(gt).any() and np.nanmax(gt[np.isfinite(gt)]) <= 64
    ev = lio.Evaluator(gt, d["nonocc"], 0.5)
    assert ev.evaluate(np.where(np.isfinite(gt), gt, 0))[0] == 0.0


def test_read_pfm_accepts_exponent_scales(tmp_path):
    """The reference parses the scale with fscanf %lf (LES/Utilities.hpp:20-82): 1e-05, -3.9e-03, +1.E+0 are valid headers."""
    from localexpstereo_amd import io as lio
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    for scale, dt in (("-1e-05", "<f4"), ("-3.9e-03", "<f4"), ("1.5E+2", ">f4"), ("-.5", "<f4")):
        p = tmp_path / f"s{scale}.pfm"
        with open(p, "wb") as f:
            f.write(f"Pf\n4 3\n{scale}\n".encode("ascii"))
            f.write(np.ascontiguousarray(a[::-1]).astype(dt).tobytes())
        assert np.array_equal(lio.read_pfm(str(p)), a)
