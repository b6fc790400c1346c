"""C++ host side above the C ABI (localexpstereo_amd/host/): LayerManager geometry against the oracle
(no GPU needed) and the les_host_demo end-to-end run (drop-in operator from OpenMP threads + device-
resident PatchMatch iterations) on the MI355X."""
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def demo():
    import os
    from localexpstereo_amd import build
    exe = os.path.join(build.HOST, "les_host_demo")
    if os.path.exists(exe) and os.path.exists(build.HIP_SO):
        return exe                     # prebuilt by __graft_entry__.build() (the GPU box only uses the prebuilt files)
    build.build_hip()
    exe = build.build_host()
    assert exe
    return exe


@pytest.mark.parametrize("W,H,units", [(450, 375, (5, 15, 25)), (1436, 992, (14, 43, 129)), (1500, 1000, (15, 45, 135)),
                                       (3000, 2000, (30, 90, 270)), (120, 96, (14, 7)), (97, 61, (10, 13))])
def test_host_layer_manager_matches_oracle(demo, oracle_mod, W, H, units):
    for u in units:
        out = subprocess.run([demo, "layers", str(W), str(H), "20", str(u)], capture_output=True, text=True, check=True).stdout.split("\n")
        wb, hb, nsets = (int(v) for v in out[0].split())
        L = oracle_mod.Layer(W, H, 20, u)
        assert (wb, hb, nsets) == (L.width_blocks, L.height_blocks, len(L.sets))
        n = wb * hb
        rects = np.array([[int(v) for v in line.split()] for line in out[1:1 + n]], np.int32)
        for k, ref in enumerate((L.unit, L.shared, L.filter)):
            got = rects[:, 4 * k:4 * k + 4]
            exp = np.stack([ref["x"], ref["y"], ref["w"], ref["h"]], 1)
            assert np.array_equal(got, exp), f"unit {u} rect kind {k}"
        for s, line in zip(L.sets, out[1 + n:1 + n + nsets]):
            assert [int(v) for v in line.split()] == s.tolist()


@pytest.mark.gpu
def test_host_demo_runs_on_gpu(demo):
    r = subprocess.run([demo, "run", "240", "160", "32", "2"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "les_host_demo: OK" in r.stdout
