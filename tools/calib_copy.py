#!/usr/bin/env python
"""Known-byte-count run for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this library's access width (one dword per
lane): copies N floats REPS times with les_calib_copy_kernel.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localexpstereo_amd import api

N = int(sys.argv[1]) if len(sys.argv) > 1 else 384_000_000          # 1.536 GB, the size of the bench volume
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = api.load()
src = torch.rand(N, device="cuda", dtype=torch.float32)
dst = torch.empty_like(src)
for _ in range(REPS):
    assert L.les_hip_calib_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(N), 0, None) == 0
torch.cuda.synchronize()
assert torch.equal(src[:1000], dst[:1000])
print("bytes read per launch", N * 4, "bytes written per launch", N * 4)
