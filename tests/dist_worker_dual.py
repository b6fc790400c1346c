"""Worker for the multi-process two-view test: the whole FastGCStereo::run mirror (localexpstereo_amd/stereo.py) on `WORLD_SIZE`
ranks -- view split x cell split, per-set tile exchange through the C ABI's pack / unpack kernels, one broadcast per view before the
post-processing -- saving rank 0's result.  Backend gloo + the CPU simulator build of the C ABI in the build container; with lib == "hip" on a
multi-GPU node the same code runs with backend nccl (RCCL over xGMI), one GPU per rank, and every cut on the GPUs."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from localexpstereo_amd import api, stereo, synth  # noqa: E402


def main():
    out, lib, H, W, D, pm_iters, gc_iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    use_gpu = lib == "hip"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if use_gpu:
        import torch
        torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl" if use_gpu else "gloo")
    imL, imR = synth.make_guide(H, W, 1234), synth.make_guide(H, W, 1235)
    e = api.HipCostVolumeEnergy(imL, imR, synth.make_volume(D, H, W, 42), synth.make_volume(D, H, W, 43), lib=None if use_gpu else lib, device=local if use_gpu else 0)
    st = stereo.FastGCStereo(e, imL, imR, dict(lambda_=0.05), device=f"cuda:{local}" if use_gpu else "cpu", rank=rank, world=world, seed=3)
    if not use_gpu:
        st.device_cuts = False             # (the simulator would spend minutes in the device cuts; on GPUs every rank cuts its cells on its own GPU)
    st.addLayer(10, [(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 2)])
    st.addLayer(30, [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)])
    lab, raw = st.run(gc_iters, (0, 1), pm_iters)
    if rank == 0:
        np.savez(out, lab=lab, raw=raw)
    if world > 1:
        # every rank must hold the same post-processed labelling
        import torch
        t = torch.from_numpy(lab.copy())
        ref = t.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(t, ref), f"rank {rank}: final labelling differs from rank 0"
        dist.barrier()
        st.close()                      # the per-view process groups of the run
        dist.destroy_process_group()
    e.close()


if __name__ == "__main__":
    main()
