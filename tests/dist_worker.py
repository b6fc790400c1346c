"""Worker for the multi-process tests: runs the sharded PatchMatch iterations (localexpstereo_amd/pm.py)
on `WORLD_SIZE` ranks and saves rank 0's result.  Backend gloo + the CPU simulator build of the C ABI in the
build container; on a multi-GPU node the same code runs with backend nccl (RCCL) and the HIP build."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from localexpstereo_amd import api, pm, synth  # noqa: E402


def main():
    out, lib, H, W, D, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    gc_iters = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    use_gpu = lib == "hip"
    if world > 1:
        dist.init_process_group("nccl" if use_gpu else "gloo")
    if use_gpu:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    e = api.HipCostVolumeEnergy(synth.make_guide(H, W, 1234), None, synth.make_volume(D, H, W, 42), None,
                                lib=None if use_gpu else lib, device=int(os.environ.get("LOCAL_RANK", "0")) if use_gpu else 0)
    table = [[(api.PROPOSE_EXPANSION, 1), (api.PROPOSE_RANSAC, 1), (api.PROPOSE_RANDOM, 3)], [(api.PROPOSE_EXPANSION, 2), (api.PROPOSE_RANSAC, 1)]]
    r = pm.PMRunner(e, (10, 30), table, seed=5, rank=rank, world=world, device="cuda" if use_gpu else "cpu")
    g = None
    if gc_iters:
        from localexpstereo_amd import gc as lgc
        g = lgc.GraphCut(synth.make_guide(H, W, 1234), None, lambda_=0.05)
    labels, cur = r.run(iters, gc_iters, g)
    if rank == 0:
        np.savez(out, labels=labels.cpu().numpy(), cur=cur.cpu().numpy(), bytes_exchanged=r.bytes_exchanged,
                 energy=(g.energy(0) if g else 0.0), host_labels=(g.labels[0] if g else 0))
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
