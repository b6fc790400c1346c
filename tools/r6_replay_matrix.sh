#!/bin/bash
# Round 6 A/B of the tiled max-flow's hand-over on dumped lock-steps (tools/_samples/r6/*.npz; LES_DUMP_TILED, pm.py):
# device-only (round 5 behaviour), the product's policy, and -- handing over as soon as <= 8 cells are open -- both host finishers and 7 row bands
# against one.  Run on the GPU box.
S=${1:-tools/_samples/r6}
O=gpurun_out/r6; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 120 python tools/tiled_cut_replay.py $S/*.npz --reps 3 --threads 16 2>&1 | grep -v "strip kernel\|amdgpu.ids"; }
{
run LES_HIP_MAXFLOW_HANDOVER=0
run LES_HIP_MAXFLOW_HANDOVER=1
run LES_HIP_MAXFLOW_HANDOVER=1 LES_HIP_MAXFLOW_HANDOVER_NO_STALL_RULE=1 LES_HIP_MAXFLOW_HANDOVER_NODES=400000
run LES_HIP_MAXFLOW_HANDOVER=1 LES_HIP_MAXFLOW_HANDOVER_NO_STALL_RULE=1 LES_HIP_MAXFLOW_HANDOVER_NODES=400000 LES_HIP_MAXFLOW_HANDOVER_SOLVER=0
run LES_HIP_MAXFLOW_HANDOVER=1 LES_HIP_MAXFLOW_HANDOVER_NO_STALL_RULE=1 LES_HIP_MAXFLOW_HANDOVER_NODES=400000 LES_GC_RESIDUAL_BAND_NODES=20000
} > $O/replay_matrix.log 2>&1
