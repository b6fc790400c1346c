# What a launch of the tiled max-flow costs besides its inner iterations: one dumped lock-step replayed with K = 1 / 2 / 8 inner iterations per launch
# under rocprofv3 --kernel-trace; per-launch durations of the first launches (RELABEL0, RELABEL..., DISCHARGE x S, ...).  Usage (GPU box): bash tools/lab/launch_cost.sh <sample.npz> <outdir>
F=${1:-tools/_samples/r6/tiled_view0_it1_layer1_110.npz}; O=${2:-gpurun_out/launch_cost}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for K in 1 2 8; do
  rm -rf $O/prof_$K
  LES_HIP_MAXFLOW_HANDOVER=0 LES_HIP_MAXFLOW_TILED_K=$K LES_HIP_MAXFLOW_TILED_K2=$K timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$K -- python tools/tiled_cut_replay.py $F --reps 1 > $O/replay_$K.log 2>&1
  tail -1 $O/replay_$K.log | cut -c1-220
  python tools/trace_summary.py $O/prof_$K maxflow_tiled_kernel --seq 100
  rm -rf $O/prof_$K
done
