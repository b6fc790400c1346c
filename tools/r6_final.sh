#!/bin/bash
# Round 6, final tree: the evidence that changes with the finest layer's new cut kernel (GPU box, through gpurun):
#   kernel-trace stats of one-view runs of both scenes, the cell-kernel A/B, the hand-over A/B on whole runs, the bench line.
# Usage: bash tools/r6_final.sh [outdir]      (the march kernel's own passes: tools/collect_profiles.sh, unchanged sources)
O=${1:-gpurun_out/r6_final}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/e2e -- python tools/e2e_bench.py > $O/e2e.log 2>&1
python tools/prof_summary.py $O/e2e --md > $O/round6_e2e_kernel_stats.md
timeout 300 rocprofv3 --kernel-trace --stats -d $O/e2e_ts -- python tools/e2e_bench.py --scene three_surfaces > $O/e2e_ts.log 2>&1
python tools/prof_summary.py $O/e2e_ts --md > $O/round6_e2e_three_surfaces_kernel_stats.md
rm -rf $O/e2e $O/e2e_ts
bash tools/lab/ab_cell_kernel.sh $O/ab_cell > $O/round6_cell_kernel_ab.log 2>&1
bash tools/r6_e2e_ab.sh $O/ab > $O/round6_e2e_ab.log 2>&1
python bench.py > $O/round6_bench.json 2> $O/round6_bench.err
tail -c 600 $O/round6_bench.json
