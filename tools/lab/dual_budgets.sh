#!/bin/bash
# two-view end-to-end run under search budgets of the banded (coarsest-layer) cells
i=0
for cfg in "LES_GC_BK_BAND_OPS_PER_NODE=12" "LES_GC_BK_BAND_OPS_PER_NODE=40" "LES_GC_BK_BAND_OPS_PER_NODE=40 LES_GC_BK_OPS_PER_NODE=12" "LES_GC_BK_BAND_OPS_PER_NODE=200 LES_GC_BK_OPS_PER_NODE=3"; do
  i=$((i+1))
  env $cfg python tools/e2e_bench.py --dual 1 > gpurun_out/e2e_db_$i.json 2>/dev/null
  echo "$cfg" > gpurun_out/e2e_db_$i.cfg
done
