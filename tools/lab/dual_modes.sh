#!/bin/bash
# two-view end-to-end run under the view schedules of stereo.py (LES_VIEWS) and OpenMP placement knobs
for v in concurrent concurrent-swapped joint serial; do
  mkdir -p gpurun_out/dm_$v
  LES_VIEWS=$v LES_GC_TRACE=gpurun_out/dm_$v/trace.txt python tools/e2e_bench.py --dual 1 > gpurun_out/dm_$v/e2e.json 2>/dev/null
done
mkdir -p gpurun_out/dm_bind
OMP_PROC_BIND=spread OMP_PLACES=cores LES_GC_TRACE=gpurun_out/dm_bind/trace.txt python tools/e2e_bench.py --dual 1 > gpurun_out/dm_bind/e2e.json 2>/dev/null
lscpu | grep -i "numa\|socket\|model name\|thread" > gpurun_out/lscpu.txt
