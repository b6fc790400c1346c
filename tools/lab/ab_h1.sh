#!/bin/bash
# H1 time only of several builds of the same ABI:  bash tools/ab_h1.sh libA libB ...   (names under csrc/, without .so)
cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  echo -n "$L: "
  LES_HIP_LIB=localexpstereo_amd/csrc/$L.so python bench.py --steps 20 --warmup 3 --cpu-planes 0 --sub-steps 0 --e2e 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done
