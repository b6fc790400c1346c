#!/bin/bash
# knobs of the host-cut lock-steps on this host (tools/cut_replay.py): solver path, team size, wait policy of the OpenMP pool
S="tools/_samples/*.npz"
for pre in 1 0; do
  LES_GC_PREPUSH=$pre python tools/cut_replay.py $S --threads 16
done
python tools/cut_replay.py $S --threads 8
python tools/cut_replay.py $S --threads 16 --gap-ms 0
OMP_WAIT_POLICY=active python tools/cut_replay.py $S --threads 16
OMP_WAIT_POLICY=active GOMP_SPINCOUNT=200000 python tools/cut_replay.py $S --threads 8
