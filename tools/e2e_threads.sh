#!/bin/bash
# sensitivity of the host max-flow phase to the OpenMP team size / placement (tools/e2e_bench.py, 2 graph-cut iterations)
for cfg in "16 -" "32 -" "64 -" "128 -" "32 spread" "64 spread" "64 close" "128 spread"; do
  set -- $cfg
  if [ "$2" = "-" ]; then unset OMP_PROC_BIND OMP_PLACES; else export OMP_PROC_BIND=$2 OMP_PLACES=cores; fi
  python tools/e2e_bench.py --iterations 2 --host-threads $1 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); g = r['gc_seconds']
print('threads $1 bind $2: optimiser %.2f s  device %.2f  host %.2f (L0 %.2f L1 %.2f L2 %.2f)' % (r['seconds_optimiser'], g['device'], g['host_cuts'], g['host_cuts_layer0'], g['host_cuts_layer1'], g['host_cuts_layer2']))"
done
