#!/bin/bash
# sensitivity of the host max-flow phase to the OpenMP team size / wait policy (tools/e2e_bench.py, 2 graph-cut iterations)
for cfg in "24 passive" "48 passive" "64 passive" "96 passive" "128 passive"; do
  set -- $cfg
  if [ "$2" = "-" ]; then unset OMP_WAIT_POLICY; else export OMP_WAIT_POLICY=$2; fi
  python tools/e2e_bench.py --iterations 2 --host-threads $1 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); g = r['gc_seconds']
print('threads $1 wait $2: optimiser %.2f s  device %.2f  host %.2f (L0 %.2f L1 %.2f L2 %.2f)' % (r['seconds_optimiser'], g['device'], g['host_cuts'], g['host_cuts_layer0'], g['host_cuts_layer1'], g['host_cuts_layer2']))"
done
