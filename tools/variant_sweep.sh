#!/bin/bash
# A/B sweep of the radius-10 strip-kernel variants compiled into liblocalexp_hip.so (LES_HIP_VARIANT=n).
for v in "$@"; do
  echo -n "variant $v: "
  LES_HIP_VARIANT=$v python bench.py --steps 5 --warmup 2 --cpu-planes 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['roofline']['kernel_ms'], 'ms  strip', r['config']['strip_width'], ' wgs', r['config']['workgroups_per_launch'])
"
done
