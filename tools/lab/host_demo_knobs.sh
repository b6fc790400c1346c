#!/bin/bash
# the C++ driver at full size under the host-cut knobs (its scene's moves flip most of a coarse cell: the hard case of the host solver)
D=localexpstereo_amd/host/les_host_demo
for cfg in "X=1" "LES_GC_PREPUSH=0 LES_GC_BK_OPS_PER_NODE=12 LES_GC_BK_BAND_OPS_PER_NODE=12" "LES_GC_BK_OPS_PER_NODE=12 LES_GC_BK_BAND_OPS_PER_NODE=12" "LES_GC_BK_OPS_PER_NODE=1" "LES_GC_PREPUSH=0"; do
  echo "== $cfg"; env $cfg $D full 1436 992 256 5 2 | grep "^full"
done
