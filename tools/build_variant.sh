#!/bin/bash
# Builds an A/B variant of the HIP library with extra compiler flags (-DLES_MARCH_LAB: the experiment switches of csrc/les_march_lab.h), next to the product:
#   bash tools/build_variant.sh role1 -DLES_MARCH_ROLE_MASK=1        -> localexpstereo_amd/csrc/libles_role1.so
#   bash tools/build_variant.sh order3 -DLES_MARCH_ROLE_ORDER=3      -> .../libles_order3.so
# The variants travel to the GPU box with the snapshot; tools/ab_time.sh / role_time.sh / order_probe.sh time them
# (LES_HIP_LIB=... selects the library for api.py).  They are git-ignored.
set -e
cd "$(dirname "$0")/../localexpstereo_amd/csrc"
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DLES_MARCH_LAB "$@" les_hip.hip -o libles_$name.so -ldl
echo "built localexpstereo_amd/csrc/libles_$name.so"
