cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 1 2 4; do
  LES_HIP_LIB=localexpstereo_amd/csrc/libles_role$m.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d gpurun_out/rolepmc$m -- python bench.py --steps 2 --warmup 1 --cpu-planes 0 --sub-steps 0 > /dev/null 2>&1
  echo "role mask $m"; python tools/prof_summary.py gpurun_out/rolepmc$m les_march_kernel --md | grep "^| SQ"
  rm -rf gpurun_out/rolepmc$m
done
