cd "$GRAFT_REPO_ROOT"
for o in 0 1 2 3 4 5; do
  L=localexpstereo_amd/csrc/libles_order$o.so; [ $o = 0 ] && L=localexpstereo_amd/csrc/libles_phase_timing.so
  echo "== role order $o"
  PHASE_LIB=$PWD/$L python tools/phase_probe.py 2>&1 | tail -3
  LES_HIP_LIB=$L python bench.py --steps 20 --warmup 3 --cpu-planes 0 --sub-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms', d['ms_per_step'])"
done
