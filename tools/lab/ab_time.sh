# A/B timing of two builds of the same ABI on the headline bench:  bash tools/ab_time.sh libA libB ...   (names under csrc/, without .so)
cd "$GRAFT_REPO_ROOT"
for L in "$@"; do
  echo -n "$L: "
  LES_HIP_LIB=localexpstereo_amd/csrc/$L.so python bench.py --steps 50 --warmup 3 --cpu-planes 0 --sub-steps 10 --e2e 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['h2']['ms_per_step'], d['h3']['ms_per_step'], d.get('h3_batched', {}).get('ms_per_step'))"
done
