cd "$GRAFT_REPO_ROOT"
for L in liblocalexp_hip libles_order1 libles_order2 libles_order3 libles_order4 libles_order5; do
  echo -n "$L: "
  LES_HIP_LIB=localexpstereo_amd/csrc/$L.so python bench.py --steps 50 --warmup 3 --cpu-planes 0 --sub-steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['h2']['ms_per_step'], d['h3']['ms_per_step'])"
done
