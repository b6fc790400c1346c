# kernel time with only some roles of les_march_kernel compiled in (builds: -DLES_MARCH_ROLE_MASK=m -> csrc/libles_role<m>.so;
# bit 0 = role A, 1 = C, 2 = D).  The output is meaningless, the time shows what each role costs alone / in pairs.
cd "$GRAFT_REPO_ROOT"
for m in 1 2 4 3 5 6; do
  echo -n "role mask $m: "
  LES_HIP_LIB=localexpstereo_amd/csrc/libles_role$m.so python bench.py --steps 20 --warmup 3 --cpu-planes 0 --sub-steps 0 --e2e 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done
echo -n "all roles: "; python bench.py --steps 20 --warmup 3 --cpu-planes 0 --sub-steps 0 --e2e 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
