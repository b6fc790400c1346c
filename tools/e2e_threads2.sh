for t in 0 4 6 8 12; do python tools/e2e_bench.py --dual 1 --host-threads $t > gpurun_out/e2e_d3_t$t.json 2>/dev/null; done
python tools/e2e_bench.py --dual 0 --host-threads 8 > gpurun_out/e2e_s3_t8.json 2>/dev/null
cat /sys/fs/cgroup/cpu.max > gpurun_out/cpu_max.txt; nproc >> gpurun_out/cpu_max.txt
