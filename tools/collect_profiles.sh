#!/bin/bash
# Collects the round's rocprofv3 evidence for the headline bench (run on the GPU box through gpurun):
#   kernel-trace stats of `python bench.py`, FETCH_SIZE / WRITE_SIZE in separate --pmc passes (no trace domains mixed
#   in), the same two counters on a known-byte-count dword copy for calibration, and the SQ counters.
# Summaries land in gpurun_out/prof/ ; traffic.json (HBM bytes per launch + the sha1 of the kernel sources it was taken on)
# is what bench.py quotes as roofline.traffic.  Usage: bash tools/collect_profiles.sh [tag]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-round6}
O=gpurun_out/prof; mkdir -p $O
B="python bench.py --steps 5 --warmup 1 --cpu-planes 0 --sub-steps 0 --e2e 0"
rocprofv3 --kernel-trace --stats -d $O/stats -- python bench.py --steps 100 --warmup 5 --cpu-planes 0 --sub-steps 0 --e2e 0 > $O/stats.log 2>&1     # enough launches that the cold first ones do not weigh on the average
python tools/prof_summary.py $O/stats --md > $O/${TAG}_kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $O/pmc_$c -- $B > $O/pmc_$c.log 2>&1
  python tools/prof_summary.py $O/pmc_$c les_march_kernel --md > $O/pmc_$c.md
  rocprofv3 --pmc $c -d $O/cal_$c -- python tools/calib_copy.py > $O/cal_$c.log 2>&1
  python tools/prof_summary.py $O/cal_$c les_calib --md > $O/cal_$c.md
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
python tools/prof_summary.py $O/pmc_sq les_march_kernel --md > $O/pmc_sq.md
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY -d $O/pmc_lds -- $B > $O/pmc_lds.log 2>&1
python tools/prof_summary.py $O/pmc_lds les_march_kernel --md > $O/pmc_lds.md
# the optimiser's geometry (H3: 240 cell-batched launches per step) and the end-to-end run (MidV3 loop, one view): kernel-trace stats
rocprofv3 --kernel-trace --stats -d $O/h3 -- python bench.py --workload h3 --steps 20 --warmup 2 --cpu-planes 0 --sub-steps 0 --e2e 0 > $O/h3.log 2>&1
python tools/prof_summary.py $O/h3 --md > $O/${TAG}_h3_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $O/e2e -- python tools/e2e_bench.py > $O/e2e.log 2>&1
python tools/prof_summary.py $O/e2e --md > $O/${TAG}_e2e_kernel_stats.md
# round 5: the hard scene (every cut on the GPU: the coarse layers by les_maxflow_tiled_kernel; __amd_rocclr_copyBuffer calls = what still crosses PCIe),
# and the launches of the tiled solver on dumped lock-steps (per-launch durations in sequence: RELABEL0 / RELABEL / DISCHARGE phases)
rocprofv3 --kernel-trace --stats -d $O/e2e_ts -- python tools/e2e_bench.py --scene three_surfaces > $O/e2e_ts.log 2>&1
python tools/prof_summary.py $O/e2e_ts --md > $O/${TAG}_e2e_three_surfaces_kernel_stats.md
if ls tools/_samples/r6/*.npz > /dev/null 2>&1; then
  rocprofv3 --kernel-trace --output-format csv -d $O/mf -- python tools/tiled_cut_replay.py tools/_samples/r6/*.npz --reps 1 > $O/${TAG}_tiled_replay.log 2>&1
  python tools/trace_summary.py $O/mf tiled_kernel --seq 160 > $O/${TAG}_tiled_maxflow_trace.md
fi
rm -rf $O/h3 $O/e2e $O/e2e_ts $O/mf
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/cal_FETCH_SIZE $O/cal_WRITE_SIZE $O/pmc_sq $O/pmc_lds
python - "$O" "$TAG" <<'PY'
import json, re, sys, os
sys.path.insert(0, os.getcwd())
O, tag = sys.argv[1], sys.argv[2]
def counter(path, name):
    for l in open(path):
        m = re.match(r"\|\s*%s\s*\|\s*([0-9.eE+-]+)\s*\|" % name, l)
        if m: return float(m.group(1))
    return None
fetch, write = counter(f"{O}/pmc_FETCH_SIZE.md", "FETCH_SIZE"), counter(f"{O}/pmc_WRITE_SIZE.md", "WRITE_SIZE")
cf, cw = counter(f"{O}/cal_FETCH_SIZE.md", "FETCH_SIZE"), counter(f"{O}/cal_WRITE_SIZE.md", "WRITE_SIZE")
known = 384_000_000 * 4.0                           # tools/calib_copy.py: bytes read = bytes written per launch
# counters are in KiB; the calibration factor (known bytes / counted bytes of the dword copy) corrects the gfx950 unit quirks
kf, kw = known / (cf * 1024.0), known / (cw * 1024.0)
bytes_per_launch = fetch * 1024.0 * kf + write * 1024.0 * kw
import bench
# co-bounds of the kernel: instruction issue, the LDS pipe, occupancy.  Issue costs per wave-instruction and SIMD at three
# waves per SIMD from tools/ubench/valu_rates.hip (profiles/round4_valu_rates_w3.log): 2.9 cycles for the dual-issue class (fp32 add / mul /
# fma, 32-bit integer add / sub, logic, shifts, moves), 4.2 for every other VALU instruction; the class shares come from the static census of
# the kernel's listing (profiles/<tag>_isa_census.json, written by tools/isa_census.py in the build container for the same source hash).
sq = lambda n: counter(f"{O}/pmc_sq.md", n)
lds = lambda n: counter(f"{O}/pmc_lds.md", n)
kus = None
for l in open(f"{O}/{tag}_kernel_stats.md"):
    m = re.match(r"\|\s*les::les_march_kernel[^|]*\|\s*\d+\s*\|\s*[0-9.]+\s*\|\s*([0-9.]+)\s*\|", l)
    if m: kus = float(m.group(1)); break
census = {}
try:
    c = json.load(open(f"profiles/{tag}_isa_census.json"))
    if c.get("kernel_source_sha1") == bench.kernel_source_hash(): census = c
except Exception:
    pass
nsimd, clk = 1024, 2.4e9
valu = sq("SQ_INSTS_VALU")
other = census.get("valu_share_outside_dual_issue_class")
co = {"valu_insts_per_launch": valu, "salu_insts_per_launch": sq("SQ_INSTS_SALU"), "lds_insts_per_launch": sq("SQ_INSTS_LDS"),
      "vmem_insts_per_launch": (sq("SQ_INSTS_VMEM_RD") or 0) + (sq("SQ_INSTS_VMEM_WR") or 0),
      "slow_pipe_share": other,
      "valu_issue_floor_ms": (round(valu / nsimd * ((1 - other) * 2.9 + other * 4.2) / clk * 1e3, 4) if valu and other is not None else None),
      "lds_busy_frac": (round(lds("SQ_LDS_IDX_ACTIVE") / (256 * kus * 1e-6 * clk), 4) if kus and lds("SQ_LDS_IDX_ACTIVE") else None),
      "lds_bank_conflict_share": (round(lds("SQ_LDS_BANK_CONFLICT") / lds("SQ_LDS_IDX_ACTIVE"), 4) if lds("SQ_LDS_IDX_ACTIVE") else None),
      "waves_per_simd": 3, "lds_bytes_per_wg": census.get("lds_bytes_per_wg"), "vgprs": census.get("vgprs"),
      "method": "SQ_* from separate rocprofv3 --pmc passes of this command; issue floor = VALU instructions / 1024 SIMDs x (2.9 cycles for the dual-issue class, 4.2 otherwise; "
                "class shares from the static listing) at 2.4 GHz with every wait hidden; lds_busy_frac = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles)"}
rec = {"h1": {"shape": [1000, 1500, 256], "kernel_source_sha1": bench.kernel_source_hash(), "bytes_per_launch": bytes_per_launch,
              "fetch_bytes": fetch * 1024.0 * kf, "write_bytes": write * 1024.0 * kw,
              "source": f"profiles/{tag}_pmc.md: FETCH_SIZE {fetch:.4g} KiB x {kf:.3f} + WRITE_SIZE {write:.4g} KiB x {kw:.3f} per launch "
                        f"(factors = known bytes / counted bytes of a 1.536 GB dword copy in the same profile run), separate --pmc passes",
              "co_bounds": co}}
json.dump(rec, open(f"{O}/traffic.json", "w"), indent=1)
with open(f"{O}/{tag}_pmc.md", "w") as f:
    f.write(f"# {tag}: PMC counters of les_march_kernel on `python bench.py` (H1, 1500x1000x256), per launch\n\n")
    for name in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "cal_FETCH_SIZE", "cal_WRITE_SIZE", "pmc_sq", "pmc_lds"):
        f.write(f"## {name}\n" + open(f"{O}/{name}.md").read() + "\n")
    f.write(f"\nHBM traffic per launch: {bytes_per_launch / 1e9:.3f} GB (fetch {fetch * 1024 * kf / 1e9:.3f} + write {write * 1024 * kw / 1e9:.3f}); algorithmic 3.144 GB\n")
print(open(f"{O}/traffic.json").read())
PY
# the other two workloads' counters (VERDICT r5 #3b): H2 (256 slanted planes, tiled-copy taps) and H3 (the optimiser's cell batches) -> traffic.json entries
bash tools/pmc_workload.sh h2 $TAG > /dev/null 2>&1
bash tools/pmc_workload.sh h3 $TAG > /dev/null 2>&1
python tools/traffic_merge.py $TAG $O
head -12 $O/${TAG}_kernel_stats.md
