#!/usr/bin/env python
"""Folds the HBM counters of the H2 / H3 workloads (tools/pmc_workload.sh -> gpurun_out/prof/<tag>_<wl>_pmc.md: FETCH_SIZE / WRITE_SIZE of the march
kernel's launches in separate rocprofv3 --pmc passes, each calibrated with the known-size dword copy of the same run) into traffic.json next to the H1
entry tools/collect_profiles.sh wrote, so that bench.py's `h2` / `h3` sub-records carry `traffic` and the wasted-traffic ratio is current.

  python tools/traffic_merge.py <tag> [dir = gpurun_out/prof]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KNOWN = 384_000_000 * 4.0          # tools/calib_copy.py: bytes read = bytes written per launch


def section(text, name):
    """-> (sum over the workload's march launches of the counter [KiB], number of those launches, calibration average [KiB])"""
    start = text.index("## %s (KiB)" % name)
    ends = [text.find(mark, start + 5) for mark in ("## FETCH_SIZE (KiB)", "## WRITE_SIZE (KiB)", "## SQ instruction counters", "## L2 (TCC) requests")]
    body = text[start: min([e for e in ends if e > start] + [len(text)])]
    work, cal = body.split("calibration copy", 1)
    rows = lambda t: [(float(a), int(n)) for a, n in re.findall(r"\|\s*%s\s*\|\s*([0-9.eE+-]+)\s*\|\s*(\d+)\s*\|" % name, t)]
    w, c = rows(work), rows(cal)
    return sum(a * n for a, n in w), sum(n for _, n in w), c[0][0]


def main():
    tag = sys.argv[1]
    d = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof")
    tf = os.path.join(d, "traffic.json")
    rec = json.load(open(tf)) if os.path.exists(tf) else {}
    for wl in ("h2", "h3"):
        f = os.path.join(d, f"{tag}_{wl}_pmc.md")
        if not os.path.exists(f):
            continue
        text = open(f).read()
        fs, fn, fc = section(text, "FETCH_SIZE")
        ws, wn, wc = section(text, "WRITE_SIZE")
        kf, kw = KNOWN / (fc * 1024.0), KNOWN / (wc * 1024.0)
        line = re.search(r"\{'ms_per_step'.*", text)
        steps = 4                                       # pmc_workload.sh: --steps 3 --warmup 1
        launches_per_step = fn / steps
        per_step = (fs * 1024.0 * kf + ws * 1024.0 * kw) / steps
        alg = None
        m = re.search(r"'algorithmic_bytes_per_launch': ([0-9.eE+]+)", text)
        if m:
            alg = float(m.group(1)) * (1.0 if wl == "h2" else 1.0)
        rec[wl] = {"kernel_source_sha1": bench.kernel_source_hash(), "shape": [1000, 1500, 256], "launches_per_step": launches_per_step,
                   "bytes_per_step": per_step, "fetch_bytes_per_step": fs * 1024.0 * kf / steps, "write_bytes_per_step": ws * 1024.0 * kw / steps,
                   "source": f"profiles/{tag}_{wl}_pmc.md: FETCH_SIZE / WRITE_SIZE summed over the march kernel's {fn} launches of {steps} steps, factors {kf:.3f} / {kw:.3f} "
                             f"from the 1.536 GB dword copy of the same profile run, separate --pmc passes",
                   "bench_line_of_the_run": line.group(0) if line else None}
    json.dump(rec, open(tf, "w"), indent=1)
    print(json.dumps({k: {a: b for a, b in v.items() if a != "co_bounds"} for k, v in rec.items()}, indent=1))


if __name__ == "__main__":
    main()
