#!/usr/bin/env python
"""Per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV: count, sum, percentiles, and the sequence (us).
  python tools/trace_summary.py gpurun_out/x/prof kernel_name_substring [--seq N]"""
import csv, glob, sys
import numpy as np
d = sys.argv[1]; name = sys.argv[2]; nseq = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "--seq" else 0
for f in sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
    if not rows:
        continue
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
    s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows])
    gap = (s[1:] - e[:-1]) / 1e3
    print(f"{f}: {len(rows)} launches of *{name}*, {dur.sum() / 1e3:.2f} ms; us mean {dur.mean():.1f} p10 {np.percentile(dur, 10):.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}; gaps p50 {np.median(gap):.1f} p90 {np.percentile(gap, 90):.1f}")
    if nseq:
        print(np.round(dur[:nseq]).astype(int).tolist())
