#!/usr/bin/env python
"""Summarise rocprofv3 output (rocpd sqlite .db or csv): per-kernel time stats and PMC counter totals.

usage: python tools/prof_summary.py <dir-or-db> [kernel-substring] [--md]
Counter values are summed over all hardware instances (SEs/XCDs) of a dispatch, then averaged over the
dispatches of the kernel.
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def short(name, n=90):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n] + "..."


def summarize_db(path, filt):
    con = sqlite3.connect(path)
    cur = con.cursor()
    out = []
    try:
        rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
        out.append(f"## {os.path.basename(path)}: kernel time (us)")
        out.append("| kernel | calls | total_us | avg_us | pct |")
        out.append("|---|---|---|---|---|")
        for name, calls, tot, avg, pct in rows[:10]:
            out.append(f"| {short(name)} | {calls} | {tot:.1f} | {avg:.1f} | {pct:.2f} |")
    except sqlite3.Error:
        pass
    try:
        rows = cur.execute("select kernel_name, dispatch_id, counter_name, value, vgpr_count, accum_vgpr_count, sgpr_count, "
                           "lds_block_size, scratch_size, workgroup_size, grid_size from counters_collection").fetchall()
    except sqlite3.Error:
        rows = []
    if rows:
        per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
        meta = {}
        for k, disp, cname, val, vg, av, sg, lds, scr, wg, grid in rows:
            if filt and filt not in k:
                continue
            per[k][cname][disp] += float(val)
            meta[k] = dict(vgpr=vg, agpr=av, sgpr=sg, lds=lds, scratch=scr, wg=wg, grid=grid)
        for k in per:
            out.append(f"## {os.path.basename(path)}: counters for {short(k)}")
            out.append(f"resources: {meta[k]}")
            out.append("| counter | avg per dispatch | dispatches |")
            out.append("|---|---|---|")
            for c in sorted(per[k]):
                vals = list(per[k][c].values())
                out.append(f"| {c} | {sum(vals) / len(vals):.4g} | {len(vals)} |")
    return out


def main():
    target = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    dbs = [target] if target.endswith(".db") else sorted(glob.glob(os.path.join(target, "**", "*.db"), recursive=True))
    for db in dbs:
        print("\n".join(summarize_db(db, filt)))


if __name__ == "__main__":
    main()
