#!/usr/bin/env python
"""Static census of les_march_kernel<10, 256, 1, 7> (the headline instantiation) from a `hipcc -S` listing, written to
profiles/<tag>_isa_census.json for the kernel sources' hash: registers, LDS, and the share of VALU instructions outside the dual-issue
class (tools/ubench/valu_rates.hip: fp32 add / mul / fma, 32-bit integer add / sub, logic, shifts and moves issue at ~2.9 cycles per
wave-instruction with three waves per SIMD, everything else at ~4.2).  tools/collect_profiles.sh folds it into profiles/traffic.json
(`co_bounds`), which bench.py quotes.  The census is STATIC (every specialisation of the three roles, weighted by code size, not by
execution counts): the loops dominate the listing, so it is a fair estimate of the dynamic mix, not a measurement of it.

  python tools/isa_census.py [tag]          (build container: needs hipcc, no GPU)
"""
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench                      # noqa: E402  (kernel_source_hash)
# the dual-issue class as measured (profiles/round4_valu_rates_w3.log: 2.8 - 3.1 cycles; every other VALU instruction 4.0 - 4.5)
FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac)_f32(_e32|_e64)?$|^v_(add|sub|subrev)_u32(_e32|_e64)?$|^v_(and|or|xor|lshlrev|lshrrev)_b32(_e32|_e64)?$|^v_ashrrev_i32(_e32|_e64)?$|^v_mov_b32(_e32|_e64)?$")
from localexpstereo_amd import build  # noqa: E402

KERNEL = "les_march_kernelILi10ELi256ELi1ELi7E"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "round4"
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "les.s")
        subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DLES_MARCH_FEW_RADII", "--cuda-device-only", "-S",
                               os.path.join(build.CSRC, "les_hip.hip"), "-o", out], cwd=build.CSRC, stderr=subprocess.DEVNULL)
        text = open(out).read()
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3les16" + KERNEL) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ops = Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.split()[0].endswith(":"):
            continue
        ops[t.split()[0]] += 1
    valu = {o: n for o, n in ops.items() if o.startswith("v_")}
    nv = sum(valu.values())
    fast = sum(n for o, n in valu.items() if FAST.match(o) and "dpp" not in o and "sdwa" not in o)
    m = re.search(r"\.name:\s+_ZN3les16" + KERNEL, text)
    meta = text[max(0, m.start() - 1500):m.start() + 1200]
    g = lambda k: int((re.findall(k + r":\s+(\d+)", meta) or ["0"])[-1])
    rec = {"kernel": "les_march_kernel<10, 256, 1, 7>", "kernel_source_sha1": bench.kernel_source_hash(),
           "instructions": sum(ops.values()), "valu": nv, "salu": sum(n for o, n in ops.items() if o.startswith("s_")),
           "lds": sum(n for o, n in ops.items() if o.startswith("ds_")), "vmem": sum(n for o, n in ops.items() if o.startswith(("buffer_", "global_"))),
           "valu_share_outside_dual_issue_class": round(1.0 - fast / nv, 4),
           "vgprs": g(r"\.vgpr_count"), "sgprs": g(r"\.sgpr_count"), "lds_bytes_per_wg": g(r"\.group_segment_fixed_size"), "vgpr_spills": g(r"\.vgpr_spill_count"),
           "top_valu": sorted(valu.items(), key=lambda kv: -kv[1])[:16]}
    path = os.path.join(ROOT, "profiles", f"{tag}_isa_census.json")
    json.dump(rec, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "top_valu"}))


if __name__ == "__main__":
    main()
