"""Static instruction census of one kernel in a hipcc -S listing: totals per class and issue-cycle estimate per basic block.
   python tools/isa_count.py /tmp/les_hip.s les_march_kernel [--blocks]
Cycle weights from tools/ubench/valu_rates.hip on MI355X: 2 cycles for fp32 / int32 add, mul, fma, logic and moves, 4 for everything else on
the VALU (fp64, conversions, DPP, 64-bit integer MAD, bit-field ops, min/max)."""
import re
import sys
from collections import Counter

FAST = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad)_(f32|u32|i32|co_u32)|^v_(and|or|xor|not|lshlrev|lshrrev|ashrrev)_b32|^v_ashrrev_i32|^v_mov_b32|^v_cndmask_b32|^v_addc|^v_subb|^v_add3|^v_lshl_add_u32|^v_lshl_or|^v_and_or|^v_or3|^v_cmp|^v_accvgpr")


def cyc(op):
    if not op.startswith("v_"):
        return 0
    if op.endswith("_dpp") or "dpp" in op:
        return 4
    return 2 if FAST.match(op) else 4


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and name in l and l.rstrip().endswith((":", ")")) or (l.startswith("_ZN") and name in l and ":" in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start + 1:end]
    blocks, cur, label = [], [], "entry"
    for l in body:
        t = l.strip()
        if not t or t.startswith((";", "//")):
            continue
        if t.endswith(":") or (t.split()[0].endswith(":")):
            if cur:
                blocks.append((label, cur))
            label, cur = t.split(":")[0], []
            continue
        if t.startswith("."):
            continue
        cur.append(t.split()[0])
    if cur:
        blocks.append((label, cur))
    tot = Counter()
    for _, ops in blocks:
        tot.update(ops)
    cats = Counter()
    for op, n in tot.items():
        k = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else op.split("_")[0]
        cats[k] += n
    print("instructions:", sum(tot.values()), dict(cats), "VALU issue cycles:", sum(cyc(o) * n for o, n in tot.items()))
    print("top:", tot.most_common(40))
    if "--blocks" in sys.argv:
        for label, ops in blocks:
            if len(ops) < 40:
                continue
            c = Counter(ops)
            v = sum(n for o, n in c.items() if o.startswith("v_"))
            print(f"{label:24s} n={len(ops):5d} valu={v:5d} valu_cyc={sum(cyc(o) * n for o, n in c.items()):6d} lds={sum(n for o, n in c.items() if o.startswith('ds_')):4d} "
                  f"vmem={sum(n for o, n in c.items() if o.startswith(('global_', 'buffer_', 'scratch_'))):4d} scratch={sum(n for o, n in c.items() if o.startswith('scratch_')):3d} "
                  f"salu={sum(n for o, n in c.items() if o.startswith('s_')):4d} barriers={c.get('s_barrier', 0)} waitcnt={c.get('s_waitcnt', 0)} nop={c.get('s_nop', 0)}")


if __name__ == "__main__":
    main()
