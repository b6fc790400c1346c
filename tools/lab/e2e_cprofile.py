#!/usr/bin/env python
"""cProfile of one end-to-end run (host-side view: which Python / C-ABI calls the wall-clock of tools/e2e_bench.py goes to)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import e2e_bench  # noqa: E402

e2e_bench.run(quiet=True)                 # warm: library loads, first-touch allocations
pr = cProfile.Profile()
pr.enable()
rec = e2e_bench.run(quiet=True)
pr.disable()
print("optimiser seconds", rec["seconds_optimiser"], rec["gc_seconds"])
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
