#!/usr/bin/env python
"""The reference's demo.bat on the MI355X path: cones (MiddV2, two views), teddy (MiddV2) -- the bundled data, here from
tests/golden/ -- with the reference's default options (iterations 5, pmIterations 2, smooth_weight 1).  Prints the
Evaluator rows (LES/Evaluator.h: index, time, energy, data, smooth, bad-0.5 all, nonocc) and writes disp0.pfm.

  python tools/run_demo.py [--out DIR] [--iterations 5] [--pm-iterations 2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--pm-iterations", type=int, default=2)
    args = ap.parse_args()
    from localexpstereo_amd import io as lio
    from localexpstereo_amd import stereo
    for name, dual in (("cones", True), ("teddy", False)):
        data = lio.load_data(os.path.join(ROOT, "tests", "golden", name))            # ndisp from info.txt, like the reference
        st, lab, raw = stereo.MidV2(data, iterations=args.iterations, pmIterations=args.pm_iterations, doDual=dual, smooth_weight=1.0)
        print(f"== {name}: {data['imL'].shape[1]}x{data['imL'].shape[0]}, ndisp {data['ndisp']}, doDual {int(dual)}  ({st.seconds:.2f} s)")
        for r in st.log:
            print("%2d %6.2f\t%.0f\t%.0f\t%.0f\t%5.2f\t%5.2f" % (r["index"], r["time"], r["energy"], r["data"], r["smooth"] if r["smooth"] == r["smooth"] else 0,
                                                              r["all"], r["nonocc"]))
        if args.out:
            os.makedirs(os.path.join(args.out, name), exist_ok=True)
            lio.write_pfm(os.path.join(args.out, name, "disp0.pfm"), stereo.disparities(lab))
            if dual:
                lio.write_pfm(os.path.join(args.out, name, "disp0raw.pfm"), stereo.disparities(raw))


if __name__ == "__main__":
    main()
