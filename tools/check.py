#!/usr/bin/env python3
"""check.py -- this repository's own validator, interface-compatible with the reference's
`check/check.py` (which cannot travel to the GPU box).

    python tools/check.py --ref-av-vels-file R_AV --ref-final-state-file R_FS \
                          --av-vels-file AV --final-state-file FS [--tolerance 1]

Same inputs, same six report lines, same exit status as /root/reference/check/check.py
(:19-50 arguments; :57-63 columns 1 of av_vels and 0,1,5 of final_state; :73-80 coordinate and
step-count checks; :83-99 metric 100*(ref-sim)/sim, worst entry; :136-148 fail when the worst
percentage is not finite or exceeds the tolerance).  The arithmetic lives in
`lbm-asynchronous_amd/__init__.py::check_rule`, shared with the test-suite.
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_package():
    spec = importlib.util.spec_from_file_location(
        "lbm_asynchronous_amd", os.path.join(ROOT, "lbm-asynchronous_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["lbm_asynchronous_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def main(argv=None):
    ap = argparse.ArgumentParser(description="LBM output validator (check.py-compatible)",
                                 fromfile_prefix_chars="@",
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("--tolerance", type=float, default=1.0,
                    help="Percentage tolerance to match against reference results")
    ap.add_argument("--ref-av-vels-file", required=True, help="reference av_vels results file")
    ap.add_argument("--ref-final-state-file", required=True, help="reference final_state results file")
    ap.add_argument("--av-vels-file", required=True, help="calculated av_vels results file")
    ap.add_argument("--final-state-file", required=True, help="calculated final_state results file")
    args = ap.parse_args(argv)
    lbm = load_package()

    av_ref = np.loadtxt(args.ref_av_vels_file, usecols=[1])
    fs_ref = np.loadtxt(args.ref_final_state_file, usecols=[0, 1, 5])
    av_sim = np.loadtxt(args.av_vels_file, usecols=[1])
    fs_sim = np.loadtxt(args.final_state_file, usecols=[0, 1, 5])

    if fs_ref.shape != fs_sim.shape or np.any(fs_ref[:, 0:2] != fs_sim[:, 0:2]):
        print("Final state files coordinates were not the same")
        return 1
    if av_ref.size != av_sim.size:
        print("Different number of steps in av_vels files")
        return 1

    av = lbm.check_rule(av_ref, av_sim)
    fs = lbm.check_rule(fs_ref[:, 2], fs_sim[:, 2])
    print("Total difference in av_vels : {:.12E}".format(av["total"]))
    print("Biggest difference (at step {:d}) : {:.12E}".format(av["index"], av["max_diff"]))
    print("  {:.12E} vs. {:.12E} = {:.2g}%".format(av["sim_val"], av["ref_val"], av["max_diff_pcnt"]))
    print()
    k = fs["index"]
    print("Total difference in final_state : {:.12E}".format(fs["total"]))
    print("Biggest difference (at coord ({:d},{:d})) : {:.12E}".format(int(fs_sim[k, 0]), int(fs_sim[k, 1]),
                                                                      fs["max_diff"]))
    print("  {:.12E} vs. {:.12E} = {:.2g}%".format(fs["sim_val"], fs["ref_val"], fs["max_diff_pcnt"]))
    print()

    fs_failed = (not np.isfinite(fs["max_diff_pcnt"])) or abs(fs["max_diff_pcnt"]) > args.tolerance
    av_failed = (not np.isfinite(av["max_diff_pcnt"])) or abs(av["max_diff_pcnt"]) > args.tolerance
    if fs_failed:
        print("final state failed check")
    if av_failed:
        print("av_vels failed check")
    if fs_failed or av_failed:
        return 1
    print("Both tests passed!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
