#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Run in the builder container (needs /root/reference).

    python tests/golden/make_golden.py [--ref-out DIR]

Inputs and where each fixture comes from:

* inputs/                      -- the reference's four data sets, copied verbatim
                                  (/root/reference/dataSet/*.params, *.dat: data, not code).
* check_goldens.npz            -- the reference's own double-precision goldens
                                  (/root/reference/check/*.av_vels.dat column 1 and
                                  *.final_state.dat column 5 = pressure), as float64 arrays.
                                  The 256x256 and 1024x1024 final_state goldens are missing upstream
                                  (.MISSING_LARGE_BLOBS).
* serialcode_<grid>.npz        -- outputs of the REFERENCE PROGRAM ITSELF: oracle/_ref/d2q9-bgk-serial
                                  (SerialCode/d2q9-bgk.c compiled unmodified by oracle/Makefile with
                                  the reference's flags) run on each data set; av_vels and the four
                                  final_state columns parsed from its text files into float32
                                  (%.12E prints 13 significant digits, so the fp32 values are
                                  recovered exactly), plus the md5 of the text files and the Reynolds
                                  number it printed.  For 1024x1024 only av_vels and pressure are kept
                                  (size).
DIR (default /tmp/ref_out) holds one sub-directory per grid with the reference program's
final_state.dat, av_vels.dat and stdout.txt; regenerate it with

    for g in 128x128 128x256 256x256 1024x1024; do (mkdir -p DIR/$g && cd DIR/$g && \
        /root/repo/oracle/_ref/d2q9-bgk-serial /root/reference/dataSet/input_$g.params \
        /root/reference/dataSet/obstacles_$g.dat > stdout.txt); done
"""
import argparse
import hashlib
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
GRIDS = ("128x128", "128x256", "256x256", "1024x1024")


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-out", default="/tmp/ref_out")
    args = ap.parse_args()

    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    for g in GRIDS:
        for name in (f"input_{g}.params", f"obstacles_{g}.dat"):
            shutil.copyfile(os.path.join(REF, "dataSet", name), os.path.join(HERE, "inputs", name))

    gold = {}
    for g in GRIDS:
        gold[f"av_vels_{g}"] = np.loadtxt(os.path.join(REF, "check", f"{g}.av_vels.dat"), usecols=[1])
        fs = os.path.join(REF, "check", f"{g}.final_state.dat")
        if os.path.exists(fs):
            gold[f"pressure_{g}"] = np.loadtxt(fs, usecols=[5])
    np.savez_compressed(os.path.join(HERE, "check_goldens.npz"), **gold)

    for g in GRIDS:
        d = os.path.join(args.ref_out, g)
        nx, ny = (int(v) for v in g.split("x"))
        av = np.loadtxt(os.path.join(d, "av_vels.dat"), usecols=[1]).astype(np.float32)
        fs = np.loadtxt(os.path.join(d, "final_state.dat"), usecols=[2, 3, 4, 5, 6])
        rec = {"av_vels": av, "pressure": fs[:, 3].astype(np.float32).reshape(ny, nx),
               "obstacle": fs[:, 4].astype(np.uint8).reshape(ny, nx),
               "md5_final_state": np.array(md5(os.path.join(d, "final_state.dat"))),
               "md5_av_vels": np.array(md5(os.path.join(d, "av_vels.dat")))}
        if g != "1024x1024":
            rec["u_x"] = fs[:, 0].astype(np.float32).reshape(ny, nx)
            rec["u_y"] = fs[:, 1].astype(np.float32).reshape(ny, nx)
            rec["u"] = fs[:, 2].astype(np.float32).reshape(ny, nx)
        with open(os.path.join(d, "stdout.txt")) as fh:
            for line in fh:
                if line.startswith("Reynolds number"):
                    rec["reynolds"] = np.array(float(line.split()[-1]))
                if line.startswith("Elapsed Compute time"):
                    rec["compute_seconds_builder_container"] = np.array(float(line.split()[-2]))
        np.savez_compressed(os.path.join(HERE, f"serialcode_{g}.npz"), **rec)
        print(g, "av_vels", av.shape, "md5", rec["md5_final_state"])


if __name__ == "__main__":
    main()
