"""tools/check.py: the repository's own validator must behave like the reference's check/check.py --
same report, same verdict, same exit status.  Where /root/reference is mounted (builder container)
the two programs are run side by side on the same files; elsewhere the tool is exercised on
files written from the golden fixtures."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

TOOL = os.path.join(ROOT, "tools", "check.py")
REF_TOOL = "/root/reference/check/check.py"


def write_outputs(lbm, tmp_path, name, perturb=0.0, prefix=""):
    """av_vels.dat / final_state.dat in the program's text format, from the SerialCode fixture."""
    ref = np.load(os.path.join(GOLDEN, f"serialcode_{name}.npz"))
    av = ref["av_vels"].astype(np.float32) * np.float32(1.0 + perturb)
    fields = {"u_x": ref["u_x"], "u_y": ref["u_y"], "u": ref["u"],
              "pressure": (ref["pressure"] * np.float32(1.0 + perturb)).astype(np.float32)}
    avf, fsf = tmp_path / f"{prefix}av_vels.dat", tmp_path / f"{prefix}final_state.dat"
    lbm.write_av_vels(str(avf), av)
    lbm.write_final_state(str(fsf), fields, ref["obstacle"].astype(np.int32))
    return str(avf), str(fsf)


def run(tool, ref_av, ref_fs, av, fs, *extra):
    return subprocess.run([sys.executable, tool, f"--ref-av-vels-file={ref_av}", f"--ref-final-state-file={ref_fs}",
                           f"--av-vels-file={av}", f"--final-state-file={fs}", *extra],
                          capture_output=True, text=True)


@pytest.mark.parametrize("perturb,passes", [(0.0, True), (0.004, True), (0.02, False)])
def test_verdict_and_report(lbm, tmp_path, perturb, passes):
    ref_av, ref_fs = write_outputs(lbm, tmp_path, "128x128", 0.0, "ref_")
    av, fs = write_outputs(lbm, tmp_path, "128x128", perturb, "sim_")
    out = run(TOOL, ref_av, ref_fs, av, fs)
    assert out.returncode == (0 if passes else 1), out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].startswith("Total difference in av_vels : ")
    assert lines[1].startswith("Biggest difference (at step ")
    assert lines[4].startswith("Total difference in final_state : ")
    assert lines[5].startswith("Biggest difference (at coord (")
    assert (lines[-1] == "Both tests passed!") == passes
    if not passes:
        assert "final state failed check" in out.stdout and "av_vels failed check" in out.stdout
    # a looser tolerance flips the verdict
    if not passes:
        assert run(TOOL, ref_av, ref_fs, av, fs, "--tolerance", "5").returncode == 0


def test_mismatched_inputs(lbm, tmp_path):
    ref_av, ref_fs = write_outputs(lbm, tmp_path, "128x128", 0.0, "ref_")
    av, fs = write_outputs(lbm, tmp_path, "128x256", 0.0, "sim_")
    out = run(TOOL, ref_av, ref_fs, av, fs)
    assert out.returncode == 1 and "coordinates were not the same" in out.stdout
    # same grid, fewer steps
    short = tmp_path / "short_av.dat"
    lbm.write_av_vels(str(short), np.load(os.path.join(GOLDEN, "serialcode_128x128.npz"))["av_vels"][:100])
    out = run(TOOL, ref_av, ref_fs, str(short), ref_fs)
    assert out.returncode == 1 and "Different number of steps" in out.stdout


@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason="reference not mounted (GPU box)")
@pytest.mark.parametrize("perturb", [0.0, 0.004, 0.02])
def test_same_output_as_reference_check_py(lbm, tmp_path, perturb):
    """Side by side with the reference's own validator on the reference's own golden files."""
    ref_av = "/root/reference/check/128x128.av_vels.dat"
    ref_fs = "/root/reference/check/128x128.final_state.dat"
    av, fs = write_outputs(lbm, tmp_path, "128x128", perturb, "sim_")
    mine = run(TOOL, ref_av, ref_fs, av, fs)
    theirs = run(REF_TOOL, ref_av, ref_fs, av, fs)
    assert mine.returncode == theirs.returncode
    assert mine.stdout == theirs.stdout
