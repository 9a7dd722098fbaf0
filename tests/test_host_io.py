"""Host side of the command line (lbm-asynchronous_amd/host/*.c and the Python twins in the
binding): argument handling, the reference's negative tests on malformed input
(SerialCode/d2q9-bgk.c:145-148, 480-506, 588-597), output formats (:722, :737) and the
check.py acceptance rule."""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding
from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def cli(lbm):
    if not os.path.exists(lbm.CLI_PATH):
        lbm.build()
    return lbm.CLI_PATH


def run(cmd, cwd, env=None):
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, env=env)


def test_usage_on_wrong_argument_count(cli, tmp_path):
    for args in ([], ["only-one"], ["a", "b", "c"]):
        out = run([cli] + args, tmp_path)
        assert out.returncode == 1
        assert out.stderr.startswith(f"Usage: {cli} <paramfile> <obstaclefile>")


def test_missing_param_file(cli, tmp_path):
    out = run([cli, "nope.params", "nope.dat"], tmp_path)
    assert out.returncode == 1
    assert "could not open input parameter file: nope.params" in out.stderr
    assert out.stderr.startswith("Error at line ")


@pytest.mark.parametrize("n_values,what", [(0, "nx"), (1, "ny"), (2, "maxIters"), (3, "reynolds_dim"),
                                           (4, "density"), (5, "accel"), (6, "omega")])
def test_truncated_param_file(cli, tmp_path, n_values, what):
    vals = ["128", "128", "10", "10", "0.1", "0.005", "1.85"][:n_values]
    (tmp_path / "in.params").write_text("\n".join(vals) + ("\n" if vals else ""))
    out = run([cli, "in.params", "x.dat"], tmp_path)
    assert out.returncode == 1
    assert f"could not read param file: {what}" in out.stderr


@pytest.mark.parametrize("line,msg", [
    ("5 5\n", "expected 3 values per line in obstacle file"),
    ("-1 5 1\n", "obstacle x-coord out of range"),
    ("16 5 1\n", "obstacle x-coord out of range"),
    ("5 8 1\n", "obstacle y-coord out of range"),
    ("5 5 2\n", "obstacle blocked value should be 1"),
])
def test_bad_obstacle_lines(cli, lbm, tmp_path, line, msg):
    (tmp_path / "in.params").write_text("16\n8\n10\n10\n0.1\n0.005\n1.85\n")
    (tmp_path / "ob.dat").write_text("0 0 1\n" + line)
    out = run([cli, "in.params", "ob.dat"], tmp_path)
    assert out.returncode == 1 and msg in out.stderr
    # the Python twin raises the same message
    with pytest.raises(lbm.LbmError, match=msg):
        lbm.read_obstacles(str(tmp_path / "ob.dat"), 16, 8)


def test_empty_and_duplicate_obstacle_lines(lbm, tmp_path):
    """An empty obstacle file is a grid without obstacles; duplicate lines are allowed
    (the reference just assigns the flag again, SerialCode/d2q9-bgk.c:588-601)."""
    (tmp_path / "empty.dat").write_text("")
    assert lbm.read_obstacles(str(tmp_path / "empty.dat"), 16, 8).sum() == 0
    (tmp_path / "dup.dat").write_text("3 2 1\n3 2 1\n15 7 1\n")
    ob = lbm.read_obstacles(str(tmp_path / "dup.dat"), 16, 8)
    assert ob.sum() == 2 and ob[2, 3] == 1 and ob[7, 15] == 1
    # the C reader agrees (through the oracle CLI, which links the product's lbm_io.c)
    (tmp_path / "in.params").write_text("16\n8\n2\n10\n0.1\n0.005\n1.85\n")
    out = run([oracle_binding.CLI, "in.params", "dup.dat"], tmp_path)
    assert out.returncode == 0, out.stderr
    flags = np.loadtxt(tmp_path / "final_state.dat", usecols=[6]).reshape(8, 16)
    assert np.array_equal(flags.astype(np.int32), ob)
    out = run([oracle_binding.CLI, "in.params", "empty.dat"], tmp_path)
    assert out.returncode == 0 and np.loadtxt(tmp_path / "final_state.dat", usecols=[6]).sum() == 0


def test_missing_obstacle_file(cli, tmp_path):
    (tmp_path / "in.params").write_text("16\n8\n10\n10\n0.1\n0.005\n1.85\n")
    out = run([cli, "in.params", "absent.dat"], tmp_path)
    assert out.returncode == 1
    assert "could not open input obstacles file: absent.dat" in out.stderr


def test_cli_without_gpu_fails_loudly(cli, lbm, tmp_path):
    if lbm.device_count() > 0:
        pytest.skip("a GPU is present")
    out = run([cli, os.path.join(GOLDEN, "inputs", "input_128x128.params"),
               os.path.join(GOLDEN, "inputs", "obstacles_128x128.dat")], tmp_path)
    assert out.returncode == 1
    assert "no HIP device" in out.stderr
    assert not (tmp_path / "final_state.dat").exists()


def test_python_readers_match_datasets(lbm):
    for name, blocked in (("128x128", 508), ("1024x1024", 5114)):
        p = lbm.read_params(os.path.join(GOLDEN, "inputs", f"input_{name}.params"))
        ob = lbm.read_obstacles(os.path.join(GOLDEN, "inputs", f"obstacles_{name}.dat"), p.nx, p.ny)
        assert (p.nx, p.ny) == tuple(int(v) for v in name.split("x"))
        assert int(ob.sum()) == blocked
    big = lbm.tile_obstacles(ob, 8192, 8192)
    assert int(big.sum()) == 327296              # 8x8 tiles
    assert np.array_equal(big[1024:2048, 3072:4096], ob)


def test_output_formats_through_oracle_cli(lbm, tmp_path):
    """The product's writers (host/lbm_io.c, linked into the oracle CLI) produce the reference's
    line formats; the Python writers produce identical bytes."""
    oracle_binding.load()
    (tmp_path / "in.params").write_text("16\n8\n5\n10\n0.1\n0.005\n1.85\n")
    (tmp_path / "ob.dat").write_text("".join(f"{x} 0 1\n" for x in range(16)) + "3 4 1\n")
    out = run([oracle_binding.CLI, "in.params", "ob.dat"], tmp_path)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0] == "==done=="
    assert lines[1].startswith("Reynolds number:\t\t") and "E" in lines[1]
    for i, label in enumerate(("Init", "Compute", "Collate", "Total")):
        assert lines[2 + i].startswith(f"Elapsed {label} time:\t\t\t") and lines[2 + i].endswith(" (s)")
    fs = (tmp_path / "final_state.dat").read_text().splitlines()
    av = (tmp_path / "av_vels.dat").read_text().splitlines()
    assert len(fs) == 16 * 8 and len(av) == 5
    import re
    num = r"-?\d\.\d{12}E[+-]\d{2}"
    assert re.fullmatch(rf"0 0 {num} {num} {num} {num} 1", fs[0])
    assert re.fullmatch(rf"3 4 {num} {num} {num} {num} 1", fs[4 * 16 + 3])
    assert re.fullmatch(rf"15 7 {num} {num} {num} {num} 0", fs[-1])
    assert re.fullmatch(rf"4:\t{num}", av[4])
    # Python twins: same bytes
    p = lbm.read_params(str(tmp_path / "in.params"))
    ob = lbm.read_obstacles(str(tmp_path / "ob.dat"), p.nx, p.ny)
    o = oracle_binding.load()
    cells = o.init_cells(p)
    avv = o.run(p, cells, ob, 5)
    lbm.write_final_state(str(tmp_path / "fs_py.dat"), o.final_state(p, cells, ob), ob)
    lbm.write_av_vels(str(tmp_path / "av_py.dat"), avv)
    assert (tmp_path / "fs_py.dat").read_bytes() == (tmp_path / "final_state.dat").read_bytes()
    assert (tmp_path / "av_py.dat").read_bytes() == (tmp_path / "av_vels.dat").read_bytes()


def test_check_rule_matches_reference_formula(lbm):
    """check.py: diff = ref - sim; pct = 100*diff/(ref-diff); worst |pct| vs tolerance; non-finite
    fails (check/check.py:83-99, 136-148)."""
    ref = np.array([1.0, 2.0, 4.0])
    sim = np.array([1.0, 2.02, 4.0])
    d = lbm.check_rule(ref, sim)
    assert d["index"] == 1
    assert d["max_diff_pcnt"] == pytest.approx(100 * (2.0 - 2.02) / 2.02)
    assert lbm.check_passes(ref, sim, 1.0)
    assert not lbm.check_passes(ref, np.array([1.0, 2.03, 4.0]), 1.0)
    assert not lbm.check_passes(ref, np.array([1.0, np.nan, 4.0]), 1.0)
    with pytest.raises(lbm.LbmError):
        lbm.check_rule(ref, sim[:2])
