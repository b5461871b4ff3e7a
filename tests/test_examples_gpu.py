"""The command-line example (examples/run_swmhd.py: the reference's driver set-ups on the host mirror) runs and reproduces the initial
energies the reference's plots show (two Gaussians amp 0.5 -> ME ~ 0.545; the vortex -> KE = 25 pi / 8)."""
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("form,ic,me0", [("jacobian", "uniform", None), ("divergence", "gaussians", 0.5461)])
def test_example_runs(tmp_path, form, ic, me0):
    csvf = tmp_path / "e.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run_swmhd.py"), "--formulation", form, "--ic", ic, "--size", "128",
                        "--stop-time", "0.5", "--every", "25", "--energies", str(csvf)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split(",") for l in csvf.read_text().strip().splitlines()[1:]]
    assert len(rows) == 3 and abs(float(rows[-1][0]) - 0.5) < 1e-9
    assert abs(float(rows[0][1]) - 25 * math.pi / 8) < 2e-3 * 25 * math.pi / 8            # KE of (u, v) = 5 (y, -x) exp(-r^2), 2nd-order quadrature
    if me0 is not None:
        assert abs(float(rows[0][2]) - me0) < 2e-3
    assert "Simulation took" in r.stdout and all(math.isfinite(float(x)) for row in rows for x in row)


def test_example_reruns_a_plotted_case_beside_its_readings():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run_swmhd.py"), "--plot-case", "divergence_formulation/64x64_two_Gaussians_high_B"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [l for l in r.stdout.splitlines() if l.startswith("t = ")][-1]
    assert last.startswith("t =  10.0")
    run, plot = last.split("err")[1].split("|")
    assert abs(float(run) - float(plot.split("+-")[0])) < 0.05          # 1.19 against the plot's 1.18
