"""bench.py's self-launcher and workload table (CPU): `python bench.py --gpus 2` with no torchrun around it must start its own
workers, reach process-group creation and print exactly one line; a failing worker must fail the whole run without a line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_self_launch_reaches_rendezvous_with_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rendezvous-only"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"launcher": "ok", "n_gpus": 2, "backend": "gloo"}


def test_failed_worker_fails_the_run_without_a_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "no-such-backend", "--rendezvous-only"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""


def test_launcher_deadline_stops_a_rank_that_never_arrives():
    """a rank stuck before / inside communicator creation: the launcher must not wait for ever (round-2 advisor finding)"""
    import time
    e = _env(); e["SWMHD_BENCH_TEST_HANG_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--rendezvous-only", "--launch-timeout", "20"], env=e,
                       capture_output=True, text=True, timeout=200)
    assert r.returncode == 124 and r.stdout.strip() == ""
    assert time.time() - t0 < 90
    assert "still running" in r.stderr and "rank 1" in r.stderr and "SWMHD_BENCH_TEST_HANG_RANK" in r.stderr     # who, and its stderr tail


def test_failed_worker_reports_its_stderr_tail():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "no-such-backend", "--rendezvous-only"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode not in (0, 124) and "--- rank 0" in r.stderr and "--- rank 1" in r.stderr


def test_world_size_mismatch_is_an_error_not_an_assert():
    e = _env(); e.update(RANK="0", WORLD_SIZE="3", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and r.stdout.strip() == ""


@pytest.mark.parametrize("config,scaling,world,want", [
    (3, None, 1, (4096, 4096, 4096, "strong")), (3, None, 8, (4096, 4096, 512, "strong")), (3, "weak", 8, (4096, 32768, 4096, "weak")),
    (4, None, 8, (8192, 8192, 1024, "strong")), (4, None, 1, (8192, 8192, 8192, "strong")), (4, "weak", 2, (8192, 2048, 1024, "weak")),
    (5, None, 1, (16384, 2048, 2048, "weak")), (5, None, 8, (16384, 16384, 2048, "weak")),
])
def test_workload_shapes(config, scaling, world, want):
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--config", str(config)] + (["--scaling", scaling] if scaling else [])
    cfg, Nx, Nyg, Nyl, form, sc, ydom = bench.workload(bench.parse(argv), world)
    assert (Nx, Nyg, Nyl, sc) == want
    y0, y1 = cfg["domain"]["y"]
    full = {3: 4096, 4: 8192, 5: 16384}[config]
    assert abs((ydom[1] - ydom[0]) / Nyg - (y1 - y0) / full) < 1e-15      # dy of the configuration is kept
    assert abs(0.5 * (ydom[0] + ydom[1]) - 0.5 * (y0 + y1)) < 1e-12       # centred on the configuration's axis
