#!/usr/bin/env python3
"""Stamp the achieved-error records the GPU tests write (gpurun_out/fullsize_parity.json from tests/test_fullsize_gpu.py,
gpurun_out/plot_parity.jsonl from tests/test_reference_plots.py with SWMHD_PLOT_PARITY_OUT set) with the kernel-source hash and put them
under profiles/r03/, where bench.py's `parity` block reads them back.   usage: python tools/parity_summary.py [--out profiles/r03]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swmhd_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--raw", default=os.path.join(ROOT, "gpurun_out"))
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03"))
a = ap.parse_args()
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
stamp = {"kernel_source_hash": _lib.source_hash(), "git_head": head}
os.makedirs(a.out, exist_ok=True)
fp = os.path.join(a.raw, "fullsize_parity.json")
if os.path.exists(fp):
    d = json.load(open(fp)); d.update(stamp, collected_by="tests/test_fullsize_gpu.py on MI355X")
    json.dump(d, open(os.path.join(a.out, "fullsize_parity.json"), "w"), indent=1, sort_keys=True)
    print("fullsize_parity.json:", len(d) - 3, "cases")
pj = os.path.join(a.raw, "plot_parity.jsonl")
if os.path.exists(pj):
    rows = [json.loads(l) for l in open(pj) if l.strip()]
    latest = {}
    for r in rows:
        latest[(r["case"], r["run"])] = r
    rows = [latest[k] for k in sorted(latest)]
    worst = max(w["ratio"] for r in rows for w in r["worst"].values())
    out = dict(stamp, collected_by="tests/test_reference_plots.py (SWMHD_PLOT_PARITY_OUT) on MI355X", slack=rows[0]["slack"], runs=len(rows),
               worst_ratio=worst, note="ratio = |run - plot| / (slack x reading tolerance); <= 1 passes", cases=rows)
    json.dump(out, open(os.path.join(a.out, "plot_parity.json"), "w"), indent=1)
    print("plot_parity.json:", len(rows), "runs, worst ratio", round(worst, 3))
