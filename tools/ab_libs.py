#!/usr/bin/env python3
"""Same-call A/B of two builds of libswmhd.so: alternates `tools/stage_times.py` between the libraries (fresh process each, SWMHD_LIBRARY)
for R rounds and prints the per-stage medians.   usage: python tools/ab_libs.py libA.so libB.so [libC.so ...] [rounds] [N] [tool.py]   (SWMHD_FORM=Conservative for the other model)"""
import os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [os.path.abspath(a) for a in sys.argv[1:] if a.endswith(".so")]
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
rounds = int(rest[0]) if len(rest) > 0 else 3
N = rest[1] if len(rest) > 1 else "4096"
tool = rest[2] if len(rest) > 2 else "stage_times.py"
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), N], env=dict(os.environ, SWMHD_LIBRARY=l),
                             capture_output=True, text=True).stdout.strip()
        print(os.path.basename(l), out, flush=True)
        m = re.search(r"\[([\d., ]+)\].*step ms: ([\d.]+)", out)
        if m:
            res[l].append([float(x) for x in m.group(1).split(",")] + [float(m.group(2))])
for l in libs:
    if res[l]:
        cols = list(zip(*res[l]))
        print(f"{os.path.basename(l):28s} median stages us {[round(statistics.median(c), 1) for c in cols[:-1]]}  step ms {statistics.median(cols[-1]):.4f}")
