#!/usr/bin/env python3
"""Discriminator sweep: run the CPU oracle with each version-dependent choice of DESIGN.md section 3 switched
and compare with the reference's energy plots (tests/golden/plot_readings.json).  Test infrastructure: uses oracle/.

    python tools/plot_sweep.py [--case jacobian_formulation/128x128_two_Gaussians_low_B] [--variants name=value,... ...] [--out file.json]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import plot_cases as P

DEFAULT_VARIANTS = ["", "rbeta_mirror=1", "js_weights=1", "vel_beta=1", "vel_beta=2", "vel_beta=3", "vel_beta=4", "vhat4=1", "no_cdivU=1",
                    "eps=1e-10", "eps=1e-2", "weno_exp=1"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="jacobian_formulation/128x128_two_Gaussians_low_B")
    ap.add_argument("--variants", nargs="*", default=DEFAULT_VARIANTS)
    ap.add_argument("--times", default="20,28,40,50,60")
    ap.add_argument("--t-end", type=float, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = P.readings()[a.case]
    ts = [float(t) for t in a.times.split(",")]
    rows = {}
    print(f"{a.case}\n{'variant':28s} " + " ".join(f"err@{t:<4g}" for t in ts) + "  KE(end)   ME(end)   worst |run-plot|/tol  K M P E")
    j = [r["times"].index(t) for t in ts]
    print(f"{'PLOT':28s} " + " ".join(f"{r['error_x100'][k][0]:8.5f}" for k in j) + f"  {r['kinetic'][j[-1]][0]:.6f} {r['magnetic'][j[-1]][0]:.6f}")
    for v in a.variants:
        kw = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in v.split(",") if kv)
        t0 = time.time()
        s = P.run_oracle(a.case, variant=kw, t_end=a.t_end or ts[-1])
        i = [s["times"].index(t) for t in ts]
        c = P.compare(s, r)
        print(f"{v or 'default':28s} " + " ".join(f"{s['error_x100'][k]:8.5f}" for k in i) + f"  {s['kinetic'][i[-1]]:.6f} {s['magnetic'][i[-1]]:.6f}   "
              + " ".join(f"{c[p][0]:6.1f}" for p in c) + f"   ({time.time() - t0:.0f} s)", flush=True)
        rows[v or "default"] = dict(series=s, worst={p: c[p] for p in c})
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(case=a.case, plot={p: r[p] for p in ("times", "kinetic", "magnetic", "potential", "error_x100") if p in r}, runs=rows), f)


if __name__ == "__main__":
    main()
