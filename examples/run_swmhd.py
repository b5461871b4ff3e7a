#!/usr/bin/env python3
"""The reference's two driver scripts as one command-line example of the host mirror (no plotting, no NetCDF/JLD2):

    python examples/run_swmhd.py --formulation jacobian   [--size 64] [--stop-time 30] [--dt 0.01] [--ic uniform|gaussians]
    python examples/run_swmhd.py --formulation divergence ...

Set-up as jacobian_formulation/SWMHD_example.jl:7-42 / divergence_formulation/divergence_sw_mhd.jl:7-40: [-5,5]^2 periodic grid,
g = 9.81, f = 1, RK3, A = 0.5|y| ("uniform B_x") or the two Gaussians, h = 1, the Gaussian vortex (u, v) = 5 (y, -x) exp(-r^2)
(multiplied by h for the conservative variables), dt = 0.01, stop time 30.  Every --every iterations one progress line like the
reference's (SWMHD_example.jl:47-61: time, iteration, max|u|, max|A|, min h, wall time) and one row of the energies the reference
sends to NetCDF (:74-77) into --energies (CSV).  --dump-every T writes the fields incl. halos as .npy (the JLD2 writer's role, :80-84).
The step loop runs through HIP-graph replays (two RK3 steps per replay).

    python examples/run_swmhd.py --plot-case jacobian_formulation/128x128_two_Gaussians_low_B
re-runs one of the twelve runs behind the reference's committed energy plots (energy_plots/*/*.png; set-up from the scripts' commented
alternatives, see tests/plot_cases.py) and prints, beside every energy row, the value read off the plot at that time
(tests/golden/plot_readings.json) -- the comparison tests/test_reference_plots.py asserts."""
import argparse, csv, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--formulation", choices=["jacobian", "divergence"], default="jacobian")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--stop-time", type=float, default=30.0)
    ap.add_argument("--dt", type=float, default=0.01)
    ap.add_argument("--ic", choices=["uniform", "gaussians"], default="uniform")
    ap.add_argument("--amp", type=float, default=None, help="A amplitude (default 0.5 for |y|, 0.1 / 0.5 for the Gaussians)")
    ap.add_argument("--every", type=int, default=100, help="iterations between progress lines / energy rows")
    ap.add_argument("--energies", default=None, help="CSV file for (time, KE, ME, PE, total)")
    ap.add_argument("--dump-every", type=float, default=0.0, help="model time between field dumps (0 = none)")
    ap.add_argument("--out", default="swmhd_out")
    ap.add_argument("--plot-case", default=None, help="one of the reference's plotted runs, e.g. jacobian_formulation/64x64_low_B_low_U")
    a = ap.parse_args()
    if a.plot_case:
        return plot_case(a)

    import torch
    import swmhd_amd as S
    from swmhd_amd import configs
    N, L = a.size, 10.0
    grid = S.RectilinearGrid(size=(N, N), x=(-L / 2, L / 2), y=(-L / 2, L / 2))
    form = "VectorInvariant" if a.formulation == "jacobian" else "Conservative"
    model = S.ShallowWaterModel(grid, configs.G, configs.F, formulation=form)
    amp = a.amp if a.amp is not None else (0.5 if a.ic == "uniform" or form == "Conservative" else 0.1)
    A0 = (lambda X, Y: amp * np.abs(Y)) if a.ic == "uniform" else configs.two_gaussians(amp)
    u0 = lambda X, Y: 5 * Y * np.exp(-(X ** 2 + Y ** 2))
    v0 = lambda X, Y: -5 * X * np.exp(-(X ** 2 + Y ** 2))
    n1, n2 = model.names[:2]
    model.set(**{n1: u0, n2: v0, "h": lambda X, Y: np.ones_like(X), "A": A0})      # h = 1: (uh, vh) = (u, v)
    nsteps = int(round(a.stop_time / a.dt))
    rows = []

    def report(wall):
        d = model.diagnostics()
        print(f"Time: {model.clock_time:9.3f}, iteration: {model.iteration}, max(|u|): {max(d['max_abs_u'], d['max_abs_v']):.2e}, "
              f"max(|A|): {d['max_abs_A']:.2e}, min(h): {d['min_h']:.2e}, wall time: {wall * 1e3:.1f} ms "
              f"| KE {d['kinetic_energy']:.6f} ME {d['magnetic_energy']:.6f} PE {d['potential_energy']:.3e} total {d['total_energy']:.6f}", flush=True)
        rows.append((model.clock_time, d["kinetic_energy"], d["magnetic_energy"], d["potential_energy"], d["total_energy"]))

    report(0.0)
    e0 = rows[0][4]
    model.time_step(a.dt)
    model.capture_graph(a.dt)
    next_dump = a.dump_every
    t_start = time.perf_counter()
    while model.iteration < nsteps:
        n = min(a.every - model.iteration % a.every, nsteps - model.iteration)
        t0 = time.perf_counter()
        model.time_steps(n, a.dt)
        model.synchronize()
        report(time.perf_counter() - t0)
        if a.dump_every > 0 and model.clock_time + 1e-12 >= next_dump:
            os.makedirs(a.out, exist_ok=True)
            model.save_checkpoint(os.path.join(a.out, f"fields_{model.iteration:07d}"))
            next_dump += a.dump_every
    total = time.perf_counter() - t_start
    print(f"Simulation took {total:.2f} s to finish running ({nsteps} iterations, {N * N * nsteps / total / 1e6:.1f} Mcell-steps/s); "
          f"energy drift abs(E - E0) * 100 = {abs(rows[-1][4] - e0) * 100:.4f}")
    if a.energies:
        with open(a.energies, "w", newline="") as f:
            w = csv.writer(f); w.writerow(["time", "kinetic", "magnetic", "potential", "total"]); w.writerows(rows)


def plot_case(a):
    """One of the reference's twelve plotted runs through the HIP engine, the plot's own readings printed beside the run's energies."""
    import json
    import swmhd_amd as S
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "plot_readings.json")) as f:
        R = json.load(f)
    if a.plot_case not in R:
        sys.exit("--plot-case must be one of: " + ", ".join(k for k in sorted(R) if not k.startswith("_")))
    r = R[a.plot_case]
    form_dir, rest = a.plot_case.split("/")
    N, ic = int(rest.split("x")[0]), rest.split("_", 1)[1]
    form = "VectorInvariant" if form_dir.startswith("jacobian") else "Conservative"
    L, dt = 10.0, 0.01
    gauss = lambda amp: (lambda X, Y: amp * np.exp(-((X - 0.5) ** 2 + Y ** 2)) - amp * np.exp(-((X + 0.5) ** 2 + Y ** 2)))
    zero = lambda X, Y: np.zeros_like(X)
    bcs, topo, u0, v0 = None, ("Periodic", "Periodic", "Flat"), zero, zero
    if ic == "low_B_low_U":          # divergence_sw_mhd.jl:34,36-37 with the commented GradientBoundaryCondition(-0.05), (Periodic, Bounded)
        A0, topo = (lambda X, Y: -0.05 * Y), ("Periodic", "Bounded", "Flat")
        u0, v0 = (lambda X, Y: Y * np.exp(-(X ** 2 + Y ** 2))), (lambda X, Y: -X * np.exp(-(X ** 2 + Y ** 2)))
        bcs = {"A": S.FieldBoundaryConditions(south=S.GradientBoundaryCondition(-0.05), north=S.GradientBoundaryCondition(-0.05))}
    else:
        A0 = gauss(0.1 if ic.endswith("low_B") else 0.5)
    grid = S.RectilinearGrid(size=(N, N), x=(-L / 2, L / 2), y=(-L / 2, L / 2), topology=topo)
    m = S.ShallowWaterModel(grid, 9.81, 1.0, formulation=form, boundary_conditions=bcs)
    n1, n2 = m.names[:2]
    m.set(**{n1: u0, n2: v0, "h": lambda X, Y: np.ones_like(X), "A": A0})
    t_end = a.stop_time if a.stop_time != 30.0 else r["times"][-1]
    step = r["times"][1] - r["times"][0]
    e0 = None
    print(f"{a.plot_case}: {form}, {N}x{N}, dt = {dt}, to t = {t_end:g}   [run | plot reading +- tolerance]")
    t0 = time.perf_counter()
    for k, t in enumerate(r["times"]):
        if t > t_end + 1e-9:
            break
        if k:
            m.time_steps(int(round(step / dt)), dt)
        d = m.diagnostics()
        e0 = d["total_energy"] if e0 is None else e0
        vals = dict(kinetic=d["kinetic_energy"], magnetic=d["magnetic_energy"], potential=d["potential_energy"], error_x100=abs(d["total_energy"] - e0) * 100)
        cells = []
        for p in ("kinetic", "magnetic", "potential", "error_x100"):
            rd = r.get(p, [None] * len(r["times"]))[k]
            ref = "      -      " if rd is None else f"{rd[0] - (490.5 if (p == 'potential' and rd[0] > 400) else 0):9.5f}+-{rd[1]:.5f}"
            cells.append(f"{p[:3]} {vals[p]:9.5f} | {ref}")
        print(f"t = {t:5.1f}  " + "   ".join(cells), flush=True)
    print(f"{m.iteration} iterations in {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()
