#!/usr/bin/env python3
"""Reads the twelve energy plots the reference commits (the only dynamic outputs it holds) into numbers.

    python tests/golden/digitize_energy_plots.py            # needs /root/reference; writes tests/golden/plot_readings.json

Source files: /root/reference/energy_plots/{jacobian,divergence}_formulation/{64x64,128x128}_{two_Gaussians_low_B,
two_Gaussians_high_B,low_B_low_U}.png -- four Makie panels each (kinetic, magnetic, potential energy and `abs(E - E0) * 100`,
plotted by SWMHD_example.jl:133-165 / divergence_sw_mhd.jl:130-162 from the per-iteration NetCDF series).  The output is DATA
(times, values, reading tolerance), not reference source.

Method: every panel frame is a 2-pixel grey (127,127,127) rectangle, its tick marks are grey stubs outside the frame; the tick
LABEL values are typed in below from the images (CALIB) and the script checks that the number of detected stubs matches the
number of typed labels; a least-squares line through (stub pixel, label) gives the axis map.  The curve is the set of pixels
of the panel's pure colour (red, blue, green, black; drawn 4 px wide); per pixel column the reading is the mean of the top and
bottom coloured pixel (the centre of the band), `tol` = what half the band height + one pixel amounts to in data units, never
less than 1.5 px.  Readings are taken at whole model times (every 1 t.u.; 0.5 for the 15-t.u. runs)."""
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = "/root/reference/energy_plots"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plot_readings.json")
PANELS = ("kinetic", "magnetic", "potential", "error_x100")            # Makie layout [1,1] [1,2] [2,1] [2,2]
COLOURS = {"kinetic": (255, 0, 0), "magnetic": (0, 0, 255), "potential": (0, 128, 0), "error_x100": (0, 0, 0)}

# Tick labels as printed in each image: (x ticks, y ticks in increasing value) per panel.
X60, X70, X35, X15, X10 = [0, 10, 20, 30, 40, 50, 60], [0, 25, 50], [0, 10, 20, 30], [0, 5, 10, 15], [0, 5, 10]
T_END = {id(X60): 60, id(X70): 70, id(X35): 35, id(X15): 15, id(X10): 10}
CALIB = {
    "jacobian_formulation/128x128_two_Gaussians_low_B": dict(x=X60, y=dict(
        kinetic=[0.000, 0.001, 0.002, 0.003], magnetic=[0.019, 0.020, 0.021, 0.022],
        potential=[0.0, 0.00005, 0.00010], error_x100=[0.000, 0.005, 0.010])),
    "jacobian_formulation/64x64_two_Gaussians_low_B": dict(x=X70, y=dict(
        kinetic=[0.000, 0.001, 0.002], magnetic=[0.019, 0.020, 0.021],
        potential=None, error_x100=[0.00, 0.01, 0.02])),   # PE panel: Float32 plot coordinates around 490.5 (steps, uneven ticks) -- unreadable
    "jacobian_formulation/128x128_two_Gaussians_high_B": dict(x=X35, y=dict(
        kinetic=[0.00, 0.02, 0.04, 0.06, 0.08], magnetic=[0.46, 0.48, 0.50, 0.52, 0.54],
        potential=[0.000, 0.002, 0.004, 0.006, 0.008], error_x100=[0, 2, 4])),
    "jacobian_formulation/64x64_two_Gaussians_high_B": dict(x=X35, y=dict(
        kinetic=[0.00, 0.02, 0.04, 0.06], magnetic=[0.46, 0.48, 0.50, 0.52, 0.54],
        potential=[490.500, 490.502, 490.504, 490.506, 490.508], error_x100=[0, 1, 2, 3])),
    "jacobian_formulation/128x128_low_B_low_U": dict(x=X15, y=dict(
        kinetic=[0.2, 0.3, 0.4], magnetic=[0.15, 0.20, 0.25, 0.30, 0.35],
        potential=[0.00, 0.01, 0.02], error_x100=[0.0, 0.2, 0.4])),
    "jacobian_formulation/64x64_low_B_low_U": dict(x=X15, y=dict(
        kinetic=[0.20, 0.25, 0.30, 0.35, 0.40], magnetic=[0.15, 0.20, 0.25, 0.30],
        potential=[490.500, 490.505, 490.510, 490.515, 490.520], error_x100=[0.0, 0.2, 0.4, 0.6, 0.8])),
    "divergence_formulation/128x128_two_Gaussians_low_B": dict(x=X60, y=dict(
        kinetic=[0.000, 0.001, 0.002, 0.003], magnetic=[0.019, 0.020, 0.021],
        potential=[0.0, 0.00005, 0.00010], error_x100=[0.0, 0.1, 0.2, 0.3])),
    "divergence_formulation/64x64_two_Gaussians_low_B": dict(x=X60, y=dict(
        kinetic=[0.000, 0.001, 0.002, 0.003], magnetic=[0.019, 0.020, 0.021],
        potential=[0.0, 0.00005, 0.00010], error_x100=[0.00, 0.05, 0.10])),
    "divergence_formulation/128x128_two_Gaussians_high_B": dict(x=X35, y=dict(
        kinetic=[0.00, 0.05, 0.10, 0.15], magnetic=[0.50, 0.55, 0.60],
        potential=[0.000, 0.002, 0.004, 0.006, 0.008], error_x100=[0, 5, 10, 15, 20])),
    "divergence_formulation/64x64_two_Gaussians_high_B": dict(x=X10, y=dict(
        kinetic=[0.00, 0.02, 0.04, 0.06, 0.08], magnetic=[0.475, 0.500, 0.525],
        potential=[0.000, 0.002, 0.004, 0.006, 0.008], error_x100=[0.0, 0.5, 1.0])),
    "divergence_formulation/128x128_low_B_low_U": dict(x=X15, y=dict(
        kinetic=[0.2, 0.3, 0.4], magnetic=[0.15, 0.20, 0.25, 0.30, 0.35],
        potential=[0.000, 0.005, 0.010, 0.015, 0.020], error_x100=[0.0, 0.1, 0.2, 0.3, 0.4])),
    "divergence_formulation/64x64_low_B_low_U": dict(x=X15, y=dict(
        kinetic=[0.20, 0.25, 0.30, 0.35, 0.40], magnetic=[0.15, 0.20, 0.25, 0.30],
        potential=[0.000, 0.005, 0.010, 0.015, 0.020], error_x100=[0.0, 0.5, 1.0])),
}


def _runs(mask):
    """[(start, stop)) of the True runs of a 1-D boolean array"""
    d = np.diff(np.concatenate(([0], mask.astype(int), [0])))
    return list(zip(np.flatnonzero(d == 1), np.flatnonzero(d == -1)))


def find_frames(grey):
    """the four panel frames as (x0, x1, y0, y1): pixel index of the OUTER grey line of each side"""
    rows = [r for r in _runs(grey.sum(axis=1) > 200)]            # horizontal frame lines: two panels wide
    cols = [c for c in _runs(grey.sum(axis=0) > 150)]
    assert len(rows) == 4 and len(cols) == 4, (rows, cols)
    ys = [(rows[0][0], rows[1][1] - 1), (rows[2][0], rows[3][1] - 1)]
    xs = [(cols[0][0], cols[1][1] - 1), (cols[2][0], cols[3][1] - 1)]
    return [(xs[c][0], xs[c][1], ys[r][0], ys[r][1]) for r in (0, 1) for c in (0, 1)]


def tick_pixels(lum, frame):
    """darkness-weighted centres of the tick stubs below the bottom side (x) and left of the left side (y, returned bottom-up);
    a stub is ~2 px wide and antialiased when it falls between pixels"""
    x0, x1, y0, y1 = frame
    wx = (255 - lum[y1 + 2:y1 + 5, x0 - 2:x1 + 3]).min(axis=0).astype(float)       # dark in all three rows under the frame
    wy = (255 - lum[y0 - 2:y1 + 3, x0 - 4:x0 - 1]).min(axis=1).astype(float)
    xt = [x0 - 2 + np.average(np.arange(a, b), weights=wx[a:b]) for a, b in _runs(wx > 40)]
    yt = [y0 - 2 + np.average(np.arange(a, b), weights=wy[a:b]) for a, b in _runs(wy > 40)]
    return xt, yt[::-1]


def axis_map(pix, val):
    assert len(pix) == len(val), f"detected {len(pix)} tick stubs for {len(val)} typed labels: {pix} {val}"
    a, b = np.polyfit(pix, val, 1)
    assert np.abs(np.polyval([a, b], pix) - np.asarray(val)).max() <= 0.004 * abs(val[-1] - val[0]), "ticks are not equidistant"
    return a, b


def read_panel(im, grey, frame, colour, xt_val, yt_val, times):
    x0, x1, y0, y1 = frame
    xt, yt = tick_pixels(im[:, :, 0], frame)
    ax, bx = axis_map(xt, xt_val)
    ay, by = axis_map(yt, yt_val)
    inside = np.zeros(im.shape[:2], bool)
    inside[y0 + 2:y1 - 1, x0 + 2:x1 - 1] = True
    m = (im == np.asarray(colour)).all(axis=2) & inside
    out = []
    for t in times:
        px = (t - bx) / ax
        c = int(round(px))
        col = np.flatnonzero(m[:, c])
        for dc in (1, -1, 2, -2):                                  # first / last sample: the line starts a pixel or two inside
            if col.size == 0 and (t == times[0] or t == times[-1]):
                col = np.flatnonzero(m[:, c + dc])
        if col.size == 0:
            out.append(None)
            continue
        top, bot = col.min(), col.max()
        v = ay * (top + bot) / 2 + by
        tol = abs(ay) * max((bot - top + 1) / 2 - 1.0, 1.5)      # a 4-px band read at its centre: >= 1.5 px
        out.append((float(f"{v:.7g}"), float(f"{tol:.4g}")))
    return out, dict(x_per_px=float(ax), y_per_px=float(abs(ay)))


def main():
    if not os.path.isdir(ROOT):
        sys.exit("needs /root/reference (the plots are the reference's own files)")
    result = {"_about": "values read off the reference's committed energy plots by tests/golden/digitize_energy_plots.py; "
                        "error_x100 is abs(E - E0) * 100 (SWMHD_example.jl:146-147); each reading is [value, tolerance]"}
    for key, cal in CALIB.items():
        path = os.path.join(ROOT, key + ".png")
        im = np.asarray(Image.open(path).convert("RGB")).astype(int)
        grey = (im == 127).all(axis=2)
        frames = find_frames(grey)
        t_end = T_END[id(cal["x"])]
        dt = 0.5 if t_end <= 15 else 1.0
        times = [float(t) for t in np.arange(0, t_end + 1e-9, dt)]
        entry = {"png": "energy_plots/" + key + ".png", "times": times}
        for name, frame in zip(PANELS, frames):
            if cal["y"][name] is None:
                continue
            vals, scale = read_panel(im, grey, frame, COLOURS[name], cal["x"], cal["y"][name], times)
            entry[name] = vals
            entry[name + "_scale"] = scale
        result[key] = entry
    with open(OUT, "w") as f:
        f.write("{\n" + ",\n".join(json.dumps(k) + ": " + json.dumps(v, separators=(",", ":")) for k, v in result.items()) + "\n}\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
