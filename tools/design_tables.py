#!/usr/bin/env python3
"""Rewrite the generated tables of DESIGN.md section 5 (between the <!-- configs:begin/end --> and <!-- benchlines:begin/end --> markers)
from profiles/r03/configs.json and profiles/r03/bench_*.json, so that the prose never drifts from the committed measurements."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r03")
d = json.load(open(os.path.join(P, "configs.json")))
rows = {k: v for k, v in d.items() if isinstance(v, dict) and "grid" in v}
g = lambda k: (rows[k]["lorentz_operator_us"], rows[k]["lorentz_operator_GBps"] / 1e3, rows[k]["tendency_kernel_us"], rows[k]["rk3_step_ms"],
               rows[k]["Mcell_steps_per_s"] / 1e3)
c2, c3 = g("config2_1024sq_divergence_uniformBx"), g("config3_4096sq_jacobian_bickley")
s2, s1, s5 = g("config3_strong_slab_4096x2048"), g("config3_strong_slab_4096x1024"), g("config3_strong_slab_4096x512")
j1, c4 = g("config3_1024sq_jacobian"), g("config4_slab_8192x1024_divergence")
c5, c5f = g("config5_slab_16384x2048_jacobian_f64"), g("config5_slab_16384x2048_jacobian_f32")
table = f"""| configuration (SURVEY §8(d)) | dtype | Lorentz operator | tendency kernel (unfused) | RK3 step | Gcell-steps/s |
|---|---|---|---|---|---|
| config2 1024² divergence uniformBx | f64 | {c2[0]:.0f} µs | {c2[2]:.0f} µs | {c2[3]:.3f} ms | **{c2[4]:.1f}** |
| config3 4096² jacobian bickley | f64 | {c3[0]:.0f} µs ({c3[1]:.1f} TB/s) | {c3[2]:.0f} µs | {c3[3]:.3f} ms | **{c3[4]:.1f}** |
| config3 strong-scaling slab 4096×2048 | f64 | {s2[0]:.0f} µs | {s2[2]:.0f} µs | {s2[3]:.3f} ms | {s2[4]:.1f} |
| config3 strong-scaling slab 4096×1024 | f64 | {s1[0]:.0f} µs | {s1[2]:.0f} µs | {s1[3]:.3f} ms | {s1[4]:.1f} |
| config3 strong-scaling slab 4096×512 | f64 | {s5[0]:.0f} µs | {s5[2]:.0f} µs | {s5[3]:.3f} ms | **{s5[4]:.1f} = {100 * s5[4] / c3[4]:.0f} % of the 4096² per-cell rate** |
| 1024² jacobian (vector-invariant) | f64 | {j1[0]:.0f} µs | {j1[2]:.0f} µs | {j1[3]:.3f} ms | {j1[4]:.1f} |
| config4 slab 8192×1024 divergence | f64 | {c4[0]:.0f} µs ({c4[1]:.1f} TB/s) | {c4[2]:.0f} µs | {c4[3]:.3f} ms | **{c4[4]:.1f}** |
| config5 slab 16384×2048 jacobian | f64 | {c5[0]:.0f} µs ({c5[1]:.1f} TB/s) | {c5[2]:.0f} µs | {c5[3]:.3f} ms | {c5[4]:.1f} |
| config5 slab 16384×2048 jacobian | f32 | {c5f[0]:.0f} µs ({c5f[1]:.1f} TB/s) | {c5f[2]:.0f} µs | {c5f[3]:.3f} ms | **{c5f[4]:.1f} = {c5f[4] / c5[4]:.2f}× fp64** |
"""
B = {f: json.loads(open(os.path.join(P, f + ".json")).read().strip().splitlines()[-1]) for f in
     ("bench_default", "bench_cons", "bench_ring", "bench_c5f32", "bench_c5f64", "bench_c4")}
v = lambda f: f"{B[f]['value'] / 1e3:.1f} ({B[f]['ms_per_step']:.3f} ms)"
tr = B["bench_default"]["roofline"].get("traffic")
lines = (f"`bench.py` lines (`profiles/r03/bench_*.json`): default {v('bench_default')}"
         + (f", `traffic` {tr / 1e9:.2f} GB per launch = {tr / (sum((64, 128, 96)) / 3 * 4096 * 4096):.2f}× the 64/128/96-byte mean the fused stages move" if tr else "")
         + f"; `--formulation Conservative` {v('bench_cons')}; `--force-ring` {v('bench_ring')} — ring of one rank through RCCL, deep-halo schedule"
         + (f", its `companion` run {B['bench_ring']['companion']['ms_per_step']:.3f} ms" if "companion" in B["bench_ring"] else "")
         + f"; `--config 5` {v('bench_c5f64')} and `--config 5 --dtype f32` {v('bench_c5f32')}; `--config 4` (the whole 8192² grid on one GPU) {v('bench_c4')}.\n")
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for tag, body in (("configs", table), ("benchlines", lines)):
    s = re.sub(rf"<!-- {tag}:begin -->.*?<!-- {tag}:end -->", lambda m: f"<!-- {tag}:begin -->\n{body}<!-- {tag}:end -->", s, flags=re.S)
open(p, "w").write(s)
print(table); print(lines)
