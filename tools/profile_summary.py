#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of tools/profile_r03.sh (gpurun_out/prof_r03/) into the small JSON / CSV summaries under
profiles/r03/ that bench.py reads back.  Every file records the kernel-source hash (swmhd_amd._lib.source_hash) and the git HEAD it
was collected at; bench.py echoes a value only while the hash still matches the kernels that are running.

    python tools/profile_summary.py [--raw gpurun_out/prof_r03] [--out profiles/r03] [--steps 100]

Counters: SQ_* are summed over all waves of a dispatch; FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte
requests at 64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section), so it is doubled.
"""
import argparse, collections, csv, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swmhd_amd import _lib  # noqa: E402

MODE_STAGE = {"1": "stage1", "7": "stage2", "3": "stage3", "4": "tendency_only"}   # stage 1 stores no tendencies since round 3 (MODE 1; it was MODE 5)


def kernel_key(name):
    m = re.search(r"k_tendency_(vi|cons)_march<(double|float), (\d), (\d+), (\d)", name)
    if m:
        return f"k_tendency_{m.group(1)}_march", m.group(5)
    m = re.search(r"(k_lorentz_\w+_march|k_halo_multi|k_tendency_tile|k_lorentz_jacobian|k_lorentz_divergence)", name)
    return (m.group(1), None) if m else (None, None)


def provenance(note):
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "swmhd_amd/csrc"], capture_output=True, text=True).stdout.strip()
    return {"kernel_source_hash": _lib.source_hash(), "git_head": head + ("+uncommitted kernel edits" if dirty else ""), "collected_by": note}


def counters(path):
    """{(kernel, mode): {counter: [values per dispatch]}} from a *_counter_collection.csv"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = kernel_key(r["Kernel_Name"])
            if k[0]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--raw", default=os.path.join(ROOT, "gpurun_out", "prof_r03"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03"))
    ap.add_argument("--steps", type=int, default=100, help="timed steps of the --stats run (the last 3*steps stage launches)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    geo = {"Nx": 4096, "rows": 4096}

    # ---- kernel trace + stats of `bench.py --steps 100 --warmup 20`
    st = os.path.join(a.raw, "stats", "step_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(a.out, "fullstep_kernel_stats.csv"))
        rows = list(csv.DictReader(open(st)))
        out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-seconds 0 --steps 100 --warmup 20",
               **provenance("tools/profile_r03.sh + tools/profile_summary.py"), "kernels": {}}
        for r in rows:
            k, mode = kernel_key(r["Name"])
            if not k:
                continue
            tag = k + (("_" + MODE_STAGE.get(mode, "mode" + mode)) if mode else "")
            out["kernels"][tag] = {"calls": int(r["Calls"]), "mean_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                   "max_us": float(r["MaxNs"]) / 1e3}
        # the timed region only: the last 3*steps fused-stage dispatches of the trace
        tr = os.path.join(a.raw, "stats", "step_kernel_trace.csv")
        if os.path.exists(tr):
            stage = []
            for r in csv.DictReader(open(tr)):
                k, mode = kernel_key(r["Kernel_Name"])
                if k and k.startswith("k_tendency") and mode in ("1", "7", "3"):
                    stage.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, mode))
            stage.sort()
            timed = stage[-3 * a.steps:]
            out["tendency_stage_mean_ms"] = mean([d for _, d, _ in timed]) / 1e3
            out["tendency_stage_mean_ms_by_stage"] = {MODE_STAGE[m]: mean([d for _, d, mm in timed if mm == m]) / 1e3 for m in ("1", "7", "3")}
            out["tendency_stage_mean_note"] = f"mean over the last {len(timed)} fused-stage dispatches (= the timed region of the run)"
        json.dump(out, open(os.path.join(a.out, "fullstep_kernel_stats.json"), "w"), indent=1)
        bj = os.path.join(a.raw, "stats_bench.json")
        if os.path.exists(bj):
            shutil.copy(bj, os.path.join(a.out, "fullstep_bench_under_rocprof.json"))

    # ---- operator kernels: kernel trace of tools/time_ops.py 4096 (300 spin-up + 30 timed launches per kernel)
    so = os.path.join(a.raw, "ops", "ops_kernel_stats.csv")
    if os.path.exists(so):
        shutil.copy(so, os.path.join(a.out, "operators_kernel_stats.csv"))
        out = {"command": "rocprofv3 --kernel-trace --stats -- python3 tools/time_ops.py 4096", **provenance("tools/profile_r03.sh + tools/profile_summary.py")}
        for r in csv.DictReader(open(so)):
            k, _ = kernel_key(r["Name"])
            if k and k.startswith("k_lorentz"):
                out[k + "_mean_ms"] = float(r["AverageNs"]) / 1e6
                out[k + "_calls"] = int(r["Calls"])
        json.dump(out, open(os.path.join(a.out, "operators_kernel_stats.json"), "w"), indent=1)
        lg = os.path.join(a.raw, "ops_time.log")
        if os.path.exists(lg):
            shutil.copy(lg, os.path.join(a.out, "operators_timing_under_rocprof.txt"))

    # ---- VALU instruction counts per wave-row
    pv = os.path.join(a.raw, "pmc_valu", "valu_counter_collection.csv")
    if os.path.exists(pv):
        acc = counters(pv)
        g = _lib_geometry(geo)
        out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -- python3 bench.py --steps 3 --warmup 1",
               **provenance("tools/profile_r03.sh + tools/profile_summary.py"), "launch_geometry": g, "by_stage": {}}
        per_row = []
        for (k, mode), d in sorted(acc.items()):
            if not k.startswith("k_tendency") or mode is None:
                continue
            waves = mean(d.get("SQ_WAVES", [])) or (g["nstrips"] * g["nseg"] * g["threads"] // 64)
            # wave-rows executed: every wave runs `rows` output iterations in total across the segments of its strip
            wave_rows = g["nstrips"] * (g["threads"] // 64) * geo["rows"]
            e = {c: mean(v) for c, v in d.items()}
            e["dispatches"] = len(next(iter(d.values())))
            e["wave_rows"] = wave_rows
            e["valu_insts_per_wave_row"] = e["SQ_INSTS_VALU"] / wave_rows
            f64 = sum(e.get("SQ_INSTS_VALU_" + t + "_F64", 0.0) for t in ("ADD", "MUL", "FMA", "TRANS"))
            e["fp64_insts_per_wave_row"] = f64 / wave_rows
            out["by_stage"][MODE_STAGE.get(mode, "mode" + mode)] = e
            if mode in ("1", "7", "3"):
                per_row.append(e["valu_insts_per_wave_row"])
        out["valu_insts_per_wave_row"] = mean(per_row)
        out["note"] = ("SQ_INSTS_VALU of one launch / (strips x waves per workgroup x rows): all VALU instructions incl. the per-segment prologue, "
                       "per output row of one wave; mean of the three fused stage kernels")
        json.dump(out, open(os.path.join(a.out, "tendency_pmc_valu.json"), "w"), indent=1)

    # ---- wait / busy fractions
    pw = os.path.join(a.raw, "pmc_wait", "wait_counter_collection.csv")
    if os.path.exists(pw):
        acc = counters(pw)
        out = {"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 1",
               **provenance("tools/profile_r03.sh + tools/profile_summary.py"), "by_stage": {}}
        for (k, mode), d in sorted(acc.items()):
            if not k.startswith("k_tendency") or mode is None:
                continue
            e = {c: mean(v) for c, v in d.items()}
            if e.get("SQ_WAVE_CYCLES"):
                e["waitcnt_fraction_of_wave_cycles"] = e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"]
            if e.get("GRBM_GUI_ACTIVE"):
                shader = e["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
                e["valu_busy_fraction_of_shader_cycles"] = 4.0 * e.get("SQ_ACTIVE_INST_VALU", 0) / (shader * 1024)   # quad-cycles, 1024 SIMDs
            out["by_stage"][MODE_STAGE.get(mode, "mode" + mode)] = e
        json.dump(out, open(os.path.join(a.out, "tendency_pmc_sq.json"), "w"), indent=1)

    # ---- HBM traffic
    pf, pwr = (os.path.join(a.raw, d, f) for d, f in (("pmc_fetch", "fetch_counter_collection.csv"), ("pmc_write", "write_counter_collection.csv")))
    if os.path.exists(pf) and os.path.exists(pwr):
        fa, wa = counters(pf), counters(pwr)
        out = {"command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1",
               **provenance("tools/profile_r03.sh + tools/profile_summary.py"),
               "note": "KiB counters; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); per launch", "by_stage": {}}
        tot = []
        for key in sorted(fa):
            k, mode = key
            if not k.startswith("k_tendency") or mode is None or key not in wa:
                continue
            fe, wr = mean(fa[key]["FETCH_SIZE"]) * 1024, mean(wa[key]["WRITE_SIZE"]) * 1024
            e = {"FETCH_SIZE_bytes_raw": fe, "WRITE_SIZE_bytes": wr, "hbm_bytes_corrected": 2 * fe + wr, "dispatches": len(fa[key]["FETCH_SIZE"])}
            out["by_stage"][MODE_STAGE.get(mode, "mode" + mode)] = e
            if mode in ("1", "7", "3"):
                tot.append(e["hbm_bytes_corrected"])
        out["hbm_bytes_per_launch_corrected"] = mean(tot)
        cells = geo["Nx"] * geo["rows"]
        out["algorithmic_bytes_per_launch"] = {"tendency_64B": 64 * cells, "fused_stage_mean_(64+128+96)/3": 288 / 3 * cells}
        json.dump(out, open(os.path.join(a.out, "tendency_pmc_traffic.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(a.out)))


def _lib_geometry(geo):
    """Launch geometry of the 4096^2 vector-invariant launch.  Needs no GPU: the CU count falls back to 256 (MI355X)."""
    try:
        return _lib.tendency_launch_geometry(geo["Nx"], geo["rows"], 1, 8, 0)
    except Exception:
        return {"kind": 2, "threads": 256, "nstrips": 17, "nseg": 45, "rows_per_segment": 92, "wg_per_cu": 3, "halo_lanes": 3, "cus": 256}


if __name__ == "__main__":
    main()
