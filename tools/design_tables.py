#!/usr/bin/env python3
"""Markdown rows of DESIGN.md §4's workload table from the round's committed bench lines (profiles/rNN_*).  usage: design_tables.py r06 [r05]"""
import json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = sys.argv[2] if len(sys.argv) > 2 else "r05"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
def load(name):
    try:
        t = open(os.path.join(root, name)).read().strip().splitlines()[-1]
        return json.loads(t)
    except Exception:
        return None
line = load(f"{R}_default_bench_line.json"); prev = load(f"{P}_default_bench_line.json")
def prev_value(w):
    if prev is None: return None
    if prev["config"]["workload"] == w: return prev["value"]
    for c in prev.get("configs", []):
        if c["workload"] == w: return c["value"]
    b = load(f"{P}_{w}_bench.json")
    return b["value"] if b else None
def row(title, w, e):
    k = e["kernel_ms_per_step"]; r = e.get("roofline") or {}
    sw = e.get("roofline_sweep_hbm")
    pv = prev_value(w)
    note = f"`{(r.get('kernel') or '?').split(' ')[0]}` {r.get('frac'):.2f}" if r.get("frac") is not None else f"`{(r.get('kernel') or '?')}`: no flop model"
    if sw: note += f"; sweep {sw['achieved']:.0f} GB/s = {sw['frac']:.3f} of HBM peak"
    print(f"| {title} | **{e['value'] / 1e3:.1f} k** ({pv / 1e3:.1f} k) | {e['ms_per_step']:.2f} | {k['k_solve']:.2f} | {k['k_count']:.2f} | {k['k_bound']:.2f} | {k['k_score']:.2f} | {k['k_lo']:.2f} | {k['k_final']:.2f} | {note} |" if pv else
          f"| {title} | **{e['value'] / 1e3:.1f} k** | {e['ms_per_step']:.2f} | {k['k_solve']:.2f} | {k['k_count']:.2f} | {k['k_bound']:.2f} | {k['k_score']:.2f} | {k['k_lo']:.2f} | {k['k_final']:.2f} | {note} |")
print("| workload | pairs/s (r05) | ms / step | `k_solve` | `k_count` | `k_bound` | `k_score` | `k_lo` | `k_final` | roofline of the dominant kernel; scoring sweep vs HBM |")
print("|---|---|---|---|---|---|---|---|---|---|")
row("calibrated P3P, N = 2000, 50 % outliers (configs[1])", "calib_p3p_n2000_i10k", {**line, "roofline": line["roofline"]})
names = {"shared_n2000_i10k": "shared focal, N = 2000 (configs[2])", "varying_n5000_i10k": "varying focal, N = 5000, shift flag (configs[3])",
         "calib_shift_n2000_i10k": "calibrated + shift solver, N = 2000", "calib_p3p_n2000_i10k_clean": "calibrated P3P, 0 % outliers"}
for c in line["configs"]:
    if c["workload"] in names: row(names[c["workload"]], c["workload"], c)
for w, t in (("relpose_5pt_n2000_i10k", "5-point `estimate_relative_pose`"), ("fundamental_7pt_n2000_i10k", "7-point `estimate_fundamental`"),
             ("shared_6pt_n2000_i10k", "6-point `estimate_shared_focal_relative_pose`, B = 256")):
    b = load(f"{R}_{w}_bench.json")
    if b: row(t, w, b)
for c in line["configs"]:
    if c["workload"].startswith("c5_"):
        pv = load(f"{P}_c5_bench_12500_pairs_1gpu.json")
        k = c["kernel_ms_per_step"]
        print(f"| headline generator, **12 500 pairs** in one call (one rank's share of configs[4]; on the driver's line) | **{c['value'] / 1e3:.1f} k** ({pv['value'] / 1e3:.1f} k) | {c['ms_per_step']:.2f} | {k['k_solve']:.2f} | {k['k_count']:.2f} | {k['k_bound']:.2f} | {k['k_score']:.2f} | {k['k_lo']:.2f} | {k['k_final']:.2f} | `k_lo` {c['roofline']['frac']:.2f} of the fp64 peak: at this size the LM's tails amortise |")
print()
print("headline:", f"{line['value']:.0f} pairs/s, {line['ms_per_step']:.3f} ms; pipelined {line['pipelined']['value']:.0f}; host {line['host_buffers']['value']:.0f} ratio {line['host_buffers']['ratio_to_resident']:.3f}; cpu {line['cpu_baseline']['value']:.2f} x{line['speedup_vs_cpu_1thread']:.0f}; "
      f"vs ref binary est x{line.get('speedup_vs_reference_binary_1thread_est', 0):.0f}; k_lo frac {line['roofline']['frac']:.3f} traffic {line['roofline']['traffic']}; count frac {line['roofline_count']['frac']:.3f} mfma busy {line['roofline_count'].get('mfma_busy_frac_pmc')}; "
      f"sweep hbm {line['roofline_sweep_hbm']['achieved']:.0f} GB/s {line['roofline_sweep_hbm']['frac']:.3f} bytes {line['roofline_sweep_hbm']['bytes_per_step']:.3e}; alg frac {line['roofline_hbm_algorithmic']['frac']:.2f}; "
      f"evals mfma {line['work']['evals_mfma_count_per_step']:.3e} bound {line['work']['evals_fp32_bound_per_step']:.3e} fp64 {line['work']['evals_fp64_sweep_per_step']:.3e} alg {line['work']['evals_algorithmic_per_step']:.3e}")
lat = line.get("latency")
if lat:
    print("latency:", {k: (round(v["median_ms"], 2), round(v.get("cpu_port_ms_per_pair", 0), 1)) for k, v in lat["single_pair"].items()}, {k: round(v["median_ms_per_call"], 2) for k, v in lat["batch_entry_point_headline_shape"].items()})
