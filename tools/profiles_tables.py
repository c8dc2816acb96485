#!/usr/bin/env python3
"""Prints the tables of profiles/README.md from the committed files of a round (profiles/<tag>_cfgN_*), so that the
prose never carries a number the files do not.   usage: tools/profiles_tables.py [r02]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def load(name):
    p = os.path.join(ROOT, "profiles", f"{tag}_{name}")
    return json.load(open(p)) if os.path.exists(p) else None


def n(x, d=0):
    s = f"{x:,.{d}f}".replace(",", " ")
    return s


print("### headline\n")
print("| config | Msamples/s | ms per step (device) | segments per path | CPU oracle (threads) | dominant kernel, HBM fraction |")
print("|---|---|---|---|---|---|")
for c in (1, 2, 3, 4, 0):
    d = load(f"cfg{c}_bench.json")
    if not d:
        continue
    r, cb = d["roofline"], d.get("cpu_baseline") or {}
    print(f"| {c} | {n(d['value'])} | {n(d['gpu_ms_rank0'], 1)} | {d['mean_path_length']:.3f} | "
          f"{cb.get('value', 0):.1f} ({cb.get('cores', '-')}) | {r['kernel'].split(' (')[0]} {100 * r['frac']:.1f} % |")

for c in (1, 3, 2, 4):
    d, dp = load(f"cfg{c}_bench.json"), load(f"cfg{c}_bench_profiled.json")
    if not d:
        continue
    print(f"\n### config {c}: kernels (HIP events of the un-profiled run; rocprofv3 mean of {tag}_cfg{c}_kernel_stats.csv)\n")
    stats = {}
    p = os.path.join(ROOT, "profiles", f"{tag}_cfg{c}_kernel_stats.csv")
    if os.path.exists(p):
        for row in csv.DictReader(open(p)):
            nm = row["Name"]
            key = ("shadow" if "ShadowIO" in nm else "extend") if ("k_trace" in nm or "k_own" in nm) else \
                  next((k for k in ("k_shade", "k_raygen", "k_accumulate", "k_scatter2", "k_scatter", "k_tile_sums") if k + "(" in nm), None)
            if key:
                stats[key] = (int(row["Calls"]), float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e3)
    print("| kernel | launches per step | ms per step (events) | µs per launch (events) | rocprofv3: calls, total ms, mean µs | B/unit | units per launch | achieved GB/s | of 8 TB/s | counters: 2·FETCH+WRITE per launch | counter GB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k in ("extend", "shade", "shadow"):
        kk = d["roofline"]["kernels"][k]
        st = stats.get("k_shade" if k == "shade" else k)
        prof = f"{st[0]}, {st[1]:.2f}, {n(st[2])}" if st else "-"
        tr = f"{n(kk['traffic'] / 1e6)} MB ({kk['traffic'] / kk['algorithmic_bytes_per_launch']:.2f} × algorithmic)" if kk.get("traffic") else "-"
        print(f"| {k} | {kk['launches']} | {d['kernel_ms_rank0'][k]:.2f} | {n(1e3 * kk['avg_launch_ms'])} | {prof} | {kk['bytes_per_unit']} | "
              f"{kk['units_per_launch'] / 1e6:.2f} M | {n(kk['achieved'])} | {100 * kk['frac']:.1f} % | {tr} | {n(kk.get('traffic_gbs', 0))} |")
    for k in ("raygen", "compact", "accumulate"):
        st = [stats[s] for s in stats if s in {"raygen": ("k_raygen",), "compact": ("k_scatter2", "k_scatter", "k_tile_sums"), "accumulate": ("k_accumulate",)}[k]]
        prof = ", ".join(f"{a}" for a in [sum(s[0] for s in st), f"{sum(s[1] for s in st):.2f}"]) if st else "-"
        print(f"| {k} | | {d['kernel_ms_rank0'][k]:.2f} | | {prof} | | | | | | |")
    r = d["roofline"]
    print(f"\nwhole pipeline: {r['pipeline_bytes_per_segment']} B per segment × {n(d['value'])} M segments/s = {n(r['pipeline_achieved'])} GB/s = "
          f"{100 * r['pipeline_frac']:.1f} % of HBM peak; device time {d['gpu_ms_rank0']:.2f} ms, kernel times add up to "
          f"{d['kernel_ms_sum_over_gpu_ms']:.3f} × that (two streams)."
          + (f" Profiled run's own line: {n(dp['value'])} Msamples/s." if dp else ""))
    v = r.get("valu_issue")
    if v:
        print(f"\nvector-ALU issue ({v['source']['file']}, SQ_ACTIVE_INST_VALU): busy " + ", ".join(f"{k} {x:.2f} ms" for k, x in v["busy_ms_per_step_at_peak_clock"].items())
              + f" = {v['busy_ms_total']:.2f} ms of {v['device_ms']:.2f} ms device time at {v['clock_ghz']} GHz = **{100 * v['frac']:.1f} %** of the issue capacity of {v['simds']} SIMDs")
    ls = load(f"cfg{c}_lane_stats.json")
    if ls:
        print("\n| lane statistics (diagnostic build) | box-pair steps per ray | lanes in a box step | leaves per ray | triangles per ray | lanes in a triangle iteration | lanes holding a ray at a vote | wave steps per 64 rays: box / leaf / triangle |")
        print("|---|---|---|---|---|---|---|---|")
        for k in ("extend", "shadow"):
            e = ls[k]; w = e["wave_steps_per_64_rays"]
            print(f"| {k} | {e['box_pair_steps_per_ray']:.2f} | {e['box_step_lane_util']:.2f} | {e['leaves_per_ray']:.2f} | {e['triangles_per_ray']:.2f} | {e['triangle_lane_util']:.2f} | {e['lanes_holding_a_ray_at_vote']:.2f} | {w['box']} / {w['leaf']} / {w['tri']} |")
    pb = load(f"cfg{c}_per_bounce.json")
    if pb:
        seg = d["segments_by_bounce_rank0"]
        print("\n| bounce | " + " | ".join(str(i) for i in range(len(seg))) + " |")
        print("|---|" + "---|" * len(seg))
        print("| M rays | " + " | ".join(f"{s / 1e6 / d['steps']:.1f}" for s in seg) + " |")
        for k in ("extend", "shade", "shadow"):
            print(f"| {k} µs | " + " | ".join(n(v) for v in pb["per_bounce"][k]) + " |")

one = load("cfg1_bench_one_stream.json")
if one:
    print(f"\n### config 1 on one stream (--overlap 0): {n(one['value'])} Msamples/s, device {one['gpu_ms_rank0']:.2f} ms, kernels "
          + ", ".join(f"{k} {v:.2f}" for k, v in one["kernel_ms_rank0"].items())
          + f" ms (sum / device = {one['kernel_ms_sum_over_gpu_ms']:.3f}); HBM fractions alone: "
          + ", ".join(f"{k} {100 * v['frac']:.1f} %" for k, v in one["roofline"]["kernels"].items()))
