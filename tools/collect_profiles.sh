#!/bin/bash
# Copies what the evidence session (tools/sessions/r03/s12.sh; round 2: tools/sessions/session15.sh) left under gpurun_out/ into profiles/ (the tracked evidence of a round).
# usage: tools/collect_profiles.sh r02
tag=${1:-r02}; cd "$(dirname "$0")/.."
for f in gpurun_out/prof_$tag/${tag}_cfg*_{bench,bench_profiled,pmc,counters,per_bounce,lane_stats}.json gpurun_out/prof_$tag/${tag}_cfg1_bench_one_stream.json \
         gpurun_out/prof_$tag/${tag}_cfg*_kernel_stats.csv; do [ -s "$f" ] && cp "$f" profiles/; done
for f in gpurun_out/prof_$tag/${tag}_upload_times.txt gpurun_out/prof_$tag/${tag}_multi.json gpurun_out/prof_$tag/${tag}_multi_gather.json gpurun_out/prof_$tag/${tag}_cfg*_lane_stats_leaves*.json gpurun_out/prof_$tag/${tag}_cfg1_per_bounce_counters_leaves*.json; do [ -s "$f" ] && cp "$f" profiles/; done
python3 - "$tag" <<'PY'
import json, sys, glob
tag = sys.argv[1]
for p in sorted(glob.glob(f"profiles/{tag}_cfg*_bench.json")):
    d = json.load(open(p)); r = d["roofline"]
    print(p, d.get("commit"), d["value"], d["ms_per_step"], r["kernel"], round(r["frac"], 4), r.get("traffic"))
PY
