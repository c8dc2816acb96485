#!/usr/bin/env python3
"""tools/readme_headline.py <tag>: rewrites the headline table of profiles/README.md (between the '## Headline (one GPU, `<tag>_cfgN_bench.json`'
line and the '**Throughput did not move' paragraph) from the committed profiles/<tag>_cfgN_bench.json files and the manifest."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = lambda c: json.load(open(os.path.join(ROOT, "profiles", f"{tag}_cfg{c}_bench.json")))
man = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_manifest.json")))
fmt = lambda v: f"{v:,.0f}".replace(",", " ")
rows = {1: ("1 (BASELINE configs[1], the headline)", "Cornell 996 tris, 1920×1080, 64 spp, 8 bounces, MIS on", "(the same pipeline over the round's boxes: 9 370 – 10 250)", "9 872"),
        2: ("2", "Cornell + 3 textured spheres, 3 876 tris, 1920×1080, 512 spp", "(6 280 – 6 650)", "6 439"),
        3: ("3", "1 M-triangle displaced grid, 1920×1080, 64 spp", "(4 790 – 5 080)", "4 996"),
        4: ("4", "Cornell, 3840×2160, 256 spp, depth of field, whole frame on ONE GPU", "(9 365 – 9 960)", "9 966")}
out = [f"## Headline (one GPU, `{tag}_cfgN_bench.json`; all from one box and one session, kernel sources of commit `{man['commit']}`)", "",
       "| config | scene, frame, spp | Msamples/s | device time per step | round 2 (its box) | CPU oracle, 16 threads | vector-ALU issue at nominal clock | lanes active per VALU instruction: extend / shadow / shade |",
       "|---|---|---|---|---|---|---|---|"]
for c, (name, scene, rng, r2) in rows.items():
    d = P(c); v = d["roofline"]["valu_issue"]; lu = v["lane_utilisation"]
    steps = d["steps"]
    t = f"{d['gpu_ms_rank0']:.1f} ms" + (f" ({steps} steps)" if steps > 1 else "")
    out.append(f"| {name} | {scene} | **{fmt(d['value'])}** {rng} | {t} | {r2} | {d['cpu_baseline']['value']:.1f} | {100 * v['frac']:.0f} % | "
               f"{lu['extend']:.2f} / {lu['shadow']:.2f} / {lu['shade']:.2f} |")
d = P(0)
out.append(f"| 0 | Cornell, 256×256, 16 spp, 4 bounces, MIS off | {fmt(d['value'])} | {d['gpu_ms_rank0']:.2f} ms | 4 483 | {d['cpu_baseline']['value']:.1f} | — | — |")
path = os.path.join(ROOT, "profiles", "README.md")
s = open(path).read()
i = s.index(f"## Headline (one GPU, `{tag}_cfgN_bench.json`"); j = s.index("**Throughput did not move")
s = s[:i] + "\n".join(out) + "\n\n" + s[j:]
s = re.sub(r"\(kernel sources of commit `[0-9a-f]{7}`; run as `tools/sessions/r03/s12\.sh [0-9a-f]{7}`\)",
           f"(kernel sources of commit `{man['commit']}`; run as `tools/sessions/r03/s12.sh {man['commit']}`)", s)
open(path, "w").write(s)
d = P(1); r = d["roofline"]
print({k: (round(v["frac"], 3), round(v.get("traffic", 0) / v["algorithmic_bytes_per_launch"], 2), v["avg_launch_ms"]) for k, v in r["kernels"].items()},
      "pipeline", round(r["pipeline_frac"], 3), round(d["value"] * 285.2 / 1e3), "valu", r["valu_issue"]["busy_ms_total"], d["gpu_ms_rank0"], r["valu_issue"]["frac"])
