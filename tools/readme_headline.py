#!/usr/bin/env python3
"""tools/readme_headline.py <tag>: rewrites the headline table of profiles/README.md (between the '## Headline (one GPU, `<tag>_cfgN_bench.json`'
line and the '**Throughput did not move' paragraph) from the committed profiles/<tag>_cfgN_bench.json files and the manifest."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
if tag >= "r04":
    # round 4 on: own leaves with the same-run comparison over the reference's leaves (bench.py `leaves_compare`)
    Pj = lambda c: json.load(open(os.path.join(ROOT, "profiles", f"{tag}_cfg{c}_bench.json")))
    fm = lambda v: f"{v:,.0f}".replace(",", " ")
    names = {1: ("1 (BASELINE configs[1], the headline)", "Cornell 996 tris, 1920×1080, 64 spp, 8 bounces, MIS on"),
             2: ("2", "Cornell + 3 textured spheres, 3 876 tris, 1920×1080, 512 spp"),
             3: ("3", "1 M-triangle displaced grid, 1920×1080, 64 spp"),
             4: ("4", "Cornell, 3840×2160, 256 spp, depth of field, whole frame on ONE GPU"),
             0: ("0", "Cornell, 256×256, 16 spp, 4 bounces, MIS off")}
    var = {102: "exact nodes, 16-bit references, 2 workgroups per CU", 112: "quantised nodes, 16-bit references, 2 workgroups per CU, 8 spilling entries", 72: "quantised nodes in LDS, 2 workgroups per CU", 71: "quantised nodes in LDS, 1 workgroup per CU, spilling stacks",
           81: "quantised nodes from memory, top of the tree in LDS", 41: "exact nodes + triangles in LDS, 1 workgroup per CU", 91: "exact nodes from memory"}
    out = ["## Headline (one GPU; all from one box and one session)", "",
           "| config | scene, frame, spp | Msamples/s, own leaves (default) | same run, reference leaves (`leaves_compare`) | gain | device time per step | vector-ALU issue at nominal clock | lanes active per VALU instruction: extend / shadow / shade | extend / shadow variant | CPU oracle (threads) |",
           "|---|---|---|---|---|---|---|---|---|---|"]
    for c in (1, 2, 3, 4, 0):
        d = Pj(c); lc = d.get("leaves_compare") or {}; v = (d["roofline"].get("valu_issue") or {}); lu = v.get("lane_utilisation") or {}
        t = f"{d['gpu_ms_rank0']:.1f} ms" + (f" ({d['steps']} steps)" if d["steps"] > 1 else "")
        gain = f"{100 * (d['value'] / lc['value'] - 1):+.1f} %" if lc.get("value") else "—"
        ev, sv = d["config"]["extend_variant"], d["config"]["shadow_variant"]
        out.append(f"| {names[c][0]} | {names[c][1]} | **{fm(d['value'])}** | {fm(lc.get('value', 0))} | **{gain}** | {t} | "
                   + (f"{100 * v['frac']:.0f} %" if v.get("frac") else "—") + " | "
                   + (f"{lu['extend']:.2f} / {lu['shadow']:.2f} / {lu['shade']:.2f}" if lu else "—")
                   + f" | {var.get(ev, ev)}" + ("" if sv == ev or not d['shadow_traced_rank0'] else f" / {var.get(sv, sv)}")
                   + f" | {d['cpu_baseline']['value']:.1f} ({d['cpu_baseline']['cores']}) |")
    path = os.path.join(ROOT, "profiles", "README.md")
    s = open(path).read()
    i = s.index("## Headline (one GPU; all from one box and one session)"); j = s.index("CPU oracle on the same workloads")
    s = s[:i] + "\n".join(out) + "\n\n" + s[j:]
    open(path, "w").write(s)
    print("\n".join(out))
    sys.exit(0)
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
