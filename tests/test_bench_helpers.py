"""bench.py's bookkeeping that needs no GPU: the byte model of DESIGN.md §5, the committed counter files it reads
(profiles/r02_cfgN_pmc.json), BASELINE.json's configurations and the committed bench lines' contract fields."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_byte_model_matches_design():
    b = _bench()
    assert b.EXTEND_BYTES_PER_RAY == 4 + 32 + 8
    # 44 + 92 + 8 (+ 2 x 44 + 2 x 12 with MIS) + (44 + 44 - 20) / mean path length; radiance at 16-byte stride: 32 and 48 + 48
    assert abs(b.pipeline_bytes_per_segment(1, 2.0) - (256 + 34)) < 1e-9
    assert abs(b.pipeline_bytes_per_segment(0, 4.0) - (144 + 17)) < 1e-9
    assert abs(b.pipeline_bytes_per_segment(1, 2.0, 16) - (264 + 38)) < 1e-9
    assert b.shadow_bytes_per_ray() == 4 + 44 + 24 and b.shadow_bytes_per_ray(16) == 4 + 44 + 32


def test_shade_byte_model():
    b = _bench()
    # queue 4 + hit 8 + O,D,C 40 read, 40 written, ballots 1/4, + 44 per emitted shadow record
    assert abs(b.shade_bytes_per_segment(1, 0.5, 0.0) - (92.25 + 22)) < 1e-9
    assert abs(b.shade_bytes_per_segment(0, 0.5, 1.0) - (92.25 - 12)) < 1e-9       # bounce 0 reads no queue, no stored throughput


def test_configs_are_baseline_json():
    """--config N = BASELINE.json configs[N]: resolution, spp, bounces, MIS, depth of field, and which scaling the N>1 run is."""
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert sorted(b.CONFIGS) == list(range(len(base)))
    c = b.CONFIGS
    assert (c[0]["width"], c[0]["height"], c[0]["spp"], c[0]["bounces"], c[0]["mis"]) == (256, 256, 16, 4, 0)
    assert (c[1]["scene"], c[1]["width"], c[1]["height"], c[1]["spp"], c[1]["bounces"], c[1]["mis"]) == ("cornell", 1920, 1080, 64, 8, 1)
    assert (c[2]["scene"], c[2]["spp"]) == ("cornell_spheres", 512) and (c[3]["scene"], c[3]["spp"]) == ("grid_1m", 64)
    assert (c[4]["width"], c[4]["height"], c[4]["spp"], c[4]["aperture"], c[4]["focus"], c[4]["scaling"]) == (3840, 2160, 256, 0.05, 2.8, "strong")
    assert all(c[i]["scaling"] == "weak" for i in range(4))
    for i in range(5):
        assert c[i]["spp"] % c[i]["fps"] == 0


def _committed(kind):
    b = _bench()
    return [(n, b.profile_path(n, kind)) for n in sorted(b.CONFIGS) if os.path.exists(b.profile_path(n, kind))]


def test_committed_counter_files_feed_the_roofline():
    b = _bench()
    have = _committed("pmc")
    assert have, f"no profiles/{b.PROFILE_TAG}_cfgN_pmc.json committed"
    for n, path in have:
        traffic, src = b.pmc_traffic(n, True)
        assert src["file"].endswith(f"{b.PROFILE_TAG}_cfg{n}_pmc.json") and src["code_commit"]
        assert 1e7 < traffic["extend"] < 2e10 and traffic["shade"] > 0                 # bytes per launch
        assert b.pmc_traffic(n, False) == ({}, None)                                   # other workloads: not applicable
    assert b.pmc_traffic(0, True)[0] == {} or os.path.exists(b.profile_path(0, "pmc"))


def test_committed_valu_counters_feed_the_second_roofline():
    """bench.py's roofline.valu_issue replays SQ_ACTIVE_INST_VALU of profiles/<tag>_cfgN_counters.json: a fraction of the
    vector ALUs' issue capacity between 0 and ~1, per kernel and in all; None for a workload without a counter file."""
    b = _bench()
    have = _committed("counters")
    assert have, f"no profiles/{b.PROFILE_TAG}_cfgN_counters.json committed"
    for n, path in have:
        bench_line = json.load(open(b.profile_path(n, "bench")))
        launches = {k: bench_line["roofline"]["kernels"][k]["launches"] for k in ("extend", "shade", "shadow")}
        v = b.valu_issue(n, True, launches, bench_line["gpu_ms_rank0"])
        assert v["bound"] == "valu issue" and 0.3 < v["frac"] < 1.1
        assert abs(sum(v["busy_ms_per_step_at_peak_clock"].values()) - v["busy_ms_total"]) < 0.01
        assert b.valu_issue(n, False, launches, 10.0) is None
    assert b.valu_issue(0, True, {"extend": 4, "shade": 4, "shadow": 0}, 1.0) is None or os.path.exists(b.profile_path(0, "counters"))


def _check_line(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "Msamples/s" and "sample" in c
    assert d["value"] > 100 * c["value"]


def test_committed_bench_lines_keep_the_contract():
    _check_line(json.load(open(os.path.join(ROOT, "profiles", "r01_bench_n1.json"))))
    b = _bench()
    have = _committed("bench")
    assert have, f"no profiles/{b.PROFILE_TAG}_cfgN_bench.json committed"
    seq = os.path.join(ROOT, "profiles", f"{b.PROFILE_TAG}_cfg1_bench_one_stream.json")       # bench.py --overlap 0: the accounting check
    d = json.load(open(seq))
    _check_line(d)
    assert d["shadow_overlapped"] is False and abs(d["kernel_ms_sum_over_gpu_ms"] - 1.0) < 0.05
    for n, path in have:
        d = json.load(open(path))
        _check_line(d)
        assert d["config"]["config_index"] == n and f"configs[{n}]" in d["config"]["workload"]
        assert d["scaling"] == b.CONFIGS[n]["scaling"]
        # the line is self-contained: the library's per-kernel HIP-event times add up to its whole-dispatch time — or to more
        # than it where the shadow kernel runs on its own stream beside the next bounce
        if d["shadow_overlapped"]:
            assert 1.0 <= d["kernel_ms_sum_over_gpu_ms"] < 2.0
        elif d["gpu_ms_rank0"] > 5.0:
            assert abs(d["kernel_ms_sum_over_gpu_ms"] - 1.0) < 0.05
        else:                                                                  # configs[0]: 0.7 ms in all, launch gaps count
            assert 0.8 < d["kernel_ms_sum_over_gpu_ms"] <= 1.01
        for k in ("extend", "shade", "shadow") if d["shadow_traced_rank0"] else ("extend", "shade"):
            e = d["roofline"]["kernels"][k]
            assert e["launches"] > 0 and abs(e["frac"] - e["achieved"] / d["roofline"]["peak"]) < 1e-4
