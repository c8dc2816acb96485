"""bench.py's bookkeeping that needs no GPU: the byte model of DESIGN.md §5, the committed counter files it reads
(profiles/r01_bench_n1_{pmc,valu}.json) and the committed bench line's contract fields."""
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
    assert b.EXTEND_BYTES_PER_RAY == 4 + 32 + 16
    # 52 + 116 + 8 (+128 with MIS) + (48 + 48 - 28) / mean path length
    assert abs(b.pipeline_bytes_per_segment(1, 2.0) - (304 + 34)) < 1e-9
    assert abs(b.pipeline_bytes_per_segment(0, 4.0) - (176 + 17)) < 1e-9


def test_committed_counter_files_feed_the_roofline():
    b = _bench()
    traffic = b.pmc_traffic(True)
    assert traffic is not None and 0.8e9 < traffic < 3e9                      # bytes per extend launch
    assert b.pmc_traffic(False) is None                                       # other workloads: counters not applicable
    v = b.valu_issue(0.8, True)
    assert v and 3.0 < v["simd_cycles_per_instruction"] < 6.0 and v["wave_instructions_per_launch"] > 1e8
    assert b.valu_issue(0.8, False) is None


def test_committed_bench_line_keeps_the_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 0.01
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "Msamples/s" and "sample" in c
    assert d["value"] > 100 * c["value"]
