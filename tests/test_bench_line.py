"""bench.py's final stdout line is what the driver parses: it must stay one short JSON object (a 20 KB line was not parsed)."""
import importlib.util
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_headline_of_a_recorded_result_is_short_json():
    bench = _bench()
    with open(os.path.join(REPO, "profiles", "r05_bench_default_flags.json")) as f:
        res = json.load(f)                                  # a full result of round 5: 20 KB, every side block present
    assert len(json.dumps(res)) > 16000
    line = bench.headline(res)
    assert len(line) < bench.HEADLINE_LIMIT <= 4096 and "\n" not in line
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "parity_checked", "projected_strong_scaling"):
        assert key in out, key
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(out["roofline"])
    assert out["roofline"]["bound"] in ("hbm", "mfma")
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(out["cpu_baseline"])
    assert "workload" in out["config"] and "model" not in out["config"]
    assert abs(out["value"] - res["value"]) <= 1e-5 * res["value"]
    assert abs(out["roofline"]["frac"] - res["roofline"]["frac"]) <= 1e-5


def test_headline_survives_padded_side_blocks():
    bench = _bench()
    with open(os.path.join(REPO, "profiles", "r05_bench_default_flags.json")) as f:
        res = json.load(f)
    res["config"]["workload"] = "w" * 5000
    res["cpu_baseline"]["sample"] = "s" * 5000
    res["roofline"]["kernel"] = "eval_slide_kernel (" + "x" * 5000
    res["pipeline"]["rows_7"] = {"run_ms": 1.0, "construct_ms": 2.0, "tsv_equal_oracle": True, "junk": "j" * 5000}
    line = bench.headline(res)
    assert len(line) < 4096
    json.loads(line)
