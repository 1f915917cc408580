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


def test_provisional_headline_is_a_complete_contract_line():
    """bench.py prints the headline once its own fields exist, before the side measurements: a process that dies in one of them leaves
    that line as the last line of stdout.  It carries every key of the contract + roofline + cpu_baseline."""
    bench = _bench()
    with open(os.path.join(REPO, "profiles", "r05_bench_default_flags.json")) as f:
        res = json.load(f)
    for side in ("variants", "pipeline", "weak_shard", "shard_shapes", "k_sweep", "side_steps", "projected_strong_scaling"):
        res.pop(side, None)                                 # what exists when the provisional line is printed
    out = json.loads(bench.headline({**res, "provisional": True}))
    assert out["provisional"] is True
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "parity_checked"):
        assert key in out, key
    src = open(os.path.join(REPO, "bench.py")).read()
    assert src.index('"provisional": True') < src.index('progress("variants")') < src.index("emit(res)")


def _fake_child(tmp_path, body):
    path = tmp_path / "child.py"
    path.write_text("import json, os, signal, sys\n" + body)
    return [__import__("sys").executable, str(path)]


def test_supervisor_keeps_the_headline_of_a_child_that_dies_in_a_side_block(tmp_path):
    """bench.py at N = 1 measures in a child process: a fatal signal after the provisional headline leaves that headline as the last line."""
    import io
    bench = _bench()
    line = {"metric": "m", "value": 1.5, "unit": "u", "roofline": {"frac": 0.3}, "cpu_baseline": {"value": 1}, "provisional": True}
    out = io.StringIO()
    rc = bench.supervise(_fake_child(tmp_path, f"print(json.dumps({line!r})); sys.stdout.flush(); print('bench_detail: {{half', end=''); sys.stdout.flush()\n"
                                               "os.kill(os.getpid(), signal.SIGSEGV)\n"), out=out)
    last = json.loads(out.getvalue().splitlines()[-1])
    assert rc == 0 and "provisional" not in last and last["value"] == 1.5 and "side_measurements" in last


def test_supervisor_passes_a_finished_child_through_and_retries_an_early_death(tmp_path):
    import io
    bench = _bench()
    out = io.StringIO()
    final = {"metric": "m", "value": 2.0, "parity_checked": False}
    rc = bench.supervise(_fake_child(tmp_path, f"print(json.dumps({{'metric': 'm', 'value': 1.0, 'provisional': True}}))\nprint('bench_detail: x')\n"
                                               f"print(json.dumps({final!r}))\nsys.exit(1)\n"), out=out)
    assert rc == 1 and json.loads(out.getvalue().splitlines()[-1]) == final          # parity failure: the child's own status
    marker = tmp_path / "second"
    out = io.StringIO()
    rc = bench.supervise(_fake_child(tmp_path, f"p = {str(marker)!r}\nif not os.path.exists(p):\n    open(p, 'w').close(); os.kill(os.getpid(), signal.SIGSEGV)\n"
                                               "print(json.dumps({'metric': 'm', 'value': 3.0}))\n"), out=out)
    assert rc == 0 and json.loads(out.getvalue().splitlines()[-1])["value"] == 3.0
