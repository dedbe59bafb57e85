"""bench.py's ONE stdout line: built through the same function the bench uses, from a result of the
shape and size a real run produces (profiles/r05_bench.json: 22 KB, the line round 5's driver could not parse), it must
stay below 4 KB, be strict JSON, and carry the headline, `roofline` and `cpu_baseline` keys the contract names."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

HEADLINE = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _fake(n_gpus=1):
    """A full result: round 5's real 22 KB line where it exists, with this round's roofline keys laid over it."""
    out = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    out["n_gpus"] = n_gpus
    out["roofline"] = {"bound": "onchip-model", "kernel": "csm_bnb_kernel<2, true, true, true> + csm_bnb_cand_kernel<2>",
                       "avg_launch_ms": 6.17123456789, "launches": 20, "achieved": 1620000.123456, "peak": 5.4e6,
                       "unit": "pairs/s through the matcher's kernels; peak = an ideally-pruned matcher (ideal_ms)",
                       "frac": 0.3, "ideal_ms": 1.85, "ideal_ms_bounds": 0.27, "ideal_ms_candidates": 1.58, "traffic": 3.5e9,
                       "stale": False, "other_ceilings": {"bounds_valu_frac": 0.41, "candidates_ta_busy_frac": 0.61, "nested": {"x": 1}},
                       "matcher": {"big": ["x" * 100] * 50}, "profile": {"stale": False}, "note": "n" * 2000}
    out["secondary"]["resid_lidar"]["roofline"].update(kernel="resid_lidar_kernel<0, true>", avg_launch_ms=0.258)
    out["secondary"]["broken_leg_error"] = "RuntimeError('x')"
    if n_gpus > 1:
        out.update(bench.scaling_fields(out["value"], {"one_gpu_same_workload_pairs_per_s": 1.8e6, "one_gpu_same_workload_ms_per_step": 554.0,
                                                       "one_gpu_records_equal_sharded_table": True, "one_gpu_steps_timed": 1}, n_gpus, n_gpus))
        out["config"]["rccl_world_size"] = n_gpus
        out["config"]["per_rank"] = [dict(out["config"]["per_rank"][0]) for _ in range(n_gpus)]
    return out


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_compact_line_is_small_strict_json_with_the_contract_keys(n_gpus):
    out = _fake(n_gpus)
    assert len(json.dumps(out)) > 15000, "the fake must be as large as a real run's result"
    line = bench.compact_line(out, "bench_details.json")
    assert "\n" not in line and len(line) < 4096

    def no_constants(c):
        raise ValueError("NaN / Infinity in the line: " + c)
    d = json.loads(line, parse_constant=no_constants)
    assert json.dumps(d, allow_nan=False)
    for k in HEADLINE:
        assert k in d, k
    for k in ROOFLINE + ("kernel", "avg_launch_ms", "ideal_ms", "stale"):
        assert k in d["roofline"], k
    for k in CPU + ("gpu_matches_oracle_on_sample",):
        assert k in d["cpu_baseline"], k
    assert d["roofline"]["frac"] == 0.3 and "nested" not in d["roofline"]["other_ceilings"]
    assert "matcher" not in d["roofline"] and "note" not in d["roofline"] and "secondary" not in d
    assert d["config"]["workload"] and "per_rank" not in d["config"] and "model" not in d["config"]
    assert d["roofline_resid_lidar"]["bound"] == "hbm" and 0 < d["roofline_resid_lidar"]["frac"] < 1
    assert d["secondary_errors"] == ["broken_leg_error"]
    assert d["details"] == "bench_details.json"
    assert d["value"] == pytest.approx(out["value"], rel=1e-5) and d["n_gpus"] == n_gpus
    assert set(d["parity_vs_f64"]) >= {"pairs", "index_agreement", "max_rel_score"}
    if n_gpus > 1:
        for k in ("one_gpu_same_workload_pairs_per_s", "speedup_vs_one_gpu", "rccl_world_size"):
            assert k in d, k
        assert d["config"]["rccl_world_size"] == n_gpus
        assert "max_over_mean_predicted_cost" in d["config"]["shard_balance"]


def test_compact_line_survives_nan_and_long_strings():
    out = _fake()
    out["roofline"]["frac"] = float("nan")
    out["kernels_ms_per_step"]["grid_build"] = float("inf")
    out["config"]["workload"] = "w" * 5000
    out["cpu_baseline"]["sample"] = "s" * 5000
    line = bench.compact_line(out, None)
    assert len(line) < 4096
    d = json.loads(line)
    assert d["roofline"]["frac"] is None and d["kernels_ms_per_step"]["grid_build"] is None
    assert len(d["config"]["workload"]) <= 400


def test_emit_prints_the_line_last_and_writes_the_details(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out = _fake()
    line = bench.emit(out)
    printed = capsys.readouterr().out.strip().splitlines()
    assert printed[-1] == line and json.loads(printed[-1])["details"] == "bench_details.json"
    full = json.load(open(tmp_path / "bench_details.json"))
    assert "secondary" in full and full["secondary"]["host_buffer_api"]
