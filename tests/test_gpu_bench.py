"""GPU tier: the bench.py contract -- one JSON line with the keys the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "13", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 13 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["status"] == "Optimal" and d["objective_relerr"] <= 1e-5
    for rf in (d["roofline"], d["sweep_roofline"]):
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in rf, k
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-12 and 0.0 < rf["frac"] < 1.0
    assert d["sweep_roofline"]["frac"] >= 0.35          # the column-blocked sweep (north star: >= 40 % of the HBM peak)


def test_bench_throughput_workload_line():
    """--workload cfg5 (BASELINE.json configs[4]): the same contract keys, instances/s, every instance Optimal"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg5", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "incl_load"):
        assert k in d, k
    assert d["unit"] == "instances/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["status"] == "Optimal"
    assert abs(d["value"] - 512 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["value"] >= 2500.0 and d["max_objective_relerr"] <= 1e-5         # (5 100/s measured with the batch resident)
    assert 0.0 < d["incl_load"]["instances_per_s"] <= d["value"]
