"""bench.py's output contract: the JSON line the driver reads (keys, types, the roofline object) and the arithmetic that
turns the in-step launch timings into `achieved` / `frac`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _FakeTimer:
    def __init__(self, ms, n):
        self._ms, self.pairs = ms, [None] * n

    def mean_ms(self):
        return self._ms if self.pairs else None


def test_in_step_roofline_rescales_every_rate_by_the_same_factor():
    import bench
    roof = {"bound": "mfma", "achieved": 80.0, "peak": 157.3, "unit": "TFLOP/s", "frac": round(80.0 / 157.3, 4),
            "effective_tflops": 180.0, "effective_over_peak": round(180.0 / 157.3, 4), "ms_per_launch": 2.5,
            "hbm_view": {"algorithmic_GB": 1.611, "achieved_GBps": 644.4, "frac_of_hbm_peak": 0.0806}}
    out = bench.in_step_roofline(dict(roof, hbm_view=dict(roof["hbm_view"])), _FakeTimer(2.0, 50))
    assert out["ms_per_launch"] == 2.0 and out["ms_per_launch_isolated"] == 2.5 and out["launches_timed_in_step"] == 50
    assert abs(out["achieved"] - 100.0) < 1e-6 and abs(out["frac"] - round(100.0 / 157.3, 4)) < 1e-9
    assert abs(out["effective_tflops"] - 225.0) < 1e-6
    assert abs(out["hbm_view"]["achieved_GBps"] - 805.5) < 0.06
    assert out["frac_isolated"] == roof["frac"]
    # no launches timed (e.g. --no-roofline runs never reach this; a shape that does not occur): the object is unchanged
    same = bench.in_step_roofline(dict(roof), _FakeTimer(2.0, 0))
    assert same["ms_per_launch"] == 2.5 and "ms_per_launch_isolated" not in same


def test_metric_names_follow_config_and_precision():
    import bench
    assert "fp32" in bench.metric_name("cfg2", "fp32", (128, 128, 128))
    assert "bf16" in bench.metric_name("cfg2", "bf16", (128, 128, 128))
    assert "160x160x128" in bench.metric_name("cfg5", "bf16", (160, 160, 128))


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys_and_in_step_roofline():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2
    assert d["unit"] == "samples/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and "workload" in d["config"]
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]  # batch 2 per GPU
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "ms_per_launch", "ms_per_launch_isolated"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches_timed_in_step"] == 3  # one forward launch of the roofline layer per timed step


@pytest.mark.gpu
def test_bench_line_bf16_reports_the_hbm_roofline_of_the_block():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--precision", "bf16", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["dtype"] == "bf16" and "bf16" in d["metric"] and d["steps"] == 3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches_timed_in_step"] == 6  # the two 32 -> 32 forward launches at the patch, per timed step
    assert abs(r["achieved"] - r["algorithmic_GB_per_launch"] / (r["ms_per_launch"] * 1e-3)) < 0.01 * r["achieved"]
