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


def test_in_step_roofline_keeps_frac_on_the_isolated_launch_and_adds_the_in_step_view():
    import bench
    roof = {"bound": "mfma", "achieved": 80.0, "peak": 157.3, "unit": "TFLOP/s", "frac": round(80.0 / 157.3, 4),
            "effective_tflops": 180.0, "ms_per_launch": 2.5}
    out = bench.in_step_roofline(dict(roof), _FakeTimer(2.0, 5))
    assert out["ms_per_launch"] == 2.5 and out["frac"] == roof["frac"] and out["achieved"] == 80.0
    assert out["ms_per_launch_in_step"] == 2.0 and out["launches_timed_in_step"] == 5
    assert abs(out["achieved_in_step"] - 100.0) < 1e-6 and abs(out["frac_in_step"] - round(100.0 / 157.3, 4)) < 1e-9
    # no launches timed (a shape that does not occur in the step): the object is unchanged
    same = bench.in_step_roofline(dict(roof), _FakeTimer(2.0, 0))
    assert same == roof


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
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "ms_per_launch",
              "ms_per_launch_in_step", "frac_in_step"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches_timed_in_step"] == 5  # one forward launch of the roofline layer in each of the 5 eager steps
    assert "hipGraph" in d["config"]["step_launch"]
    s16 = d["secondary"]["bf16"]            # the default run appends the bf16 steps (never in `value`)
    assert s16["dtype"] == "bf16" and s16["steps"] == 20 and s16["value"] > d["value"]
    b = s16["roofline"]
    assert b["bound"] == "hbm" and abs(b["frac"] - b["algorithmic_GB_per_launch"] / (b["block_ms"] * 1e-3) / 8000.0) < 2e-3
    assert b["conv_only"]["ms_per_launch"] <= b["block_ms"]


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
    assert r["launches_timed_in_step"] == 10  # the two 32 -> 32 forward launches at the patch, in 5 eager steps
    # the BLOCK (conv + every InstanceNorm launch it needs) against the block's byte model; the conv alone beside it
    assert abs(r["achieved"] - r["algorithmic_GB_per_launch"] / (r["block_ms"] * 1e-3)) < 0.01 * r["achieved"]
    assert r["conv_only"]["ms_per_launch"] < r["block_ms"] and r["conv_only"]["frac"] > r["frac"]
