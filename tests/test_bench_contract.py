"""bench.py's output contract (GPU): ONE JSON line with the keys the driver parses, sane values, and the roofline /
cpu_baseline objects -- run as the driver runs it (a subprocess), on a reduced batch so it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fractions(o, path=""):
    """(path, value) of every float in the line whose key names a fraction."""
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _fractions(v, path + "/" + k)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _fractions(v, f"{path}[{i}]")
    elif isinstance(o, float) and "frac" in path.rsplit("/", 1)[-1]:
        yield path, o


@pytest.mark.gpu
def test_bench_line_with_its_secondary_legs_holds_no_fraction_above_one():
    """The whole line as the driver gets it (secondary configurations included; the CPU baseline skipped for time): every
    key that names a fraction is one, also where the launches move less than SURVEY's model bytes (7-DoF table, k steps per
    launch), where the model-bytes fraction is withheld instead of exceeding 1."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout)
    seen = list(_fractions(d))
    assert len(seen) > 10 and all(0.0 <= v <= 1.0 for _, v in seen), [p_ for p_ in seen if not 0.0 <= p_[1] <= 1.0]
    cfg7 = [v for k, v in d["secondary"]["other_configs"].items() if "7-DoF" in k][0]
    assert cfg7["survey_model_gbs"] > 0 and (cfg7["frac_survey_model"] is None or cfg7["frac_survey_model"] <= 1.0)


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--envs-per-gpu", "262144", "--no-secondary"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("\n") == 1 and out.stdout.endswith("\n")     # nothing but the line (native chatter -> stderr)
    d = json.loads(out.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_device_timeline", "device_ms_per_step"):
        assert key in d, key
    assert d["metric"] == "env-steps/sec (whole node), 1M parallel 4-DoF arms, random actions"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["gathers_in_timed_region"] >= 1
    assert d["config"]["repeats"] >= 5 and d["config"]["prewarm_launches"] > 0
    # value = envs x steps / (ms_per_step x steps)
    assert d["value"] == pytest.approx(262144 / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_env_step", "avg_kernel_us",
                "frac_survey_model", "regime", "achievable_gbs", "achievable_gbs_same_size", "steps_per_kernel_launch"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.05 < r["frac"] < 1.0
    # 262 144 arms: mt_rollout runs five steps per launch there (one chain), so a step moves its outputs (12 K + 17 B) and a
    # fifth of the state traffic (8 D + 12 K + 16 B per launch): achieved = those bytes / the device time of one step;
    # SURVEY's own 249-byte figure beside it
    assert r["steps_per_kernel_launch"] == 5.0 and r["launches_per_step"] == 1 and r["envs_per_launch"] == 262144
    assert d["config"]["dispatch"]["rollout"] == {"form": "multi_step", "steps_per_launch": 5, "graph": False, "lanes_per_env": 1, "chains": 1,
                                                "absorbs_reset": True, "writes_snapshot": True}
    assert "5 steps per launch" in r["kernel"]
    moved = 101 + 132 / 5
    assert r["bytes_per_env_step"] == pytest.approx(moved) and r["bytes_per_env_step_survey_model"] == 249
    assert r["achieved"] == pytest.approx(moved * 262144 / (r["avg_kernel_us"] * 1e-6) / 1e9, rel=1e-6)
    model_frac = r["frac"] * 249 / moved        # SURVEY's model bytes over the same time: reported only while it is a fraction
    assert r["frac_survey_model"] == (pytest.approx(model_frac, rel=1e-6) if model_frac <= 1.0 else None)
    assert r["achieved_survey_model"] == pytest.approx(model_frac * r["peak"], rel=1e-6)
    assert r["avg_kernel_us"] * 1e-3 <= d["ms_per_step"]                 # the step's device time fits inside the wall-clock step
    assert r["bytes_per_step"] == pytest.approx(moved * 262144)
    assert r["bytes_per_launch"] == pytest.approx(moved * 5 * 262144)
    # the region on the device's clock: between the step launches alone and the fenced wall clock
    assert r["avg_kernel_us"] * 1e-3 <= d["device_ms_per_step"] <= d["ms_per_step"]
    assert d["value_device_timeline"] == pytest.approx(262144 / (d["device_ms_per_step"] * 1e-3), rel=1e-6)
    assert r["achievable_gbs"] > 2000 and r["achievable_gbs_same_size"] > 2000
    # no figure called a fraction exceeds 1 anywhere in the line
    seen = list(_fractions(d))
    assert seen and all(0.0 <= v <= 1.0 for _, v in seen), [p_ for p_ in seen if not 0.0 <= p_[1] <= 1.0]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "numpy_multiprocess_value", "numpy_multiprocess_cores"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e4
    assert c["numpy_multiprocess_value"] is not None and c["numpy_multiprocess_value"] > 1e3, c["numpy_multiprocess_sample"]
