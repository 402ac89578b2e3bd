"""bench.py's output contract (GPU): ONE JSON line with the keys the driver parses, sane values, and the roofline /
cpu_baseline objects -- run as the driver runs it (a subprocess), on a reduced batch so it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--envs-per-gpu", "262144", "--no-secondary"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("\n") == 1 and out.stdout.endswith("\n")     # nothing but the line (native chatter -> stderr)
    d = json.loads(out.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
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
                "frac_survey_model", "regime"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.05 < r["frac"] < 1.0
    # achieved = the bytes the timed kernel really moves (SURVEY's 249 B minus the 16 B action read it does not do) per
    # step / the device time of one step; SURVEY's own figure beside it
    assert r["bytes_per_env_step"] == 233 and r["bytes_per_env_step_survey_model"] == 249
    assert r["achieved"] == pytest.approx(233 * 262144 / (r["avg_kernel_us"] * 1e-6) / 1e9, rel=1e-6)
    assert r["frac_survey_model"] == pytest.approx(r["frac"] * 249 / 233, rel=1e-6)
    assert r["avg_kernel_us"] * 1e-3 <= d["ms_per_step"]                 # the step's device time fits inside the wall-clock step
    assert r["launches_per_step"] == 2 and r["envs_per_launch"] == 131072 and "2 chains of 131072 envs" in r["kernel"]
    assert r["bytes_per_step"] == 233 * 262144 and r["bytes_per_launch"] * r["launches_per_step"] == r["bytes_per_step"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "numpy_multiprocess_value", "numpy_multiprocess_cores"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e4
    assert c["numpy_multiprocess_value"] is not None and c["numpy_multiprocess_value"] > 1e3, c["numpy_multiprocess_sample"]
