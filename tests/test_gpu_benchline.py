"""The one JSON line `bench.py` prints at N = 1 (the driver's contract): keys, types and the line's own consistency, on a small grid so
that the test takes seconds.  (`--gpus N` is covered by the one-GPU rehearsal in test_gpu_hostgroup.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--size", "512", "--substeps", "24", *extra],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.strip()]
    return json.loads(lines[-1]), lines


def test_single_gpu_line_keeps_the_contract():
    d, lines = _line()
    assert lines[-1].lstrip().startswith("{") and sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1      # ONE JSON line, the last thing printed
    assert d["metric"] == "EVP sub-cycle cell-updates/s" and d["unit"] == "cell-updates/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["scaling"] in ("weak", "strong")
    assert isinstance(d["config"], dict) and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value = cells x sub-steps x steps / time
    assert abs(d["value"] - 512 * 512 * 24 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the dominant kernel's time comes from the timed launches: launches x average <= the step
    assert r["avg_launch_source"].startswith("HIP events") or "timed" in r["avg_launch_source"], r["avg_launch_source"]
    assert r["launches_x_avg_ms"] <= d["ms_per_step"] * 1.0001
    assert abs(r["achieved"] - r["kernel_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert set(r["clock"]) >= {"sclk_mhz", "power_w", "samples", "source"}      # (values may be None where sysfs is not readable)
    assert 0.0 < r["unfused"]["frac"] < 1.0                                       # the SURVEY 8(d)-literal three-kernel figure of the same run
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "cell-updates/s" and isinstance(c["sample"], str)
    assert d["value"] > 20 * c["value"]                                           # a GPU path, not a fallback
    assert all(v["finite"] for v in d["result_check"].values() if isinstance(v, dict) and "finite" in v)


def test_line_without_the_optional_legs():
    d, _ = _line("--no-cpu-baseline", "--no-full-step", "--no-unfused")
    assert "roofline" in d and d["value"] > 0
    assert not d.get("cpu_baseline") and "unfused" not in d["roofline"]
