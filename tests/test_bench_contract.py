"""bench.py against its contract: ONE JSON line with BASELINE.json's metric, the whole-job value,
the roofline of the dominant kernel and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _baseline():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        return json.load(f)


def test_bench_names_the_baseline_metric():
    sys.path.insert(0, ROOT)
    import bench
    b = _baseline()
    assert bench.BASELINE_METRIC == b["metric"]
    # configs[1]: the workload the metric is quoted on
    assert bench.N_PER_GPU == 10 ** 6 and bench.K == 3 and bench.NB_VARS == 100 and bench.SEL == 5000


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1",
                          "--cpu-sample", "20000"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"] == _baseline()["metric"] and d["unit"] == "candidates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # whole-job throughput = candidates of all steps / timed region
    assert abs(d["value"] - 10 ** 6 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert 5e8 < d["value"] < 1e10
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.2 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["flops_per_candidate"] * 10 ** 6 / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-9 * r["achieved"]
    assert r["kernel_ms"] < d["ms_per_step"]
    assert r["traffic"] is None or r["traffic"] >= r["bytes_per_candidate"] * 10 ** 6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "candidates/s" and c["value"] > 1e4 and c["sample"]
