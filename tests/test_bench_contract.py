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
    c2 = bench.CONFIGS["c2"]
    assert c2["per_gpu"] == 10 ** 6 and c2["k"] == 3 and c2["nb_vars"] == 100 and bench.SEL == 5000 and c2["scaling"] == "weak"
    # configs[3]: 1e8 candidates in total over the ranks, n = 1000 (strong scaling); its 8-GPU shard
    c4 = bench.CONFIGS["c4"]
    assert c4["total"] == 10 ** 8 and c4["nb_vars"] == 1000 and c4["scaling"] == "strong"
    assert bench.CONFIGS["c4-shard"]["per_gpu"] * 8 == 10 ** 8


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
    assert d["config"]["bracket"].startswith("host point")           # SURVEY 8 d: host to host
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "candidates/s" and c["value"] > 1e3 and c["sample"]
    # (structure only: on a 20 000-candidate sample the pool's speed is process start-up and host load, not a property to assert)
    assert c["all_cores"]["cores"] > 1 and c["all_cores"]["value"] > 0
    # SURVEY 8 d: the k = 2, 4, 5 single-GPU rates ride on the same line
    for k in (2, 4, 5):
        s = d["secondary"]["k%d" % k]
        assert s["unit"] == "candidates/s" and s["value"] > 2e8 and 0.2 < s["roofline_frac"] < 1.0


@pytest.mark.gpu
def test_bench_config_c4_shard():
    """BASELINE.json configs[3] on one GPU: one of the eight shards (1.25e7 candidates of the on-device generator)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4-shard", "--steps", "3", "--warmup", "1",
                          "--cpu-sample", "20000"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["config"]["candidates_per_gpu"] == 12_500_000 and d["config"]["nb_vars"] == 1000 and d["scaling"] == "weak"
    assert abs(d["value"] - 12_500_000 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"] and d["value"] > 5e8
    assert d["roofline"]["candidates_per_launch"] == 12_500_000 and 0.2 < d["roofline"]["frac"] < 1.0
    assert d["cpu_baseline"]["value"] > 1e4


@pytest.mark.gpu
def test_bench_under_torchrun_with_real_rccl_prints_one_json_line():
    """The driver's N > 1 launch form (torch.distributed.run, one rank per GPU) with one rank: the
    process group is RCCL, the collectives of the sharded round are issued for real
    (SDPCUT_FORCE_COLLECTIVES=1), and stdout still carries exactly one JSON line (RCCL's version
    banner goes to stderr)."""
    env = dict(os.environ, SDPCUT_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5",
                          "--warmup", "1", "--no-cpu-baseline", "--no-secondary"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and 5e8 < d["value"] < 1e10


@pytest.mark.gpu
@pytest.mark.parametrize("config,extra", [("c2", []), ("c4", ["--steps", "2"])])
def test_bench_two_ranks_sharing_the_gpu(config, extra):
    """The N = 2 form of the driver's launch (torch.distributed.run, two ranks) rehearsed on the one-GPU
    box: both ranks on cuda:0 (SDPCUT_BENCH_ONE_DEVICE=1), the all-gather staged through gloo
    (SDPCUT_BENCH_BACKEND=gloo).  One JSON line from rank 0, n_gpus = 2, the whole-job value = the
    candidates of BOTH ranks over the slowest rank's time; c4 splits its 1e8 candidates between the ranks
    (strong scaling, ids from the on-device generator)."""
    env = dict(os.environ, SDPCUT_BENCH_ONE_DEVICE="1", SDPCUT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
           "--no-cpu-baseline", "--no-secondary", "--config", config] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    per_rank = 10 ** 6 if config == "c2" else 10 ** 8 // 2
    assert d["n_gpus"] == 2 and d["config"]["candidates_per_gpu"] == per_rank
    assert d["scaling"] == ("weak" if config == "c2" else "strong")
    assert abs(d["value"] - 2 * per_rank / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["value"] > 3e8          # two ranks time-share one GPU: roughly the one-GPU rate in total


def _plain(args, env_extra, timeout=600):
    """`python3 bench.py --gpus N ...` exactly as the driver's one-GPU command spells it: no launcher, WORLD_SIZE unset"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


@pytest.mark.parametrize("n", [2, 8])
def test_plain_bench_gpus_n_launches_its_own_ranks(n):
    """a bare `python bench.py --gpus N` used to exit non-zero.  Now the GPU-free parent starts N ranks, relays
    rank 0's ONE JSON line (stray stdout lines go to stderr) and returns 0.  Rehearsed here without a GPU: the ranks only
    rendezvous over gloo and run one collective (SDPCUT_BENCH_LAUNCH_ONLY)."""
    out = _plain(["--gpus", str(n), "--steps", "5"], {"SDPCUT_BENCH_LAUNCH_ONLY": "1"})
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["rank_sum"] == n * (n + 1) // 2 and d["self_launched"] and d["steps"] == 5
    assert "a stray line" in out.stderr.decode()


def test_plain_bench_returns_the_worst_rank_exit_code_and_leaves_nobody_behind():
    """a rank that dies: the launcher ends the others (they would wait in the collective for ever) and reports its code"""
    import time
    t0 = time.monotonic()
    out = _plain(["--gpus", "3"], {"SDPCUT_BENCH_LAUNCH_ONLY": "1", "SDPCUT_BENCH_FAIL_RANK": "1"}, timeout=300)
    assert out.returncode == 3, (out.returncode, out.stderr.decode()[-2000:])
    assert not [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert time.monotonic() - t0 < 120


@pytest.mark.gpu
@pytest.mark.parametrize("config,extra", [("c2", []), ("c4", ["--steps", "2"])])
def test_plain_bench_gpus_2_on_the_one_gpu_box(config, extra):
    """The exact form a SCALE run may use -- `python3 bench.py --gpus 2 --steps 5` -- on the one-GPU box: both ranks on cuda:0
    (SDPCUT_BENCH_ONE_DEVICE=1), the all-gather through gloo.  rc 0, one JSON line, n_gpus 2, phases and the 8-GPU budget on it."""
    out = _plain(["--gpus", "2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--config", config] + extra,
                 {"SDPCUT_BENCH_ONE_DEVICE": "1", "SDPCUT_BENCH_BACKEND": "gloo"}, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    per_rank = 10 ** 6 if config == "c2" else 10 ** 8 // 2
    assert d["n_gpus"] == 2 and d["config"]["candidates_per_gpu"] == per_rank
    assert d["config"]["launcher"].startswith("self") and d["config"]["collective_backend"] == "gloo" and d["config"]["rccl_ranks"] == 0
    assert abs(d["value"] - 2 * per_rank / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"] and d["value"] > 3e8
    assert set(d["phases"]["device_us"]) == {"score_and_head", "all_gather", "merge_and_rows"}
    assert "step_ms" in d["expected_8gpu"]
