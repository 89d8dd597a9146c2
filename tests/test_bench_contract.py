"""bench.py as the driver runs it: the command line parses without a GPU, the environment the HIP runtime reads is set
before torch is imported, and (GPU) a short run prints ONE JSON line with the fields of the contract."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_help_runs_without_a_gpu():
    out = subprocess.run([sys.executable, BENCH, "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--streams"):
        assert flag in out.stdout


def test_hardware_queues_are_requested_before_torch_is_imported():
    src = open(BENCH).read()
    env = src.index('os.environ.setdefault("GPU_MAX_HW_QUEUES"')
    first_torch = min(m.start() for m in re.finditer(r"^\s*(import torch|from torch|from pypore_amd)", src, re.M))
    assert env < first_torch


def test_gpus_n_starts_by_itself_and_relays_rank0_line():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment becomes the launcher of two ranks (children under
    torch.distributed.run, 127.0.0.1) before anything touches a GPU; --selftest-dist makes the ranks run only the N > 1
    plumbing of bench.py -- JobGather, the sharded-trace join + repair gather, the files gather, the max-over-ranks clock
    -- on CPU tensors over gloo.  One JSON line comes back on stdout, the exit code is the children's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-dist"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["ok"] is True and d["ranks_seen"] == 2 and all(d["checks"].values()), d


def test_world_of_eight_over_gloo_incl_a_forced_seam_repair_and_the_64_file_shard():
    """The first SCALE run must not die on plumbing: `bench.py --gpus 8 --selftest-dist` -- eight ranks
    through the real launcher path on CPU tensors over gloo: the job gather, the sharded-trace join, the same join with a
    seam that shares no anchor (the upstream rank re-segments, dist.stitch_pieces' repair), BASELINE config 4's shard of 64
    files (every file once, balanced), the max-over-ranks clock."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--selftest-dist"], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["ok"] is True and d["ranks_seen"] == 8 and all(d["checks"].values()), d
    assert d["files_units"] == 64 and d["checks"]["sharded_trace_join_with_seam_repair"] and d["shard_imbalance"] < 1.1


def test_self_launch_command_line(monkeypatch):
    """the launcher's command: one node, N processes, rendezvous on 127.0.0.1, this file with the caller's arguments; no exec"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.self_launch(["--gpus", "4", "--steps", "3"], 4)
    cmd = seen["cmd"]
    assert rc == 7
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the ranks pick their own number of hardware queues (20 under a communicator) unless the caller set one
    assert seen["env"].get("GPU_MAX_HW_QUEUES") == os.environ.get("GPU_MAX_HW_QUEUES") or not bench._QUEUES_FROM_CALLER
    src = open(BENCH).read()
    assert "os.exec" not in src and "execv" not in src.replace("no exec", "")


@pytest.mark.gpu
def test_short_run_prints_the_contract_line():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu", "--no-h2d"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_measured_in_run", "evaluations_per_s"):
        assert key in r, key
    assert d["config"]["ranks_seen"] == 1 and len(d["config"]["ms_per_step_per_rank"]) == 1
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert d["value"] == pytest.approx(1e8 / (d["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    assert d["config"]["checks"]["g7_sha256_equal"] is True and d["config"]["checks"]["all_streams_equal_single_stream"] is True
    assert d["host"]["gpu_max_hw_queues"] == os.environ.get("GPU_MAX_HW_QUEUES", "16")
