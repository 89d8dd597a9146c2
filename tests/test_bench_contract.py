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
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    assert d["value"] == pytest.approx(1e8 / (d["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    assert d["config"]["checks"]["g7_sha256_equal"] is True and d["config"]["checks"]["all_streams_equal_single_stream"] is True
    assert d["host"]["gpu_max_hw_queues"] == os.environ.get("GPU_MAX_HW_QUEUES", "8")
