"""bench.py's own N-rank launcher (`--gpus N` without WORLD_SIZE): argument checks and failure propagation on the CPU,
the two-rank rehearsal on one card (`-m gpu`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LIP_DIST_BACKEND")}
    env.update(kw)
    return env


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="CPU-only behaviour")
def test_refuses_more_ranks_than_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "refusing" in r.stderr


def test_gpus_flag_must_match_world_size():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "disagrees" in r.stderr


@pytest.mark.skipif(not _no_gpu(), reason="CPU-only behaviour")
def test_failing_rank_fails_the_launch():
    """The children cannot initialise a GPU here: the parent must come back non-zero, promptly, with no rank left."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=_env(LIP_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=180)
    assert r.returncode not in (0, 2)


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_card():
    """`python bench.py --gpus 2` starts its own two ranks (gloo collectives, both on the one card of the box) and
    rank 0 prints one line with n_gpus = 2 in the whole-data-set unit."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--probes", "32", "--samples", "0",
                        "--no-cpu-baseline", "--no-resnet50"], env=_env(LIP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["examples_total"] == 100
    assert abs(line["per_shard_products_per_s"] - 2 * line["value"]) < 1e-6 * line["value"]
    assert abs(line["value"] - 32 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
