"""bench.py --gpus N must start its own N ranks (the driver may call it without torchrun).

CPU: `--launch-check` goes through the real self-launch (child torch.distributed.run, rendezvous on 127.0.0.1,
all-reduce, one JSON line from rank 0) without rendering.  GPU: the full 2-rank bench over gloo with both ranks on
the single GPU of the test box -- weak-scaling value, the sharded 800x800 frame and its equality with the 1-rank image.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}: {out.stdout[-500:]}"
    return json.loads(lines[0])


def test_self_launch_two_ranks_gloo():
    line = _run("--gpus", "2", "--backend", "gloo", "--launch-check")
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2 and line["launch_check"] is True


def test_single_rank_needs_no_launcher():
    line = _run("--gpus", "1", "--backend", "gloo", "--launch-check")
    assert line["n_gpus"] == 1 and line["rccl_ranks_seen"] == 1


def test_mismatched_launcher_world_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_full_bench():
    line = _run("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                timeout=900)
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2
    assert line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["global_rays_per_step"] == 2 * 4096
    frame = line["frame"]
    assert "error" not in frame, frame
    assert frame["rays"] == 640000 and frame["scaling"] == "strong"
    assert frame["equals_one_rank_image"] is True
    assert frame["image_sha256_16"] == frame["image_sha256_16_one_rank"]
