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


def _run(*flags, timeout=600, launcher=(), lines_expected=1):
    """Run bench.py (optionally under `python -m torch.distributed.run ...`); returns the parsed JSON lines of rank 0.
    With a process group up the bench prints the headline line first and the complete line last."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *flags], capture_output=True,
                         text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == lines_expected, f"expected {lines_expected} JSON line(s) from rank 0: {out.stdout[-800:]}"
    return lines[-1] if lines_expected == 1 else lines


def test_self_launch_two_ranks_gloo():
    line = _run("--gpus", "2", "--backend", "gloo", "--launch-check")
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2 and line["launch_check"] is True


def test_self_launch_eight_ranks_gloo():
    """The shape of the driver's 8-GPU run, on CPU: eight ranks through the real self-launch, every collective of the
    N-rank line on stand-in data (slab all-gather per step, per-rank time gather, the frame leg's statistics gather)."""
    line = _run("--gpus", "8", "--backend", "gloo", "--launch-check", "--steps", "3")
    assert line["n_gpus"] == 8 and line["rccl_ranks_seen"] == 8
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and pr["ms_per_step_min"] <= pr["ms_per_step_max"] and 0 <= pr["rank_of_max"] < 8
    assert line["frame"]["rays_per_rank"] == [80000] * 8 and line["frame"]["launches_per_rank"] == [1] * 8


def test_single_rank_needs_no_launcher():
    line = _run("--gpus", "1", "--backend", "gloo", "--launch-check")
    assert line["n_gpus"] == 1 and line["rccl_ranks_seen"] == 1


def test_mismatched_launcher_world_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr


def test_a_rank_that_dies_before_the_first_collective_ends_the_run():
    """The failure path of the N-rank launch: rank 1 exits between the rendezvous and the first
    data collective.  The run must END -- launcher return code != 0 well inside --dist-timeout + launcher teardown --
    with no headline line, instead of hanging in the surviving rank's all-reduce.  Child processes only."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--launch-check", "--fault-rank", "1", "--dist-timeout", "20"],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout[-500:]
    assert time.time() - t0 < 120


def test_a_dying_rank_among_eight_ends_the_run():
    """The same failure with the driver's rank count: rank 5 of 8 exits before the first collective."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo",
                          "--launch-check", "--fault-rank", "5", "--dist-timeout", "20"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout[-500:]
    assert time.time() - t0 < 150


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_full_bench():
    first, line = _run("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                       timeout=900, lines_expected=2)
    pr = line["per_rank"]        # what a bad N-GPU number is diagnosed from
    assert len(pr["ms_per_step"]) == 2 and pr["ms_per_step_min"] <= pr["ms_per_step_max"] and pr["rank_of_max"] in (0, 1)
    assert line["frame"]["gather_ms"] > 0
    assert len(line["frame"]["ms_per_rank"]) == 2 and min(line["frame"]["ms_per_rank"]) > 0
    assert line["frame"]["rays_per_rank"] == [320000, 320000] and line["frame"]["launches_per_rank"] == [3, 3]
    assert "error" not in line["train"], line["train"]          # data-parallel leg: on by default since round 4
    # the headline line leaves before any secondary (collective) leg; the last line repeats it and adds the legs
    assert first["partial"] and "frame" not in first and first["value"] == line["value"] and "partial" not in line
    assert line["n_gpus"] == 2 and line["rccl_ranks_seen"] == 2
    assert line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["global_rays_per_step"] == 2 * 4096
    frame = line["frame"]
    assert "error" not in frame, frame
    assert frame["rays"] == 640000 and frame["scaling"] == "strong"
    assert frame["equals_one_rank_image"] is True
    assert frame["image_sha256_16"] == frame["image_sha256_16_one_rank"]


@pytest.mark.gpu
def test_one_rank_rccl_collectives():
    """The first RCCL calls this code ever issues must not be the driver's 8-GPU run: one rank under
    torch.distributed.run with backend nccl (= RCCL), process group created with device_id and a timeout,
    all_gather_into_tensor in every step, all_reduce / barrier around the timing, the frame leg's collectives."""
    launcher = ("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", "29571")
    first, line = _run("--gpus", "1", "--force-dist", "--backend", "nccl", "--steps", "3", "--warmup", "1",
                       "--no-cpu-baseline", "--no-train", "--no-bf16", "--no-stages", "--no-configs",
                       "--no-runner-loop", timeout=900, launcher=launcher, lines_expected=2)
    assert first["partial"] and first["value"] > 0
    assert line["backend"] == "nccl" and line["rccl_ranks_seen"] == 1 and line["n_gpus"] == 1
    assert line["config"]["parallelism"].endswith("all-gather")
    frame = line["frame"]
    assert "error" not in frame, frame
    assert frame["equals_one_rank_image"] is True and frame["rays"] == 640000


@pytest.mark.gpu
def test_eight_ranks_on_one_gpu_through_the_real_kernels():
    """The driver's rank count through the REAL kernels: eight ranks, all on the box's one GPU,
    gloo between them -- every rank renders its own 4096-ray steps and its 80 000-pixel range of the 800x800 frame with
    the fused render kernel, the slabs meet in the all-gather and the assembled image is the one-rank image bit for
    bit.  (No scaling figure: eight processes share one GPU.)"""
    first, line = _run("--gpus", "8", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                       "--no-train", "--no-bf16", "--no-f16x2", timeout=1500, lines_expected=2)
    assert first["partial"] and "frame" not in first and first["value"] == line["value"] and "partial" not in line
    assert line["n_gpus"] == 8 and line["rccl_ranks_seen"] == 8 and line["scaling"] == "weak"
    assert line["config"]["global_rays_per_step"] == 8 * 4096 and line["value"] > 0
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and min(pr["ms_per_step"]) > 0 and 0 <= pr["rank_of_max"] < 8
    frame = line["frame"]
    assert "error" not in frame, frame
    assert frame["rays"] == 640000 and frame["scaling"] == "strong"
    assert frame["rays_per_rank"] == [80000] * 8 and frame["launches_per_rank"] == [1] * 8
    assert len(frame["ms_per_rank"]) == 8 and min(frame["ms_per_rank"]) > 0 and frame["gather_ms"] > 0
    assert frame["equals_one_rank_image"] is True
    assert frame["image_sha256_16"] == frame["image_sha256_16_one_rank"]
