"""Checkpoints in the reference's on-disk layout.

The reference's runners write `ckpt_<epoch:06d>.pth` files holding
    {"epoch", "optimizer_state_dict", ["scheduler_state_dict"], "scene_default", ["scene_fine"]}
where the scene entries are the radiance field's `state_dict()` (runners/runner_utils.py:737-775) and
resume from the lexicographically last file of the directory (:778-830).  These two functions read and write
exactly that, so that checkpoints move between the reference and this package in both directions:
the NeRF module here has the reference's parameter names, and `torch_nerf.amd.optim.FusedAdam` has
`torch.optim.Adam`'s state layout.

Tensors are written as detached CPU copies: with FusedAdam all parameters (and both moment estimates) are
views into one device blob, and a checkpoint should neither drag that blob along 44 times nor pin GPU memory
when it is loaded elsewhere.
"""
import os
from pathlib import Path
from typing import Optional, Union

import torch

__all__ = ["checkpoint_path", "save_checkpoint", "load_checkpoint"]


def checkpoint_path(ckpt_dir: Union[str, Path], epoch: int) -> Path:
    return Path(ckpt_dir) / f"ckpt_{str(epoch).zfill(6)}.pth"


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().to("cpu", copy=True)
    if isinstance(obj, dict):
        return type(obj)((k, _to_cpu(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def save_checkpoint(ckpt_dir: Union[str, Path], epoch: int, default_scene, fine_scene=None, optimizer=None,
                    scheduler=None) -> Path:
    """Write one checkpoint file; returns its path.  `*_scene` are scene primitives (`.radiance_field`)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    ckpt = {"epoch": int(epoch)}
    if optimizer is not None:
        ckpt["optimizer_state_dict"] = _to_cpu(optimizer.state_dict())
    if scheduler is not None:
        ckpt["scheduler_state_dict"] = _to_cpu(scheduler.state_dict())
    ckpt["scene_default"] = _to_cpu(default_scene.radiance_field.state_dict())
    if fine_scene is not None:
        ckpt["scene_fine"] = _to_cpu(fine_scene.radiance_field.state_dict())
    path = checkpoint_path(ckpt_dir, epoch)
    torch.save(ckpt, path)
    return path


def load_checkpoint(ckpt_dir: Optional[Union[str, Path]], default_scene, fine_scene=None, optimizer=None,
                    scheduler=None, device=None) -> int:
    """Restore the latest checkpoint of `ckpt_dir` into the given objects; returns the epoch to continue from
    (0 if there is nothing to load, like the reference).  Networks end up on `device` (default: the current
    GPU if one is visible, else they stay where they are)."""
    if ckpt_dir is None or not Path(ckpt_dir).exists():
        return 0
    files = sorted(p for p in Path(ckpt_dir).iterdir() if p.is_file())
    if not files:
        return 0
    ckpt = torch.load(files[-1], map_location="cpu")
    if device is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    default_scene.radiance_field.load_state_dict(ckpt["scene_default"])
    if device is not None:
        default_scene.radiance_field.to(device)
    if fine_scene is not None:
        fine_scene.radiance_field.load_state_dict(ckpt["scene_fine"])
        if device is not None:
            fine_scene.radiance_field.to(device)
    if optimizer is not None:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if scheduler is not None:
            scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return int(ckpt["epoch"])
