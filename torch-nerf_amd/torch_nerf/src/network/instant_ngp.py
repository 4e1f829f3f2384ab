"""Names kept so that `network.InstantNeRF` resolves (runners/runner_utils.py:617).
Instant-NGP is an alternative model family outside the hot path."""
import torch.nn as nn

__all__ = ["InstantNeRF"]


class InstantNeRF(nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("InstantNeRF is out of scope of the MI355X volume-rendering hot path")
