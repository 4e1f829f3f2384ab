"""Fused, data-parallel Adam for the two NeRF networks.

The reference builds ``torch.optim.Adam(params, lr=init_lr, eps=eps)`` over the parameters of the
coarse and the fine network (runners/runner_utils.py:683-695), decays it with ``ExponentialLR``
(:701-711) and calls ``optimizer.step(); scheduler.step()`` once per batch (runners/train.py:215-218).
``FusedAdam`` takes the same constructor arguments and is a ``torch.optim.Optimizer`` (``zero_grad``,
``param_groups`` -- so torch's LR schedulers drive it unchanged -- ``state_dict`` /
``load_state_dict`` with torch.optim.Adam's per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` keys),
but:

* all parameters of a group live in ONE contiguous fp32 blob (each ``p.data`` becomes a view into it,
  values preserved), and so do both moment estimates: a step is one launch of ``nerf_adam_step`` over
  2 x 595 844 values instead of ~10 multi-tensor launches over 44 tensors; the networks' kernels read the
  same blob as their flat parameter image (``NeRF._stream``), so nothing is re-concatenated per step;
* with ``torch.distributed`` initialised, every rank back-propagates its own shard of the ray batch and the
  step starts with ONE all-reduce (RCCL) of the joined gradient blob (4.77 MB); the 1/world average is
  folded into the update kernel.

The learning rate is a host scalar read from ``param_groups`` at every step: no device synchronisation,
nothing like ``.item()``.  No CPU fallback: parameters must be on the GPU.
"""
from typing import Optional

import torch
import torch.distributed as dist

from . import ops

__all__ = ["FusedAdam"]


def _flat_view(t: torch.Tensor, count: int) -> torch.Tensor:
    """`count` consecutive floats starting at t's first element, sharing t's storage."""
    return t.as_strided((count,), (1,), t.storage_offset())


class _Arena:
    """Contiguous parameter / gradient / moment blobs of one param group."""

    def __init__(self, params, state):
        self.params = list(params)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += p.numel()
        self.n = n
        dev = self.params[0].device
        self.P = torch.empty(n, dtype=torch.float32, device=dev)
        # gradient staging; the tail of len(params) floats carries one "this rank has a gradient" flag per parameter
        # through the data-parallel all-reduce (same collective, no second one)
        self.G_flags = torch.zeros(n + len(self.params), dtype=torch.float32, device=dev)
        self.G = self.G_flags[:n]
        self.M = torch.zeros(n, dtype=torch.float32, device=dev)
        self.V = torch.zeros(n, dtype=torch.float32, device=dev)
        # per-parameter step counts, like torch.optim.Adam's state["step"]; host integers between checkpoints
        self.steps = [int(state[p]["step"]) if (p in state and "step" in state[p]) else 0 for p in self.params]
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                k = p.numel()
                self.P[off:off + k].copy_(p.detach().reshape(-1))
                old = state.get(p, {})
                if "exp_avg" in old:        # after load_state_dict
                    self.M[off:off + k].copy_(old["exp_avg"].reshape(-1))
                    self.V[off:off + k].copy_(old["exp_avg_sq"].reshape(-1))
                p.data = self.P[off:off + k].view(p.shape)
                state[p] = {"step": torch.tensor(0.0), "exp_avg": self.M[off:off + k].view(p.shape),
                            "exp_avg_sq": self.V[off:off + k].view(p.shape)}

    def intact(self) -> bool:
        base = self.P.data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, off in zip(self.params, self.offsets))

    def publish_steps(self, state) -> None:
        for p, t in zip(self.params, self.steps):
            if p in state and "step" in state[p]:
                state[p]["step"].fill_(float(t))

    def gradient_runs(self):
        """[(first, last, a, b, tensor-or-None)]: maximal runs params[first:last] = blob[a:b] of consecutive
        parameters with equal step counts whose gradients are consecutive in memory (they are when one
        backward kernel wrote them), or that have none."""
        runs = []
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            g, k = p.grad, p.numel()
            if g is not None and not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                raise RuntimeError("FusedAdam: gradients must be contiguous fp32 GPU tensors")
            if runs:
                first, _, a, _, head, end_ptr = runs[-1]
                if self.steps[i] == self.steps[first] and (
                        (head is None and g is None) or
                        (head is not None and g is not None and g.data_ptr() == end_ptr
                         and g.untyped_storage().data_ptr() == head.untyped_storage().data_ptr())):
                    runs[-1] = (first, i + 1, a, off + k, head, None if g is None else end_ptr + 4 * k)
                    continue
            runs.append((i, i + 1, off, off + k, g, None if g is None else g.data_ptr() + 4 * k))
        return [(first, last, a, b, None if head is None else _flat_view(head, b - a))
                for first, last, a, b, head, _ in runs]


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 data_parallel: Optional[bool] = None, process_group: Optional[dist.ProcessGroup] = None):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid Adam hyper-parameters: lr={lr}, betas={betas}, eps={eps}")
        # torch.optim.Adam's remaining group keys ride along at their defaults, so that a checkpoint written by one
        # optimizer loads into the other (runner_utils.py:758-775 saves optimizer.state_dict()); settings the
        # kernel does not implement are refused at step time
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False,
                                      fused=None, decoupled_weight_decay=False))
        self._data_parallel = data_parallel
        self._process_group = process_group
        self._arenas = {}

    # ------------------------------------------------------------------ layout
    def _arena(self, index: int, group) -> _Arena:
        arena = self._arenas.get(index)
        same = arena is not None and len(arena.params) == len(group["params"]) and \
            all(a is b for a, b in zip(arena.params, group["params"]))
        if not same or not arena.intact():
            for p in group["params"]:
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameters must be on the GPU (no CPU fallback)")
                if p.dtype != torch.float32:
                    raise ValueError("FusedAdam: fp32 parameters only")
            if arena is not None:
                # the step counts live as host integers in the arena between checkpoints: hand them (and, through
                # the state's views, the moments) to the replacement -- e.g. after an EMA swap reassigned p.data
                arena.publish_steps(self.state)
            arena = self._arenas[index] = _Arena(group["params"], self.state)
        return arena

    def state_dict(self):
        for arena in self._arenas.values():
            arena.publish_steps(self.state)
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._arenas.clear()        # moments now live in freshly copied tensors: re-home them lazily

    def _world(self) -> int:
        on = dist.is_available() and dist.is_initialized()
        if self._data_parallel is False or not on:
            if self._data_parallel and not on:
                raise RuntimeError("FusedAdam(data_parallel=True) needs torch.distributed initialised")
            return 1
        return dist.get_world_size(self._process_group)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        world = self._world()
        for index, group in enumerate(self.param_groups):
            if not group["params"]:
                continue
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FusedAdam implements Adam as the reference configures it: "
                                          "no weight decay, no amsgrad, no maximize")
            arena = self._arena(index, group)
            runs = arena.gradient_runs()
            if world == 1 and all(g is None for *_, g in runs):
                continue          # (with world > 1 a rank without gradients still joins the all-reduce, with zeros)
            b1, b2 = group["betas"]
            if world > 1:
                # every rank contributes its shard's gradient; a parameter without one contributes zeros and a 0
                # flag.  A parameter steps iff SOME rank has a gradient for it (what DDP + torch.optim.Adam do:
                # no gradient anywhere -> no update, no step count).  A rank that holds every gradient itself
                # knows the answer without looking; only a rank with a hole reads the reduced flags back (one
                # device->host copy, on the rare ragged-shard step).
                n_par = len(arena.params)
                local = [True] * n_par
                for first, last, a, b, g in runs:
                    if g is None:
                        arena.G[a:b].zero_()
                        local[first:last] = [False] * (last - first)
                    else:
                        arena.G[a:b].copy_(g)
                if all(local):       # the usual step: one fill kernel, no host-to-device copy (a pageable copy would
                    arena.G_flags[arena.n:].fill_(1.0)          # drain the stream and stall the host behind the backward)
                else:
                    arena.G_flags[arena.n:].copy_(torch.tensor([float(x) for x in local]).pin_memory(),
                                                  non_blocking=True)
                dist.all_reduce(arena.G_flags, op=dist.ReduceOp.SUM, group=self._process_group)
                has = local if all(local) else [x > 0.0 for x in arena.G_flags[arena.n:].tolist()]
                i = 0
                while i < n_par:      # maximal ranges of stepping parameters with equal step counts: normally ONE
                    if not has[i]:
                        i += 1
                        continue
                    j = i + 1
                    while j < n_par and has[j] and arena.steps[j] == arena.steps[i]:
                        j += 1
                    a = arena.offsets[i]
                    b = arena.offsets[j] if j < n_par else arena.n
                    t = arena.steps[i] + 1
                    arena.steps[i:j] = [t] * (j - i)
                    ops.adam_step(arena.P[a:b], arena.G[a:b], arena.M[a:b], arena.V[a:b], t, group["lr"], b1, b2,
                                  group["eps"], grad_scale=1.0 / world)
                    i = j
            else:
                for first, last, a, b, g in runs:
                    if g is None:
                        continue                      # torch.optim.Adam skips parameters without gradient
                    if (g.data_ptr() - arena.P[a:b].data_ptr()) % 16:
                        arena.G[a:b].copy_(g)         # the kernel wants one common 16-byte phase
                        g = arena.G[a:b]
                    t = arena.steps[first] + 1
                    arena.steps[first:last] = [t] * (last - first)
                    ops.adam_step(arena.P[a:b], g, arena.M[a:b], arena.V[a:b], t, group["lr"], b1, b2,
                                  group["eps"])
            # the kernel wrote through raw pointers: tell autograd (and NeRF._stream's pack cache)
            torch.autograd.graph.increment_version(arena.P)
            for p in arena.params:
                torch.autograd.graph.increment_version(p)
        return loss
