"""Ray sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-device.  Rays of a frame are independent once the two networks are
replicated, so a frame is cut into `world` contiguous ranges of the row-major pixel index,
each rank renders its range (coarse pass, then fine pass on the same rank: per-ray weights
never cross GPUs) and ONE all-gather of (H*W/world, 3) fp32 assembles the image
(960 KB per rank at 800x800 on 8 GPUs: latency bound, no ring needed).  There is no other
collective on the data path.

Random draws are a pure function of (seed, stream, global ray index, sample index), so an
image does not depend on how many GPUs rendered it.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

_M1 = -4658895280553007687   # 0xBF58476D1CE4E5B9 as int64
_M2 = -7723592293110705685   # 0x94D049BB133111EB
_G1 = -7046029254386353131   # 0x9E3779B97F4A7C15
_G2 = -3372029247567499371   # 0xD1342543DE82EF95


# rays per launch of render_frame's kernel chain (non-fused networks): 16 384 x 192 samples of points, directions, sigma,
# radiance (and, for encoders without a raw-point entry such as SHEncoder, the row-major encodings) stay well under 1 GB
CHAIN_RAYS_PER_LAUNCH = 16384


def _lsr(x: torch.Tensor, k: int) -> torch.Tensor:
    """logical shift right on int64"""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix(x: torch.Tensor) -> torch.Tensor:
    x = (x ^ _lsr(x, 30)) * _M1
    x = (x ^ _lsr(x, 27)) * _M2
    return x ^ _lsr(x, 31)


def _mix_int(x: int) -> int:
    mask = (1 << 64) - 1
    x &= mask
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & mask
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & mask
    return x ^ (x >> 31)


def counter_uniform(seed: int, stream: int, first: int, count: int, device) -> torch.Tensor:
    """fp32 uniforms in [0,1): element i equals synth.counter_uniform(seed, stream, n)[first + i]."""
    base = _mix_int(seed * 0x9E3779B97F4A7C15 + stream)
    if torch.device(device).type == "cuda":      # one HIP launch instead of ~20 tensor ops
        from torch_nerf.amd import ops
        return ops.counter_uniform(base + 1, first, count, device)
    if base >= 1 << 63:
        base -= 1 << 64
    idx = torch.arange(first, first + count, dtype=torch.int64, device=device)
    bits = _mix(idx * _G2 + (base + 1))
    return (_lsr(bits, 40).to(torch.float64) * (1.0 / 16777216.0)).to(torch.float32)


def ray_draws(seed: int, first_ray: int, n_rays: int, n_coarse: int, n_fine: int, device):
    """(u1_coarse_pass, u1, u2, u3) for rays [first_ray, first_ray + n_rays) -- G-independent."""
    def block(stream, width):
        if width == 0:
            return torch.empty((n_rays, 0), dtype=torch.float32, device=device)
        return counter_uniform(seed, stream, first_ray * width, n_rays * width, device).view(n_rays, width)
    return block(0, n_coarse), block(1, n_coarse), block(2, n_fine), block(3, n_fine)


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of the row-major pixel index owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_image(local_rgb: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All-gather the per-rank (n_r, 3) slabs into the (total, 3) image, on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert local_rgb.shape[0] == total
        return local_rgb
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    longest = (total + world - 1) // world
    slab = local_rgb.new_zeros((longest, 3))
    slab[: local_rgb.shape[0]] = local_rgb
    out = local_rgb.new_empty((world * longest, 3))
    dist.all_gather_into_tensor(out, slab, group=group)
    if total == world * longest:
        return out
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(out[r * longest: r * longest + (hi - lo)])
    return torch.cat(parts, 0)


def allreduce_gradients(parameters, group: Optional[dist.ProcessGroup] = None, average: bool = True) -> None:
    """Data-parallel training glue: every rank back-propagates its own shard of the
    ray batch, then ONE all-reduce of the concatenated gradients of both networks (2 x 595 844 fp32 =
    4.77 MB) makes them identical everywhere.  In place on `.grad`; no-op for a single process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def _scene_parts(x):
    """(network, scene | None): render_frame takes the two NeRF modules (encoders inferred from their widths) or the two
    scene primitives the runners hold (runner_utils.py:872-908 passes default_scene / fine_scene)."""
    field = getattr(x, "radiance_field", None)
    return (x, None) if field is None else (field, x)


@torch.no_grad()
def render_frame(camera, coarse_net, fine_net, n_coarse: int, n_fine: int, project_to_ndc: bool, seed: int,
                 group: Optional[dist.ProcessGroup] = None, rays_per_launch: int = 131072,
                 bf16: bool = False, single_rank: bool = False, stats: Optional[dict] = None,
                 f16x2: bool = False) -> torch.Tensor:
    """Full frame (H*W, 3) on every rank; each rank renders only its pixel range on its own GPU.
    stats (optional dict) receives what a slow N-GPU frame is diagnosed from: this rank's pixel range, its number of
    launches and two GPU events bracketing its own rendering, the collective excluded (`render_events`).
    coarse_net / fine_net: NeRF modules, or PrimitiveCube scenes (network + encoders).  Networks of the fused family
    behind PositionalEncoders run ONE kernel per pass and launch; every other combination the scene can evaluate --
    coord_encode_level >= 11, dir_encode_level >= 5, signal_encoder: sh -- runs the kernel chain sampling ->
    scene.query_points (raw points into the layered network kernel behind PositionalEncoders; encoder kernel + network
    kernel otherwise) -> integral per launch, in smaller launches (a launch's points and directions live in HBM).
    Same pixel ranges, same counter draws, same all-gather.
    single_rank=True: the calling rank renders the whole frame alone, no collective (the 1-GPU image a sharded
    image must equal bit for bit)."""
    from torch_nerf.amd import ops
    from torch_nerf.src.renderer.ray_samplers import StratifiedSampler

    on = dist.is_available() and dist.is_initialized() and not single_rank
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    device = torch.device("cuda", torch.cuda.current_device())
    total = camera.img_height * camera.img_width
    lo, hi = shard_range(total, rank, world)
    sampler = StratifiedSampler()
    t_bins, ps = sampler._create_t_bins(camera.t_near, camera.t_far, n_coarse, device)
    (coarse_net, coarse_scene), (fine_net, fine_scene) = _scene_parts(coarse_net), _scene_parts(fine_net)
    spec_c = coarse_scene.fused_net() if coarse_scene is not None else coarse_net.inferred_net()
    spec_f = fine_scene.fused_net() if fine_scene is not None else fine_net.inferred_net()
    fused = spec_c is not None and spec_f is not None
    if not fused and (coarse_scene is None or fine_scene is None):
        raise RuntimeError("render_frame: these networks are outside the fused family (feat_dim 256, pos_dim <= 64, "
                           "view_dir_dim <= 32 behind PositionalEncoders); pass the scene primitives (network + "
                           "encoders) instead of the bare networks so that the kernel chain can encode for them")
    if bf16 and not fused:
        raise RuntimeError("render_frame(bf16=True): the bf16 kernel serves the fused family only")
    if f16x2 and not fused:
        # the kernel chain: scene.query_points picks the split-f16 kernel from the networks' own flag (feat_dim 256 behind
        # PositionalEncoders with pos_dim <= 128, view_dir_dim <= 64); anything else would silently stay fp32 -- refuse
        for sc in (coarse_scene, fine_scene):
            spec = sc.raw_net()
            if spec is None or not spec.f16x2_ok:
                raise RuntimeError("render_frame(f16x2=True): the split-f16 kernel serves feat_dim 256 behind two "
                                   "PositionalEncoders with pos_dim <= 128 and view_dir_dim <= 64")
        saved_flags = [(net, net.f16x2_inference) for net in (coarse_net, fine_net)]
        for net, _ in saved_flags:
            net.f16x2_inference = True
        try:
            return render_frame(camera, coarse_scene, fine_scene, n_coarse, n_fine, project_to_ndc, seed, group,
                                rays_per_launch, False, single_rank, stats, False)
        finally:
            for net, flag in saved_flags:
                net.f16x2_inference = flag
    if fused:
        _, flat_c, packed_c = coarse_net._stream()
        _, flat_f, packed_f = fine_net._stream()
        if bf16:  # BASELINE configs[2]: bf16 weights / layer inputs on the bf16 MFMA path
            packed_c, packed_f = ops.mlp_pack_bf16(flat_c, spec_c), ops.mlp_pack_bf16(flat_f, spec_f)
        if f16x2:  # the fp32 bound on the f16 matrix pipe (operands split in two f16 parts)
            packed_c, packed_f = ops.mlp_pack_f16x2(flat_c, spec_c), ops.mlp_pack_f16x2(flat_f, spec_f)
    else:
        rays_per_launch = min(rays_per_launch, CHAIN_RAYS_PER_LAUNCH)
    out = torch.empty((hi - lo, 3), dtype=torch.float32, device=device)
    # equal launches of at most rays_per_launch rays (a multiple of 4: the fused pass walks bunches of four rays): the
    # persistent kernel then ends every launch with the same, small, last-round imbalance instead of one short tail
    launches = max(1, -(-(hi - lo) // max(1, rays_per_launch)))
    per_launch = -(-(hi - lo) // launches)
    per_launch = min(max(4, (per_launch + 3) // 4 * 4), max(4, rays_per_launch))
    if stats is not None:
        stats.update(rank=rank, first=lo, rays=hi - lo, launches=len(range(lo, hi, per_launch)), rays_per_launch=per_launch,
                     fused=fused, render_events=(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        stats["render_events"][0].record()
    for first in range(lo, hi, per_launch):
        n = min(per_launch, hi - first)
        bundle = sampler.generate_rays_from_pixels(camera, project_to_ndc, first=first, count=n, device=device)
        u1c, u1, u2, u3 = ray_draws(seed, first, n, n_coarse, n_fine, device)
        if fused:
            rgb, w = ops.render_rays(packed_c, bundle.ray_origin, bundle.ray_dir, t_bins, ps, u1c, bf16=bf16, net=spec_c, f16x2=f16x2)
            if n_fine > 0:     # n_fine == 0: coarse-only frame (BASELINE configs[0])
                rgb, _ = ops.render_rays(packed_f, bundle.ray_origin, bundle.ray_dir, t_bins, ps, u1, weights=w, u2=u2,
                                         u3=u3, bf16=bf16, net=spec_f, f16x2=f16x2)
        else:
            pts, dirs, delta = ops.sample_stratified(bundle.ray_origin, bundle.ray_dir, t_bins, ps, u1c)
            sigma, radiance = coarse_scene.query_points(pts, dirs)
            rgb, w = ops.composite_forward(sigma, radiance, delta)
            if n_fine > 0:
                pts, dirs, delta = ops.sample_hierarchical(bundle.ray_origin, bundle.ray_dir, t_bins, ps, w, u1, u2, u3)
                sigma, radiance = fine_scene.query_points(pts, dirs)
                rgb, _ = ops.composite_forward(sigma, radiance, delta)
        out[first - lo: first - lo + n] = rgb
    if stats is not None:
        stats["render_events"][1].record()
    return out if single_rank else gather_image(out, total, group)
