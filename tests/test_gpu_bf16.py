"""BASELINE configs[2]: bf16 MLP weights (and layer inputs) on the bf16 MFMA path.  The 1e-5 bound is
unreachable in bf16 by construction; the parity target is restated as a PSNR bound on pixel colours
against the fp32 path: >= 40 dB, plus loose element-wise bounds on sigma / rgb."""
import numpy as np
import pytest
import torch

from torch_nerf.amd import ops, shard, synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def psnr(a, b):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


@pytest.mark.parametrize("M", [1, 63, 256, 257, 5000, 70000])
def test_bf16_forward_close_to_fp32(M):
    rng = np.random.RandomState(M)
    pts = dev(rng.uniform(-4, 4, (M, 3)).astype(np.float32))
    dirs = dev(rng.uniform(-1, 1, (M, 3)).astype(np.float32))
    flat = dev(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0))
    s32, c32 = ops.mlp_forward(ops.mlp_pack(flat), pts, dirs, encoded=False)
    s16, c16 = ops.mlp_forward_bf16(ops.mlp_pack_bf16(flat), pts, dirs)
    assert torch.isfinite(s16).all() and torch.isfinite(c16).all()
    # bf16 has 8 mantissa bits: ~0.4 % per rounding, a few % after ten layers
    assert (c16 - c32).abs().max().item() < 3e-2
    assert ((s16 - s32).abs() <= 0.05 * s32.abs() + 0.05).all()
    assert psnr(c16, c32) > 40.0


def test_bf16_rendered_pixels_psnr():
    """Coarse + fine pass of a 2048-ray batch with bf16 networks vs fp32 networks, same draws."""
    H = W = 800
    focal = float(synth.blender_focal(W))
    pose = torch.from_numpy(synth.pose_spherical(37.0, -30.0, 4.0))
    n = 2048
    pix = dev(synth.pixel_batch(2, H, W, n))
    k4 = (np.float32(focal), np.float32(focal), W / 2.0, H / 2.0)
    o, d = ops.generate_rays(H, W, k4, pose, False, focal, 2.0, "cuda", pix=pix)
    t_bins = torch.linspace(2.0, 6.0, 65, device="cuda")[:-1]
    ps = 4.0 / 64
    u1c, u1, u2, u3 = shard.ray_draws(11, 0, n, 64, 128, "cuda")
    flats = [dev(synth.nerf_flat_params(seed=s, sigma_bias=1.0, sigma_gain=30.0)) for s in (3, 4)]

    def render(fwd, packs):
        pts, dirs, delta = ops.sample_stratified(o, d, t_bins, ps, u1c)
        s, c = fwd(packs[0], pts.reshape(-1, 3), dirs.reshape(-1, 3))
        _, w = ops.composite_forward(s.reshape(n, 64), c.reshape(n, 64, 3), delta)
        pts, dirs, delta = ops.sample_hierarchical(o, d, t_bins, ps, w, u1, u2, u3)
        s, c = fwd(packs[1], pts.reshape(-1, 3), dirs.reshape(-1, 3))
        rgb, _ = ops.composite_forward(s.reshape(n, 192), c.reshape(n, 192, 3), delta)
        return rgb

    ref = render(lambda p, a, b: ops.mlp_forward(p, a, b, encoded=False), [ops.mlp_pack(f) for f in flats])
    got = render(ops.mlp_forward_bf16, [ops.mlp_pack_bf16(f) for f in flats])
    assert psnr(got, ref) > 40.0, psnr(got, ref)
    assert (got - ref).abs().max().item() < 5e-2


def test_bf16_through_the_module_flag():
    import torch_nerf.src.network as network
    net = network.NeRF(63, 27)
    flat = synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()})
    net = net.cuda()
    pts = torch.rand(1000, 3, device="cuda") * 8 - 4
    dirs = torch.rand(1000, 3, device="cuda") * 2 - 1
    with torch.no_grad():
        s32, c32 = net.forward_fused(pts, dirs)
        net.bf16_inference = True
        s16, c16 = net.forward_fused(pts, dirs)
    assert not torch.equal(c16, c32) and psnr(c16, c32) > 40.0
    # training keeps fp32 kernels even when the flag is set
    s, c = net.forward_fused(pts, dirs)
    assert torch.equal(s.detach(), s32) and c.requires_grad
