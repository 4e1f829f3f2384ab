"""Edge cases through the C ABI on the GPU: empty inputs, ragged sizes, error codes."""
import ctypes

import numpy as np
import pytest
import torch

from torch_nerf.amd import _lib, ops, synth

pytestmark = pytest.mark.gpu


def test_empty_inputs_are_no_ops():
    z3 = torch.zeros((0, 3), device="cuda")
    t_bins = torch.linspace(2.0, 6.0, 65, device="cuda")[:-1]
    pts, dirs, delta = ops.sample_stratified(z3, z3, t_bins, 4.0 / 64, torch.zeros((0, 64), device="cuda"))
    assert pts.shape == (0, 64, 3) and delta.shape == (0, 64)
    rgb, w = ops.composite_forward(torch.zeros((0, 64), device="cuda"), torch.zeros((0, 64, 3), device="cuda"),
                                   torch.zeros((0, 64), device="cuda"))
    assert rgb.shape == (0, 3) and w.shape == (0, 64)
    packed = ops.mlp_pack(torch.from_numpy(synth.nerf_flat_params(seed=0)).cuda())
    s, c = ops.mlp_forward(packed, z3, z3, encoded=False)
    assert s.shape == (0,) and c.shape == (0, 3)
    assert ops.posenc(z3, 10, True).shape == (0, 63)
    torch.cuda.synchronize()


def test_error_codes_instead_of_crashes():
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.nerf_composite_forward(None, None, None, 4, 64, None, None, st) == 1       # NERF_ERR_ARG
    assert b"null" in lib.nerf_amd_last_error()
    assert lib.nerf_sample_stratified(None, None, -1, 64, None, 0.1, None, None, None, None, None, st) == 1
    assert lib.nerf_mlp_forward(None, None, None, None, 10, 0, None, None, None, st) == 1
    x = torch.zeros(8, device="cuda")
    big = 20000  # more samples per ray than the LDS row buffer holds
    assert lib.nerf_sample_stratified(x.data_ptr(), x.data_ptr(), 1, big, x.data_ptr(), 0.1, x.data_ptr(), None,
                                      x.data_ptr(), x.data_ptr(), x.data_ptr(), st) == 2   # NERF_ERR_UNSUPPORTED
    ext = (ctypes.c_float * 12)(*([0.0] * 12))
    assert lib.nerf_generate_rays(None, None, 0, 4, 8, 8, 1.0, 1.0, 4.0, 4.0, ext, 1, 1.0, -1.0, x.data_ptr(),
                                  x.data_ptr(), st) == 1                                  # z_near < 0 under NDC
    with pytest.raises(ValueError):
        ops.mlp_pack(torch.zeros(10, device="cuda"))


@pytest.mark.parametrize("S", [1, 5, 63, 65, 100, 257])
def test_integrator_ragged_sample_counts(oracle, S):
    rng = np.random.RandomState(S)
    n = 37
    sigma = rng.gamma(0.7, 3.0, (n, S)).astype(np.float32)
    c = rng.rand(n, S, 3).astype(np.float32)
    t = np.sort(rng.uniform(2, 6, (n, S)).astype(np.float32), axis=1)
    delta = np.diff(np.concatenate([t, np.full((n, 1), 1e8, np.float32)], 1), axis=1).astype(np.float32)
    g = rng.standard_normal((n, 3)).astype(np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()
    rgb, w = ops.composite_forward(dev(sigma), dev(c), dev(delta))
    ro, wo = oracle.composite_forward(sigma, c, delta)
    np.testing.assert_allclose(rgb.cpu().numpy(), ro, rtol=0, atol=1e-5)
    np.testing.assert_allclose(w.cpu().numpy(), wo, rtol=0, atol=1e-6)
    gs, gc = ops.composite_backward(dev(sigma), dev(c), dev(delta), dev(g))
    gso, gco = oracle.composite_backward(sigma, c, delta, g)
    np.testing.assert_allclose(gc.cpu().numpy(), gco, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gs.cpu().numpy(), gso, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 7, 5), (2, 64, 0), (5, 33, 200)])
def test_sampling_ragged_shapes_vs_oracle(oracle, shape):
    n, Sc, Sf = shape
    rng = np.random.RandomState(n * 1000 + Sc)
    o = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    u1 = rng.rand(n, Sc).astype(np.float32)
    t_bins = torch.linspace(2.0, 6.0, Sc + 1)[:-1].numpy()
    ps = 4.0 / Sc
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pts, dirs, delta = ops.sample_stratified(dev(o), dev(d), dev(t_bins), ps, dev(u1))
    _, po, do, dlo = oracle.stratified_sample(o, d, t_bins, ps, u1)
    assert np.array_equal(pts.cpu().numpy(), po) and np.array_equal(delta.cpu().numpy(), dlo)
    if Sf > 0:
        w = (rng.rand(n, Sc) ** 4).astype(np.float32)
        u2, u3 = rng.rand(n, Sf).astype(np.float32), rng.rand(n, Sf).astype(np.float32)
        wt = dev(w)
        pts, dirs, delta, idx = ops.sample_hierarchical(dev(o), dev(d), dev(t_bins), ps, wt, dev(u1), dev(u2), dev(u3),
                                                        want_idx=True)
        io, to, po, _, dlo, wo = oracle.hierarchical_sample(o, d, t_bins, ps, w, u1, u2, u3)
        assert np.array_equal(idx.cpu().numpy(), io)
        assert np.array_equal(pts.cpu().numpy(), po) and np.array_equal(delta.cpu().numpy(), dlo)
        assert np.array_equal(wt.cpu().numpy(), wo)


def test_counter_uniform_kernel_is_the_numpy_stream_bit_for_bit():
    """csrc/draws.hip against torch_nerf.amd.synth.counter_uniform (numpy) and against its own slices:
    the draws of a ray range must not depend on which GPU asks for them."""
    import numpy as np
    import torch
    from torch_nerf.amd import shard, synth
    for seed, stream in ((0, 0), (123, 2), (2 ** 40 + 7, 3)):
        ref = synth.counter_uniform(seed, stream, 70001)
        full = shard.counter_uniform(seed, stream, 0, 70001, "cuda").cpu().numpy()
        assert np.array_equal(full, ref)
        part = shard.counter_uniform(seed, stream, 31337, 1025, "cuda").cpu().numpy()
        assert np.array_equal(part, ref[31337:31337 + 1025])
    assert shard.counter_uniform(1, 1, 5, 0, "cuda").numel() == 0
    a = shard.ray_draws(9, 40, 20, 64, 128, "cuda")
    b = shard.ray_draws(9, 0, 100, 64, 128, "cpu")
    for x, y in zip(a, b):
        assert torch.equal(x.cpu(), y[40:60])


@pytest.mark.parametrize("S", [64, 192, 300])
def test_backward_transmittance_equals_forward_for_sharp_densities(oracle, S):
    """The backward kernel rebuilds T_i per 64-sample step; its prefix must be the forward kernel's, bit for
    bit, also when densities are sharp (sigma ~ 1e2..1e4) and the last sample carries tau = sigma * 1e8.
    With g_rgb = (1, 0, 0), g_radiance[..., 0] IS the backward's weight w_i."""
    rng = np.random.RandomState(S)
    n = 256
    sigma = (rng.gamma(0.5, 1.0, (n, S)) * 10.0 ** rng.uniform(0, 4, (n, 1))).astype(np.float32)
    sigma[:, -1] = rng.uniform(50, 2e4, n).astype(np.float32)          # a dense far end: tau_last up to 2e12
    c = rng.rand(n, S, 3).astype(np.float32)
    t = np.sort(rng.uniform(2, 6, (n, S)).astype(np.float32), axis=1)
    delta = np.diff(np.concatenate([t, np.full((n, 1), 1e8, np.float32)], 1), axis=1).astype(np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()
    _, w = ops.composite_forward(dev(sigma), dev(c), dev(delta))
    g = np.zeros((n, 3), np.float32)
    g[:, 0] = 1.0
    gs, gc = ops.composite_backward(dev(sigma), dev(c), dev(delta), dev(g))
    assert torch.equal(gc[..., 0], w), (gc[..., 0] - w).abs().max().item()
    # g_sigma against the oracle (all-double arithmetic).  Rays whose first samples are already opaque have
    # gradients ~1e-18 that consist of cancellation noise in fp32 (the reference's fp32 autograd has the same);
    # so the yardstick is the ray's largest gradient, floored at 1e-6 of the batch's largest
    gso, _ = oracle.composite_backward(sigma, c, delta, g)
    scale = np.maximum(np.abs(gso).max(axis=1, keepdims=True), 1e-6 * np.abs(gso).max())
    assert np.max(np.abs(gs.cpu().numpy() - gso) / scale) < 1e-4      # fp32 T / w (as in the forward), double sums
