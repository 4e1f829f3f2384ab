"""Backward parity on the GPU: hand-written MLP / integrator gradients against the gradients
the reference's autograd produced (golden digests) and against the CPU oracle."""
import numpy as np
import pytest
import torch

import torch_nerf.src.network as network
import torch_nerf.src.scene as scene
import torch_nerf.src.renderer.cameras as cameras
import torch_nerf.src.renderer.integrators.quadrature_integrator as integrators
import torch_nerf.src.renderer.ray_samplers as ray_samplers
from torch_nerf.src.renderer.volume_renderer import VolumeRenderer
from torch_nerf.src.signal_encoder import PositionalEncoder
from torch_nerf.amd import ops, synth
from helpers import (F7_TIGHT, assert_grads_match_given_masks, check_grad_digest, f7_oracle_chain, f7_oracle_grad,
                     fused_masks, reference_relu_decisions)

pytestmark = pytest.mark.gpu

MLP_KW = {"default": dict(seed=1), "dense": dict(seed=2, sigma_bias=1.0, sigma_gain=30.0)}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def flat_grad(net):
    return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()


def make_net(flat):
    net = network.NeRF(63, 27)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat).items()})
    return net.cuda()


@pytest.mark.parametrize("tag", ["default", "dense"])
def test_mlp_backward_golden(golden, tag):
    """Same inputs and upstream gradients as the reference's autograd run (fixture F5)."""
    g = golden("f5_mlp")
    flat = synth.nerf_flat_params(**MLP_KW[tag])
    fp = dev(flat)
    packed = ops.mlp_pack(fp)
    pts, dirs = dev(g["pts"]), dev(g["dirs"])
    sigma, rgb, saved = ops.mlp_forward(packed, pts, dirs, encoded=False, save=True)
    np.testing.assert_allclose(sigma.cpu().numpy(), g[tag + "_sigma"], rtol=0, atol=1e-5)
    grad = ops.mlp_backward(packed, fp, pts, dirs, False, sigma, rgb, saved, dev(g[tag + "_g_sigma"]),
                            dev(g[tag + "_g_rgb"]))
    check_grad_digest(grad.cpu().numpy(), g, tag + "_grad_", rtol=2e-4, atol_scale=2e-3)


@pytest.mark.parametrize("M", [1, 100, 128, 1000, 5000, 20000])
def test_mlp_backward_vs_oracle_full_tensor(oracle, M):
    """Every element of every gradient tensor against the CPU oracle, ragged sizes included.
    A pre-activation that sits within an ulp of zero can take the other ReLU branch under a different fp32 summation
    order.  That is measured, not assumed: the kernel's own ReLU decisions are decoded from the record's mask planes,
    must differ from the oracle's in < 1e-5 of all (sample, unit) pairs, and are then handed to the oracle's backward
    (force_masks) -- both sides differentiate the same piecewise-linear function, and every element of every tensor
    must agree within 2e-5 relative + 2e-5 of the tensor's rms (helpers.assert_grads_match_given_masks; summation-order rounding: the oracle sums in double)."""
    from helpers import assert_grads_match_given_masks, fused_masks
    rng = np.random.RandomState(M + 3)
    pts = rng.uniform(-4, 4, (M, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    gs = rng.standard_normal(M).astype(np.float32)
    gc = rng.standard_normal((M, 3)).astype(np.float32)
    flat = synth.nerf_flat_params(seed=11, sigma_bias=0.3, sigma_gain=20.0)
    fp = dev(flat)
    packed = ops.mlp_pack(fp)
    sigma, rgb, saved = ops.mlp_forward(packed, dev(pts), dev(dirs), encoded=False, save=True)
    got = ops.mlp_backward(packed, fp, dev(pts), dev(dirs), False, sigma, rgb, saved, dev(gs), dev(gc)).cpu().numpy()
    # the oracle on the KERNEL's encodings (the record's PE / DE planes would do too; nerf_posenc's rows differ from the
    # in-register Cody-Waite values by <= 1.5e-7, which moves gradients by more than the bound below)
    pe, de = oracle.posenc(pts, 10), oracle.posenc(dirs, 4)
    _, _, _, own = oracle.mlp_backward_ex(flat, pe, de, gs, gc, want_inputs=False, want_masks=True)
    masks = fused_masks(saved, sigma, M)
    flips = masks != own
    assert flips.mean() < 1e-5, f"{flips.sum()} of {flips.size} ReLU decisions differ from the oracle's"
    ref = oracle.mlp_backward_ex(flat, pe, de, gs, gc, want_inputs=False, force_masks=masks)[0]
    # encodings differ by <= 1.5e-7 abs between the kernel (Cody-Waite in registers) and the oracle (libm): that alone
    # is a 1e-6-relative perturbation of fc_in / fc_5 / fc_9 inputs, inside the bound
    assert_grads_match_given_masks(got, ref, synth.split_flat_params, f"M={M} ")


@pytest.mark.parametrize("M,levels", [(1, (10, 4, True)), (129, (10, 4, True)), (5000, (10, 4, True)),
                                      (1000, (6, 2, False)), (777, (4, 4, True))])
def test_fused_input_gradients_vs_oracle(oracle, M, levels):
    """g_pos / g_view_dir out of the fused dX chain (the three thin GEMMs of csrc/mlp_backward.hip, IG variant) --
    what autograd returns for the two inputs of NeRF.forward (nerf.py:102, :108, :116) -- every element against the
    CPU oracle under the kernel's own ReLU decisions, together with the parameter gradients of the same launch (which
    must equal the plain chain's bit for bit: same kernels on the same planes)."""
    from helpers import assert_grads_match_given_masks, fused_masks
    lp, ld, inc = levels
    e_p, e_d = 6 * lp + 3 * inc, 6 * ld + 3 * inc
    rng = np.random.RandomState(M + 17)
    pts = rng.uniform(-3, 3, (M, 3)).astype(np.float32)
    dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    gs = rng.standard_normal(M).astype(np.float32)
    gc = rng.standard_normal((M, 3)).astype(np.float32)
    flat = synth.nerf_flat_params(seed=13, pos_dim=e_p, view_dir_dim=e_d, sigma_bias=0.3, sigma_gain=20.0)
    pe, de = oracle.posenc(pts, lp, inc), oracle.posenc(dirs, ld, inc)
    spec = ops.Net(e_p, e_d, 256, lp, inc, ld, inc)
    fp = dev(flat)
    packed = ops.mlp_pack(fp, spec)
    sigma, rgb, saved = ops.mlp_forward(packed, dev(pe), dev(de), encoded=True, save=True, net=spec)
    plain = ops.mlp_backward(packed, fp, dev(pe), dev(de), True, sigma, rgb, saved, dev(gs), dev(gc), net=spec)
    got, g_pos, g_dir = ops.mlp_backward(packed, fp, dev(pe), dev(de), True, sigma, rgb, saved, dev(gs), dev(gc),
                                         net=spec, want_pos=True, want_dir=True)
    assert torch.equal(got, plain)
    assert tuple(g_pos.shape) == (M, e_p) and tuple(g_dir.shape) == (M, e_d)
    masks = fused_masks(saved, sigma, M)
    _, _, _, own = oracle.mlp_backward_ex(flat, pe, de, gs, gc, want_inputs=False, want_masks=True)
    assert (masks != own).mean() < 1e-5
    ref, ref_pos, ref_dir, _ = oracle.mlp_backward_ex(flat, pe, de, gs, gc, want_inputs=True, force_masks=masks)
    split = lambda f: synth.split_flat_params(f, e_p, e_d, 256)
    assert_grads_match_given_masks(got.cpu().numpy(), ref, split, f"M={M} ")
    for name, a, b in (("g_pos", g_pos.cpu().numpy(), ref_pos), ("g_view_dir", g_dir.cpu().numpy(), ref_dir)):
        rms = np.sqrt(np.mean(b.astype(np.float64) ** 2)) + 1e-30
        bad = np.abs(a - b) > 2e-5 * np.abs(b) + 2e-5 * rms
        assert not bad.any(), f"{name}: {bad.sum()} of {bad.size} beyond the bound, worst {np.abs(a - b).max() / rms:.2e} rms"
    # one side only: the other pointer NULL
    _, only_pos, none_dir = ops.mlp_backward(packed, fp, dev(pe), dev(de), True, sigma, rgb, saved, dev(gs), dev(gc),
                                             net=spec, want_pos=True)
    assert none_dir is None and torch.equal(only_pos, g_pos)


def test_raw_point_gradients_through_the_fused_query(golden):
    """PrimitiveCube.query_points on raw points that require grad (the reference's autograd reaches them through
    cube.py:59-72): fused record forward -> IG dX chain -> reverse of PositionalEncoder.encode, against fixture F11's
    gradients w.r.t. the raw points / directions captured from the reference."""
    from helpers import NET_VARIANTS, variant_params
    g = golden("f11_net_variants")
    for tag in ("l6_l2", "l4_l4", "l10_l4_noinput"):
        lp, ld, inc, feat = NET_VARIANTS[tag]
        flat, dims = variant_params(g, tag)
        net = network.NeRF(*dims)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat, *dims).items()})
        net = net.cuda()
        cube = scene.PrimitiveCube(net, {"coord_enc": PositionalEncoder(3, lp, inc), "dir_enc": PositionalEncoder(3, ld, inc)})
        M = g["pts"].shape[0]
        pts = dev(g["pts"]).view(M // 4, 4, 3).requires_grad_(True)
        dirs = dev(g["dirs"]).view(M // 4, 4, 3).requires_grad_(True)
        sigma, rgb = cube.query_points(pts, dirs)
        np.testing.assert_allclose(sigma.detach().cpu().numpy().reshape(-1), g[tag + "_sigma"], rtol=0, atol=1e-5)
        ((sigma.reshape(-1) * dev(g["g_sigma"])).sum() + (rgb.reshape(-1, 3) * dev(g["g_rgb"])).sum()).backward()
        check_grad_digest(flat_grad(net), g, tag + "_grad_", rtol=2e-4, atol_scale=2e-3, dims=dims)
        for got, want in ((pts.grad.reshape(M, 3), g[tag + "_g_pts"]), (dirs.grad.reshape(M, 3), g[tag + "_g_dirs"])):
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=2e-5 * np.abs(want).max())


def test_autograd_function_paths(golden):
    """NeRF.forward (encoded inputs) and forward_fused (raw inputs) give the same gradients."""
    g = golden("f5_mlp")
    flat = synth.nerf_flat_params(**MLP_KW["dense"])
    pts, dirs = dev(g["pts"]), dev(g["dirs"])
    gs, gc = dev(g["dense_g_sigma"]), dev(g["dense_g_rgb"])
    grads = []
    for fused in (True, False):
        net = make_net(flat)
        if fused:
            sigma, rgb = net.forward_fused(pts, dirs)
        else:
            sigma, rgb = net(PositionalEncoder(3, 10, True).encode(pts), PositionalEncoder(3, 4, True).encode(dirs))
        ((sigma * gs).sum() + (rgb * gc).sum()).backward()
        grads.append(flat_grad(net))
        check_grad_digest(grads[-1], g, "dense_grad_", rtol=2e-4, atol_scale=2e-3)
    # the two paths differ only in who evaluates sin/cos (a few ulp): gradients agree to fp32 noise
    np.testing.assert_allclose(grads[0], grads[1], rtol=1e-4, atol=5e-6)


class _Replay:
    def __init__(self, draws):
        self.draws = [dev(d) for d in draws]

    def __call__(self, shape, device=None, **kw):
        d = self.draws.pop(0)
        assert tuple(d.shape) == tuple(shape)
        return d


def test_training_step_gradients_match_reference(golden, oracle, monkeypatch):
    """loss = MSE(coarse) + MSE(fine), backward through integrator + MLP of both networks,
    exactly as runners/train.py:172-215; compared with the reference's gradients (fixture F7).

    F7 holds a digest per sample of the ReLU decisions the reference's autograd differentiated through.  At 53 M
    decisions no fp32 evaluation reproduces all of them (~4e-7 sit within rounding of zero: the kernels differ from the
    reference in ~20 samples, and so do the oracle's plain loops), and one toggled decision moves a gradient row by
    ~1e-3 of its rms -- which is why this check used to need rtol 5e-4 / atol 2e-2 rms.  Now the decisions are accounted
    for exactly:
      1. the reference's decisions are rebuilt from the digests (helpers.reference_relu_decisions) and the kernels' are
         decoded from the records the two training forwards wrote: they must differ in < 1e-5 of the units;
      2. the chain is cut where autograd hands the integrator's gradients to the MLP backward (the arguments of
         ops.mlp_backward, captured): (a) those upstream gradients vs the oracle's double-precision reverse of the
         quadrature rule on the kernels' own sigma / radiance (rtol 1e-5 + 2e-6 of the ray's largest, as for F6);
         (b) the parameter gradients vs the oracle differentiating with the KERNELS' decisions and the same upstream:
         every element of every tensor at rtol 2e-5 + 2e-5 rms;
      3. the parameter gradients, corrected by what the oracle attributes to the differing decisions (oracle with the
         reference's decisions minus oracle with the kernels'), vs the reference's digests at F7_TIGHT: rtol 2e-5,
         6e-4 rms, norms 1e-5 -- the reference's own sgemm summation noise (tests/test_oracle_golden.py measures it)."""
    g = golden("f7_e2e")
    H, W, focal, near, far = g["meta"]
    cam = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H},
                                    torch.from_numpy(g["pose"]), float(near), float(far))
    enc = {"coord_enc": PositionalEncoder(3, 10, True), "dir_enc": PositionalEncoder(3, 4, True)}
    net_c = make_net(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0))
    net_f = make_net(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0))
    vr = VolumeRenderer(integrators.QuadratureIntegrator(), ray_samplers.StratifiedSampler(), cam)
    pix = torch.from_numpy(g["pix"])
    gt = dev(g["gt"])
    dev_i = torch.cuda.current_device()
    monkeypatch.setattr(torch, "rand", _Replay([g["u1c"], g["u1"], g["u2"], g["u3"]]))
    records = []                                     # (record, sigma, M) of each training forward, in call order
    real_forward = ops.mlp_forward

    def recording_forward(packed, pos, view_dir, encoded, save=False, net=None):
        out = real_forward(packed, pos, view_dir, encoded, save=save, net=net)
        if save:
            records.append((out[2], out[0], pos.shape[0]))
        return out

    monkeypatch.setattr(ops, "mlp_forward", recording_forward)
    upstream = []                                    # (g_sigma, g_rgb) of each MLP backward, in call order (fine first)
    real_backward = ops.mlp_backward

    def recording_backward(packed, flat_params, pos, view_dir, encoded, sigma, rgb, saved, g_sigma, g_rgb, **kw):
        upstream.append((sigma, rgb, g_sigma, g_rgb))
        return real_backward(packed, flat_params, pos, view_dir, encoded, sigma, rgb, saved, g_sigma, g_rgb, **kw)

    monkeypatch.setattr(ops, "mlp_backward", recording_backward)
    c_rgb, c_idx, c_w = vr.render_scene(scene.PrimitiveCube(net_c, enc), len(pix), 64, False, dev_i,
                                        pixel_indices=pix)
    # fine bins from the reference's coarse weights (sampling carries no gradient)
    f_rgb, _, _ = vr.render_scene(scene.PrimitiveCube(net_f, enc), len(pix), (64, 128), False, dev_i,
                                  pixel_indices=c_idx, weights=dev(g["coarse_w"]))
    mse = torch.nn.MSELoss()
    loss = mse(gt, c_rgb) + mse(gt, f_rgb)
    assert abs(loss.item() - float(g["loss"][0])) < 1e-6
    loss.backward()
    assert [r[2] for r in records] == [96 * 64, 96 * 192]
    assert [u[0].shape[0] for u in upstream] == [96 * 192, 96 * 64]
    chain = f7_oracle_chain(oracle, g)
    for (saved, sigma, M), up, pixels, tag, net in zip(records, upstream[::-1], (c_rgb, f_rgb), ("coarse", "fine"),
                                                       (net_c, net_f)):
        c = chain[tag]
        n, S = c["delta"].shape
        k_sigma, k_rgb, k_gs, k_gc = (t.detach().cpu().numpy() for t in up)
        # 2a: the integrator's reverse pass on the kernels' own forward values
        g_pixels = (2.0 * (pixels.detach().cpu().numpy() - g["gt"]) / np.float32(n * 3)).astype(np.float32)
        want_s, want_c = oracle.composite_backward(k_sigma.reshape(n, S), k_rgb.reshape(n, S, 3), c["delta"], g_pixels)
        err = np.abs(k_gs.reshape(n, S) - want_s)
        # F6's bound (test_gpu_kernels.py: rtol 1e-5 + 2e-6 of the ray's largest + an absolute floor), the floor scaled
        # to THIS loss's pixel gradients: d sigma_i = delta_i (T_{i+1} G_i - sum_{k>i} w_k G_k) cancels two terms of size
        # delta |g| whose fp32 exp() carry 6e-8 each -- on rays where they cancel completely (row maximum ~ 0) that
        # leaves ~1e-8 |g| (measured worst: 1.07e-8 |g|); 3e-8 |g| allowed
        bound = 1e-5 * np.abs(want_s) + 2e-6 * np.abs(want_s).max(axis=1, keepdims=True) + 3e-8 * np.abs(g_pixels).max()
        assert np.all(err <= bound), (tag, float((err / bound).max()))
        np.testing.assert_allclose(k_gc.reshape(n, S, 3), want_c, rtol=1e-5, atol=1e-7 * np.abs(g_pixels).max())
        # 1: decisions
        ref_masks, _, unresolved = reference_relu_decisions(oracle, c["params"], c["pe"], c["de"], c["masks"],
                                                            g[tag + "_relu_hash"])
        assert unresolved.size == 0
        mine = fused_masks(saved, sigma, M)
        flips = int((mine != ref_masks).sum())
        assert flips < 1e-5 * ref_masks.size, f"{tag}: {flips} of {ref_masks.size} ReLU decisions differ from the reference's"
        # 2b: the MLP's reverse pass, same decisions and same upstream on both sides
        ck = dict(c, g_sigma=k_gs.reshape(-1), g_rgb=k_gc.reshape(-1, 3))
        got = flat_grad(net)
        with_mine = f7_oracle_grad(oracle, ck, mine)
        assert_grads_match_given_masks(got, with_mine, synth.split_flat_params, tag=f"{tag} ({flips} flips): ")
        # 3: against the reference's numbers
        corrected = got + (f7_oracle_grad(oracle, ck, ref_masks) - with_mine) if flips else got
        check_grad_digest(corrected, g, tag + "_grad_", **F7_TIGHT)
