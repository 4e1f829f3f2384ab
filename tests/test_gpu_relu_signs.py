"""The kernels' ReLU decisions against the REFERENCE's own.

Fixtures F5 / F11 / F12 hold, next to the reference's outputs and gradient digests, the sign of every activation the
reference's autograd differentiated through (captured by a forward hook on its nn.ReLU, make_golden.py:ReluSigns).  A
pre-activation within an ulp of zero may take the other branch under another fp32 summation order -- that is the only
way a correct kernel can differ, it is rare (~1e-7 of the units) and it is MEASURED here: the kernels' masks, decoded
from their records, must differ from the reference's in < 1e-5 of the units.  On a fixture where they are identical,
kernel and reference differentiate the same piecewise-linear function and the gradient digests must agree to
summation-order rounding: rtol 2e-5 + 2e-5 of the tensor's rms (the flip-blind comparison needed 2e-4 / 2e-3)."""
import numpy as np
import pytest
import torch

from torch_nerf.amd import ops, synth
from helpers import NET_VARIANTS, check_grad_digest, fused_masks, layered_masks, variant_params

pytestmark = pytest.mark.gpu
TIGHT = dict(rtol=2e-5, atol_scale=2e-5, norm_rtol=2e-5)
LOOSE = dict(rtol=2e-4, atol_scale=2e-3)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def ref_bits(g, tag):
    shape = tuple(int(v) for v in g[tag + "_relu_shape"])
    return np.unpackbits(g[tag + "_relu_bits"])[:shape[0] * shape[1]].reshape(shape)


def compare(masks, want, grad, g, prefix, dims):
    flips = int((masks != want).sum())
    assert flips < 1e-5 * want.size + 1, f"{flips} of {want.size} ReLU decisions differ from the reference's"
    check_grad_digest(grad, g, prefix, dims=dims, **(TIGHT if flips == 0 else LOOSE))
    return flips


@pytest.mark.parametrize("tag,kw", [("default", dict(seed=1)), ("dense", dict(seed=2, sigma_bias=1.0, sigma_gain=30.0))])
def test_shipped_network_raw_entry(golden, tag, kw):
    g = golden("f5_mlp")
    fp = dev(synth.nerf_flat_params(**kw))
    packed = ops.mlp_pack(fp)
    pts, dirs = dev(g["pts"]), dev(g["dirs"])
    sigma, rgb, saved = ops.mlp_forward(packed, pts, dirs, encoded=False, save=True)
    grad = ops.mlp_backward(packed, fp, pts, dirs, False, sigma, rgb, saved, dev(g[tag + "_g_sigma"]), dev(g[tag + "_g_rgb"]))
    compare(fused_masks(saved, sigma, pts.shape[0]), ref_bits(g, tag), grad.cpu().numpy(), g, tag + "_grad_", (63, 27, 256))


@pytest.mark.parametrize("tag", sorted(NET_VARIANTS))
def test_network_variants(golden, tag):
    g = golden("f11_net_variants")
    flat, dims = variant_params(g, tag)
    spec = ops.Net.dims_only(*dims)
    fp, pe, de = dev(flat), dev(g[tag + "_pe"]), dev(g[tag + "_de"])
    M = pe.shape[0]
    gs, gc = dev(g["g_sigma"]), dev(g["g_rgb"])
    want = ref_bits(g, tag)
    if spec.fused:
        packed = ops.mlp_pack(fp, spec)
        sigma, rgb, saved = ops.mlp_forward(packed, pe, de, True, save=True, net=spec)
        grad = ops.mlp_backward(packed, fp, pe, de, True, sigma, rgb, saved, gs, gc, net=spec)
        compare(fused_masks(saved, sigma, M), want, grad.cpu().numpy(), g, tag + "_grad_", dims)
    sigma, rgb, rec = ops.mlp_layered_forward(fp, pe, de, spec, record=True)      # the layered family serves every network
    grad, g_pos, g_dir = ops.mlp_layered_backward(fp, pe, de, spec, sigma, rgb, rec, gs, gc, want_pos=True, want_dir=True)
    flips = compare(layered_masks(rec, sigma, M, spec), want, grad.cpu().numpy(), g, tag + "_grad_", dims)
    if flips == 0:   # the input gradients autograd returned, at the same tolerance
        for got, ref in ((g_pos, g[tag + "_g_pe"]), (g_dir, g[tag + "_g_de"])):
            rms = np.sqrt(np.mean(ref.astype(np.float64) ** 2))
            np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * rms)


def test_sh_scene_network(golden):
    g = golden("f12_sh_encoder")
    spec = ops.Net.dims_only(16, 16, 256)
    fp = dev(synth.nerf_flat_params(seed=6, pos_dim=16, view_dir_dim=16, sigma_bias=0.5, sigma_gain=4.0))
    pe, de = ops.shenc(dev(g["pts"]), 4), ops.shenc(dev(g["dirs"]), 4)
    packed = ops.mlp_pack(fp, spec)
    sigma, rgb, saved = ops.mlp_forward(packed, pe, de, True, save=True, net=spec)
    grad = ops.mlp_backward(packed, fp, pe, de, True, sigma, rgb, saved, dev(g["net_g_sigma"]), dev(g["net_g_rgb"]), net=spec)
    compare(fused_masks(saved, sigma, pe.shape[0]), ref_bits(g, "net"), grad.cpu().numpy(), g, "net_grad_", (16, 16, 256))
