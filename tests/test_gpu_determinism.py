"""Determinism + hazard sweep of every asm-bearing entry.

The `save_plane` hazard of round 3 (a v_readfirstlane-written SGPR read by an asm-issued store too early) corrupted
one store in ~60 -- differently on every run -- and reached a green suite because nothing ran the affected entry
twice.  Round 4 met two more of the same family while building new kernels (asm-issued loads whose destinations hipcc
copied or re-used before the data landed; a wide store's data registers overwritten one instruction later).  The static
audit (scripts/audit_asm_loads.py) knows these classes; THIS test is the dynamic net under it: every entry that issues
loads / stores / LDS reads by hand runs REPS times on the same inputs, and every output -- record planes included --
must be bit-identical across runs.  There are no atomics on any of these paths, so any difference is a hazard."""
import numpy as np
import pytest
import torch

from torch_nerf.amd import ops, synth

pytestmark = pytest.mark.gpu
REPS = 50
M = 3001        # ragged: the last tile is partly padding


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def same_bits(x, y):
    """Bit patterns, not values: the padded rows of a record may hold NaN-patterned filler, and NaN != NaN."""
    if x is None or y is None:
        return x is y
    return torch.equal(x.view(torch.int32), y.view(torch.int32))


def _inputs(e_p, e_d, seed=0):
    rng = np.random.RandomState(seed)
    return (dev(rng.uniform(-2, 2, (M, 3)).astype(np.float32)), dev(rng.uniform(-1, 1, (M, 3)).astype(np.float32)),
            dev(rng.uniform(-1, 1, (M, e_p)).astype(np.float32)), dev(rng.uniform(-1, 1, (M, e_d)).astype(np.float32)),
            dev(rng.standard_normal(M).astype(np.float32)), dev(rng.standard_normal((M, 3)).astype(np.float32)))


FUSED_CASES = [
    ("raw-shipped", (63, 27, 256, 10, 1, 4, 1), False),
    ("raw-levels", (39, 15, 256, 6, 1, 2, 1), False),
    ("raw-noinput", (60, 24, 256, 10, 0, 4, 0), False),
    ("pre-shipped", (63, 27, 256, 10, 1, 4, 1), True),
    ("pre-levels", (27, 27, 256, 4, 1, 4, 1), True),
    ("pre-sh-widths", (16, 16, 256, -1, 0, -1, 0), True),       # SHEncoder(3, 4) on both inputs: levels unknown
]


@pytest.mark.parametrize("name,key,encoded", FUSED_CASES, ids=[c[0] for c in FUSED_CASES])
def test_fused_family_record_forward_and_backward_are_bit_reproducible(name, key, encoded):
    spec = ops.Net(*key)
    e_p, e_d = key[0], key[1]
    pts, dirs, pe, de, gs, gc = _inputs(e_p, e_d)
    a, b = (pe, de) if encoded else (pts, dirs)
    fp = dev(synth.nerf_flat_params(seed=3, pos_dim=e_p, view_dir_dim=e_d, sigma_bias=0.5, sigma_gain=8.0))
    packed = ops.mlp_pack(fp, spec)
    first = None
    for rep in range(REPS):
        sigma, rgb, saved = ops.mlp_forward(packed, a, b, encoded, save=True, net=spec)
        s_inf, c_inf = ops.mlp_forward(packed, a, b, encoded, save=False, net=spec)
        g, g_pos, g_dir = ops.mlp_backward(packed, fp, a, b, encoded, sigma, rgb, saved, gs, gc, net=spec,
                                           want_pos=True, want_dir=True)
        g_plain = ops.mlp_backward(packed, fp, a, b, encoded, sigma, rgb, saved, gs, gc, net=spec)
        out = (sigma, rgb, saved, s_inf, c_inf, g, g_pos, g_dir, g_plain)
        if first is None:
            first = [t.clone() for t in out]
            assert torch.equal(sigma, s_inf) and torch.equal(rgb, c_inf) and torch.equal(g, g_plain)
            assert all(torch.isfinite(t).all() for t in (sigma, rgb, g, g_pos, g_dir))
        else:
            for k, (x, y) in enumerate(zip(first, out)):
                assert same_bits(x, y), f"{name}: output {k} differs in repetition {rep}"


LAYERED_CASES = [("narrow-128", (63, 27, 128)), ("narrow-3-pos-blocks", (75, 27, 100)), ("reg-256-level-12", (75, 27, 256)),
                 ("reg-256-dir-level-5", (63, 33, 256)), ("reg-256-level-16-dir-level-10", (99, 63, 256)),
                 ("wide-512", (63, 27, 512)),
                 ("ragged-160", (40, 40, 160)), ("thin-64", (75, 39, 64)), ("reg-64", (63, 27, 64))]


@pytest.mark.parametrize("name,dims", LAYERED_CASES, ids=[c[0] for c in LAYERED_CASES])
def test_layered_family_is_bit_reproducible(name, dims):
    spec = ops.Net.dims_only(*dims)
    _, _, pe, de, gs, gc = _inputs(dims[0], dims[1], seed=1)
    fp = dev(synth.nerf_flat_params(seed=4, pos_dim=dims[0], view_dir_dim=dims[1], feat_dim=dims[2], sigma_bias=0.5,
                                    sigma_gain=8.0))
    first = None
    for rep in range(REPS // 2):
        sigma, rgb, rec = ops.mlp_layered_forward(fp, pe, de, spec, record=True)
        s_inf, c_inf = ops.mlp_layered_forward(fp, pe, de, spec)
        g, g_pos, g_dir = ops.mlp_layered_backward(fp, pe, de, spec, sigma, rgb, rec, gs, gc, want_pos=True, want_dir=True)
        # without input gradients (every training step) the 256-feature networks take the register-resident reverse chain
        g_plain, _, _ = ops.mlp_layered_backward(fp, pe, de, spec, sigma, rgb, rec, gs, gc)
        out = (sigma, rgb, rec, s_inf, c_inf, g, g_pos, g_dir, g_plain)
        if first is None:
            first = [t.clone() for t in out]
            assert torch.equal(sigma, s_inf) and torch.equal(rgb, c_inf) and torch.equal(g, g_plain)
        else:
            for k, (x, y) in enumerate(zip(first, out)):
                assert same_bits(x, y), f"{name}: output {k} differs in repetition {rep}"


def test_render_passes_are_bit_reproducible():
    """The single-kernel render pass (fp32) and the bf16 chain, coarse + fine, incl. the floored weights."""
    from torch_nerf.amd import shard
    n, Sc, Sf = 1027, 64, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    o = torch.randn(n, 3, device="cuda", generator=g)
    d = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda", generator=g), dim=-1)
    t_bins = torch.linspace(2.0, 6.0, Sc + 1, device="cuda")[:-1]
    u1c, u1, u2, u3 = shard.ray_draws(7, 0, n, Sc, Sf, "cuda")
    fp = dev(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0))
    for bf16 in (False, True):
        packed = ops.mlp_pack_bf16(fp) if bf16 else ops.mlp_pack(fp)
        first = None
        for rep in range(REPS // 2):
            rgb_c, w_c = ops.render_rays(packed, o, d, t_bins, 4.0 / Sc, u1c, bf16=bf16)
            w_in = w_c.clone()
            rgb_f, w_f = ops.render_rays(packed, o, d, t_bins, 4.0 / Sc, u1, weights=w_in, u2=u2, u3=u3, bf16=bf16)
            out = (rgb_c, w_c, w_in, rgb_f, w_f)
            if first is None:
                first = [t.clone() for t in out]
            else:
                for k, (x, y) in enumerate(zip(first, out)):
                    assert same_bits(x, y), f"bf16={bf16}: output {k} differs in repetition {rep}"
