"""CPU emulation of split-operand MLP arithmetic on the 16-bit matrix pipe -- the go / no-go table of DESIGN 4.4.

Question (round-5 verdict, item 1): can the eleven layers of NeRF.forward (nerf.py:102-119) run on the bf16 / f16 MFMA
pipe (16x the fp32 MFMA rate) and still meet north_star's 1e-5 bound on sigma / rgb / pixels?  Every operand of every
matrix-pipe layer is split into k parts of a 16-bit type (part_0 = round(x), part_1 = round(x - part_0), ...), a chosen
set of part products is formed (exact in fp32: 8 x 8 or 11 x 11 significand bits) and accumulated in fp32, exactly as
`v_mfma_f32_32x32x16_{bf16,f16}` would.  Arrangements:

  bf16x2/3p   two bf16 parts, products hi.hi + hi.lo + lo.hi              (3 MFMAs per fp32 MFMA-equivalent)
  bf16x3/6p   three bf16 parts, the six products with i + j <= 2          (6)
  f16x2/3p    two f16 parts (11 + 11 significand bits), hi.hi + hi.lo + lo.hi   (3); weights pre-scaled per layer by a
              power of two so that their low parts stay normal; `ftz` = the same with f16 subnormal parts flushed
              (what a pipe without subnormal support would do)
  f16x1       plain f16 operands (1) -- for scale
  bf16x1      plain bf16 operands (1) -- the shipped configs[2] arithmetic

What stays fp32 in every arrangement (as in csrc/mlp_forward_bf16.hip): encodings, biases, accumulation, the density row
of fc_8 (from the unrounded h7), fc_out and the sigmoid.

Inputs: goldens F5 (both weight sets), F11 (the feat-256 variants) and F7 end to end through oracle/torch_port.py with
the MLP swapped for the emulation (pixels, compositing weights, fine-bin flips against the port's own indices).
Runs in the build container only (torch CPU); nothing here is imported by the product.

    python scripts/split_emulate.py            # prints the table, writes profiles/r06_split_emulate.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "torch-nerf_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import torch_port as TP          # noqa: E402  (test infrastructure: this script is a checker)
from torch_nerf.amd import synth             # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
F16_MIN_NORMAL = 2.0 ** -14


def parts(x, dtype, k, ftz=False):
    """x (fp32) -> k fp32 tensors holding its successive `dtype` roundings (round-to-nearest-even)."""
    out, r = [], x
    for _ in range(k):
        p = r.to(dtype).to(torch.float32)
        if ftz:
            p = torch.where(p.abs() < F16_MIN_NORMAL, torch.zeros_like(p), p)
        out.append(p)
        r = r - p
    return out


class Arrangement:
    def __init__(self, name, dtype, k, keep, scale_weights=False, ftz=False, mfma_per_product=None):
        self.name, self.dtype, self.k, self.keep = name, dtype, k, keep
        self.scale_weights, self.ftz = scale_weights, ftz
        self.products = len(keep)

    def weight_scale(self, w):
        """Power of two that lifts max|w| to [2^13, 2^14): the low f16 part of a weight then sits ~2^-11 below, in the
        normal range for every weight above 2^-16 of the largest."""
        if not self.scale_weights:
            return 1.0
        m = float(w.abs().max())
        return 2.0 ** (13 - int(np.floor(np.log2(m)))) if m > 0 else 1.0

    def linear(self, x, w, b):
        """x (M, K) @ w (O, K)^T + b with split operands; fp32 accumulation."""
        s = self.weight_scale(w)
        xp = parts(x, self.dtype, self.k, self.ftz)
        wp = parts(w * s, self.dtype, self.k, self.ftz)
        acc = None
        for i, j in self.keep:            # small terms first would be marginally better; the MFMA chain adds as it goes
            t = xp[i] @ wp[j].t()
            acc = t if acc is None else acc + t
        return acc * (1.0 / s) + b


ARRANGEMENTS = [
    Arrangement("bf16x1/1p", torch.bfloat16, 1, [(0, 0)]),
    Arrangement("f16x1/1p", torch.float16, 1, [(0, 0)], scale_weights=True),
    Arrangement("bf16x2/3p", torch.bfloat16, 2, [(0, 0), (0, 1), (1, 0)]),
    Arrangement("bf16x3/6p", torch.bfloat16, 3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
    Arrangement("f16x2/3p", torch.float16, 2, [(0, 0), (0, 1), (1, 0)], scale_weights=True),
    Arrangement("f16x2/3p ftz", torch.float16, 2, [(0, 0), (0, 1), (1, 0)], scale_weights=True, ftz=True),
    Arrangement("f16x2/3p noscale", torch.float16, 2, [(0, 0), (0, 1), (1, 0)], scale_weights=False),
    Arrangement("f16x2/4p", torch.float16, 2, [(0, 0), (0, 1), (1, 0), (1, 1)], scale_weights=True),
]


def mlp_split(arr, p, pos, view):
    """nerf.py:102-119 with the matrix-pipe layers through `arr`; density row, fc_out, sigmoid in fp32."""
    def lin(name, x):
        return arr.linear(x, p[name + ".weight"], p[name + ".bias"])

    x = torch.relu(lin("fc_in", pos))
    for name in ("fc_1", "fc_2", "fc_3", "fc_4"):
        x = torch.relu(lin(name, x))
    x = torch.cat([pos, x], -1)
    for name in ("fc_5", "fc_6", "fc_7"):
        x = torch.relu(lin(name, x))
    w8, b8 = p["fc_8.weight"], p["fc_8.bias"]
    sigma = torch.relu(x @ w8[0] + b8[0])                                   # fp32 vector ALU row
    x8 = arr.linear(x, w8[1:], b8[1:])
    x = torch.relu(lin("fc_9", torch.cat([x8, view], -1)))
    return sigma, torch.sigmoid(torch.nn.functional.linear(x, p["fc_out.weight"], p["fc_out.bias"]))


def tparams(flat, dims=(63, 27, 256)):
    return {k: torch.from_numpy(v.copy()) for k, v in synth.split_flat_params(flat, *dims).items()}


def encode(x, levels, include_input=True):
    feats = [x] if include_input else []
    for lv in range(levels):
        f = float(2 ** lv)
        feats += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(feats, -1)


def run():
    torch.set_num_threads(8)
    rows = []
    g5 = np.load(os.path.join(GOLDEN, "f5_mlp.npz"))
    g11 = np.load(os.path.join(GOLDEN, "f11_net_variants.npz"))
    g7 = np.load(os.path.join(GOLDEN, "f7_e2e.npz"))
    pe5, de5 = encode(torch.from_numpy(g5["pts"]), 10), encode(torch.from_numpy(g5["dirs"]), 4)
    variants = {"l6_l2": (6, 2, True), "l4_l4": (4, 4, True), "l10_l4_noinput": (10, 4, False)}

    # the port's own F7 run: its fine-bin indices are the reference's (tests/test_torch_port.py)
    H, W, focal, near, far = g7["meta"]
    draws = tuple(torch.from_numpy(g7[k]) for k in ("u1c", "u1", "u2", "u3"))
    pc = tparams(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0))
    pf = tparams(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0))

    def f7(mlp):
        saved = TP.mlp
        TP.mlp = mlp
        try:
            with torch.no_grad():
                return TP.render_batch(pc, pf, torch.from_numpy(g7["pix"]), int(H), int(W), float(focal),
                                       torch.from_numpy(g7["pose"]), float(near), float(far), 64, 128, draws)
        finally:
            TP.mlp = saved

    ref7 = f7(TP.mlp)
    assert np.abs(ref7[2].numpy() - g7["fine_rgb"]).max() <= 1e-5

    with torch.no_grad():
        for arr in ARRANGEMENTS:
            row = {"arrangement": arr.name, "products": arr.products}
            ds = dc = 0.0
            for tag, kw in (("default", dict(seed=1)), ("dense", dict(seed=2, sigma_bias=1.0, sigma_gain=30.0))):
                s, c = mlp_split(arr, tparams(synth.nerf_flat_params(**kw)), pe5, de5)
                ds = max(ds, float(np.abs(s.numpy() - g5[tag + "_sigma"]).max()))
                dc = max(dc, float(np.abs(c.numpy() - g5[tag + "_rgb"]).max()))
                # sigma spans 0 .. ~40 in `dense`: also relative to the value
                row["f5_" + tag + "_sigma_rel"] = float((np.abs(s.numpy() - g5[tag + "_sigma"]) /
                                                         np.maximum(np.abs(g5[tag + "_sigma"]), 1.0)).max())
            row["f5_sigma"], row["f5_rgb"] = ds, dc
            ds = dc = 0.0
            for tag, (lp, ld, inc) in variants.items():
                e_p, e_d, feat = (int(v) for v in g11[tag + "_dims"][:3])
                flat = synth.nerf_flat_params(seed=5, pos_dim=e_p, view_dir_dim=e_d, feat_dim=feat, sigma_bias=0.5,
                                              sigma_gain=4.0)
                s, c = mlp_split(arr, tparams(flat, (e_p, e_d, feat)), torch.from_numpy(g11[tag + "_pe"]),
                                 torch.from_numpy(g11[tag + "_de"]))
                ds = max(ds, float(np.abs(s.numpy() - g11[tag + "_sigma"]).max()))
                dc = max(dc, float(np.abs(c.numpy() - g11[tag + "_rgb"]).max()))
            row["f11_sigma"], row["f11_rgb"] = ds, dc
            c_rgb, c_w, f_rgb, f_w, idx = f7(lambda p, a, b, arr=arr: mlp_split(arr, p, a, b))
            row["f7_coarse_pixels"] = float(np.abs(c_rgb.numpy() - g7["coarse_rgb"]).max())
            row["f7_coarse_w"] = float(np.abs(c_w.numpy() - g7["coarse_w_after"]).max())
            row["f7_fine_pixels"] = float(np.abs(f_rgb.numpy() - g7["fine_rgb"]).max())
            row["f7_fine_w"] = float(np.abs(f_w.numpy() - g7["fine_w"]).max())
            row["f7_bin_flips"] = int((idx != ref7[4]).sum())
            row["f7_bins"] = int(idx.numel())
            worst = max(row["f5_rgb"], row["f11_rgb"], row["f11_sigma"], row["f7_coarse_pixels"], row["f7_fine_pixels"],
                        row["f5_default_sigma_rel"], row["f5_dense_sigma_rel"])
            row["worst"] = worst
            row["clears_3e-6"] = bool(worst <= 3e-6)
            row["clears_1e-5"] = bool(worst <= 1e-5)
            rows.append(row)
    return rows


def main():
    rows = run()
    cols = ["f5_sigma", "f5_rgb", "f11_sigma", "f11_rgb", "f7_coarse_pixels", "f7_fine_pixels", "f7_fine_w"]
    print("%-18s %2s  " % ("arrangement", "P") + " ".join("%-16s" % c for c in cols) + " flips  <=3e-6 <=1e-5")
    for r in rows:
        print("%-18s %2d  " % (r["arrangement"], r["products"]) + " ".join("%-16.3g" % r[c] for c in cols) +
              " %d/%d  %s %s" % (r["f7_bin_flips"], r["f7_bins"], r["clears_3e-6"], r["clears_1e-5"]))
    out = os.path.join(ROOT, "profiles", "r06_split_emulate.json")
    with open(out, "w") as f:
        json.dump(rows, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
