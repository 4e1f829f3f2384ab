#!/usr/bin/env python3
"""How far the whole-chain gradients of fixture F7 are from the reference's, tensor by tensor, and how many samples'
ReLU decisions differ (the numbers behind the tolerance of tests/test_gpu_backward.py::test_training_step_gradients_match_reference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import test_gpu_backward as T
from helpers import fused_masks, relu_row_hashes
from torch_nerf.amd import ops, synth

g = np.load(os.path.join(ROOT, "tests/golden/f7_e2e.npz"))
H, W, focal, near, far = g["meta"]
cam = T.cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H}, torch.from_numpy(g["pose"]), float(near), float(far))
enc = {"coord_enc": T.PositionalEncoder(3, 10, True), "dir_enc": T.PositionalEncoder(3, 4, True)}
nets = [T.make_net(synth.nerf_flat_params(seed=s, sigma_bias=1.0, sigma_gain=30.0)) for s in (3, 4)]
vr = T.VolumeRenderer(T.integrators.QuadratureIntegrator(), T.ray_samplers.StratifiedSampler(), cam)
pix = torch.from_numpy(g["pix"]); gt = T.dev(g["gt"])
torch.rand = T._Replay([g["u1c"], g["u1"], g["u2"], g["u3"]])
records = []; real = ops.mlp_forward
def rec(packed, pos, vd, encoded, save=False, net=None):
    out = real(packed, pos, vd, encoded, save=save, net=net)
    if save: records.append((out[2], out[0], pos.shape[0]))
    return out
ops.mlp_forward = rec
c_rgb, c_idx, c_w = vr.render_scene(T.scene.PrimitiveCube(nets[0], enc), len(pix), 64, False, 0, pixel_indices=pix)
f_rgb, _, _ = vr.render_scene(T.scene.PrimitiveCube(nets[1], enc), len(pix), (64, 128), False, 0, pixel_indices=c_idx, weights=T.dev(g["coarse_w"]))
mse = torch.nn.MSELoss(); (mse(gt, c_rgb) + mse(gt, f_rgb)).backward()
for (saved, sigma, M), tag, net in zip(records, ("coarse", "fine"), nets):
    diff = int((relu_row_hashes(fused_masks(saved, sigma, M)) != g[tag + "_relu_hash"]).sum())
    print(tag, "samples with differing decisions:", diff, "of", M)
    grads = synth.split_flat_params(T.flat_grad(net))
    for k, v in grads.items():
        v = v.reshape(-1); pre = tag + "_grad_" + k
        norm_ref = float(g[pre + ".norm"][0]); rms = norm_ref / np.sqrt(v.size)
        head, st = g[pre + ".head"], g[pre + ".stride"]; step = max(1, v.size // 192)
        got = np.concatenate([v[:head.size], v[::step][:st.size]]); ref = np.concatenate([head, st])
        e = np.abs(got - ref)
        print(f"  {k:14s} rms {rms:.2e} max|err|/rms {e.max()/rms:.2e}  max(|err|-2e-5|ref|)/rms {np.maximum(e-2e-5*np.abs(ref),0).max()/rms:.2e}  norm rel {abs(np.linalg.norm(v.astype(np.float64))-norm_ref)/norm_ref:.2e}")
