"""Default network, a call whose inputs require grad (NeRF.forward's record forward + the input-gradient dX chain + dW):
time and fraction of the fp32 MFMA peak, next to the plain training call.  FLOPs: forward 2 mac M, backward 4 mac M
(+ 2 ig M for the three input-gradient GEMMs: fc_in^T, fc_5[:, :E_p]^T, fc_9[:, 256:]^T)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
from torch_nerf.amd import ops, synth
torch.cuda.set_device(0)
PEAK = 157.3e12
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = int(sys.argv[1]) if len(sys.argv) > 1 else 786432
e_p, e_d, F, H = 63, 27, 256, 128
mac = e_p * F + 4 * F * F + (F + e_p) * F + 2 * F * F + F * (F + 1) + (F + e_d) * H + 3 * H
ig = 2 * e_p * F + e_d * H
flat = torch.from_numpy(synth.nerf_flat_params(seed=1)).cuda()
packed = ops.mlp_pack(flat)
pe, de = torch.randn(M, e_p, device="cuda"), torch.randn(M, e_d, device="cuda")
gs, gc = torch.randn(M, device="cuda"), torch.randn(M, 3, device="cuda")
sigma, rgb, saved = ops.mlp_forward(packed, pe, de, True, save=True)
fwd = t(lambda: ops.mlp_forward(packed, pe, de, True, save=True))
bwd = t(lambda: ops.mlp_backward(packed, flat, pe, de, True, sigma, rgb, saved, gs, gc))
bwd_ig = t(lambda: ops.mlp_backward(packed, flat, pe, de, True, sigma, rgb, saved, gs, gc, want_pos=True, want_dir=True))
print(f"NeRF(63,27,256) M={M}: record forward {fwd:.3f} ms = {2*mac*M/fwd/1e9/PEAK*1e12:.3f} of peak")
print(f"  backward (parameters)          {bwd:.3f} ms = {4*mac*M/bwd/1e9/PEAK*1e12:.3f}")
print(f"  backward (parameters + inputs) {bwd_ig:.3f} ms = {(4*mac+2*ig)*M/bwd_ig/1e9/PEAK*1e12:.3f}")
print(f"  input-gradient call, forward + backward: {fwd+bwd_ig:.3f} ms = {(6*mac+2*ig)*M/(fwd+bwd_ig)/1e9/PEAK*1e12:.3f} of the fp32 MFMA peak")
