"""Throughput of the layered family (csrc/mlp_layered.hip) next to the fused one on the same network."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import numpy as np, torch
from torch_nerf.amd import ops, synth
torch.cuda.set_device(0)
def t(fn, window_ms=120.0):   # mean over ~window_ms of back-to-back calls (as bench.py's net_variants leg)
    def run(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    fn(); torch.cuda.synchronize()
    return run(max(3, min(200, int(window_ms / max(run(2), 1e-3)))))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
NETS = ((63, 27, 256), (75, 27, 256), (99, 27, 256), (63, 27, 128), (63, 27, 64), (75, 39, 64), (40, 40, 160), (63, 27, 512))
if len(sys.argv) > 2:   # e.g. "128,512": feat_dims to run (profiling passes)
    NETS = tuple(n for n in NETS if str(n[2]) in sys.argv[2].split(",") or f"{n[0]}x{n[2]}" in sys.argv[2].split(","))
PEAK = 157.3
for (e_p, e_d, F) in NETS:
    net = ops.Net.dims_only(e_p, e_d, F)
    flat = torch.from_numpy(synth.nerf_flat_params(seed=1, pos_dim=e_p, view_dir_dim=e_d, feat_dim=F)).cuda()
    pe, de = torch.randn(M, e_p, device="cuda"), torch.randn(M, e_d, device="cuda")
    gs, gc = torch.randn(M, device="cuda"), torch.randn(M, 3, device="cuda")
    H = F // 2
    mac = e_p * F + 4 * F * F + (F + e_p) * F + 2 * F * F + F * (F + 1) + (F + e_d) * H + 3 * H
    fwd = t(lambda: ops.mlp_layered_forward(flat, pe, de, net))
    sigma, rgb, rec = ops.mlp_layered_forward(flat, pe, de, net, record=True)
    bwd = t(lambda: ops.mlp_layered_backward(flat, pe, de, net, sigma, rgb, rec, gs, gc))
    rfwd = t(lambda: ops.mlp_layered_forward(flat, pe, de, net, record=True))
    line = (f"NeRF({e_p},{e_d},{F}) M={M}: layered fwd {fwd:7.2f} ms = {2*mac*M/fwd/1e9:6.1f} TFLOP/s ({2*mac*M/fwd/1e9/PEAK:.2f})   "
            f"record fwd {rfwd:7.2f} ms ({2*mac*M/rfwd/1e9/PEAK:.2f})   bwd {bwd:7.2f} ms = {4*mac*M/bwd/1e9:6.1f} TFLOP/s ({4*mac*M/bwd/1e9/PEAK:.2f})")
    if net.fused:
        packed = ops.mlp_pack(flat, net)
        f2 = t(lambda: ops.mlp_forward(packed, pe, de, True, net=net))
        line += f"   | fused fwd {f2:6.2f} ms = {2*mac*M/f2/1e9:6.1f} TFLOP/s"
    print(line, flush=True)
