import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
from torch_nerf.amd import ops, synth, _lib
import ctypes
flat = synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)
packed = ops.mlp_pack(torch.from_numpy(flat).cuda())
lib = _lib.load()
for M in (4096*64, 4096*192):
    pts = torch.rand(M, 3, device="cuda") * 8 - 4
    dirs = torch.rand(M, 3, device="cuda") * 2 - 1
    sigma = torch.empty(M, device="cuda"); rgb = torch.empty(M, 3, device="cuda")
    saved = torch.empty(lib.nerf_mlp_saved_bytes(None, M) // 4, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ts = []
    for it in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.nerf_mlp_forward(None, packed.data_ptr(), pts.data_ptr(), dirs.data_ptr(), M, 0, sigma.data_ptr(), rgb.data_ptr(), saved.data_ptr(), st)
        e1.record(); torch.cuda.synchronize(); ts.append(round(e0.elapsed_time(e1), 3))
    print(M, ts)
