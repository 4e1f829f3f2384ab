"""Cycle timeline of workgroup 0 of the fp32 forward kernel (needs an -DX_TIMELINE build of mlp_forward.hip:
NERF_AMD_LIB=<variant .so>).  Usage: python scripts/timeline.py [save]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import numpy as np, torch
from torch_nerf.amd import ops, synth, _lib
save = len(sys.argv) > 1 and sys.argv[1] == "save"
flat = torch.from_numpy(synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)).cuda()
packed = ops.mlp_pack(flat)
lib = _lib.load()
M = 4096 * 192
pts = torch.rand(M, 3, device="cuda") * 8 - 4
dirs = torch.rand(M, 3, device="cuda") * 2 - 1
sigma = torch.empty(M, device="cuda")
rgb = torch.zeros(3 * M + 65536, device="cuda")          # stamps land behind the 3 M colours
saved = torch.empty(lib.nerf_mlp_saved_bytes(None, M) // 4, device="cuda") if save else None
for _ in range(3):
    rgb[3 * M:].zero_()
    rc = lib.nerf_mlp_forward(None, packed.data_ptr(), pts.data_ptr(), dirs.data_ptr(), M, 0, sigma.data_ptr(), rgb.data_ptr(),
                              saved.data_ptr() if save else None, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
t = rgb[3 * M:].view(torch.int64).cpu().numpy()
t = t[t != 0]
per_tile = 2 + 8 * 4 + 2
n = len(t) // per_tile
print("record mode" if save else "inference", "stamps", len(t), "tiles", n)
t = t[: n * per_tile].reshape(n, per_tile)
d = np.diff(t, axis=1)[1:-1].mean(0)
print("tile period", np.diff(t[:, 0]).mean(), " ideal MFMA cycles/tile", 9280 * 64)
names = ["encodings", "fc_in: acquire + bias + pair"]
for l in range(1, 9):
    names += [f"L{l}: acquire", "  seam (ReLU, record, bias)", "  pair 0 (+pos pair for L5)", "  pairs 1-3"]
names += ["fc_9 (4.5 pairs)", ]
for k, v in enumerate(d):
    print(f"{names[k] if k < len(names) else str(k):36s} {v:9.0f}")
print("heads + tile turnover", (t[1:, 0] - t[:-1, -1]).mean())
