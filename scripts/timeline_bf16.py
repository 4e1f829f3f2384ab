"""Cycle timeline of workgroup 0 of the bf16 kernel (needs the X_TIMELINE variant: NERF_AMD_LIB=.../lib_TL.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import numpy as np, torch
from torch_nerf.amd import ops, synth
flat = torch.from_numpy(synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)).cuda()
packed = ops.mlp_pack_bf16(flat)
M = 4096 * 192
pts = torch.rand(M, 3, device="cuda") * 8 - 4
dirs = torch.rand(M, 3, device="cuda") * 2 - 1
from torch_nerf.amd import _lib
lib = _lib.load()
sigma = torch.empty(M, device="cuda")
rgb = torch.zeros(3 * M + 65536, device="cuda")        # stamps land behind the 3 M colours
for _ in range(3):
    rgb[3 * M:].zero_()
    lib.nerf_mlp_forward_bf16(None, packed.data_ptr(), pts.data_ptr(), dirs.data_ptr(), M, sigma.data_ptr(), rgb.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
t = rgb[3 * M:].view(torch.int64).cpu().numpy()
t = t[t != 0]
per_tile = int(sys.argv[1]) if len(sys.argv) > 1 else 1 + 5 + 8 * 5
n = len(t) // per_tile
print("stamps", len(t), "tiles", n)
t = t[: n * per_tile].reshape(n, per_tile)
d = np.diff(t, axis=1)[1:-1].mean(0)
print("tile period", np.diff(t[:, 0]).mean(), "stamps/tile", per_tile)
names = ["load raw+enc+arm", "acq fc_in", "mma fc_in A", "mma fc_in B (+drain A)", "arm S0"]
for l in range(1, 9):
    names += [f"L{l}: (enc)", "  (pos A) + acquire A", "  mma A (+drain B')", "  (pos B) + acquire B", "  mma B (+drain A)"]
for k, v in enumerate(d):
    print(f"{names[k] if k < len(names) else str(k):24s} {v:9.0f}")
print("rest of tile (fc_5, fc_9, heads)", (t[1:, 0] - t[:-1, -1]).mean())
