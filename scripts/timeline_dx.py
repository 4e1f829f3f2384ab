"""Cycle timeline of workgroup 0 of the backward dX-chain kernel (needs an -DX_TIMELINE build: NERF_AMD_LIB=...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import numpy as np, torch
from torch_nerf.amd import ops, synth, _lib
flat = torch.from_numpy(synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)).cuda()
packed = ops.mlp_pack(flat)
lib = _lib.load()
M = 4096 * 192
pts = torch.rand(M, 3, device="cuda") * 8 - 4
dirs = torch.rand(M, 3, device="cuda") * 2 - 1
gs = torch.randn(M, device="cuda"); gc = torch.randn(M, 3, device="cuda")
sigma, rgb, saved = ops.mlp_forward(packed, pts, dirs, False, save=True)
ws = torch.zeros(lib.nerf_mlp_backward_workspace_bytes(None, M) // 4, device="cuda")
g = torch.empty(lib.nerf_mlp_param_count(None), device="cuda")
for _ in range(2):
    ws.zero_()
    rc = lib.nerf_mlp_backward(None, packed.data_ptr(), flat.data_ptr(), pts.data_ptr(), dirs.data_ptr(), M, 0, sigma.data_ptr(),
                               rgb.data_ptr(), saved.data_ptr(), gs.data_ptr(), gc.data_ptr(), g.data_ptr(), None, None, ws.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
MP = (M + 127) // 128 * 128
off = (MP * 2533 + 255) // 256 * 256 + (2 * 256 + 13) * (256 * 256 + 256) - 65536
t = ws[off:off + 65536].view(torch.int64).cpu().numpy()
t = t[t != 0]
per_tile = 5 + 8 * 4 + 1
n = len(t) // per_tile
print("stamps", len(t), "tiles", n)
t = t[: n * per_tile].reshape(n, per_tile)
d = np.diff(t, axis=1)[1:-1].mean(0)
print("tile period", np.diff(t[:, 0]).mean())
names = ["head: loads issued, gy", "head: acquire (vmcnt(0): the loads, the previous tile's dY0 stores)", "head: dY9 on the vector ALU", "head: fc_9^T (2 pairs)"]
for l in range(8, 0, -1):
    names += [f"L{l}: tail of previous pairs", "  acquire", "  seam (mask, store dY, zero acc)", "  pair 0", ]
    names[-4] = f"L{l}: pairs 1-3 of previous layer" if l < 8 else "L8: -"
for k, v in enumerate(d):
    print(f"{names[k] if k < len(names) else str(k):44s} {v:9.0f}")
print("tile end (pairs 1-3 of L1, dY0) -> next tile", (t[1:, 0] - t[:-1, -1]).mean())
