"""Cycle timeline of one workgroup of the forward kernel (needs the instrumented variant library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch, numpy as np
from torch_nerf.amd import ops, synth, _lib
flat = synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)
packed = ops.mlp_pack(torch.from_numpy(flat).cuda())
lib = _lib.load()
M = 4096 * 192
pts = torch.rand(M, 3, device="cuda") * 8 - 4; dirs = torch.rand(M, 3, device="cuda") * 2 - 1
sigma = torch.empty(M, device="cuda"); rgb = torch.empty(M, 3, device="cuda")
dbg = torch.zeros(4096, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    dbg.zero_()
    lib.nerf_mlp_forward(packed.data_ptr(), pts.data_ptr(), dirs.data_ptr(), M, 0, sigma.data_ptr(), rgb.data_ptr(), dbg.data_ptr(), st)
    torch.cuda.synchronize()
t = dbg.cpu().numpy(); t = t[t != 0]
per_tile = 3 + 8 * 8 + 2
n = len(t) // per_tile
t = t[:n * per_tile].reshape(n, per_tile)
d = np.diff(t, axis=1)
names = ["encodings", "fc_in issue"] + [x for l in range(1, 9) for x in (f"L{l-1} tail -> acquire L{l}", f"L{l} seam", f"L{l} pair0 mfma", "  acquire1", "  pair1 mfma", "  acquire2", "  pair2 mfma", "  acquire3")] + ["L8 pair3 mfma", "fc_9 + heads"]
print("tiles", n, "tile period (ticks):", np.diff(t[:, 0]).mean() if n > 1 else None)
for k in range(d.shape[1]):
    print(f"{names[k] if k < len(names) else k:28s} mean {d[1:, k].mean():10.0f}")
print("end of tile -> next tile start", (t[1:, 0] - t[:-1, -1]).mean())
