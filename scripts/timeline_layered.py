"""Where workgroup 0 / wavefront 0 of the layered family's GENERAL kernel spends its cycles (needs an
-DX_LAYERED_TIMELINE build: make EXTRA=-DX_LAYERED_TIMELINE BUILD=build_lt OUT=../lib/variants/lt.so, then
NERF_AMD_LIB=torch-nerf_amd/lib/variants/lt.so python scripts/timeline_layered.py 63,27,512 786432)."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
from torch_nerf.amd import _lib, ops, synth
dims = tuple(int(v) for v in sys.argv[1].split(","))
M = int(sys.argv[2]) if len(sys.argv) > 2 else 786432
lib = _lib.load()
lib.nerf_debug_layered_timeline.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 8)()
net = ops.Net.dims_only(*dims)
flat = torch.from_numpy(synth.nerf_flat_params(seed=1, pos_dim=dims[0], view_dir_dim=dims[1], feat_dim=dims[2])).cuda()
pe, de = torch.randn(M, dims[0], device="cuda"), torch.randn(M, dims[1], device="cuda")
gs, gc = torch.randn(M, device="cuda"), torch.randn(M, 3, device="cuda")
names = ["store drain", "operand issue + init", "acquire waits", "MFMA loops", "epilogue", "pass program", "passes"]
def show(tag):
    lib.nerf_debug_layered_timeline(buf, 1)
    v = np.array(buf[:7], dtype=np.float64)
    n = max(v[6], 1)
    tot = v[:6].sum()
    print(f"{tag}: {int(n)} passes of wavefront 0, {tot / n:.0f} cycles per pass: " +
          ", ".join(f"{names[k]} {v[k] / n:.0f} ({100 * v[k] / tot:.1f} %)" for k in range(6)))
sigma, rgb, rec = ops.mlp_layered_forward(flat, pe, de, net, record=True)
lib.nerf_debug_layered_timeline(buf, 1)
sigma, rgb, rec = ops.mlp_layered_forward(flat, pe, de, net, record=True)
show(f"NeRF{dims} forward")
ops.mlp_layered_backward(flat, pe, de, net, sigma, rgb, rec, gs, gc)
lib.nerf_debug_layered_timeline(buf, 1)
ops.mlp_layered_backward(flat, pe, de, net, sigma, rgb, rec, gs, gc)
show(f"NeRF{dims} reverse chain")
