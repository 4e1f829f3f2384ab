"""Is the record of the pre-encoded forward deterministic, and where does it differ from the raw-input one? (debugging aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import numpy as np, torch
from oracle import oracle as O
from torch_nerf.amd import ops, synth
O.build()
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
flat = synth.nerf_flat_params(seed=5, sigma_bias=0.5, sigma_gain=4.0)
fp = dev(flat); packed = ops.mlp_pack(fp)
M = 1000; MP = 1024
rng = np.random.RandomState(M)
pts = rng.uniform(-3, 3, (M, 3)).astype(np.float32); dirs = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
pe, de = dev(O.posenc(pts, 10)), dev(O.posenc(dirs, 4))
gs, gc = dev(rng.standard_normal(M).astype(np.float32)), dev(rng.standard_normal((M, 3)).astype(np.float32))
recs = []
for it in range(3):
    sigma, rgb, saved = ops.mlp_forward(packed, pe, de, True, save=True)
    torch.cuda.synchronize()
    recs.append(saved.clone())
_, _, raw = ops.mlp_forward(packed, dev(pts), dev(dirs), False, save=True)
planes = [("PE", 0, 64)] + [(f"H{l}", MP * (64 + 256 * l), 256) for l in range(8)] + [("Y8", MP * (64 + 2048), 256), ("H9", MP * (64 + 2304), 128), ("DE", MP * (64 + 2304 + 128), 32)]
for name, off, w in planes:
    a, b, c = (r[off:off + MP * w] for r in recs)
    r0 = raw[off:off + MP * w]
    print(name, "enc run0==run1", bool(torch.equal(a, b)), "run1==run2", bool(torch.equal(b, c)), "max|enc-raw|", float((a - r0).abs().max()),
          "n differing run0/run1", int((a != b).sum()))
# backward determinism on ONE record
g = [ops.mlp_backward(packed, fp, pe, de, True, sigma, rgb, recs[0], gs, gc) for _ in range(3)]
print("backward on the same record: equal", bool(torch.equal(g[0], g[1])), bool(torch.equal(g[1], g[2])))
g2 = [ops.mlp_backward(packed, fp, pe, de, True, sigma, rgb, raw, gs, gc) for _ in range(3)]
print("backward on the raw-input record: equal", bool(torch.equal(g2[0], g2[1])), bool(torch.equal(g2[1], g2[2])), "vs enc record", float((g[0]-g2[0]).abs().max()))
for name, off, w in (planes[0], planes[-1]):
    for j in (1, 2):
        a, b = recs[0][off:off + MP * w], recs[j][off:off + MP * w]
        idx = torch.nonzero(a != b).flatten().cpu().numpy()
        if idx.size:
            print(name, "run0 vs run", j, "differing floats", idx.size, "range", idx.min(), idx.max(), "-> 32-sample tile", idx.min() // (32 * w),
                  "slot in tile", (idx.min() % (32 * w)) // 256, "values a", a[idx[:3]].tolist(), "b", b[idx[:3]].tolist(), "raw", raw[off:off + MP * w][idx[:3]].tolist())
