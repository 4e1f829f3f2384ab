import sys, numpy as np, torch
sys.path.insert(0, "torch-nerf_amd")
from torch_nerf.amd import ops, synth
def tf_index(width, M):
    m = np.arange(M)[:, None]; k = np.arange(width)[None, :]
    return (m >> 5) * 32 * width + ((((k >> 5) * 4 + ((k >> 3) & 3)) << 8)) + 4 * ((2 * (m & 31) + ((k >> 2) & 1)) ^ (2 * ((k >> 3) & 3))) + (k & 3)
flat = torch.from_numpy(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)).cuda()
p32, px = ops.mlp_pack(flat), ops.mlp_pack_f16x2(flat)
for M in [int(a) for a in sys.argv[1:]] or [1000, 5000, 20001, 40000]:
    rng = np.random.RandomState(M)
    xs = torch.from_numpy(rng.uniform(-3, 3, (M, 3)).astype(np.float32)).cuda()
    vs = torch.from_numpy(rng.uniform(-1, 1, (M, 3)).astype(np.float32)).cuda()
    s32, c32, r32 = ops.mlp_forward(p32, xs, vs, False, save=True)
    for rep in range(3):
        sx, cx, rx = ops.mlp_forward_f16x2(px, xs, vs, save=True)
        a, b = r32.cpu().numpy(), rx.cpu().numpy()
        MP = (M + 127) // 128 * 128
        planes = [("pe", 0, 64)] + [(f"h{l}", MP * (64 + 256 * l), 256) for l in range(8)] + [("y8", MP * (64 + 2048), 256), ("h9", MP * (64 + 2304), 128), ("de", MP * (64 + 2304 + 128), 32)]
        msg = []
        for name, off, w in planes:
            idx = tf_index(w, M)
            d = np.abs(a[off + idx] - b[off + idx]) / np.maximum(np.abs(a[off + idx]), 1.0)
            bad = np.argwhere(d > 1e-5)
            if len(bad): msg.append(f"{name}: {len(bad)} bad, rows {sorted(set(bad[:,0]))[:6]} cols {sorted(set(bad[:,1]))[:8]} max {d.max():.3g}")
        ma = a.view(np.uint32)[MP * 2528:].reshape(9, MP, 2, 4)[:, :M]; mb = b.view(np.uint32)[MP * 2528:].reshape(9, MP, 2, 4)[:, :M]
        x = ma ^ mb
        nbits = int(np.unpackbits(x.view(np.uint8)).sum())
        rows = sorted(set(np.argwhere(x != 0)[:, 1]))[:8]
        print(M, rep, "outputs", float((cx - c32).abs().max()), "| planes:", msg or "ok", "| mask bits differing", nbits, rows)
