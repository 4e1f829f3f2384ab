import sys, numpy as np, torch
sys.path.insert(0, "torch-nerf_amd")
from torch_nerf.amd import ops, synth, _lib
lib = _lib.load()
def tf_index(width, M):
    m = np.arange(M)[:, None]; k = np.arange(width)[None, :]
    return (m >> 5) * 32 * width + ((((k >> 5) * 4 + ((k >> 3) & 3)) << 8)) + 4 * ((2 * (m & 31) + ((k >> 2) & 1)) ^ (2 * ((k >> 3) & 3))) + (k & 3)
flat = torch.from_numpy(synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)).cuda()
p32, px = ops.mlp_pack(flat), ops.mlp_pack_f16x2(flat)
P = lambda t: None if t is None else t.data_ptr()
for M in [int(a) for a in sys.argv[1:]] or [128, 1000]:
    rng = np.random.RandomState(M)
    xs = torch.from_numpy(rng.uniform(-3, 3, (M, 3)).astype(np.float32)).cuda()
    vs = torch.from_numpy(rng.uniform(-1, 1, (M, 3)).astype(np.float32)).cuda()
    gs = torch.from_numpy(rng.standard_normal(M).astype(np.float32)).cuda()
    gc = torch.from_numpy(rng.standard_normal((M, 3)).astype(np.float32)).cuda()
    s, c, rec = ops.mlp_forward(p32, xs, vs, False, save=True)
    nb = lib.nerf_mlp_backward_workspace_bytes(None, M)
    outs = []
    for which in (0, 1):
        ws = torch.zeros(nb // 4, dtype=torch.float32, device="cuda")
        gp = torch.empty(595844, dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        if which == 0:
            rc = lib.nerf_mlp_backward(None, P(p32), P(flat), P(xs), P(vs), M, 0, P(s), P(c), P(rec), P(gs), P(gc), P(gp), None, None, P(ws), st)
        else:
            rc = lib.nerf_mlp_backward_f16x2(None, P(p32), P(px), M, P(s), P(c), P(rec), P(gs), P(gc), P(gp), P(ws), st)
        torch.cuda.synchronize(); assert rc == 0, lib.nerf_amd_last_error()
        outs.append((ws.cpu().numpy(), gp.cpu().numpy()))
    (a, ga), (b, gb) = outs
    MP = (M + 127) // 128 * 128
    planes = [("dy9", MP * 256 * 9, 128)] + [(f"dy{l}", MP * 256 * l, 256) for l in (8, 7, 6, 5, 4, 3, 2, 1, 0)]
    for name, off, w in planes:
        idx = tf_index(w, M)
        pa, pb = a[off + idx], b[off + idx]
        d = np.abs(pa - pb); rel = d.max() / (np.abs(pa).max() + 1e-30)
        bad = np.argwhere(d > 1e-5 * np.abs(pa).max())
        print(M, name, "max|fp32|", float(np.abs(pa).max()), "max diff rel", float(rel), "bad", len(bad), "rows", sorted(set(bad[:, 0]))[:5], "cols", sorted(set(bad[:, 1]))[:10])
    o = MP * (256 * 9 + 128)
    print("dsig diff", np.abs(a[o:o + M] - b[o:o + M]).max(), "gy diff", np.abs(a[o + MP:o + MP + 4 * M] - b[o + MP:o + MP + 4 * M]).max())
    print("grad rel L2", np.linalg.norm(ga - gb) / np.linalg.norm(ga))
