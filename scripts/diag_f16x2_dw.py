import sys, numpy as np, torch
sys.path.insert(0, "torch-nerf_amd")
from torch_nerf.amd import ops, synth
flat_np = synth.nerf_flat_params(seed=4, sigma_bias=1.0, sigma_gain=30.0)
flat = torch.from_numpy(flat_np).cuda()
p32, px = ops.mlp_pack(flat), ops.mlp_pack_f16x2(flat)
for M in [int(a) for a in sys.argv[1:]] or [1, 128, 1000]:
    rng = np.random.RandomState(M)
    xs = torch.from_numpy(rng.uniform(-3, 3, (M, 3)).astype(np.float32)).cuda()
    vs = torch.from_numpy(rng.uniform(-1, 1, (M, 3)).astype(np.float32)).cuda()
    gs = torch.from_numpy(rng.standard_normal(M).astype(np.float32)).cuda()
    gc = torch.from_numpy(rng.standard_normal((M, 3)).astype(np.float32)).cuda()
    s, c, rec = ops.mlp_forward(p32, xs, vs, False, save=True)
    a = ops.mlp_backward(p32, flat, xs, vs, False, s, c, rec, gs, gc).cpu().numpy()
    b = ops.mlp_backward(p32, flat, xs, vs, False, s, c, rec, gs, gc, packed_f16x2=px).cpu().numpy()
    A, B = synth.split_flat_params(a), synth.split_flat_params(b)
    for k in A:
        d = np.abs(A[k] - B[k]); rms = np.sqrt(np.mean(A[k].astype(np.float64) ** 2)) + 1e-30
        if d.max() / rms > 1e-4:
            bad = np.argwhere(d > 1e-4 * rms)
            print(M, k, "max diff / rms", d.max() / rms, "bad", len(bad), "rows", sorted(set(bad[:, 0]))[:6], "cols", sorted(set(bad[:, -1]))[:12], "..", sorted(set(bad[:, -1]))[-3:])
    print(M, "total rel L2", np.linalg.norm(a - b) / np.linalg.norm(a))
    if "fc_9.weight" in A:
        np.set_printoptions(precision=4, suppress=False, linewidth=200)
        print("fp32 row0 dir:", A["fc_9.weight"][0, 256:283][:12]); print("x2   row0 dir:", B["fc_9.weight"][0, 256:283][:12])
        print("fp32 col256 rows:", A["fc_9.weight"][:8, 256]); print("x2   col256 rows:", B["fc_9.weight"][:8, 256])
        r = B["fc_9.weight"][:, 256:283] / (A["fc_9.weight"][:, 256:283] + 1e-30)
        print("ratio stats", np.median(r), r.min(), r.max())
