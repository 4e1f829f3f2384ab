"""A/B timing of the training kernels on ONE box: record-mode forward and backward (dX + dW + vec + reduce) at the fine-pass
size, default library vs every torch-nerf_amd/lib/variants/*.so, interleaved rounds.   python scripts/ab_train.py [--rounds 3]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, numpy as np, torch
sys.path[:0] = [%r, %r]
from torch_nerf.amd import ops, synth
flat = torch.from_numpy(synth.nerf_flat_params(seed=9, sigma_bias=0.5, sigma_gain=20.0)).cuda()
packed = ops.mlp_pack(flat)
M = 4096 * 192
g = torch.Generator(device="cuda").manual_seed(0)
pts = torch.rand(M, 3, device="cuda", generator=g) * 8 - 4
dirs = torch.rand(M, 3, device="cuda", generator=g) * 2 - 1
gs = torch.randn(M, device="cuda", generator=g); gc = torch.randn(M, 3, device="cuda", generator=g)
def med(fn, reps=12):
    for _ in range(3): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in ev)
    return t[len(t) // 2]
sigma, rgb, saved = ops.mlp_forward(packed, pts, dirs, False, save=True)
fwd = med(lambda: ops.mlp_forward(packed, pts, dirs, False, save=True))
bwd = med(lambda: ops.mlp_backward(packed, flat, pts, dirs, False, sigma, rgb, saved, gs, gc))
gp = ops.mlp_backward(packed, flat, pts, dirs, False, sigma, rgb, saved, gs, gc)
print(json.dumps({"fwd_record_ms": fwd, "backward_ms": bwd, "chk": float(gp.double().abs().sum().item())}))
'''
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
arms = {"default": os.path.join(ROOT, "torch-nerf_amd", "lib", "libnerf_amd.so")}
for p in sorted(glob.glob(os.path.join(ROOT, "torch-nerf_amd", "lib", "variants", "*.so"))):
    arms[os.path.basename(p)[:-3]] = p
res = {k: [] for k in arms}
for r in range(rounds):
    for name, lib in arms.items():
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "torch-nerf_amd"))],
                             env=dict(os.environ, NERF_AMD_LIB=lib), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        res[name].append(json.loads(line[-1]) if line else {"error": out.stderr[-300:]})
for name, rs in res.items():
    if any("error" in x for x in rs):
        print(f"{name:12s} ERROR {rs}"); continue
    f = sorted(x["fwd_record_ms"] for x in rs); b = sorted(x["backward_ms"] for x in rs)
    print(f"{name:12s} fwd-record {f[len(f)//2]:.4f} ms {[round(x['fwd_record_ms'],3) for x in rs]}   backward {b[len(b)//2]:.4f} ms {[round(x['backward_ms'],3) for x in rs]}   chk {rs[0]['chk']:.6e}")
