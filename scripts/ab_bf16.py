"""A/B timing of kernel variants on ONE box: every variant library (torch-nerf_amd/lib/variants/*.so, built with
`make EXTRA=-DX_... BUILD=build_x OUT=../lib/variants/x.so`) and the default library time the bf16 (or fp32) fused MLP
kernel at the fine-pass size, in interleaved rounds so that clock drift hits all arms alike.
    python scripts/ab_bf16.py [--fp32] [--rounds 3]"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, numpy as np, torch
sys.path[:0] = [%r, %r]
from torch_nerf.amd import ops, synth
fp32 = %r
flat = torch.from_numpy(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)).cuda()
if %r: flat = flat * 0      # all-zero weights: the same instruction stream with (almost) no datapath toggling
pk = ops.mlp_pack(flat) if fp32 else ops.mlp_pack_bf16(flat)
M = 4096 * 192
g = torch.Generator(device="cuda").manual_seed(0)
pts = torch.rand(M, 3, device="cuda", generator=g) * 8 - 4
dirs = torch.rand(M, 3, device="cuda", generator=g) * 2 - 1
run = (lambda: ops.mlp_forward(pk, pts, dirs, encoded=False)) if fp32 else (lambda: ops.mlp_forward_bf16(pk, pts, dirs))
s, c = run()
chk = float(c.double().sum().item()), float(s.double().sum().item())
for _ in range(5): run()
ev = []
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); ev.append((a, b))
torch.cuda.synchronize()
t = sorted(x.elapsed_time(y) for x, y in ev)
print(json.dumps({"median_ms": t[len(t)//2], "min_ms": t[0], "chk": chk}))
'''
fp32 = "--fp32" in sys.argv
zero = "--zero" in sys.argv
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
arms = {"default": os.path.join(ROOT, "torch-nerf_amd", "lib", "libnerf_amd.so")}
for p in sorted(glob.glob(os.path.join(ROOT, "torch-nerf_amd", "lib", "variants", "*.so"))):
    arms[os.path.basename(p)[:-3]] = p
res = {k: [] for k in arms}
for r in range(rounds):
    for name, lib in arms.items():
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "torch-nerf_amd"), fp32, zero)],
                             env=dict(os.environ, NERF_AMD_LIB=lib), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        res[name].append(json.loads(line[-1]) if line else {"error": out.stderr[-300:]})
for name, rs in res.items():
    if any("error" in x for x in rs):
        print(f"{name:12s} ERROR {rs}")
        continue
    med = sorted(x["median_ms"] for x in rs)
    print(f"{name:12s} median of medians {med[len(med)//2]:.4f} ms   all {[round(x['median_ms'], 4) for x in rs]}   min {min(x['min_ms'] for x in rs):.4f}   chk {rs[0]['chk'][0]:.3f}")
