"""Host-side cost of one render_scene call in training mode (what sits between a .item() and the first kernel)."""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch, bench
torch.cuda.set_device(0); torch.set_num_threads(1)
device = torch.device("cuda", 0)
renderer, scene_c, scene_f, nets, _, _, focal, _ = bench.build_scene(device)
pix = torch.randperm(640000)[:4096]
def coarse():
    return renderer.render_scene(scene_c, 4096, 64, False, 0, pixel_indices=pix)
def fine(ci, cw):
    return renderer.render_scene(scene_f, 4096, (64, 128), False, 0, pixel_indices=ci, weights=cw)
for _ in range(3):
    c = coarse(); f = fine(c[1], c[2])
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    c = coarse()
torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / N
cw = c[2].detach()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(N):
    f = fine(c[1], cw.clone())
t_host = (time.perf_counter() - t0) / N
torch.cuda.synchronize()
pr.disable()
print(f"coarse call+gpu {tc*1e3:.3f} ms; fine call host-side {t_host*1e3:.3f} ms")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
