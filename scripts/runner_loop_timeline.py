"""Host and GPU timeline of ONE runner-shaped training step (statement boundaries of runners/train.py:120-218):
host perf_counter stamps and CUDA events recorded at the same points, printed relative to the step start."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import numpy as np, torch
import bench
from bench import RAYS, N_COARSE, N_FINE, H, W, NEAR, FAR
import torch_nerf.src.renderer.cameras as cameras
from torch_nerf.amd import synth
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
renderer, scene_c, scene_f, nets, _, _, focal, _ = bench.build_scene(device)
params = [p for net in nets for p in net.parameters()]
optimizer = torch.optim.Adam(params, lr=5e-4, eps=1e-8)
scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, 0.99999)
loss_func = torch.nn.MSELoss()
pixel_gt = torch.rand((H, W, 3)).reshape(-1, 3)
extrinsic = torch.from_numpy(synth.pose_spherical(10.0, -30.0, 4.0))
marks = []
MODE = sys.argv[1] if len(sys.argv) > 1 else "plain"
from torch_nerf.amd import ops
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append((name, time.perf_counter(), e))
def step():
    mark("start")
    optimizer.zero_grad()
    renderer.camera = cameras.PerspectiveCamera({"f_x": focal, "f_y": focal, "img_width": W, "img_height": H}, extrinsic, NEAR, FAR)
    mark("camera set")
    cp, ci, cw = renderer.render_scene(scene_c, num_pixels=RAYS, num_samples=N_COARSE, project_to_ndc=False, pixel_indices=None, device=torch.cuda.current_device())
    mark("render_scene coarse returned")
    g = pixel_gt[ci, ...]
    mark("cpu gather")
    g = g.cuda()
    mark(".cuda() #1 returned")
    cl = loss_func(g, cp); cl.item()
    mark("coarse loss .item()")
    fp, fi, _ = renderer.render_scene(scene_f, num_pixels=RAYS, num_samples=(N_COARSE, N_FINE), project_to_ndc=False, pixel_indices=ci, weights=cw, device=torch.cuda.current_device())
    mark("render_scene fine returned")
    g2 = pixel_gt[fi, ...]
    mark("cpu gather #2")
    if MODE == "sync-first":
        torch.cuda.synchronize()
        mark("synchronize() returned")
    g2 = g2.cuda()
    mark(".cuda() #2 returned")
    fl = loss_func(g2, fp); fl.item(); loss = cl + fl; loss.item()
    mark("losses .item()")
    loss.backward()
    mark("backward() returned")
    optimizer.step(); scheduler.step()
    mark("optimizer.step() returned")
import gc
print("torch threads", torch.get_num_threads(), "cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
if MODE == "one-thread":
    torch.set_num_threads(1)       # what the reference's runners do: runner_utils.py:427 (_init_torch)
def throttled():
    try:
        return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat")}.get("nr_throttled")
    except OSError:
        return None
print("nr_throttled before", throttled())
if MODE == "nogc":
    gc.disable()
gc_log = []
gc.callbacks.append(lambda phase, info: gc_log.append((phase, info.get("generation"), time.perf_counter())))
for k in range(12):
    marks.clear()
    ops.KERNEL_EVENTS = []
    n0 = torch.cuda.memory_stats()["num_device_alloc"]
    ta = time.perf_counter()
    step()
    torch.cuda.synchronize()
    gcs = [(g, round((t1 - t0_) * 1e3, 1)) for (p0, g, t0_), (p1, _, t1) in zip(gc_log[::2], gc_log[1::2]) if t0_ >= ta]
    print(f"step {k}: {(time.perf_counter() - ta) * 1e3:7.1f} ms   hipMallocs {torch.cuda.memory_stats()['num_device_alloc'] - n0}   gc (generation, ms) {gcs}", flush=True)
torch.cuda.synchronize()
print("nr_throttled after", throttled())
mark("gpu drained")
t0, e0 = marks[0][1], marks[0][2]
print(f"{'statement':34s} {'host ms':>9s} {'gpu-reaches-this-point ms':>26s}")
for name, t, e in marks:
    print(f"{name:34s} {(t - t0) * 1e3:9.2f} {e0.elapsed_time(e):26.2f}")

for tag, M, a, b in ops.KERNEL_EVENTS:
    print(f"   kernel-group {tag:14s} M={M:7d}  gpu start {e0.elapsed_time(a):8.2f}  end {e0.elapsed_time(b):8.2f}")
