import os, time, sys, torch
sys.path[:0]=['.','torch-nerf_amd']
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max","/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|^CPU\\(s\\)' ; nproc; free -g | head -2")
from oracle import torch_port as TP
from torch_nerf.amd import synth, shard
import numpy as np
params=[{k: torch.from_numpy(v.copy()) for k,v in synth.split_flat_params(synth.nerf_flat_params(seed=s,sigma_bias=1.0,sigma_gain=30.0)).items()} for s in (3,4)]
pose=torch.from_numpy(synth.pose_spherical(37.,-30.,4.)); focal=float(synth.blender_focal(800))
def run(n):
    pix=torch.from_numpy(synth.pixel_batch(0,800,800,n)); draws=shard.ray_draws(7,0,n,64,128,"cpu")
    t0=time.perf_counter()
    with torch.no_grad(): TP.render_batch(params[0],params[1],pix,800,800,focal,pose,2.,6.,64,128,draws)
    return time.perf_counter()-t0
for th in (1,8,16,32,64,128):
    torch.set_num_threads(th); run(32); n=256 if th>1 else 64; t=run(n); print("threads",th,"rays/s",n/t, flush=True)
