import sys, time, numpy as np, torch
sys.path[:0]=['.', 'torch-nerf_amd']
from torch_nerf.amd import ops, synth
flat = torch.from_numpy(synth.nerf_flat_params(seed=3, sigma_bias=1.0, sigma_gain=30.0)).cuda()
pk = ops.mlp_pack_bf16(flat)
pk32 = ops.mlp_pack(flat)
for M in (262144, 786432):
    pts = (torch.rand(M,3,device='cuda')*8-4); dirs = torch.rand(M,3,device='cuda')*2-1
    s,c = ops.mlp_forward_bf16(pk, pts, dirs)
    s32,c32 = ops.mlp_forward(pk32, pts, dirs, encoded=False)
    mse = torch.mean((c.double()-c32.double())**2).item()
    print("M",M,"psnr vs fp32", 10*np.log10(1/mse), "max sigma rel", ((s-s32).abs()/(s32.abs()+1)).max().item())
    torch.cuda.synchronize()
    ev=[]
    for r in range(20):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); ops.mlp_forward_bf16(pk, pts, dirs); e1.record(); ev.append((e0,e1))
    torch.cuda.synchronize()
    t=np.median([a.elapsed_time(b) for a,b in ev])
    print("  bf16 ms", t, "TF/s", M*1186816/t/1e9, "frac", M*1186816/t/1e9/2500)
