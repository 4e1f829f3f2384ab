"""cProfile of the runner-shaped loop: where does the HOST spend the step? (debugging aid for bench.runner_loop_leg)"""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
import bench
torch.cuda.set_device(0)
pr = cProfile.Profile()
pr.enable()
out = bench.runner_loop_leg(torch.device("cuda", 0), 0, 10, 2)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue()[:6000])
print({k: v for k, v in out.items() if k != "what"})
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"],
      torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_allocated() / 2**30)
