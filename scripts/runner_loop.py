"""The runner-shaped training loop of bench.py on its own (for rocprofv3 --kernel-trace --stats / timelines).
    python scripts/runner_loop.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "torch-nerf_amd")]
import torch
import bench
torch.cuda.set_device(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
print(json.dumps(bench.runner_loop_leg(torch.device("cuda", 0), 0, steps, 2)))
