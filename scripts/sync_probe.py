"""Wall time of a blocking wait vs the GPU time it waits for (does the host wake up late on this box?)."""
import time, torch
torch.cuda.set_device(0)
x = torch.zeros(8, device="cuda"); h = torch.zeros(4096, 3)
torch.cuda.synchronize()
for how in ("synchronize", ".cuda() of a pageable tensor", ".item()"):
    for ms in (1, 5, 10, 20, 40):
        walls, gpus = [], []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(); torch.cuda._sleep(int(ms * 2.1e6)); e1.record()
            if how == "synchronize": torch.cuda.synchronize()
            elif how == ".item()": x[0].item()
            else: h.cuda()
            walls.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize(); gpus.append(e0.elapsed_time(e1))
        print(f"{how:32s} gpu {sum(gpus)/5:7.2f} ms   wall {sum(walls)/5:7.2f} ms   (min {min(walls):.2f}, max {max(walls):.2f})")
