#!/usr/bin/env python3
"""How far the IMPORTED REFERENCE drifts from its own golden F14 trajectory when only its thread count changes (another
sgemm summation order): the yardstick for the bounds of tests/test_gpu_trajectory.py.  Build container only.

    python tests/golden/f14_self_drift.py 1 4
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (imports the reference from /root/reference)

want = np.load(os.path.join(HERE, "f14_train_loop.npz"))
for threads in [int(a) for a in sys.argv[1:]] or [1, 4]:
    torch.set_num_threads(threads)
    saved = {}
    G.save = lambda name, **arrays: saved.update(arrays)          # capture instead of writing the fixture
    G.f14_train_loop("f14_self", (10, 4))
    loss = max(float(np.abs(saved[k] - want[k]).max()) for k in ("coarse_loss", "fine_loss"))
    pix = max(float(np.abs(saved[k] - want[k]).max()) for k in saved if k.endswith("_rgb"))
    out = {"threads": threads, "loss": loss, "pixel": pix}
    for tag in ("coarse", "fine"):
        # the digests of the two runs against each other: norm, 99th percentile and worst element of the update
        a = np.concatenate([saved[f"{tag}_dp.head"], saved[f"{tag}_dp.stride"]])
        b = np.concatenate([want[f"{tag}_dp.head"], want[f"{tag}_dp.stride"]])
        n_ref = float(want[f"{tag}_dp.norm"][0])
        rms = n_ref / np.sqrt(595844)
        out[tag] = {"dp_norm_rel": abs(float(saved[f"{tag}_dp.norm"][0]) - n_ref) / n_ref,
                    "dp_p99_rel_rms": float(np.quantile(np.abs(a - b), 0.99) / rms), "dp_rel_rms": float(np.abs(a - b).max() / rms)}
    print(out)
