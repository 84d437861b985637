#!/usr/bin/env python3
"""Launch the TV prox with K = 1..5 fused steps (one launch each) so a kernel trace separates the
per-launch fixed cost (load/store of the region) from the per-step compute cost.
  rocprofv3 --kernel-trace --output-format csv -d out -- python tools/prox_steps.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-blind-image-deblurring-problems-with-tv_amd"))
import numpy as np
import torch
import sbtv

ctx = sbtv.default_context(0)
man = np.load(os.path.join(ROOT, "tests", "golden", "man_512.npy")).astype(np.float64)
size = int(os.environ.get("SIZE", "2048"))
img = np.tile(man, (size // 512, size // 512)) + np.random.default_rng(0).standard_normal((size, size))
gd = sbtv.to_device(img)
for rep in range(6):
    for K in (1, 2, 3, 4, 5):
        sbtv.chambolle_prox_TV_stop(gd, "lambda", 10.0, "maxiter", K)
torch.cuda.synchronize()
print("done")
