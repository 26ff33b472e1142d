#!/usr/bin/env python3
"""What the GPU-side widening of raw luma costs: 1080p batches resident in HBM as uint8 (extract_device_raw: widen kernel +
feature kernels) against the same frames resident as fp32 (extract_device), synchronous calls.  Never bench.py's `value`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vmaf_amd  # noqa: E402
from vmaf_amd import synth  # noqa: E402

W, H, B, K = 1920, 1080, 30, 40
r8, d8 = synth.make_clip(W, H, B, seed=1234)
dev = torch.device("cuda", 0)
t_r8, t_d8 = torch.from_numpy(r8).to(dev), torch.from_numpy(d8).to(dev)
t_rf, t_df = t_r8.float(), t_d8.float()
out = np.zeros((B, 20))
with vmaf_amd.Engine(W, H, max_batch=B) as e:
    res = {}
    for name, fn in (("fp32 resident", lambda c: e.extract_device(t_rf, t_df, B, continue_sequence=c, out=out)),
                     ("uint8 resident (widened on the GPU)", lambda c: e.extract_device_raw(t_r8, t_d8, B, 1, continue_sequence=c, out=out))):
        for i in range(3):
            fn(i > 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = out.copy()
        print(f"{name}: {K * B / dt:.0f} frames/s ({1e3 * dt / K:.3f} ms per {B}-frame batch, synchronous calls)", flush=True)
    a, b = list(res.values())
    print("identical results:", np.array_equal(a, b))
