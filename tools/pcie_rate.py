#!/usr/bin/env python3
"""PCIe-inclusive rate of the kernel ABI (never bench.py's `value`): 1080p batches in page-locked HOST memory through
vmaf_hip_extract_host (fp32 planes, 8 B/px over PCIe) and vmaf_hip_extract_host_raw (uint8 planes, 2 B/px, widened on the
GPU), synchronous calls (copy and kernels of one batch do not overlap).  No file I/O, no host conversion."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vmaf_amd  # noqa: E402
from vmaf_amd import synth  # noqa: E402

W, H, B, K = 1920, 1080, 60, 15
r8, d8 = synth.make_clip(W, H, 30, seed=1234)
r8, d8 = np.concatenate([r8, r8]), np.concatenate([d8, d8])


def pinned(a):
    t = torch.empty(a.shape, dtype=torch.from_numpy(a[:1]).dtype, pin_memory=True)
    t.numpy()[...] = a
    return t


p_r8, p_d8 = pinned(r8), pinned(d8)
p_rf, p_df = pinned(r8.astype(np.float32)), pinned(d8.astype(np.float32))
with vmaf_amd.Engine(W, H, max_batch=B) as e:
    for name, fn, bpp in (("fp32 planes in pinned host memory", lambda: e.extract(p_rf.numpy(), p_df.numpy(), continue_sequence=True), 8),
                          ("uint8 planes in pinned host memory", lambda: e.extract_raw(p_r8.numpy(), p_d8.numpy(), continue_sequence=True), 2)):
        e.extract_raw(p_r8.numpy()[:1], p_d8.numpy()[:1])
        fn()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        dt = time.perf_counter() - t0
        fps = K * B / dt
        print(f"{name}: {fps:.0f} frames/s, {fps * W * H * bpp / 1e9:.1f} GB/s over PCIe ({B}-frame batches, synchronous)", flush=True)
