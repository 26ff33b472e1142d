#!/usr/bin/env python3
"""Host time of one asynchronous batch enqueue (13 kernel launches + events over four streams) against the GPU time of the
batch, at small batches: is the launch path ever the bottleneck (would a captured hipGraph pay)?  python tools/enqueue_cost.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vmaf_amd  # noqa: E402
from vmaf_amd import synth  # noqa: E402

W, H = 1920, 1080
NMAX = 32
r8, d8 = synth.make_clip(W, H, NMAX, seed=3)
ref = torch.from_numpy(synth.to_luma_f32(r8, 8)).cuda()
dis = torch.from_numpy(synth.to_luma_f32(d8, 8)).cuda()
for B in [int(x) for x in sys.argv[1:]] or (2, 4, 8, 16):
    with vmaf_amd.Engine(W, H, max_batch=B) as e:
        out = np.zeros((B, 20))
        for i in range(5):
            e.extract_device(ref, dis, B, continue_sequence=i > 0, out=out)
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for k in range(200):
            h0 = time.perf_counter()
            e.extract_device(ref, dis, B, continue_sequence=True, out=out, sync=False)
            host.append(time.perf_counter() - h0)
            if k > 0:
                e.sync_oldest()
        e.sync()
        wall = (time.perf_counter() - t0) / 200
    print("batch %2d frames: host enqueue %6.1f us (median), whole batch %7.1f us -> %7.0f frames/s" %
          (B, 1e6 * float(np.median(host)), 1e6 * wall, B / wall), flush=True)
