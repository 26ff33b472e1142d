#!/usr/bin/env python3
"""Largest SSIM / MS-SSIM / PSNR difference between the HIP engine and the oracle over a few sizes and bit depths (GPU box).
The test suite holds them to 1e-6 / 1e-9; this prints what they actually are.  python tools/metrics_deviation.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import vmaf_amd  # noqa: E402
from vmaf_amd import synth  # noqa: E402

worst = np.zeros(3)
for w, h, bits in [(576, 324, 8), (577, 325, 8), (320, 180, 10), (1280, 720, 8), (1920, 1080, 8), (1920, 1080, 10)]:
    r, d = synth.make_clip(w, h, 2, bits=bits, seed=50 + w)
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    with vmaf_amd.Engine(w, h, max_batch=2) as e:
        e.enable_metrics(ten_bit=bits == 10)
        e.extract(ref, dis)
        got = e.metrics.copy()
    for f in range(2):
        want = np.array([oracle.psnr(ref[f], dis[f], bits), oracle.ssim(ref[f], dis[f])[0], oracle.ms_ssim(ref[f], dis[f])[0]])
        dev = np.abs(got[f, :3] - want)
        worst = np.maximum(worst, dev)
        print("%4dx%-4d %2d-bit frame %d  psnr %.2e  ssim %.2e  ms_ssim %.2e" % (w, h, bits, f, *dev), flush=True)
print("worst: psnr %.2e  ssim %.2e  ms_ssim %.2e" % tuple(worst))
