#!/usr/bin/env python3
"""Development check: the `motion` atom of the HIP engine against the oracle, bit for bit (motion.c:44-60 order),
over sizes, orders, bit depths and batch / chunk splits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import numpy as np
import oracle
import vmaf_amd
from vmaf_amd import synth

bad = 0
for (w, h, frames, bits) in [(64, 48, 5, 8), (577, 325, 4, 8), (200, 32, 3, 8), (33, 130, 3, 10), (1280, 720, 3, 8), (1920, 1080, 7, 8),
                             (488, 57, 9, 8), (61, 60, 40, 8), (3840, 2160, 2, 10)]:
    for order in ("avx", "scalar"):
        r, d = synth.make_clip(w, h, frames, bits=bits, seed=w + h)
        ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
        want = oracle.clip_atoms(ref, dis, oracle.ORDER_AVX if order == "avx" else oracle.ORDER_SCALAR)[:, 10]
        for mb in (frames, 2):
            with vmaf_amd.Engine(w, h, max_batch=mb, order=order) as e:
                got = e.extract(ref, dis)[:, 10]
            ok = np.array_equal(got, want)
            bad += not ok
            print(w, h, frames, bits, order, "batch", mb, "OK" if ok else f"DIFF got {got} want {want}", flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
