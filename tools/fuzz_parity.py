#!/usr/bin/env python3
"""One-off randomized parity sweep on the GPU box: random frame sizes (incl. sizes around the tile pitches of the
kernels: 48x32 and 240x8 VIF, 61x14 ADM coefficients, 60x32 motion), both summation orders, against the CPU oracle.
    python tools/fuzz_parity.py [cases=80] [seed=1]
Prints the worst relative atom error and the worst absolute feature error; exits 1 on a tolerance breach, or when the motion
atom (always) or the VIF atoms (engine with vmaf_hip_engine_set_sums_exact) are not bit-equal to the oracle's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (shares its HIP runtime with libvmaf_amd.so: import first)
import oracle  # noqa: E402
import vmaf_amd  # noqa: E402
from vmaf_amd import synth  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    special = [32, 33, 47, 48, 49, 60, 61, 64, 95, 96, 97, 120, 122, 123, 128, 176, 192, 244, 245, 256]
    wide = [240, 248, 256, 264, 304, 320, 328, 400, 472, 480, 488, 496, 552, 640, 720, 728, 736]   # workgroup-wide VIF tiles, narrow last tiles
    worst_rel, worst_feat, bad = 0.0, 0.0, 0
    for c in range(cases):
        u = rng.random()
        w = int(rng.choice(special)) if u < 0.3 else int(rng.choice(wide)) if u < 0.65 else int(rng.integers(32, 420))
        h = int(rng.choice(special)) if rng.random() < 0.5 else int(rng.integers(32, 300))
        bits = 10 if rng.random() < 0.25 else 8
        order = "avx" if rng.random() < 0.6 else "scalar"
        frames = int(rng.integers(1, 4))
        r, d = synth.make_clip(w, h, frames, bits=bits, seed=int(rng.integers(1, 1 << 30)))
        ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
        want = oracle.clip_atoms(ref, dis, oracle.ORDER_AVX if order == "avx" else oracle.ORDER_SCALAR)
        with vmaf_amd.Engine(w, h, max_batch=2, order=order) as e:
            got = e.extract_raw(r, d)
        with vmaf_amd.Engine(w, h, max_batch=2, order=order, sums_exact=True) as e:
            lit = e.extract_raw(r, d)
        # motion always, and with the VIF sums in vif_sum_s's order all nine VIF atoms: the reference's bits
        if not (np.array_equal(got[:, 10], want[:, 10]) and np.array_equal(lit[:, 10:20], want[:, 10:20]) and np.array_equal(lit[:, :10], got[:, :10])):
            bad += 1
            print(f"NOT BIT-EQUAL {w}x{h} bits={bits} order={order} frames={frames}", flush=True)
        rel = float(np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))))
        fg, fw = oracle.derived_features(got), oracle.derived_features(want)
        feat = max(float(np.max(np.abs(fg[k] - fw[k]))) for k in fg)
        worst_rel, worst_feat = max(worst_rel, rel), max(worst_feat, feat)
        if rel > 2e-6 or feat > 1e-4 or not np.all(np.isfinite(got)):
            bad += 1
            print(f"BREACH {w}x{h} bits={bits} order={order} frames={frames}: rel {rel:.3g} feat {feat:.3g}", flush=True)
    print(f"{cases} cases, worst relative atom error {worst_rel:.3g}, worst feature error {worst_feat:.3g}, breaches {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
