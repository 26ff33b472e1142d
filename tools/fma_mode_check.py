#!/usr/bin/env python3
"""How far the opt-in fused-multiply-add order (conv order 2) lands from the reference's two orders, on every golden clip
and the 18 test images: max |feature| and |VMAF| differences vs the AVX order, with scalar-vs-AVX (the reference's own
spread) beside it.  GPU only; writes gpurun_out/fma_mode.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (its ROCm runtime must be the one the library binds to)
import vmaf_amd  # noqa: E402
from vmaf_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ATOMS = json.load(open(os.path.join(GOLD, "atoms.json")))


def clips():
    for name, case in ATOMS.items():
        if not isinstance(case.get("params"), dict):
            continue
        kw = dict(case["params"])
        bits = kw.pop("bits", 8)
        if "cuts" in kw:
            kw["cuts"] = tuple(kw["cuts"])
        r, d = synth.make_clip(bits=bits, **kw)
        div = 4.0 if bits == 10 else 1.0
        yield name, r.astype(np.float32) / div, d.astype(np.float32) / div
    imgs = sorted(f[:-4] for f in os.listdir(os.path.join(GOLD, "images")) if f.endswith(".yuv") and not f.endswith("_noisy.yuv"))
    for n in imgs:
        def luma(p):
            return np.fromfile(p, np.uint8)[:481 * 321].reshape(1, 321, 481).astype(np.float32)
        yield f"image_{n}", luma(os.path.join(GOLD, "images", f"{n}.yuv")), luma(os.path.join(GOLD, "images", f"{n}_noisy.yuv"))


out = {}
for name, r, d in clips():
    res = {}
    for order in ("avx", "scalar", "fma"):
        with vmaf_amd.Engine(r.shape[2], r.shape[1], max_batch=len(r), order=order) as e:
            a = e.extract(r, d)
        res[order] = vmaf_amd.predict_from_atoms(a)
    row = {}
    for o in ("scalar", "fma"):
        row[o] = {"feature": float(np.abs(res[o][1] - res["avx"][1]).max()), "vmaf": float(np.abs(res[o][0] - res["avx"][0]).max())}
    out[name] = row
    print(f"{name:28s} scalar-avx: feat {row['scalar']['feature']:.2e} vmaf {row['scalar']['vmaf']:.2e} | fma-avx: feat {row['fma']['feature']:.2e} vmaf {row['fma']['vmaf']:.2e}", flush=True)
for o in ("scalar", "fma"):
    print(o, "worst feature", max(v[o]["feature"] for v in out.values()), "worst vmaf", max(v[o]["vmaf"] for v in out.values()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fma_mode.json"), "w"), indent=1)
