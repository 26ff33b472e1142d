#!/usr/bin/env python3
"""bench.py's end_to_end figures alone (page-locked host memory -> scores; the CLI on /dev/shm).  python tools/e2e_pinned.py"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import bench  # noqa: E402
from vmaf_amd import synth  # noqa: E402
r8, d8 = synth.make_clip(1920, 1080, 30, seed=1234)
for _ in range(2):
    print(json.dumps(bench.end_to_end(r8, d8, "avx", 0)), flush=True)
