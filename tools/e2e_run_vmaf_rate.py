#!/usr/bin/env python3
"""File-to-score rate of the Python launcher: ./run_vmaf on a 1080p yuv420p pair in /dev/shm (whole process, second run).
python tools/e2e_run_vmaf_rate.py [frames=600]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vmaf_amd import synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    r, d = synth.make_clip(1920, 1080, 30, seed=1234)
    rp, dp = os.path.join(tmp, "ref.yuv"), os.path.join(tmp, "dis.yuv")
    synth.write_yuv(rp, r, "yuv420p")
    synth.write_yuv(dp, d, "yuv420p")
    one_r, one_d = open(rp, "rb").read(), open(dp, "rb").read()
    with open(rp, "ab") as f, open(dp, "ab") as g:
        for _ in range(frames // 30 - 1):
            f.write(one_r)
            g.write(one_d)
    for env in ({}, {"VMAF_HIP_DEVICES": "0,0"}):
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, os.path.join(ROOT, "run_vmaf"), "yuv420p", "1920", "1080", rp, dp, "--out-fmt", "json"],
                               capture_output=True, text=True, env=dict(os.environ, **env))
            best = min(best, time.perf_counter() - t0)
        assert p.returncode == 0, p.stderr
        print(f"run_vmaf {frames} frames {env}: wall {best:.3f} s = {frames / best:.0f} frames/s", flush=True)
