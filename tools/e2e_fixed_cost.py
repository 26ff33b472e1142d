#!/usr/bin/env python3
"""Fixed cost vs steady state of the CLI: vmaf_amd/vmafossexec on 1080p yuv420p pairs of several lengths in /dev/shm
(whole-process wall time; second run of each, page cache warm).  python tools/e2e_fixed_cost.py"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vmaf_amd import synth  # noqa: E402

exe = os.path.join(ROOT, "vmaf_amd", "vmafossexec")
model = os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl")
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    r, d = synth.make_clip(1920, 1080, 30, seed=1234)
    rp, dp = os.path.join(tmp, "ref.yuv"), os.path.join(tmp, "dis.yuv")
    synth.write_yuv(rp, r, "yuv420p")
    synth.write_yuv(dp, d, "yuv420p")
    one_r, one_d = open(rp, "rb").read(), open(dp, "rb").read()
    have = 30
    for frames in (30, 600, 2400):
        with open(rp, "ab") as f, open(dp, "ab") as g:
            for _ in range((frames - have) // 30):
                f.write(one_r)
                g.write(one_d)
        have = frames
        for env in ({}, {"VMAF_HIP_READ_THREADS": "16"}, {"VMAF_HIP_DEVICES": "0,0"}, {"VMAF_HIP_DEVICES": "0,0,0,0"}):
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                p = subprocess.run([exe, "yuv420p", "1920", "1080", rp, dp, model], capture_output=True, text=True, env=dict(os.environ, **env))
                best = min(best, time.perf_counter() - t0)
            fps = float(re.search(r"Exec FPS: ([0-9.]+)", p.stdout).group(1))
            print(f"{frames:5d} frames {env}: wall {best:.3f} s = {frames / best:7.0f} frames/s; library timer {fps:7.0f} frames/s ({frames / fps:.3f} s)", flush=True)
