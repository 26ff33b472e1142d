#!/usr/bin/env python3
"""PCIe- and file-inclusive rate of the CLI boundary (never bench.py's `value`):
vmaf_amd/vmafossexec on a synthetic 1080p yuv420p pair in a RAM-backed directory, with the raw-luma hand-over
(default) and with the reference's fp32 read_frame contract (VMAF_HIP_FP32_CALLBACK=1).

    python tools/e2e_cli_rate.py [frames=120]
"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vmaf_amd import synth  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as tmp:
        r, d = synth.make_clip(1920, 1080, 30, seed=1234)
        rp, dp = os.path.join(tmp, "ref.yuv"), os.path.join(tmp, "dis.yuv")
        synth.write_yuv(rp, r, "yuv420p")
        synth.write_yuv(dp, d, "yuv420p")
        one_r, one_d = open(rp, "rb").read(), open(dp, "rb").read()
        with open(rp, "ab") as f, open(dp, "ab") as g:          # repeat the 30 frames up to the requested length
            for _ in range(frames // 30 - 1):
                f.write(one_r)
                g.write(one_d)
        exe = os.path.join(ROOT, "vmaf_amd", "vmafossexec")
        model = os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl")
        for label, env in (("raw luma hand-over (compute_vmaf_raw), 8 reader threads", dict(VMAF_HIP_FP32_CALLBACK="0")),
                           ("raw luma hand-over, 16 reader threads", dict(VMAF_HIP_FP32_CALLBACK="0", VMAF_HIP_READ_THREADS="16")),
                           ("raw luma hand-over, 8 reader threads, fused order (VMAF_HIP_FAST=1)", dict(VMAF_HIP_FP32_CALLBACK="0", VMAF_HIP_FAST="1")),
                           ("raw luma hand-over, one reader thread per file (fread)", dict(VMAF_HIP_FP32_CALLBACK="0", VMAF_HIP_READ_THREADS="1")),
                           ("fp32 read_frame callback (compute_vmaf)", dict(VMAF_HIP_FP32_CALLBACK="1"))):
            for rep in range(2):                                 # second run: page cache and GPU warm
                t0 = time.perf_counter()
                p = subprocess.run([exe, "yuv420p", "1920", "1080", rp, dp, model], capture_output=True, text=True,
                                   env=dict(os.environ, **env))
                wall = time.perf_counter() - t0
            fps = float(re.search(r"Exec FPS: ([0-9.]+)", p.stdout).group(1))
            score = float(re.search(r"VMAF score = ([0-9.]+)", p.stdout).group(1))
            print(f"{label}: {frames // 30 * 30} frames, Exec FPS {fps:.1f} (library timer), process wall {wall:.2f} s, VMAF {score:.4f}", flush=True)


if __name__ == "__main__":
    main()
