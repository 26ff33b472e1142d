#!/usr/bin/env python3
"""Batch-mode rate (BASELINE.json configs[3]: many 1280x720 pairs, whole pairs per GPU, no exchange step):
./run_vmaf_in_batch on a list of N lines pointing at synthetic 48-frame 720p pairs in a RAM-backed directory.
    python tools/batch_rate.py [pairs=64]
Never bench.py's `value` (file reading, Python launcher and process start-up are inside)."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vmaf_amd import synth  # noqa: E402


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as tmp:
        lines = []
        for k in range(4):                                   # four distinct pairs, listed round-robin
            r, d = synth.make_clip(1280, 720, 48, seed=100 + k)
            rp, dp = os.path.join(tmp, f"ref{k}.yuv"), os.path.join(tmp, f"dis{k}.yuv")
            synth.write_yuv(rp, r, "yuv420p")
            synth.write_yuv(dp, d, "yuv420p")
            lines.append(f"yuv420p 1280 720 {rp} {dp}")
        lst = os.path.join(tmp, "pairs.txt")
        open(lst, "w").write("\n".join(lines[i % 4] for i in range(pairs)) + "\n")
        for rep in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, os.path.join(ROOT, "run_vmaf_in_batch"), lst, "--out-fmt", "json"],
                               capture_output=True, text=True)
            dt = time.perf_counter() - t0
        assert p.returncode == 0, p.stderr[-2000:]
        n = p.stdout.count('"aggregate"')
        print(f"{n} pairs x 48 frames of 1280x720 in {dt:.2f} s: {n / dt:.1f} pairs/s, {n * 48 / dt:.0f} frames/s (one GPU, one process)")


if __name__ == "__main__":
    main()
