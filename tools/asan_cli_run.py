#!/usr/bin/env python3
"""Host-side AddressSanitizer / UBSan pass over the CLI and the host layer behind it (reader threads, sharding, model
parsers, log writers) ON THE GPU BOX.  The executable is built here on the CPU with the sanitizers on the host half only:
    make -C vmaf_amd/csrc asan_cli          (-> vmaf_amd/vmafossexec_asan, 9 MB, git-ignored; delete it afterwards: it travels with every gpurun call)
    python tools/asan_cli_run.py          (exit 1 when a run reports an error)
quarantine_size_mb: with the default 256 MB the 1080p run ends in a CHECK of ROCm's own ASan runtime (its device allocator
recycles page-locked blocks from the quarantine after the HSA runtime has unloaded, inside the HIP library's exit handler)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vmaf_amd import synth  # noqa: E402

EXE = os.path.join(ROOT, "vmaf_amd", "vmafossexec_asan")
MODEL = os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:quarantine_size_mb=4096", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
bad = 0


def run(name, args, env=None, expect_ok=True):
    global bad
    p = subprocess.run([EXE] + args, capture_output=True, env=dict(ENV, **(env or {})), timeout=300)
    p.stderr = p.stderr.decode("utf-8", "replace")
    report = "AddressSanitizer" in p.stderr or "runtime error" in p.stderr
    ok = not report and (p.returncode == 0) == expect_ok
    print("%-44s rc=%-4d %s" % (name, p.returncode, "ok" if ok else "FAILED"), flush=True)
    if not ok:
        bad += 1
        os.makedirs(os.path.join(ROOT, "gpurun_out", "asan"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "asan", name.replace(" ", "_").replace(",", "") + ".txt"), "w") as f:
            f.write(p.stderr)
        print(p.stderr[:1500], flush=True)
    return p


with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    def clip(w, h, n, bits, fmt, tag):
        r, d = synth.make_clip(w, h, n, bits=bits, seed=7)
        rp, dp = os.path.join(tmp, tag + "_r.yuv"), os.path.join(tmp, tag + "_d.yuv")
        synth.write_yuv(rp, r, fmt)
        synth.write_yuv(dp, d, fmt)
        return rp, dp
    r8, d8 = clip(576, 324, 37, 8, "yuv420p", "a")
    r10, d10 = clip(320, 180, 21, 10, "yuv422p10le", "b")
    rhd, dhd = clip(1920, 1080, 40, 8, "yuv420p", "c")
    base = ["yuv420p", "576", "324", r8, d8, MODEL]
    run("default", base)
    run("xml log", base + ["--log", os.path.join(tmp, "o.xml")])
    run("json log + metrics", base + ["--log", os.path.join(tmp, "o.json"), "--log-fmt", "json", "--psnr", "--ssim", "--ms-ssim"])
    run("phone, transform, no clip, harmonic", base + ["--phone-model", "--enable-transform", "--disable-clip", "--pool", "harmonic_mean"])
    run("scalar order", base + ["--disable-avx"])
    run("three shards", base + ["--psnr", "--ssim", "--ms-ssim"], {"VMAF_HIP_DEVICES": "0,0,0"})
    run("more shards than batches", base, {"VMAF_HIP_DEVICES": "0,0,0,0,0,0,0,0", "VMAF_HIP_BATCH": "3"})
    run("sequential path, tiny batches", base, {"VMAF_HIP_SEEKABLE": "0", "VMAF_HIP_BATCH": "2"})
    run("one reader thread", base, {"VMAF_HIP_READ_THREADS": "1"})
    run("many reader threads", base, {"VMAF_HIP_READ_THREADS": "32", "VMAF_HIP_DEVICES": "0,0"})
    run("10-bit 4:2:2", ["yuv422p10le", "320", "180", r10, d10, MODEL, "--psnr", "--ssim", "--ms-ssim"])
    run("1080p two shards", ["yuv420p", "1920", "1080", rhd, dhd, MODEL], {"VMAF_HIP_DEVICES": "0,0"})
    # truncated dis file: the last frame's luma is incomplete
    with open(d8, "r+b") as f:
        f.truncate(os.path.getsize(d8) - 576 * 324 * 3 // 2 - 1000)
    run("truncated dis", base)
    run("truncated dis, sequential", base, {"VMAF_HIP_SEEKABLE": "0"})
    run("missing file", ["yuv420p", "576", "324", r8, os.path.join(tmp, "nope.yuv"), MODEL], expect_ok=False)
    run("bad model", base[:5] + [r8], expect_ok=False)
    run("bad format", ["yuv411p"] + base[1:], expect_ok=False)
    run("too small", ["yuv420p", "8", "8", r8, d8, MODEL], expect_ok=False)
print("sanitizer runs with errors: %d" % bad)
sys.exit(1 if bad else 0)
