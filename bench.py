#!/usr/bin/env python3
"""bench.py - frames/sec of full VMAF v0.6.1 (VIF x4 + ADM + motion + nu-SVR) on synthetic 1080p.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

Workload (BASELINE.json configs[1]): synthetic 1920x1080 yuv420p 8-bit ref/dis pair, luma resident in
HBM as fp32 (what a read_frame callback delivers) before the timed region starts.  A step = one batch of
FRAMES_PER_STEP consecutive frames through the whole hot path (120: 30 distinct synthetic frames tiled four times in HBM,
every one of them scored); default 300 steps x 120 frames = 36 000 frames (a timed region of about a second, long
enough for an outside GPU-utilisation sampler to see it).
Multi-GPU: contiguous frame ranges per rank with a one-frame motion halo (weak scaling: every rank gets
steps x 120 frames), per-frame atom rows gathered to rank 0 over RCCL, SVR fusion for all frames on rank 0 -
all inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel:
vif_stat scale 0, HIP events on the engine's own stream) and `cpu_baseline` (the reference's own C code
compiled under oracle/_ref when present, else the oracle port, one core, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

try:
    import torch  # first: torch and libvmaf_amd.so must share one HIP runtime
    import torch.distributed as dist
except Exception as exc:  # pragma: no cover
    print(f"bench.py needs torch for device memory / RCCL: {exc}", file=sys.stderr)
    raise

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import vmaf_amd  # noqa: E402
from vmaf_amd import sharding, synth  # noqa: E402

W, H = 1920, 1080
FRAMES_PER_STEP = 120
BITS = 8
# BASELINE.json configs[1], [2] (per-GPU share), [3] (one pair) and [0]:
# name -> width, height, bits, distinct synthetic frames, repeats.  One step = one batch of distinct x repeats frames resident
# in HBM (the distinct frames tiled: every frame of the batch is scored, nothing is cached).  Batches of ~100 1080p frames
# are what fills the GPU in the small kernels of the path (VIF scales 2-3, ADM scales 1-3): 26.5k frames/s at 30 frames
# per step, 28.0k at 60, 28.8k at 120.
WORKLOADS = {"1080p": (1920, 1080, 8, 30, 4), "4k10": (3840, 2160, 10, 8, 6), "720p": (1280, 720, 8, 48, 4),
             "576x324": (576, 324, 8, 48, 20),      # configs[0], the reference's own CPU-runnable test clip size
             "1080p10": (1920, 1080, 10, 30, 4), "4k": (3840, 2160, 8, 8, 6)}     # the other bit depth of each size
MODEL = os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl")
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PATH_BYTES_PER_PX = 26.5        # algorithmic bytes of the whole path per luma pixel
PATH_FLOP_PER_PX = 620.0       # algorithmic fp32 flops of the whole path per luma pixel
VIF0_BYTES_PER_PX = 10.0        # dominant kernel, fp32 input: reads the fp32 ref and dis planes once (8 B/px) and writes the
                                # two decimated planes of the next VIF scale (2 x 4 B / 4 px = 2 B/px); raw input: 2 x
                                # sample bytes read instead of 8 (main() sets the figure for the run)


VALU_PEAK_WINST = 1024 * 2.4e9 / 2   # wave64 fp32 VALU instructions/s: 256 CUs x 4 SIMDs, one per 2 cycles at 2.4 GHz
                                     # (MI355X_MICROARCH.md: 157 TFLOP/s fp32 vector = this rate x 64 lanes x 2 flop)


def committed_profile(frames_per_launch, input_kind="raw"):
    """Per-launch PMC figures of the dominant kernel from the committed rocprofv3 passes (profiles/traffic.json,
    written by tools/summarize_profiles.py from separate FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU runs of this
    bench; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  {} when no profile matches."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if (t["width"], t["height"]) != (W, H) or t.get("input", "f32") != input_kind:
            return {}
        k = frames_per_launch / t["frames_per_launch"]
        return {"traffic": int(t["hbm_bytes_per_launch"] * k), "valu": int(t.get("valu_wave_insts_per_launch", 0) * k)}
    except (OSError, KeyError, ValueError):
        return {}


def cpu_baseline(ref, dis, frames_per_thread=12):
    """The per-frame feature loop of the reference on the host cores of this box, same clip, same run: its own C code
    compiled under oracle/_ref (AVX path, as it runs on x86) when that travelled with the repo, else the oracle's C
    port.  The reference is single-threaded per video and parallel over videos / frame ranges only
    (python/src/vmaf/tools/misc.py:246-276), so the multi-core figure runs one independent frame range per thread
    (ctypes releases the GIL); `cores` = threads used = this job's CPU share (16 per GPU on the bench boxes)."""
    import concurrent.futures as cf
    import oracle
    kind = "reference" if oracle.ref_lib() is not None else "port"
    threads = max(1, min(16, os.cpu_count() or 1))

    def run_range(start):
        prev = np.zeros((H, W), np.float32)
        for i in range(frames_per_thread):
            f = (start + i) % len(ref)
            if kind == "reference":
                _, prev = oracle.ref_frame(ref[f], dis[f], prev, i == 0, avx=1)
            else:
                oracle.frame(ref[f], dis[f], prev, i == 0)
        return frames_per_thread

    t0 = time.perf_counter()
    run_range(0)
    one = frames_per_thread / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(threads) as ex:
        n = sum(ex.map(run_range, range(0, threads * frames_per_thread, frames_per_thread)))
    dt = time.perf_counter() - t0
    sample = (f"{n} frames (adm + motion + vif per frame) of the same {W}x{H} clip on {threads} threads in {dt:.1f}s; "
              f"one thread alone: {one:.2f} frames/sec")
    return {"value": round(n / dt, 3), "unit": "frames/sec", "cores": threads, "kind": kind, "sample": sample,
            "one_core_value": round(one, 3)}


def end_to_end(r8, d8, order, device):
    """Beside the headline, never `value`: the same workload when the frames start on the HOST (SURVEY.md 8d "kernel-only and
    end-to-end").  (1) luma planes in page-locked host memory through the kernel ABI with two calls in flight, so that the
    H2D copies of batch k+1 overlap the kernels of batch k: uint8 (2 B/px over PCIe, widened on the GPU) and fp32 (the
    reference's read_frame contract, 8 B/px).  (2) the CLI: vmaf_amd/vmafossexec on a 600-frame yuv420p pair in /dev/shm,
    the whole process (start-up, engine creation, file reading, scoring, fusion)."""
    import re
    import shutil
    import subprocess
    import tempfile
    from vmaf_amd.binding import PinnedArray
    out = {}
    B, K = 60, 10
    reps = (B + len(r8) - 1) // len(r8)
    src_r, src_d = np.concatenate([r8] * reps)[:B], np.concatenate([d8] * reps)[:B]
    for name, dt, bpp in (("pinned_u8", r8.dtype, 2 * r8.dtype.itemsize), ("pinned_f32", np.float32, 8)):
        pr, pd = PinnedArray(src_r.shape, dt), PinnedArray(src_d.shape, dt)
        try:
            if dt == np.float32:
                pr.array[...] = synth.to_luma_f32(src_r, BITS)
                pd.array[...] = synth.to_luma_f32(src_d, BITS)
            else:
                pr.array[...] = src_r
                pd.array[...] = src_d
            atoms = np.zeros((K, B, 20))
            with vmaf_amd.Engine(W, H, max_batch=B, device=device, order=order) as e:
                for i in range(2):                                  # warm-up: allocates both staging sets
                    e.extract_host_async(pr.array, pd.array, atoms[i], continue_sequence=i > 0)
                e.sync()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(K):
                    e.extract_host_async(pr.array, pd.array, atoms[k], continue_sequence=True)
                    if k > 0:
                        e.sync_oldest()
                e.sync()
                vmaf_amd.predict_from_atoms(atoms.reshape(K * B, 20), MODEL, device=device)
                dt_s = time.perf_counter() - t0
            fps = K * B / dt_s
            out[name] = {"value": round(fps, 1), "unit": "frames/sec", "pcie_gbs": round(fps * W * H * bpp / 1e9, 1),
                         "what": f"{np.dtype(dt).name} luma in page-locked host memory -> vmaf_hip_extract_host{'' if dt == np.float32 else '_raw'}_async, "
                                 f"two calls in flight, {B}-frame batches, {K * B} frames, SVR fusion included"}
        finally:
            pr.close()
            pd.close()
    exe = os.path.join(ROOT, "vmaf_amd", "vmafossexec")
    frame_bytes = W * H * (2 if BITS == 10 else 1) * 3 // 2
    shm_ok = os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * 1200 * frame_bytes + (1 << 30)
    if os.path.exists(exe) and shm_ok:
        fmt = "yuv420p10le" if BITS == 10 else "yuv420p"
        with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
            rp, dp = os.path.join(tmp, "ref.yuv"), os.path.join(tmp, "dis.yuv")
            synth.write_yuv(rp, r8, fmt)
            synth.write_yuv(dp, d8, fmt)
            one_r, one_d = open(rp, "rb").read(), open(dp, "rb").read()
            env = dict(os.environ, VMAF_HIP_DEVICES=str(device))
            if order == "fma":
                env["VMAF_HIP_FAST"] = "1"
            args = [exe, fmt, str(W), str(H), rp, dp, MODEL] + (["--disable-avx"] if order == "scalar" else [])
            have, walls = len(r8), {}
            for frames in (600, 1200):                              # 600 = BASELINE.json configs[1]; the longer clip separates the fixed cost
                n = max(1, frames // len(r8))
                with open(rp, "ab") as f, open(dp, "ab") as g:
                    for _ in range(n - have // len(r8)):
                        f.write(one_r)
                        g.write(one_d)
                have = n * len(r8)
                best = None
                for _ in range(2):                                  # second run: binary, library and page cache warm
                    t0 = time.perf_counter()
                    p = subprocess.run(args, capture_output=True, text=True, env=env, timeout=600)
                    wall = time.perf_counter() - t0
                    if p.returncode == 0 and (best is None or wall < best[0]):
                        best = (wall, float(re.search(r"Exec FPS: ([0-9.]+)", p.stdout).group(1)))
                if best:
                    walls[have] = best
            if len(walls) == 2:
                (f0, (w0, l0)), (f1, (w1, _)) = sorted(walls.items())
                out["vmafossexec"] = {"value": round(f0 / w0, 1), "unit": "frames/sec", "library_timer_fps": round(l0, 1),
                                      "steady_state_fps": round((f1 - f0) / max(w1 - w0, 1e-9), 1), "fixed_cost_s": round(w0 - f0 * (w1 - w0) / (f1 - f0), 3),
                                      "what": f"vmaf_amd/vmafossexec {fmt} {W} {H} on a {f0}-frame pair in /dev/shm, whole process wall time (start-up, "
                                              f"engine creation, 8 reader threads, raw luma over PCIe, scoring, fusion); steady state and fixed cost from "
                                              f"the difference to a {f1}-frame pair"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-memory / CLI figures reported beside the headline")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="1080p",
                    help="1080p = BASELINE.json configs[1], the headline workload (default); the others are for DESIGN.md's table")
    ap.add_argument("--order", choices=["avx", "scalar", "fma"], default="avx",
                    help="summation order of the convolutions: the reference's AVX order (default, bit-exact), its scalar order, "
                         "or fused multiply-adds (fast mode, within tolerance but not bit-exact; not the headline workload)")
    ap.add_argument("--repeat", type=int, default=0,
                    help="how many times the distinct synthetic frames are tiled into one batch (default: the workload's own, "
                         "4 at 1080p = 120 frames per step; 1 = 30 frames per step)")
    ap.add_argument("--distinct", type=int, default=0,
                    help="use only the first N distinct synthetic frames (with --repeat 1: N frames per step; batch-size experiments)")
    ap.add_argument("--sums-exact", action="store_true",
                    help="VIF plane sums in vif_sum_s's own order (vmaf_hip_engine_set_sums_exact; what --disable-avx also selects): VIF "
                         "atoms bit-identical to the reference's; not the headline workload")
    ap.add_argument("--metrics", action="store_true", help="also compute PSNR / SSIM / MS-SSIM per frame (not the headline workload)")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="beside the headline (never `value`): the same pipelined loop repeated three times for at least this long "
                         "each, rate of every repeat and their spread - so that a short --steps run still carries a >= 1 s figure "
                         "(0 = skip)")
    ap.add_argument("--input", choices=["raw", "f32"], default="raw",
                    help="how the luma planes are resident in HBM before the timed region: raw = the samples as the YUV file holds "
                         "them (uint8, or uint16 words for 10-bit; widened in the kernels' loads - the headline), f32 = the fp32 planes "
                         "a read_frame callback delivers (round 1-2's headline, still reported beside it as `f32_input`)")
    args = ap.parse_args()

    global W, H, BITS, FRAMES_PER_STEP
    W, H, BITS, DISTINCT, REPEAT = WORKLOADS[args.workload]
    if args.repeat > 0:
        REPEAT = args.repeat
    if args.distinct > 0:
        DISTINCT = min(DISTINCT, args.distinct)
    FRAMES_PER_STEP = DISTINCT * REPEAT
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # BENCH_REHEARSAL=1: rehearse the N > 1 control flow on a box with ONE GPU - every rank uses cuda:0 and the
    # gather goes over gloo with host tensors.  Never a measurement; the driver's runs use RCCL, one rank per GPU.
    rehearsal = world > 1 and os.environ.get("BENCH_REHEARSAL", "0") not in ("", "0")
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = torch.device("cpu") if rehearsal else dev
    B, K = FRAMES_PER_STEP, args.steps
    global VIF0_BYTES_PER_PX, PATH_BYTES_PER_PX
    if args.input == "raw":                             # "the first term drops from 8N to 2N/4N"
        in_bytes = 2.0 * (2 if BITS == 10 else 1)
        VIF0_BYTES_PER_PX = in_bytes + 2.0
        PATH_BYTES_PER_PX = 26.5 - 8.0 + in_bytes

    # ---- synthetic clip, resident in HBM: raw luma (uint8 / uint16, as the file holds it) and the fp32 planes of the read_frame contract
    r8, d8 = synth.make_clip(W, H, DISTINCT, bits=BITS, seed=1234)
    ref, dis = synth.to_luma_f32(r8, BITS), synth.to_luma_f32(d8, BITS)
    t_ref = torch.from_numpy(ref).to(dev).repeat(REPEAT, 1, 1).contiguous()
    t_dis = torch.from_numpy(dis).to(dev).repeat(REPEAT, 1, 1).contiguous()
    BPS = r8.dtype.itemsize
    as_torch = (lambda a: torch.from_numpy(a)) if BPS == 1 else (lambda a: torch.from_numpy(a.view(np.int16)))   # torch has no uint16 arithmetic; bits only
    q_ref = as_torch(r8).to(dev).repeat(REPEAT, 1, 1).contiguous()
    q_dis = as_torch(d8).to(dev).repeat(REPEAT, 1, 1).contiguous()
    torch.cuda.synchronize()
    raw_in = args.input == "raw"

    def run_batch(e, continue_sequence, out, sync=True, raw=raw_in):
        if raw:
            return e.extract_device_raw(q_ref, q_dis, B, BPS, continue_sequence=continue_sequence, out=out, sync=sync)
        return e.extract_device(t_ref, t_dis, B, continue_sequence=continue_sequence, out=out, sync=sync)

    def pipelined(e, K_, out, first_continues=False, raw=raw_in):
        """K_ batches, two in flight: batch k is enqueued before batch k-1 is collected"""
        for k in range(K_):
            run_batch(e, k > 0 or first_continues, out[k], sync=False, raw=raw)
            if k > 0:
                e.sync_oldest()
        e.sync()

    eng = vmaf_amd.Engine(W, H, max_batch=B, device=local_rank, order=args.order, sums_exact=args.sums_exact)
    if args.metrics:
        eng.L.vmaf_hip_engine_enable_metrics(eng._h, 1, 1, 1, 0)
    atoms = np.zeros((K, B, 20))
    scratch = np.zeros((B, 20))
    for i in range(args.warmup):
        run_batch(eng, i > 0, scratch)
    # untimed warm-up of everything the timed region touches for the first time otherwise: the RCCL communicator
    # (lazy initialisation at the first collective costs seconds) and the fusion call's one-off allocations
    warm = sharding.gather_rows(scratch, dist if world > 1 else None, coll_dev)
    if rank == 0:
        vmaf_amd.predict_from_atoms(warm, MODEL, device=local_rank)
    eng.arm_timing()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    k_ms = []
    barrier()
    t0 = time.perf_counter()
    if rank > 0:                                        # halo: the frame before this rank's range
        if raw_in:
            eng.motion_seed_device_raw(q_ref[B - 1], BPS)
        else:
            eng.motion_seed(t_ref[B - 1])
    for k in range(K):                                  # two batches in flight: batch k is enqueued before batch k-1 is collected
        run_batch(eng, k > 0 or rank > 0, atoms[k], sync=False)
        if k > 0:
            eng.sync_oldest()
            ms = eng.last_kernel_ms(1)
            if ms > 0:
                k_ms.append(ms)
    eng.sync()
    ms = eng.last_kernel_ms(1)
    if ms > 0:
        k_ms.append(ms)
    flat = sharding.gather_rows(atoms.reshape(K * B, 20), dist if world > 1 else None, coll_dev)   # one RCCL gather, rank order = frame order
    if rank == 0:
        vmaf, _ = vmaf_amd.predict_from_atoms(flat, MODEL, device=local_rank)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    excl_ms = None
    if rank == 0:
        # the same kernel alone on the GPU (one stream, outside the timed region): in the timed region it shares the
        # CUs with the ADM and motion streams, which stretches its duration without touching whole-job throughput
        prev_serial = os.environ.get("VMAF_HIP_SERIAL")
        os.environ["VMAF_HIP_SERIAL"] = "1"
        try:
            with vmaf_amd.Engine(W, H, max_batch=B, device=local_rank, order=args.order) as eng1:
                run_batch(eng1, False, scratch)
                eng1.arm_timing()
                xs = []
                for i in range(40):                    # the first launches after the idle gap of the engine set-up run at ramping clocks
                    run_batch(eng1, True, scratch)
                    xs.append(eng1.last_kernel_ms(1))
                xs = [x for x in xs[16:] if x > 0]
                excl_ms = float(np.mean(xs)) if xs else None
        finally:
            if prev_serial is None:
                del os.environ["VMAF_HIP_SERIAL"]
            else:
                os.environ["VMAF_HIP_SERIAL"] = prev_serial
    fast = None
    if rank == 0 and world == 1 and args.order == "avx" and not args.metrics and not args.sums_exact:
        # beside the headline: the opt-in fused-multiply-add order (conv order 2, VMAF_HIP_FAST=1) on the same workload,
        # same pipelined loop + fusion, outside the timed region; and how far its numbers are from the headline run's
        with vmaf_amd.Engine(W, H, max_batch=B, device=local_rank, order="fma") as engf:
            fa = np.zeros((K, B, 20))
            for i in range(args.warmup):
                run_batch(engf, i > 0, scratch)
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            pipelined(engf, K, fa)
            fv, ff = vmaf_amd.predict_from_atoms(fa.reshape(K * B, 20), MODEL, device=local_rank)
            torch.cuda.synchronize()
            tf = time.perf_counter() - tf0
        _, f0 = vmaf_amd.predict_from_atoms(flat, MODEL, device=local_rank)
        fast = {"order": "fma (fused multiply-add taps; within tolerance, not bit-identical to the reference)",
                "value": round(K * B / tf, 2), "unit": "frames/sec", "ms_per_step": round(1e3 * tf / K, 4),
                "max_abs_feature_diff_vs_headline": float(np.abs(ff - f0).max()),
                "max_abs_vmaf_diff_vs_headline": float(np.abs(fv - vmaf).max())}
    sustained = None
    if rank == 0 and world == 1 and args.min_seconds > 0:
        # the headline loop again, outside the timed region: three repeats of >= min_seconds each (same engine, same input)
        per_step = elapsed / K
        ks = max(K, int(np.ceil(args.min_seconds / per_step)))
        sa = np.zeros((ks, B, 20))
        rates = []
        for _ in range(3):
            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            pipelined(eng, ks, sa, first_continues=True)
            vmaf_amd.predict_from_atoms(sa.reshape(ks * B, 20), MODEL, device=local_rank)
            torch.cuda.synchronize()
            rates.append(ks * B / (time.perf_counter() - ts0))
        sustained = {"values": [round(x, 1) for x in rates], "mean": round(float(np.mean(rates)), 1), "unit": "frames/sec",
                     "spread_pct": round(100.0 * (max(rates) - min(rates)) / float(np.mean(rates)), 2),
                     "steps_per_repeat": ks, "seconds_per_repeat": round(ks * B / float(np.mean(rates)), 3),
                     "what": "the timed loop (two batches in flight + SVR fusion) repeated three times after the timed region"}
        del sa
    other = None
    if rank == 0 and world == 1 and args.order == "avx" and not args.metrics and not args.sums_exact:
        # beside the headline: the same loop with the luma resident in HBM in the other form (fp32 planes <-> raw samples)
        with vmaf_amd.Engine(W, H, max_batch=B, device=local_rank, order=args.order) as engo:
            oa = np.zeros((K, B, 20))
            for i in range(args.warmup):
                run_batch(engo, i > 0, scratch, raw=not raw_in)
            engo.arm_timing()
            torch.cuda.synchronize()
            to0 = time.perf_counter()
            pipelined(engo, K, oa, raw=not raw_in)
            vmaf_amd.predict_from_atoms(oa.reshape(K * B, 20), MODEL, device=local_rank)
            torch.cuda.synchronize()
            to = time.perf_counter() - to0
            o_ms = engo.last_kernel_ms(1)                            # the dominant kernel's launch of the last batch (HIP events)
        o_bytes = (10.0 if raw_in else 2.0 * BPS + 2.0) * W * H * B   # this form's algorithmic bytes of that launch
        other = {"value": round(K * B / to, 2), "unit": "frames/sec", "ms_per_step": round(1e3 * to / K, 4),
                 "atoms_identical_to_headline": bool(np.array_equal(oa, atoms)),
                 "roofline": ({"launch_ms": round(o_ms, 4), "algorithmic_bytes": int(o_bytes), "achieved": round(o_bytes / (o_ms * 1e-3) / 1e9, 1),
                               "unit": "GB/s", "frac": round(o_bytes / (o_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} if o_ms > 0 else None),
                 "what": ("fp32 luma planes resident in HBM (the read_frame contract; the headline of rounds 1-2)" if raw_in else
                          f"raw uint{8 * BPS} luma resident in HBM, widened in the kernels' loads") + ", same loop and fusion, outside the timed region"}
    if rank == 0:
        frames = world * K * B
        fps = frames / elapsed
        kernel_ms = float(np.mean(k_ms)) if k_ms else None
        roof = None
        if kernel_ms:
            achieved = VIF0_BYTES_PER_PX * W * H * B / (kernel_ms * 1e-3) / 1e9
            prof = committed_profile(B, args.input) if args.order == "avx" else {}     # the committed counters are of the default order
            roof = {"bound": "hbm", "kernel": (f"vif_stat_wg_kernel<17> (VIF scale 0, fused, workgroup-wide 240x8 tiles; {args.order} order)" if W % 8 == 0 and W >= 240 else
                               f"vif_stat_avx_kernel<17> / vif_stat_kernel<17> (VIF scale 0, fused, wave-private 48x32 tiles; {args.order} order)"),
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": prof.get("traffic"), "algorithmic_bytes": int(VIF0_BYTES_PER_PX * W * H * B),
                    "launch_ms": round(kernel_ms, 4), "frames_per_launch": B,
                    "path_achieved_gbs": round(PATH_BYTES_PER_PX * W * H * fps / world / 1e9, 1),
                    # ~620 fp32 flop per pixel over the whole path; mul and add stay unfused by the parity
                    # rule, so the VALU ceiling is half the FMA peak (157.3 / 2 TFLOP/s)
                    "path_achieved_tflops": round(PATH_FLOP_PER_PX * W * H * fps / world / 1e12, 2),
                    "fp32_unfused_peak_tflops": 78.6}
            if excl_ms:
                roof["exclusive"] = {"launch_ms": round(excl_ms, 4), "achieved": round(VIF0_BYTES_PER_PX * W * H * B / (excl_ms * 1e-3) / 1e9, 1),
                                     "frac": round(VIF0_BYTES_PER_PX * W * H * B / (excl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if prof.get("valu"):
                # what actually binds this kernel: fp32 VALU issue, unfused mul + add by the parity rule
                rate = prof["valu"] / (kernel_ms * 1e-3)
                roof["valu_issue"] = {"wave_insts_per_launch": prof["valu"], "achieved": round(rate / 1e12, 3),
                                      "peak": round(VALU_PEAK_WINST / 1e12, 3), "unit": "T wave-inst/s",
                                      "frac": round(rate / VALU_PEAK_WINST, 3)}
                if excl_ms:
                    roof["valu_issue"]["exclusive_frac"] = round(prof["valu"] / (excl_ms * 1e-3) / VALU_PEAK_WINST, 3)
        line = {"metric": "frames/sec (whole node) 1080p VMAF v0.6.1" if args.workload == "1080p" else f"frames/sec (whole node) {args.workload} VMAF v0.6.1", "value": round(fps, 2), "unit": "frames/sec",
                "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / K, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"synthetic {W}x{H} yuv420p {BITS}-bit ref/dis pair, vmaf_v0.6.1, " +
                                       (f"raw uint{8 * BPS} luma resident in HBM (widened to fp32 in the kernels' loads)" if raw_in else "fp32 luma resident in HBM") +
                                       (" + psnr/ssim/ms_ssim" if args.metrics else "") + (", VIF sums in the reference's order" if args.sums_exact else ""),
                           "frames_per_step_per_gpu": B, "distinct_frames": DISTINCT, "frames_total": frames,
                           "parallelism": f"frame-range x{world}, 1-frame motion halo" if world > 1 else "single GPU"},
                "mean_vmaf": round(float(np.mean(vmaf)), 4), "roofline": roof}
        if fast:
            line["fast_mode"] = fast
        if other:
            line["f32_input" if raw_in else "raw_input"] = other
        if sustained:
            line["sustained"] = sustained
        if args.order != "avx":
            line["config"]["conv_order"] = args.order
        # the two side figures must never cost the headline line: a full /dev/shm or a missing oracle build is reported, not raised
        if world == 1 and not args.no_end_to_end and not args.metrics:
            try:
                line["end_to_end"] = end_to_end(r8, d8, args.order, local_rank)
            except Exception as exc:                                  # noqa: BLE001
                line["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(ref, dis)
            except Exception as exc:                                  # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
