"""Frame-range sharding of ONE video over several engines (BASELINE.json configs[2] and [4]): the file-backed
CLIs split a seekable pair into contiguous ranges, one host thread + engine + reader per entry of VMAF_HIP_DEVICES, each
range with its one-frame motion halo, motion2 after the merge.  On the one-GPU box the list repeats device 0; what is
checked is that sharding changes nothing: logs and outputs byte-identical to the unsharded run, with scene cuts placed
exactly on the shard edges k*F/G and one frame either side of them, with the phone model (score transform)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import vmaf_amd
from vmaf_amd import cli, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "vmaf_amd", "vmafossexec")
RUN = [sys.executable, os.path.join(ROOT, "run_vmaf")]
MODEL = os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl")


@pytest.fixture(scope="module")
def clip_1080p(tmp_path_factory):
    """12 frames of 1920x1080; shard edges: 6 (G=2), 4 and 8 (G=3).  Cuts on them and one frame either side."""
    tmp = tmp_path_factory.mktemp("shard1080")
    r, d = synth.make_clip(1920, 1080, 12, seed=99, cuts=(3, 4, 6, 7, 9))
    rp, dp = str(tmp / "ref.yuv"), str(tmp / "dis.yuv")
    synth.write_yuv(rp, r, "yuv420p")
    synth.write_yuv(dp, d, "yuv420p")
    return rp, dp, r, d


def _exec(rp, dp, w, h, fmt, log, env, extra=()):
    p = subprocess.run([EXE, fmt, str(w), str(h), rp, dp, MODEL, "--log", log] + list(extra), capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, **env))
    assert p.returncode == 0, p.stderr
    return re.sub(r"Exec FPS: [0-9.]+", "", p.stdout), re.sub(r'execFps="[^"]*"', "", open(log).read())


def test_vmafossexec_shards_give_byte_identical_logs(clip_1080p, tmp_path):
    rp, dp, r, d = clip_1080p
    base = _exec(rp, dp, 1920, 1080, "yuv420p", str(tmp_path / "a.xml"), {"VMAF_HIP_BATCH": "5"}, ["--phone-model"])
    assert 'numOfFrames="12"' in base[1]
    # the sequential (FIFO) path is the same arithmetic
    assert _exec(rp, dp, 1920, 1080, "yuv420p", str(tmp_path / "s.xml"), {"VMAF_HIP_SEEKABLE": "0"}, ["--phone-model"]) == base
    for devs in ("0,0", "0,0,0", "0,0,0,0,0"):
        got = _exec(rp, dp, 1920, 1080, "yuv420p", str(tmp_path / "b.xml"), {"VMAF_HIP_DEVICES": devs}, ["--phone-model"])
        assert got == base, devs
    # motion really changes at the cuts (otherwise the edges would prove nothing), and the shards agree with one engine
    with vmaf_amd.Engine(1920, 1080, max_batch=4) as e:
        atoms = e.extract_raw(r, d)
    assert atoms[4, 10] > 3 * atoms[5, 10] and atoms[6, 10] > 3 * atoms[5, 10]
    vm, _ = vmaf_amd.predict_from_atoms(atoms, MODEL, enable_transform=True)
    frames = re.findall(r'vmaf="([^"]+)"', base[1])
    assert len(frames) == 12 and all(abs(float(v) - x) < 5e-4 * max(1.0, x) for v, x in zip(frames, vm))


def test_four_shards_keep_their_reader_threads(clip_1080p, tmp_path):
    """The reader threads are a per-shard resource: four shards of one video (here on the one GPU) each get at least four
    (round 2 divided a process-wide default of 8 over the shards: 8 GPUs -> one copying thread per shard, i.e. the whole
    node at about the rate of one well-fed GPU).  The shard lines of VMAF_HIP_TRACE name range and readers; the log of the
    sharded run is the unsharded one's byte for byte."""
    rp, dp, _, _ = clip_1080p
    env = {k: v for k, v in os.environ.items() if k != "VMAF_HIP_READ_THREADS"}
    base = _exec(rp, dp, 1920, 1080, "yuv420p", str(tmp_path / "a.xml"), {})
    log = str(tmp_path / "b.xml")
    p = subprocess.run([EXE, "yuv420p", "1920", "1080", rp, dp, MODEL, "--log", log], capture_output=True, text=True, timeout=300,
                       env=dict(env, VMAF_HIP_DEVICES="0,0,0,0", VMAF_HIP_TRACE="1"))
    assert p.returncode == 0, p.stderr
    assert (re.sub(r"Exec FPS: [0-9.]+", "", p.stdout), re.sub(r'execFps="[^"]*"', "", open(log).read())) == base
    shards = sorted((int(a), int(b), int(n)) for a, b, n in re.findall(r"shard: frames \[(\d+), (\d+)\), (\d+) reader threads", p.stderr))
    assert [(a, b) for a, b, _ in shards] == [(0, 3), (3, 6), (6, 9), (9, 12)]
    assert all(n >= 4 for _, _, n in shards), shards
    assert cli._read_threads(8) >= 4 and cli._read_threads(1) >= 4


def test_sharded_metrics_and_more_shards_than_frames(tmp_path):
    r, d = synth.make_clip(576, 324, 5, seed=5, cuts=(2,))
    rp, dp = str(tmp_path / "r.yuv"), str(tmp_path / "d.yuv")
    synth.write_yuv(rp, r, "yuv422p")
    synth.write_yuv(dp, d, "yuv422p")
    flags = ["--psnr", "--ssim", "--ms-ssim", "--log-fmt", "json", "--pool", "harmonic_mean"]
    base = _exec(rp, dp, 576, 324, "yuv422p", str(tmp_path / "a.json"), {}, flags)
    for devs in ("0,0", "0,0,0,0,0,0,0,0"):                   # 8 entries for 5 frames: the list is cut to one frame per shard
        assert _exec(rp, dp, 576, 324, "yuv422p", str(tmp_path / "b.json"), {"VMAF_HIP_DEVICES": devs, "VMAF_HIP_BATCH": "2"}, flags) == base
    assert {"psnr", "ssim", "ms_ssim", "vmaf"} <= set(json.loads(base[1])["metrics"])


def test_run_vmaf_shards_give_identical_output(clip_1080p):
    rp, dp, _, _ = clip_1080p
    outs = []
    for env in ({}, {"VMAF_HIP_DEVICES": "0,0"}, {"VMAF_HIP_DEVICES": "0,0,0"}):
        p = subprocess.run(RUN + ["yuv420p", "1920", "1080", rp, dp, "--phone-model", "--out-fmt", "json"], capture_output=True,
                           text=True, timeout=300, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr
        outs.append(p.stdout)
    assert outs[1] == outs[0] and outs[2] == outs[0]
    assert len(json.loads(outs[0])["frames"]) == 12


def test_4k_10bit_pair_is_streamed_not_loaded(tmp_path, monkeypatch):
    """configs[2] in small: a 3840x2160 yuv420p10le pair scored through the launcher's code path while no more than one
    batch of frames per file is ever copied out of the file maps, on three shards; equal to the one-shard result."""
    r, d = synth.make_clip(3840, 2160, 3, bits=10, seed=8)
    rp, dp = str(tmp_path / "r.yuv"), str(tmp_path / "d.yuv")
    synth.write_yuv(rp, np.concatenate([r, r[::-1], r]), "yuv420p10le")      # 9 frames
    synth.write_yuv(dp, np.concatenate([d, d[::-1], d]), "yuv420p10le")
    asked = []
    real = cli.LumaFile.planes

    def spy(self, start, stop, out=None):
        asked.append(stop - start)
        assert out is not None                                  # always into a page-locked batch buffer, never a fresh array
        return real(self, start, stop, out)

    monkeypatch.setattr(cli.LumaFile, "planes", spy)
    monkeypatch.setattr(cli, "read_luma", lambda *a, **k: (_ for _ in ()).throw(AssertionError("whole-file read")))
    asset = cli.Asset("cmd", 0, 0, rp, dp, 3840, 2160, "yuv420p10le")
    one = cli.score_asset(asset, None, False)
    assert max(asked) <= cli._batch_size(3840, 2160) == 4
    three = cli.score_asset(asset, None, False, devices=[0, 0, 0])
    assert one.scores == three.scores and len(one.scores["VMAF_scores"]) == 9
    # frames 0..2 and 6..8 are the same pictures: same VIF/ADM features, motion differs only at the joins
    a = np.array(one.scores["VMAF_feature_vif_scale0_scores"])
    assert np.array_equal(a[:3], a[6:])


def test_bench_multi_rank_control_flow_rehearsal():
    """bench.py's N > 1 path (frame ranges per rank, motion halo, gather to rank 0, fusion there) run once with two ranks on
    the one GPU (BENCH_REHEARSAL=1: gloo for the gather, both ranks on cuda:0), so that the driver's multi-GPU run is not
    the first execution of that control flow.  Not a measurement."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "720p", "--repeat", "1", "--no-cpu-baseline", "--no-end-to-end"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=dict(os.environ, BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["value"] > 0 and doc["scaling"] == "weak"
    assert doc["config"]["frames_total"] == 2 * 3 * doc["config"]["frames_per_step_per_gpu"]
    assert 0.0 < doc["mean_vmaf"] <= 100.0


def test_seekable_entry_from_a_non_cli_caller(tmp_path, monkeypatch):
    """compute_vmaf_raw_seekable through ctypes with a positional callback that owns the frames in memory (what a decoder
    with random access binds): same log as the pull-based compute_vmaf on the same frames, for 1 and 3 shards x 1 and 4
    reader threads; every frame is asked for exactly once plus one halo frame per later shard; a failing read ends the
    call with the frame named instead of a hang or a partial score."""
    from vmaf_amd import binding
    r, d = synth.make_clip(320, 180, 13, bits=10, seed=31, cuts=(5,))
    log0 = str(tmp_path / "pull.xml")
    want = binding.compute_vmaf_from_arrays(synth.to_luma_f32(r, 10), synth.to_luma_f32(d, 10), fmt="yuv420p10le", log_path=log0, phone_model=True)
    base = re.sub(r'execFps="[^"]*"', "", open(log0).read())
    for devs, threads in (("0", "1"), ("0", "4"), ("0,0,0", "1"), ("0,0,0", "12")):
        monkeypatch.setenv("VMAF_HIP_DEVICES", devs)
        monkeypatch.setenv("VMAF_HIP_READ_THREADS", threads)
        monkeypatch.setenv("VMAF_HIP_BATCH", "4")
        log = str(tmp_path / "seek.xml")
        got, seen = binding.compute_vmaf_seekable_from_arrays(r, d, fmt="yuv420p10le", log_path=log, phone_model=True)
        assert got == want
        assert re.sub(r'execFps="[^"]*"', "", open(log).read()) == base
        shards = devs.count(",") + 1
        halos = sorted(13 * g // shards - 1 for g in range(1, shards))
        assert seen == sorted(list(range(13)) + halos)
        with pytest.raises(vmaf_amd.VmafHipError, match="read_frame failed at frame 9"):
            binding.compute_vmaf_seekable_from_arrays(r, d, fmt="yuv420p10le", fail_at=9)
