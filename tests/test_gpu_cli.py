"""GPU tests of the CLI boundary: vmaf_amd/vmafossexec (this repo's CLI) and oracle/_ref/vmafossexec_dropin
(the REFERENCE's own main.cpp linked against libvmaf_amd.so) on raw YUV files, against what the reference's
vmafossexec printed and logged for the same files (tests/golden/exec.json).

Tolerances: per-frame features 1e-4, VMAF 1e-3 (BASELINE.json); XML values carry 6 significant digits."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from vmaf_amd import synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
EXEC = json.load(open(os.path.join(GOLD, "exec.json")))
ATOMS = json.load(open(os.path.join(GOLD, "atoms.json")))
OURS = os.path.join(ROOT, "vmaf_amd", "vmafossexec")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "vmafossexec_dropin")


def _files(case, tmp_path):
    if case["clip"].startswith("images/"):
        n = case["clip"].split("/")[1]
        return os.path.join(GOLD, "images", f"{n}.yuv"), os.path.join(GOLD, "images", f"{n}_noisy.yuv"), 481, 321
    kw = dict(ATOMS[case["clip"]]["params"])
    bits = kw.pop("bits", 8)
    if "cuts" in kw:
        kw["cuts"] = tuple(kw["cuts"])
    r, d = synth.make_clip(bits=bits, **kw)
    rp, dp = str(tmp_path / "ref.yuv"), str(tmp_path / "dis.yuv")
    synth.write_yuv(rp, r, case["fmt"])
    synth.write_yuv(dp, d, case["fmt"])
    return rp, dp, r.shape[2], r.shape[1]


def _run(exe, case, tmp_path, log_fmt="xml"):
    rp, dp, w, h = _files(case, tmp_path)
    log = str(tmp_path / f"log.{log_fmt}")
    cmd = [exe, case["fmt"], str(w), str(h), rp, dp, os.path.join(ROOT, case["model"]), "--log", log] + case["args"]
    if log_fmt == "json":
        cmd += ["--log-fmt", "json"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr + p.stdout
    return p.stdout, open(log).read()


def _check_against_golden(stdout, xml, case):
    assert "Start calculating VMAF score..." in stdout and "Exec FPS:" in stdout
    m = re.search(r"VMAF score(?: \((\w+)\))? = ([-0-9.]+)", stdout)
    assert m and m.group(1) == case["pool"]
    assert abs(float(m.group(2)) - case["stdout_score"]) < 1e-3
    assert xml.startswith('<?xml version="1.0"?>\n<VMAF version="0.3.2">\n\t<params model="')
    frames = [dict(re.findall(r'(\w+)="([^"]*)"', fm.group(1))) for fm in re.finditer(r"<frame ([^/]*)/>", xml)]
    assert len(frames) == len(case["frames"])
    for got, want in zip(frames, case["frames"]):
        assert list(got.keys()) == list(want.keys())              # same attributes, same (alphabetical) order
        for k, v in want.items():
            tol = 1e-3 if k == "vmaf" else 1e-4
            assert abs(float(got[k]) - v) <= tol + 6e-6 * abs(v), (k, got[k], v)
    agg = float(re.search(r'aggregateVMAF="([^"]*)"', xml).group(1))
    assert abs(agg - case["aggregate_xml"]) < 1e-3


@pytest.mark.parametrize("name", sorted(EXEC))
def test_vmafossexec_matches_reference_outputs(name, tmp_path):
    case = EXEC[name]
    stdout, xml = _run(OURS, case, tmp_path)
    _check_against_golden(stdout, xml, case)


@pytest.mark.parametrize("name", ["default", "tenbit", "image_100007", "metrics", "metrics_tenbit"])
def test_reference_main_runs_on_this_library(name, tmp_path):
    """Drop-in proof: the reference's wrapper/src/main.cpp + common/frame.c, unmodified, on libvmaf_amd.so."""
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/vmafossexec_dropin not built (needs /root/reference at build time)")
    case = EXEC[name]
    stdout, xml = _run(DROPIN, case, tmp_path)
    _check_against_golden(stdout, xml, case)


LOGS = json.load(open(os.path.join(GOLD, "exec_logs.json")))
_NUM = re.compile(r"-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?")


def _same_text_up_to_numbers(ours, ref, what):
    """Token for token: everything that is not a number must be the same bytes (element and attribute names, order,
    quoting, indentation, line breaks); numbers must agree within the parity tolerances (features 1e-4, scores 1e-3, plus
    the rounding step of the log format) and whole numbers must be PRINTED the same way ("0" in the XML, "0.0" in the
    JSON log - pugixml's and ptools' number styles)."""
    mask = lambda t: re.sub(r'(execFps="|Exec FPS: )[^"\n]*', r"\1#", t)     # noqa: E731 - run time is not content
    ours, ref = mask(ours), mask(ref)
    assert _NUM.sub("#", ours) == _NUM.sub("#", ref), what
    for a, b in zip(_NUM.findall(ours), _NUM.findall(ref)):
        va, vb = float(a), float(b)
        if vb == int(vb):
            assert a == b, (what, a, b)
        else:
            assert abs(va - vb) <= (2e-4 if abs(vb) < 2 else 1.5e-3), (what, a, b)


@pytest.mark.parametrize("name", sorted(LOGS))
def test_log_text_matches_the_reference_token_for_token(name, tmp_path):
    """tests/golden/exec_logs.json holds what the reference's vmafossexec printed and logged (make_golden.py) as TEXT:
    pins the writers of wrapper/src/vmaf.cpp:664-750, number formatting included."""
    case, want = EXEC[name], LOGS[name]
    stdout, xml = _run(OURS, case, tmp_path, "xml")
    _same_text_up_to_numbers(stdout, want["stdout"], f"{name} stdout")
    _same_text_up_to_numbers(xml, want["xml"], f"{name} xml")
    _, js = _run(OURS, case, tmp_path, "json")
    _same_text_up_to_numbers(js, want["json"], f"{name} json")


def test_json_log_schema(tmp_path):
    case = EXEC["default"]
    _, text = _run(OURS, case, tmp_path, "json")
    doc = json.loads(text)
    assert list(doc.keys()) == ["version", "params", "metrics", "frames"]          # vmaf.cpp:709-713
    assert doc["version"] == "0.3.2" and doc["params"] == {"model": "vmaf_v0.6.1.pkl", "scaledWidth": 576, "scaledHeight": 324}
    assert doc["metrics"] == ["adm2", "motion2", "vif_scale0", "vif_scale1", "vif_scale2", "vif_scale3", "vmaf"]
    for i, (fr, want) in enumerate(zip(doc["frames"], case["frames"])):
        assert fr["frameNum"] == i
        for k in doc["metrics"]:
            assert abs(fr["metrics"][k] - want[k]) <= (1e-3 if k == "vmaf" else 1e-4) + 1e-5
            assert round(fr["metrics"][k], 5) == fr["metrics"][k]                    # rounded to 5 digits, vmaf.cpp:702
    assert text.startswith('{\n  "version":"0.3.2",\n  "params":{\n    "model":"vmaf_v0.6.1.pkl",')


def test_multi_batch_and_multi_engine_paths(tmp_path):
    """compute_vmaf's batching (VMAF_HIP_BATCH) and its round-robin over engines with a motion halo
    (VMAF_HIP_DEVICES listing the same GPU twice) must not change a single printed digit."""
    case = EXEC["default"]
    rp, dp, w, h = _files(case, tmp_path)
    outs = []
    for env in ({}, {"VMAF_HIP_BATCH": "1"}, {"VMAF_HIP_BATCH": "3"}, {"VMAF_HIP_DEVICES": "0,0", "VMAF_HIP_BATCH": "1"}):
        log = str(tmp_path / "l.xml")
        p = subprocess.run([OURS, case["fmt"], str(w), str(h), rp, dp, os.path.join(ROOT, case["model"]), "--log", log],
                           capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr
        outs.append(re.sub(r'execFps="[^"]*"', "", open(log).read()))
    assert all(o == outs[0] for o in outs[1:])


def test_truncated_and_missing_inputs(tmp_path):
    case = EXEC["default"]
    rp, dp, w, h = _files(case, tmp_path)
    model = os.path.join(ROOT, case["model"])
    # dis one frame shorter: scoring stops at the first failed luma read, like frame.c:48-78 (EOF -> 2)
    frame_bytes = w * h * 3 // 2
    data = open(dp, "rb").read()
    open(dp, "wb").write(data[:3 * frame_bytes])
    p = subprocess.run([OURS, case["fmt"], str(w), str(h), rp, dp, model, "--log", str(tmp_path / "t.xml")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'numOfFrames="3"' in open(tmp_path / "t.xml").read()
    # unknown model -> exception text + usage, exit -1 (main.cpp:225-230)
    p = subprocess.run([OURS, case["fmt"], str(w), str(h), rp, dp, "/nonexistent.pkl"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 255 and "Error: Error loading model (.pkl)" in p.stderr and "Usage:" in p.stderr
    # --ms-ssim on a frame too small for five scales: the reference fails in compute_ms_ssim ("scale below 1x1",
    # ms_ssim.c:172-177 -> combo error -> exception -> exit -1); here the same message, loudly, not a silent skip
    r, d = synth.make_clip(64, 48, 1, seed=1)
    synth.write_yuv(rp, r, "yuv420p")
    synth.write_yuv(dp, d, "yuv420p")
    p = subprocess.run([OURS, "yuv420p", "64", "48", rp, dp, model, "--ms-ssim"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 255 and "scale below 1x1" in p.stderr


@pytest.mark.parametrize("name", ["default", "tenbit"])
def test_raw_and_fp32_callbacks_give_identical_logs(name, tmp_path):
    """vmafossexec hands raw luma to compute_vmaf_raw by default (GPU-side widening); VMAF_HIP_FP32_CALLBACK=1 keeps the
    reference's fp32 read_frame contract.  Same frames, same numbers, to the last printed digit."""
    case = EXEC[name]
    rp, dp, w, h = _files(case, tmp_path)
    logs = []
    for env_val in ("0", "1"):
        log = str(tmp_path / f"log{env_val}.json")
        cmd = [OURS, case["fmt"], str(w), str(h), rp, dp, os.path.join(ROOT, case["model"]), "--log", log, "--log-fmt", "json"] + case["args"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, VMAF_HIP_FP32_CALLBACK=env_val))
        assert p.returncode == 0, p.stderr + p.stdout
        logs.append(json.load(open(log))["frames"])
    assert logs[0] == logs[1] and len(logs[0]) == len(case["frames"])


def test_fifo_inputs_and_reader_thread_settings_give_the_same_log(tmp_path):
    """Regular files go through the pread reader pool (VMAF_HIP_READ_THREADS, default 8), pipes through the sequential
    fread path that skips chroma by reading (frame.c:80-124 does the same): identical logs either way."""
    case = EXEC["default"]
    rp, dp, w, h = _files(case, tmp_path)
    model = os.path.join(ROOT, case["model"])
    logs = []
    for threads in ("8", "3", "1"):
        log = str(tmp_path / f"t{threads}.xml")
        p = subprocess.run([OURS, case["fmt"], str(w), str(h), rp, dp, model, "--log", log], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, VMAF_HIP_READ_THREADS=threads))
        assert p.returncode == 0, p.stderr
        logs.append(re.sub(r'execFps="[^"]*"', "", open(log).read()))
    fr, fd = str(tmp_path / "ref.fifo"), str(tmp_path / "dis.fifo")
    os.mkfifo(fr)
    os.mkfifo(fd)
    # the feeders open the FIFOs for writing in their own processes (that open blocks until vmafossexec opens the read end)
    feeders = [subprocess.Popen(["sh", "-c", 'exec cat "$0" > "$1"', src, dst]) for src, dst in ((rp, fr), (dp, fd))]
    log = str(tmp_path / "fifo.xml")
    try:
        p = subprocess.run([OURS, case["fmt"], str(w), str(h), fr, fd, model, "--log", log], capture_output=True, text=True, timeout=60)
        for f in feeders:
            f.wait(timeout=30)
    finally:
        for f in feeders:                                         # never leave a feeder blocked on its FIFO behind
            if f.poll() is None:
                f.kill()
    assert p.returncode == 0, p.stderr
    logs.append(re.sub(r'execFps="[^"]*"', "", open(log).read()))
    assert all(x == logs[0] for x in logs[1:])


def test_plain_c_caller_on_the_drop_in_boundary(tmp_path):
    """tests/c_caller/c_caller.c: a C program (gcc, no C++ runtime of its own, no HIP headers) that owns the frames and hands
    them over through read_frame, linked against libvmaf_amd.so like a C host links the reference's libvmaf.  Its score is the
    reference executable's; a callback failure (return 1) comes back as an error code + message instead of an exception."""
    case = EXEC["default"]
    rp, dp, w, h = _files(case, tmp_path)
    exe = str(tmp_path / "c_caller")
    libdir = os.path.join(ROOT, "vmaf_amd")
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                         os.path.join(HERE, "c_caller", "c_caller.c"), "-o", exe, "-L", libdir, "-lvmaf_amd", f"-Wl,-rpath,{libdir}"],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr
    model = os.path.join(ROOT, case["model"])
    p = subprocess.run([exe, rp, dp, str(w), str(h), model], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    got = float(re.search(r"^score ([-0-9.]+)$", p.stdout, re.M).group(1))
    assert abs(got - case["stdout_score"]) < 1e-3
    p = subprocess.run([exe, rp, dp, str(w), str(h), model, "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 3 and re.search(r"^error .+", p.stdout, re.M), p.stdout + p.stderr


def test_fast_mode_env_keeps_the_cli_inside_the_tolerances(tmp_path):
    """VMAF_HIP_FAST=1: compute_vmaf runs the fused order (conv order 2).  Same schema, values within 1e-4 / 1e-3 of what
    the reference's vmafossexec printed for the same files; --disable-avx still selects the reference's scalar order."""
    case = EXEC["default"]
    rp, dp, w, h = _files(case, tmp_path)
    log = str(tmp_path / "fast.xml")
    cmd = [OURS, case["fmt"], str(w), str(h), rp, dp, os.path.join(ROOT, case["model"]), "--log", log] + case["args"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, VMAF_HIP_FAST="1"))
    assert p.returncode == 0, p.stderr + p.stdout
    _check_against_golden(p.stdout, open(log).read(), case)


FEATURE_TOOL = json.load(open(os.path.join(GOLD, "feature_tool.json")))


@pytest.mark.parametrize("name", sorted(FEATURE_TOOL))
def test_feature_tool_speaks_the_reference_grammar(name, tmp_path):
    """vmaf_amd/vmaf = the reference's per-feature tool `feature/vmaf app fmt ref dis w h` (vmaf_main.c): the same lines
    in the same order ("<atom>: <frame> <%f>", all.c:138-226) as the reference tool printed for the same files
    (tests/golden/feature_tool.json), values within 1e-4 absolute / 2e-6 relative (the ADM / ANSNR plane sums are reduced in fp64 here); the vif_* and motion
    lines - summed in the reference's own order - identical as text."""
    case = FEATURE_TOOL[name]
    kw = dict(ATOMS[case["clip"]]["params"])
    bits = kw.pop("bits", 8)
    if "cuts" in kw:
        kw["cuts"] = tuple(kw["cuts"])
    r, d = synth.make_clip(bits=bits, **kw)
    rp, dp = str(tmp_path / "ref.yuv"), str(tmp_path / "dis.yuv")
    synth.write_yuv(rp, r, case["fmt"])
    synth.write_yuv(dp, d, case["fmt"])
    p = subprocess.run([os.path.join(ROOT, "vmaf_amd", "vmaf"), case["app"], case["fmt"], rp, dp, str(r.shape[2]), str(r.shape[1])],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr + p.stdout
    got, want = p.stdout.splitlines(), case["stdout"].splitlines()
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        gk, gf, gv = g.split()
        wk, wf, wv = w_.split()
        assert (gk, gf) == (wk, wf), (g, w_)
        assert abs(float(gv) - float(wv)) <= 1e-4 + 2e-6 * abs(float(wv)), (g, w_)
        if gk.startswith("vif") or gk.startswith("motion"):            # summed in the reference's own order: the same text
            assert g == w_, (g, w_)
    # usage errors: exit 2 (vmaf_main.c:161-176)
    assert subprocess.run([os.path.join(ROOT, "vmaf_amd", "vmaf"), "all"], capture_output=True, timeout=300).returncode == 2
