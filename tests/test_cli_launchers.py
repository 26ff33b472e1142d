"""run_vmaf / run_vmaf_in_batch (vmaf_amd/cli.py): argv contract and exit codes on CPU (run_vmaf:29-72,
run_vmaf_in_batch:31-96), output schema and scores on the GPU (python/src/vmaf/core/result.py:55-293)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from vmaf_amd import cli, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = [sys.executable, os.path.join(ROOT, "run_vmaf")]
BATCH = [sys.executable, os.path.join(ROOT, "run_vmaf_in_batch")]
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "exec.json")))
ATOMS = json.load(open(os.path.join(ROOT, "tests", "golden", "atoms.json")))


def test_run_vmaf_usage_errors_exit_2():
    for args in ([], ["yuv420p", "64", "48", "a"], ["yuv420p", "x", "48", "a", "b"], ["nv12", "64", "48", "a", "b"],
                 ["yuv420p", "64", "48", "a", "b", "--out-fmt", "csv"], ["yuv420p", "64", "48", "a", "b", "--pool", "max"],
                 ["yuv420p", "-1", "48", "a", "b"]):
        p = subprocess.run(RUN + args, capture_output=True, text=True, timeout=300)
        assert p.returncode == 2, args
    assert "usage: run_vmaf fmt width height ref_path dis_path" in subprocess.run(RUN, capture_output=True, text=True, timeout=300).stdout


def test_batch_file_grammar(tmp_path):
    f = tmp_path / "in.txt"
    f.write_text("# comment\n\nyuv420p 576 324 r0.yuv d0.yuv\nyuv422p10le 64 48 r1.yuv d1.yuv\n")
    assets, rc = cli.parse_batch_file(str(f))
    assert rc == 0 and [a.asset_id for a in assets] == [0, 1]
    assert assets[1].asset_dict == {"width": 64, "height": 48, "yuv_type": "yuv422p10le"}
    f.write_text("yuv420p 576 324 r0.yuv d0.yuv\nbogus 1 2 a b\n")
    assert subprocess.run(BATCH + [str(f)], capture_output=True, text=True, timeout=300).returncode == 1
    assert subprocess.run(BATCH, capture_output=True, text=True, timeout=300).returncode == 2
    assert subprocess.run(BATCH + [str(f), "--out-fmt", "csv"], capture_output=True, text=True, timeout=300).returncode == 2


def test_result_formats_follow_the_reference_schema():
    a = cli.Asset("cmd", 0, 3, "/x/ref_clip.yuv", "/y/dis_clip.yuv", 576, 324, "yuv420p")
    r = cli.Result(a, {"VMAF_scores": [70.0, 80.0], "VMAF_feature_adm2_scores": [0.9, 0.95], "VMAF_feature_motion2_scores": [0.0, 3.5]})
    text = r.to_string().splitlines()
    assert text[0].startswith('Asset: {"asset_dict": {"height": 324, "width": 576, "yuv_type": "yuv420p"}, "asset_id": 3,')
    assert text[1] == "Executor: VMAF_VF0.2.4b-0.6.1" and text[2] == "Result:"
    assert text[3] == "Frame 0: VMAF_feature_adm2_score:0.900000, VMAF_feature_motion2_score:0.000000, VMAF_score:70.000000"
    assert text[5] == "Aggregate (mean): VMAF_feature_adm2_score:0.925000, VMAF_feature_motion2_score:1.750000, VMAF_score:75.000000"
    d = r.to_dict()
    assert list(d.keys()) == ["executorId", "asset", "frames", "aggregate"]
    assert d["asset"]["identifier"] == "cmd_0_3_ref_clip_576x324_vs_dis_clip_576x324_q_576x324"
    assert d["frames"][1] == {"frameNum": 1, "VMAF_feature_adm2_score": 0.95, "VMAF_feature_motion2_score": 3.5, "VMAF_score": 80.0}
    assert d["aggregate"]["method"] == "mean" and d["aggregate"]["VMAF_score"] == 75.0
    r.set_pool("harmonic_mean")
    assert abs(r.aggregate("VMAF_scores") - (1.0 / np.mean(1.0 / (np.array([70.0, 80.0]) + 1.0)) - 1.0)) < 1e-12
    r.set_pool("perc10")
    assert r.aggregate("VMAF_scores") == np.percentile([70.0, 80.0], 10)
    xml = r.to_xml().splitlines()
    assert xml[1] == '<result executorId="VMAF_VF0.2.4b-0.6.1">'
    # Python 2's str(float) and attribute order: whole numbers keep '.0', frameNum / method come last
    assert xml[4] == '    <frame VMAF_feature_adm2_score="0.9" VMAF_feature_motion2_score="0.0" VMAF_score="70.0" frameNum="0"/>'
    assert xml[7] == '  <aggregate VMAF_feature_adm2_score="0.905" VMAF_feature_motion2_score="0.35" VMAF_score="71.0" method="perc10"/>'


def test_result_matches_the_reference_held_golden():
    """python/test/result_test.py:148-200: ResultFormattingTest loads python/test/resource/ssim_result_for_test.txt and
    pins to_xml() as a string and to_dict() as a value.  The same values (tests/golden/result_fixture.json) through
    cli.Result must give the same bytes."""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "result_fixture.json")))
    a = fx["asset"]
    asset = cli.Asset(a["dataset"], a["content_id"], a["asset_id"], a["ref_path"], a["dis_path"], a["width"], a["height"], "yuv420p")
    r = cli.Result(asset, fx["scores"], executor_id=fx["executor_id"])
    assert r.to_xml().strip() == "\n".join(fx["xml"])
    assert r.to_dict() == fx["dict"]
    assert list(r.to_dict()["frames"][0].keys())[0] == "frameNum" and list(r.to_dict()["aggregate"].keys())[-1] == "method"
    text = r.to_json()
    assert json.loads(text) == fx["dict"]
    for line in fx["json_aggregate_lines"]:                   # aggregates with numpy.float64's 17 digits, as Python 2 printed them
        assert line in text.split("\n")
    assert cli.py2_str_float(100.0) == "100.0" and cli.py2_str_float(1e-5) == "1e-05" and cli.py2_str_float(76.69927127248) == "76.6992712725"
    assert cli.py2_str_float(float("inf")) == "inf" and cli.py2_str_float(float("nan")) == "nan"


def _fake_worker(kind, idx, q):
    if kind == "ok":
        q.put((idx, {"VMAF_scores": [1.0 + idx]}))
        q.put(None)
    elif kind == "raises":
        q.put(("error", idx, "VmafHipError: short file"))
        q.put(None)
    elif kind == "dies":
        os._exit(3)                                           # no sentinel, no message: the OOM killer, a GPU fault


@pytest.mark.parametrize("kinds,want", [(("ok", "ok"), None), (("ok", "raises"), "asset 1 (r1 vs d1): VmafHipError: short file"),
                                        (("dies", "ok"), "ended with exit code 3")])
def test_batch_parent_never_waits_on_a_dead_worker(kinds, want):
    """run_vmaf_in_batch --parallelize (cli._collect_from_workers): a worker that fails or dies before posting its
    sentinel ends the run with a message, within the poll interval, instead of leaving the parent in q.get() forever."""
    import multiprocessing as mp
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    assets = [cli.Asset("cmd", 0, i, f"r{i}", f"d{i}", 64, 48, "yuv420p") for i in range(len(kinds))]
    results = [None] * len(kinds)
    procs = [ctx.Process(target=_fake_worker, args=(k, i, q)) for i, k in enumerate(kinds)]
    for p in procs:
        p.start()
    t0 = time.time()
    failure = cli._collect_from_workers(procs, q, assets, results, poll_s=0.2)
    assert time.time() - t0 < 60
    assert all(not p.is_alive() for p in procs)
    if want is None:
        assert failure is None and [r.scores["VMAF_scores"][0] for r in results] == [1.0, 2.0]
    else:
        assert failure is not None and want in failure


def test_read_luma_formats(tmp_path):
    r8, _ = synth.make_clip(64, 48, 2, seed=1)
    r10, _ = synth.make_clip(64, 48, 2, seed=1, bits=10)
    for fmt, planes, bits in (("yuv420p", r8, 8), ("yuv444p", r8, 8), ("yuv422p10le", r10, 10)):
        p = str(tmp_path / f"{fmt}.yuv")
        synth.write_yuv(p, planes, fmt)
        got = cli.read_luma(p, fmt, 64, 48)
        assert np.array_equal(got, synth.to_luma_f32(planes, bits))


def _clip_files(tmp_path, clip, fmt, tag=""):
    kw = dict(ATOMS[clip]["params"])
    bits = kw.pop("bits", 8)
    if "cuts" in kw:
        kw["cuts"] = tuple(kw["cuts"])
    r, d = synth.make_clip(bits=bits, **kw)
    rp, dp = str(tmp_path / f"ref{tag}.yuv"), str(tmp_path / f"dis{tag}.yuv")
    synth.write_yuv(rp, r, fmt)
    synth.write_yuv(dp, d, fmt)
    return rp, dp, r.shape[2], r.shape[1]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "phone", "tenbit"])
def test_run_vmaf_scores_match_reference(name, tmp_path):
    case = GOLD[name]
    rp, dp, w, h = _clip_files(tmp_path, case["clip"], case["fmt"])
    args = [case["fmt"], str(w), str(h), rp, dp, "--out-fmt", "json"] + (["--phone-model"] if name == "phone" else [])
    p = subprocess.run(RUN + args, capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr
    doc = json.loads(p.stdout)
    assert doc["executorId"] == "VMAF_VF0.2.4b-0.6.1"
    assert list(doc["aggregate"].keys()) == ["VMAF_feature_adm2_score", "VMAF_feature_motion2_score"] + \
        [f"VMAF_feature_vif_scale{s}_score" for s in range(4)] + ["VMAF_score", "method"]
    for fr, want in zip(doc["frames"], case["frames"]):
        assert abs(fr["VMAF_score"] - want["vmaf"]) < 1e-3
        for k in ("adm2", "motion2", "vif_scale0", "vif_scale1", "vif_scale2", "vif_scale3"):
            assert abs(fr[f"VMAF_feature_{k}_score"] - want[k]) < 1e-4 + 6e-6 * abs(want[k])
    assert abs(doc["aggregate"]["VMAF_score"] - case["stdout_score"]) < 1e-3


@pytest.mark.gpu
def test_run_vmaf_in_batch_text_output(tmp_path):
    c0, c1 = GOLD["default"], GOLD["harmonic"]
    r0, d0, w0, h0 = _clip_files(tmp_path, c0["clip"], c0["fmt"], "0")
    r1, d1, w1, h1 = _clip_files(tmp_path, c1["clip"], c1["fmt"], "1")
    f = tmp_path / "batch.txt"
    f.write_text(f"# two pairs\n{c0['fmt']} {w0} {h0} {r0} {d0}\n\n{c1['fmt']} {w1} {h1} {r1} {d1}\n")
    outputs = []
    # --parallelize with VMAF_HIP_DEVICES=0,0: two worker processes (spawned, one per listed device) on the one GPU
    for extra, env in (([], {}), (["--parallelize"], {"VMAF_HIP_DEVICES": "0,0"})):
        p = subprocess.run(BATCH + [str(f)] + extra, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr
        outputs.append(p.stdout)
        blocks = p.stdout.split("============================\nAsset ")
        assert len(blocks) == 3 and blocks[1].startswith("0:\n") and blocks[2].startswith("1:\n")
        for blk, case in ((blocks[1], c0), (blocks[2], c1)):
            frames = re.findall(r"^Frame (\d+): (.*)$", blk, flags=re.M)
            assert len(frames) == len(case["frames"])
            for (num, body), want in zip(frames, case["frames"]):
                vals = dict(kv.split(":") for kv in body.split(", "))
                assert abs(float(vals["VMAF_score"]) - want["vmaf"]) < 1e-3
                assert abs(float(vals["VMAF_feature_adm2_score"]) - want["adm2"]) < 1e-4
            agg = re.search(r"^Aggregate \(mean\): .*VMAF_score:([0-9.]+)$", blk, flags=re.M)
            assert agg and abs(float(agg.group(1)) - np.mean([fr["vmaf"] for fr in case["frames"]])) < 1e-3
    assert outputs[0] == outputs[1]                             # worker processes change nothing in the output
    # a pair that cannot be scored (short file) under --parallelize: exit 1 with the asset named, no hang
    bad = tmp_path / "short.yuv"
    bad.write_bytes(b"\0" * 100)
    f.write_text(f"{c0['fmt']} {w0} {h0} {r0} {d0}\n{c1['fmt']} {w1} {h1} {bad} {bad}\n")
    p = subprocess.run(BATCH + [str(f), "--parallelize"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, VMAF_HIP_DEVICES="0,0"))
    assert p.returncode == 1 and "asset 1" in p.stderr, (p.returncode, p.stderr[-500:])


@pytest.mark.gpu
def test_batch_of_720p_pairs_over_three_workers(tmp_path):
    """BASELINE.json configs[3] in small: 18 pairs of 1280x720 (two lengths, 8 and 5 frames, so that the longest-first
    assignment has something to balance) through run_vmaf_in_batch --parallelize with three worker processes on the one
    GPU (VMAF_HIP_DEVICES=0,0,0).  Output in input order (asset_id = line index, run_vmaf_in_batch:87-96), every pair's
    scores equal to the single-process run, and the workers' loads apart by at most one pair's cost."""
    r, d = synth.make_clip(1280, 720, 8, seed=720)
    lines, sizes = [], []
    for i in range(18):
        n = 8 if i % 3 != 1 else 5                                   # 12 long pairs, 6 short ones
        sh = i % 4                                                   # four different distorted clips
        rp, dp = str(tmp_path / f"r{i}.yuv"), str(tmp_path / f"d{i}.yuv")
        synth.write_yuv(rp, r[:n], "yuv420p")
        synth.write_yuv(dp, np.roll(d[:n], sh, axis=2), "yuv420p")
        lines.append(f"yuv420p 1280 720 {rp} {dp}")
        sizes.append(n)
    f = tmp_path / "batch.txt"
    f.write_text("# eighteen pairs\n" + "\n".join(lines) + "\n")
    one = subprocess.run(BATCH + [str(f), "--out-fmt", "json"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    par = subprocess.run(BATCH + [str(f), "--out-fmt", "json", "--parallelize"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, VMAF_HIP_DEVICES="0,0,0", VMAF_HIP_TRACE="1"))
    assert par.returncode == 0, par.stderr
    assert par.stdout == one.stdout                                  # same bytes: order, frame counts, every score
    docs, dec, text = [], json.JSONDecoder(), one.stdout[one.stdout.index("{"):].strip()  # after "Skip commented line: ..."
    while text:                                                      # one indented JSON document per asset, back to back
        doc, end = dec.raw_decode(text)
        docs.append(doc)
        text = text[end:].lstrip()
    assert len(docs) == 18
    for i, (doc, n) in enumerate(zip(docs, sizes)):
        assert len(doc["frames"]) == n
    # pairs with the same distortion and length score the same, different shifts differ
    assert docs[0]["aggregate"]["VMAF_score"] == docs[12]["aggregate"]["VMAF_score"]        # i = 0 and 12: same shift, same length
    assert docs[0]["aggregate"]["VMAF_score"] != docs[2]["aggregate"]["VMAF_score"]
    workers = re.findall(r"worker (\d+) \(device 0\): assets \[([0-9, ]*)\], cost (\d+)", par.stderr)
    assert len(workers) == 3
    seen = sorted(int(x) for _, a, _ in workers for x in a.split(",") if x.strip())
    assert seen == list(range(18))
    loads = [int(c) for _, _, c in workers]
    one_pair = 1280 * 720 * os.path.getsize(str(tmp_path / "r0.yuv"))
    assert max(loads) - min(loads) <= one_pair, loads


@pytest.mark.gpu
def test_run_vmaf_xml_is_what_python2_printed(tmp_path):
    """run_vmaf --out-fmt xml (result.py:178-224 under Python 2, pinned by python/test/result_test.py:148-160):
    attributes in plain string order - the VMAF_* scores, then frameNum / method - and values through str(float)."""
    case = GOLD["default"]
    rp, dp, w, h = _clip_files(tmp_path, case["clip"], case["fmt"])
    p = subprocess.run(RUN + [case["fmt"], str(w), str(h), rp, dp, "--out-fmt", "xml", "--pool", "harmonic_mean"],
                       capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines[0] == '<?xml version="1.0" ?>' and lines[1] == '<result executorId="VMAF_VF0.2.4b-0.6.1">'
    names = ["VMAF_feature_adm2_score", "VMAF_feature_motion2_score"] + [f"VMAF_feature_vif_scale{s}_score" for s in range(4)] + ["VMAF_score"]
    frames = [l for l in lines if l.strip().startswith("<frame ")]
    assert len(frames) == len(case["frames"])
    for i, (l, want) in enumerate(zip(frames, case["frames"])):
        attrs = re.findall(r'(\w+)="([^"]*)"', l)
        assert [a for a, _ in attrs] == names + ["frameNum"] and attrs[-1][1] == str(i)
        vals = dict(attrs)
        assert abs(float(vals["VMAF_score"]) - want["vmaf"]) < 1e-3
        for a, v in attrs[:-1]:
            assert v == cli.py2_str_float(float(v)) and len(v.replace(".", "").replace("-", "").lstrip("0")) <= 12
    agg = [l for l in lines if l.strip().startswith("<aggregate ")]
    assert len(agg) == 1
    attrs = re.findall(r'(\w+)="([^"]*)"', agg[0])
    assert [a for a, _ in attrs] == names + ["method"] and attrs[-1][1] == "harmonic_mean"
