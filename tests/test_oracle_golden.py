"""CPU tests: the oracle (oracle/) against the committed golden fixtures (tests/golden/*.json, generated
from the compiled reference by tests/golden/make_golden.py) and against the numbers the reference's own
tests pin for inputs that can be re-synthesised (flat frames, python/test/feature_test.py:659-719,
python/test/quality_runner_test.py:391-432).  Runs without a GPU and without /root/reference."""
import json
import os

import numpy as np
import pytest

import oracle
from oracle.fusion import pool
from vmaf_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")

ATOMS = json.load(open(os.path.join(GOLD, "atoms.json")))
EXEC = json.load(open(os.path.join(GOLD, "exec.json")))
SVM = json.load(open(os.path.join(GOLD, "svm.json")))

ADM_COLS = list(range(0, 10))
OTHER_COLS = list(range(10, 20))


def clip_from_params(name):
    if name.startswith("flat"):
        return np.zeros((2, 72, 128), np.float32), np.full((2, 72, 128), 10.0, np.float32), 8
    kw = dict(ATOMS[name]["params"])
    bits = kw.pop("bits", 8)
    if "cuts" in kw:
        kw["cuts"] = tuple(kw["cuts"])
    r, d = synth.make_clip(bits=bits, **kw)
    return synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits), bits


@pytest.mark.parametrize("name", sorted(ATOMS))
@pytest.mark.parametrize("mode", ["avx", "scalar"])
def test_oracle_atoms_match_reference_goldens(name, mode):
    ref, dis, _ = clip_from_params(name)
    got = oracle.clip_atoms(ref, dis, oracle.ORDER_AVX if mode == "avx" else oracle.ORDER_SCALAR)
    want = np.array(ATOMS[name][mode])
    # VIF and motion: the oracle restates the reference bit for bit (both convolution orders)
    assert np.array_equal(got[:, OTHER_COLS], want[:, OTHER_COLS])
    # ADM: the oracle divides exactly where the reference uses rcpss + one Newton step (adm_tools.c:34-38)
    assert np.allclose(got[:, ADM_COLS], want[:, ADM_COLS], rtol=3e-7, atol=0)


def test_reference_avx_and_scalar_goldens_differ_only_slightly():
    """The reference's own two code paths agree to ~1e-6 relative (quality_runner_test.py:1279-1376)."""
    for name in ATOMS:
        a, s = np.array(ATOMS[name]["avx"]), np.array(ATOMS[name]["scalar"])
        assert np.allclose(a, s, rtol=2e-5, atol=1e-6), name


@pytest.mark.parametrize("model_file", sorted(SVM))
def test_svm_oracle_matches_reference_libsvm(model_file):
    svm = oracle.load_svm_model(os.path.join(ROOT, model_file))
    got = oracle.svm_predict(svm, np.array(SVM[model_file]["x"]))
    assert np.array_equal(got, np.array(SVM[model_file]["y"]))


def _sig6(a, b):
    """XML log values carry 6 significant digits (pugixml %g)."""
    return abs(a - b) <= 6e-6 * max(1.0, abs(b)) + 1e-12


@pytest.mark.parametrize("name", sorted(EXEC))
def test_fusion_oracle_matches_reference_vmafossexec(name):
    case = EXEC[name]
    if case["clip"].startswith("images/"):
        n = case["clip"].split("/")[1]
        ref = np.fromfile(os.path.join(GOLD, "images", f"{n}.yuv"), np.uint8)[:481 * 321].reshape(1, 321, 481).astype(np.float32)
        dis = np.fromfile(os.path.join(GOLD, "images", f"{n}_noisy.yuv"), np.uint8)[:481 * 321].reshape(1, 321, 481).astype(np.float32)
    else:
        ref, dis, _ = clip_from_params(case["clip"])
    order = oracle.ORDER_SCALAR if "--disable-avx" in case["args"] else oracle.ORDER_AVX
    atoms = oracle.clip_atoms(ref, dis, order)
    pkl = oracle.load_pkl_model(os.path.join(ROOT, case["model"]))
    svm = oracle.load_svm_model(os.path.join(ROOT, case["model"] + ".model"))
    transform = "--phone-model" in case["args"] or "--enable-transform" in case["args"]
    vm, feats = oracle.vmaf_from_atoms(atoms, pkl, svm, enable_transform=transform, disable_clip="--disable-clip" in case["args"])
    assert len(vm) == len(case["frames"])
    bits = 8 if case["clip"].startswith("images/") else ATOMS[case["clip"]]["params"].get("bits", 8)
    metric = {"psnr": lambda a, b: oracle.psnr(a, b, bits), "ssim": lambda a, b: oracle.ssim(a, b)[0],
              "ms_ssim": lambda a, b: oracle.ms_ssim(a, b)[0]}      # --psnr / --ssim / --ms-ssim columns (combo.c:168-230)
    for i, fr in enumerate(case["frames"]):
        assert _sig6(vm[i], fr["vmaf"]), (i, vm[i], fr["vmaf"])
        for k, v in fr.items():
            if k in metric:
                assert _sig6(metric[k](ref[i], dis[i]), v), (k, i, v)
            elif k not in ("frameNum", "vmaf"):
                assert _sig6(feats[k][i], v), (k, i, feats[k][i], v)
    # "%f" on stdout; 2e-5 covers the exact-division vs rcpss+Newton difference in adm (<= 3e-7 relative)
    assert abs(pool(vm, case["pool"]) - case["stdout_score"]) < 2e-5


def test_flat_frames_pinned_by_the_reference_tests():
    """feature_test.py:702-719 (Y=0 vs Y=10) and :671-688 (identical), 1920x1080; quality_runner_test.py:431-432."""
    z = np.zeros((1, 1080, 1920), np.float32)
    t = np.full((1, 1080, 1920), 10.0, np.float32)
    a10 = oracle.clip_atoms(z, t, oracle.ORDER_AVX)[0]
    a00 = oracle.clip_atoms(z, z, oracle.ORDER_AVX)[0]
    names = oracle.ATOM_NAMES
    g = lambda a, n: a[names.index(n)]  # noqa: E731
    assert abs(g(a10, "vif_num_scale0") - 2073600.0) < 5e-5 and abs(g(a10, "vif_den_scale0") - 2073600.0) < 5e-5
    assert abs(g(a10, "vif_num_scale3") - 32400.0) < 5e-5 and abs(g(a10, "vif_den_scale3") - 32400.0) < 5e-5
    assert abs(g(a10, "adm_num") - 149.780313) < 5e-5
    assert abs(g(a10, "adm_num_scale0") - 65.573967) < 5e-4 and abs(g(a10, "adm_num_scale3") - 16.667711) < 5e-4
    assert g(a10, "motion") == 0.0
    # identical flat frames: the reference's pinned 2073600.125 is fp32 rounding noise of ITS build (1 ulp of the
    # fp32 plane sum is 0.125); the reference compiled here gives 2073600.0, as does the oracle
    assert abs(g(a00, "vif_num_scale0") - 2073600.125) <= 0.125 and abs(g(a00, "vif_num_scale3") - 32400.0) < 5e-5
    assert abs(g(a00, "adm_num_scale3") - 16.667747) < 5e-4
    # on a constant image the ADM detail bands are pure fp32 rounding noise, so adm_den itself is compiler
    # dependent; the reference pins 149.780392 at places=4, the ratio adm2 is what the model sees
    assert abs(g(a10, "adm_den") - 149.780392) < 5e-4
    pkl = oracle.load_pkl_model(os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl"))
    svm = oracle.load_svm_model(os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl.model"))
    v10, f10 = oracle.vmaf_from_atoms(a10[None], pkl, svm)
    v00, _ = oracle.vmaf_from_atoms(a00[None], pkl, svm)
    assert abs(f10["adm2"][0] - 1.0) < 5e-5 and abs(f10["vif_scale0"][0] - 1.0) < 5e-5
    assert abs(v10[0] - 97.427927701008869) < 5e-4
    assert abs(v00[0] - 97.428042675471147) < 5e-4


def test_pkl_reader_parses_the_model_dict():
    pkl = oracle.load_pkl_model(os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl"))
    md = pkl["model_dict"]
    assert md["model_type"] == "LIBSVMNUSVR" and md["norm_type"] == "linear_rescale"
    assert md["feature_names"] == ["VMAF_feature_adm2_score", "VMAF_feature_motion2_score"] + \
        [f"VMAF_feature_vif_scale{s}_score" for s in range(4)]
    assert len(md["slopes"]) == 7 and len(md["intercepts"]) == 7 and md["score_clip"] == [0.0, 100.0]
    assert md["score_transform"]["out_gte_in"] == "true" and md["score_transform"]["p2"] == -0.00705305
