#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE ITSELF, compiled here by oracle/Makefile:

  atoms.json  per-frame 20-atom vectors (full-precision doubles) of seeded synthetic clips, from the
              reference's compute_adm / convolution_f32_c_s + compute_motion / compute_vif
              (oracle/_ref/libvmaf_ref_feature.so), in both of its convolution modes (AVX = what
              compute_vmaf runs on x86, scalar = disable_avx)
  exec.json   end-to-end outputs of the reference's vmafossexec (oracle/_ref/vmafossexec) on raw YUV
              files: stdout aggregate, per-frame XML values; default / phone / no-clip / pooling /
              10-bit / the reference's transform, motion2 and norm_type-none fixture models / two
              image pairs of python/test/resource/test_image_yuv
  svm.json    svm_predict known answers of the reference's libsvm (oracle/_ref/libref_svm.so)

Inputs are not stored: they are re-created bit-identically by vmaf_amd/synth.py (integer arithmetic).
Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from vmaf_amd import synth  # noqa: E402

# name -> make_clip kwargs (+ bits)
CLIPS = {
    "c64x48": dict(width=64, height=48, frames=5, seed=101),
    "c577x325_odd": dict(width=577, height=325, frames=3, seed=102),
    "c576x324_cut": dict(width=576, height=324, frames=4, seed=103, cuts=(2,)),
    "c320x180_10bit": dict(width=320, height=180, frames=3, seed=104, bits=10),
    "c128x80_static": dict(width=128, height=80, frames=3, seed=105, static=True),
    "c96x64_strong": dict(width=96, height=64, frames=3, seed=106, dis_noise=25, quant=24),
}

# name -> (clip, pixel format, model file relative to repo, extra vmafossexec args)
EXEC = {
    "default": ("c576x324_cut", "yuv420p", "model/vmaf_v0.6.1.pkl", []),
    "phone": ("c576x324_cut", "yuv420p", "model/vmaf_v0.6.1.pkl", ["--phone-model"]),
    "noclip": ("c128x80_static", "yuv420p", "model/vmaf_v0.6.1.pkl", ["--disable-clip"]),
    "harmonic": ("c64x48", "yuv444p", "model/vmaf_v0.6.1.pkl", ["--pool", "harmonic_mean"]),
    "min": ("c64x48", "yuv422p", "model/vmaf_v0.6.1.pkl", ["--pool", "min"]),
    "tenbit": ("c320x180_10bit", "yuv422p10le", "model/vmaf_v0.6.1.pkl", []),
    "scalar_path": ("c576x324_cut", "yuv420p", "model/vmaf_v0.6.1.pkl", ["--disable-avx"]),
    "transform_add40": ("c96x64_strong", "yuv420p", "tests/golden/models/test_model_transform_add40.pkl", ["--enable-transform"]),
    "transform_outltein": ("c96x64_strong", "yuv420p", "tests/golden/models/test_model_transform_add40_outltein.pkl", ["--enable-transform"]),
    "motion2_model": ("c64x48", "yuv420p", "tests/golden/models/test_motion2.pkl", []),
    "norm_none": ("c64x48", "yuv420p", "tests/golden/models/nflxtrain_norm_type_none.pkl", []),
    "vmafv4": ("c64x48", "yuv420p", "tests/golden/models/nflxall_vmafv4.pkl", []),
    # optional per-frame metrics (combo.c:168-230): what the Python VmafossExecQualityRunner always asks for
    "metrics": ("c576x324_cut", "yuv420p", "model/vmaf_v0.6.1.pkl", ["--psnr", "--ssim", "--ms-ssim"]),
    "metrics_tenbit": ("c320x180_10bit", "yuv420p10le", "model/vmaf_v0.6.1.pkl", ["--psnr", "--ssim", "--ms-ssim"]),
    "psnr_only": ("c64x48", "yuv420p", "model/vmaf_v0.6.1.pkl", ["--psnr"]),
}
# name -> (app, clip, pixel format) for the per-feature tool
FEATURE_TOOL = {
    "all": ("all", "c576x324_cut", "yuv420p"),
    "all_tenbit": ("all", "c320x180_10bit", "yuv444p10le"),
    "adm": ("adm", "c64x48", "yuv420p"),
    "vif": ("vif", "c64x48", "yuv422p"),
    "motion": ("motion", "c576x324_cut", "yuv420p"),
    "ansnr": ("ansnr", "c96x64_strong", "yuv420p"),
}
IMAGES = ["100007", "100039"]      # python/test/resource/test_image_yuv/<n>.yuv vs <n>_noisy.yuv, 481x321 yuv444p


def clip_arrays(name):
    kw = dict(CLIPS[name])
    bits = kw.pop("bits", 8)
    r, d = synth.make_clip(bits=bits, **kw)
    return r, d, bits


def run_exec(fmt, w, h, ref_path, dis_path, model, extra):
    with tempfile.TemporaryDirectory() as td:
        log = os.path.join(td, "log.xml")
        out = subprocess.run([oracle.ref_exec_path(), fmt, str(w), str(h), ref_path, dis_path, os.path.join(ROOT, model),
                              "--log", log] + extra, capture_output=True, text=True, check=True).stdout
        m = re.search(r"VMAF score(?: \((\w+)\))? = ([-0-9.eE+naif]+)", out)
        xml = open(log).read()
    frames = []
    for fm in re.finditer(r"<frame ([^/]*)/>", xml):
        frames.append({k: float(v) for k, v in re.findall(r'(\w+)="([^"]*)"', fm.group(1))})
    agg = re.search(r'aggregateVMAF="([^"]*)"', xml).group(1)
    return {"stdout_score": float(m.group(2)), "pool": m.group(1), "aggregate_xml": float(agg), "frames": frames}


LOG_CASES = ["default", "phone", "harmonic", "tenbit", "metrics", "noclip", "vmafv4"]


def exec_logs():
    """exec_logs.json: the TEXT the reference's vmafossexec wrote for a few EXEC cases - stdout, the XML log
    (wrapper/src/vmaf.cpp:719-750, pugixml number style) and the JSON log (vmaf.cpp:679-717, ptools JSONPrint number
    style, values rounded to 5 digits) - so that the formatting itself is pinned, not only the values."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name in LOG_CASES:
            clip, fmt, model, extra = EXEC[name]
            r, d, bits = clip_arrays(clip)
            rp, dp = os.path.join(td, "ref.yuv"), os.path.join(td, "dis.yuv")
            synth.write_yuv(rp, r, fmt)
            synth.write_yuv(dp, d, fmt)
            F, h, w = r.shape
            res = {}
            for log_fmt in ("xml", "json"):
                log = os.path.join(td, "log." + log_fmt)
                p = subprocess.run([oracle.ref_exec_path(), fmt, str(w), str(h), rp, dp, os.path.join(ROOT, model), "--log", log,
                                    "--log-fmt", log_fmt] + extra, capture_output=True, text=True, check=True)
                res[log_fmt] = open(log).read()
                res["stdout"] = p.stdout
            out[name] = res
    json.dump(out, open(os.path.join(HERE, "exec_logs.json"), "w"), indent=0)
    print("wrote exec_logs.json")


def main():
    assert oracle.ref_lib() is not None and oracle.ref_exec_path(), "run `make -C oracle ref ref_exec` first"
    if "--logs-only" in sys.argv:
        exec_logs()
        return
    atoms = {}
    for name in CLIPS:
        r, d, bits = clip_arrays(name)
        rf, df = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
        atoms[name] = {"params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in CLIPS[name].items()},
                       "avx": oracle.ref_clip_atoms(rf, df, avx=1).tolist(),
                       "scalar": oracle.ref_clip_atoms(rf, df, avx=0).tolist()}
    flat_r = np.zeros((2, 72, 128), np.float32)
    flat_d = np.full((2, 72, 128), 10.0, np.float32)
    atoms["flat_0_vs_10_128x72"] = {"params": "ref Y=0, dis Y=10, 2 frames",
                                    "avx": oracle.ref_clip_atoms(flat_r, flat_d, avx=1).tolist(),
                                    "scalar": oracle.ref_clip_atoms(flat_r, flat_d, avx=0).tolist()}
    json.dump(atoms, open(os.path.join(HERE, "atoms.json"), "w"), indent=0)

    ex = {}
    with tempfile.TemporaryDirectory() as td:
        for name, (clip, fmt, model, extra) in EXEC.items():
            r, d, bits = clip_arrays(clip)
            rp, dp = os.path.join(td, "ref.yuv"), os.path.join(td, "dis.yuv")
            synth.write_yuv(rp, r, fmt)
            synth.write_yuv(dp, d, fmt)
            F, h, w = r.shape
            ex[name] = dict(clip=clip, fmt=fmt, model=model, args=extra, **run_exec(fmt, w, h, rp, dp, model, extra))
    img_dir = "/root/reference/python/test/resource/test_image_yuv"
    for n in IMAGES:
        res = run_exec("yuv444p", 481, 321, f"{img_dir}/{n}.yuv", f"{img_dir}/{n}_noisy.yuv", "model/vmaf_v0.6.1.pkl", [])
        ex[f"image_{n}"] = dict(clip=f"images/{n}", fmt="yuv444p", model="model/vmaf_v0.6.1.pkl", args=[], **res)
    json.dump(ex, open(os.path.join(HERE, "exec.json"), "w"), indent=0)

    # the reference's per-feature tool feature/vmaf (oracle/_ref/vmaf, `make -C oracle ref_feature_tool`): stdout as is
    tool = os.path.join(ROOT, "oracle", "_ref", "vmaf")
    ft = {}
    with tempfile.TemporaryDirectory() as td:
        for name, (app, clip, fmt) in FEATURE_TOOL.items():
            r, d, bits = clip_arrays(clip)
            rp, dp = os.path.join(td, "ref.yuv"), os.path.join(td, "dis.yuv")
            synth.write_yuv(rp, r, fmt)
            synth.write_yuv(dp, d, fmt)
            F, h, w = r.shape
            out = subprocess.run([tool, app, fmt, rp, dp, str(w), str(h)], capture_output=True, text=True, check=True).stdout
            ft[name] = dict(app=app, clip=clip, fmt=fmt, stdout=out)
    json.dump(ft, open(os.path.join(HERE, "feature_tool.json"), "w"), indent=0)

    rng = np.random.default_rng(7)
    sv = {}
    for model in ["model/vmaf_v0.6.1.pkl.model", "tests/golden/models/test_motion2.pkl.model"]:
        dim = oracle.load_svm_model(os.path.join(ROOT, model))["sv"].shape[1]
        x = rng.uniform(-0.3, 1.3, (16, dim))
        x[0] = 0.0
        sv[model] = {"x": x.tolist(), "y": oracle.ref_svm_predict(os.path.join(ROOT, model), x).tolist()}
    json.dump(sv, open(os.path.join(HERE, "svm.json"), "w"), indent=0)
    print("wrote atoms.json exec.json svm.json")
    exec_logs()


if __name__ == "__main__":
    main()
