"""GPU parity tests proper: the HIP path, called through the C-ABI (vmaf_amd/libvmaf_amd.so), against
the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): per-frame features within 1e-4 absolute of the reference C
code, aggregate VMAF within 1e-3.  The kernels keep the reference's tap orders, so the raw atoms are
additionally held to 2e-6 relative (only the final VIF / ADM plane sums are ordered differently) and the
`motion` atom, whose sum the kernel orders as motion.c does, to equality.
"""
import os

import numpy as np
import pytest

import oracle
import vmaf_amd
from vmaf_amd import synth

pytestmark = pytest.mark.gpu

FEATURE_TOL = 1e-4
ATOM_RTOL = 2e-6


def _feats(atoms):
    return oracle.derived_features(atoms)


def _check(got, want, what=""):
    # raw atoms: relative (values span 1e0 .. 1e6); motion absolute
    for k, name in enumerate(vmaf_amd.ATOM_NAMES):
        g, w = got[:, k], want[:, k]
        tol = ATOM_RTOL * np.maximum(1.0, np.abs(w))
        assert np.all(np.abs(g - w) <= tol), f"{what} atom {name}: got {g} want {w}"
    # motion is the one plane sum that reaches the model directly: the kernel adds it in the reference's own order
    # (motion.c:44-60: fp32 per row, left to right, then the rows top to bottom), so the atom is the reference's bit for bit
    assert np.array_equal(got[:, 10], want[:, 10]), f"{what} motion: got {got[:, 10]!r} want {want[:, 10]!r}"
    fg, fw = _feats(got), _feats(want)
    for k in ("adm2", "motion", "motion2", "vif_scale0", "vif_scale1", "vif_scale2", "vif_scale3",
              "adm_scale0", "adm_scale1", "adm_scale2", "adm_scale3", "vif"):
        assert np.all(np.abs(fg[k] - fw[k]) <= FEATURE_TOL), f"{what} feature {k}: {fg[k]} vs {fw[k]}"


def _clip(w, h, frames, bits=8, **kw):
    r, d = synth.make_clip(w, h, frames, bits=bits, **kw)
    return synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)


ORDERS = {"avx": oracle.ORDER_AVX, "scalar": oracle.ORDER_SCALAR}


@pytest.mark.parametrize("order", ["avx", "scalar"])
@pytest.mark.parametrize("w,h,frames", [(64, 48, 5), (577, 325, 3), (576, 324, 3), (200, 32, 2), (33, 130, 2), (100, 75, 2),
                                        (1280, 720, 2), (1920, 1080, 2)])
def test_atoms_match_oracle(w, h, frames, order):
    """Both summation orders of the reference (AVX kernels = its default on x86, scalar loops = disable_avx)."""
    ref, dis = _clip(w, h, frames, seed=w + h)
    want = oracle.clip_atoms(ref, dis, ORDERS[order])
    with vmaf_amd.Engine(w, h, max_batch=4, order=order) as e:
        got = e.extract(ref, dis)
    _check(got, want, f"{w}x{h} {order}")


# A fixed slice of tools/fuzz_parity.py's sweep: sizes that straddle the tile pitches of the kernels (VIF 48x32 tiles in
# 8-row waves, ADM 61x14 interior coefficients = 122x28 pixels, motion 60x32) and the 8-column blocks of the AVX order,
# both orders, 8 and 10 bit, through the raw-luma ingest.
FUZZ_CASES = [(47, 33, 8, "avx", 1), (48, 32, 8, "scalar", 2), (49, 65, 10, "avx", 2), (60, 61, 8, "avx", 3), (61, 60, 8, "scalar", 1),
              (95, 97, 8, "avx", 2), (96, 96, 10, "scalar", 2), (97, 95, 8, "avx", 1), (121, 57, 8, "avx", 2), (122, 56, 8, "scalar", 2),
              (123, 57, 10, "avx", 1), (244, 56, 8, "avx", 2), (245, 113, 8, "scalar", 1), (32, 32, 8, "avx", 3), (257, 131, 10, "avx", 2),
              (419, 299, 8, "scalar", 2),
              # wide tiles whose last tile of a row is narrow: 8, 16 and 248 columns, and 2 tiles + 1 strip
              (248, 40, 8, "avx", 2), (256, 33, 10, "avx", 2), (488, 57, 8, "avx", 1), (488, 57, 8, "scalar", 1), (304, 33, 8, "avx", 2)]


@pytest.mark.parametrize("w,h,bits,order,frames", FUZZ_CASES)
def test_sizes_around_the_tile_pitches(w, h, bits, order, frames):
    r, d = synth.make_clip(w, h, frames, bits=bits, seed=1000 * w + h)
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    want = oracle.clip_atoms(ref, dis, ORDERS[order])
    with vmaf_amd.Engine(w, h, max_batch=2, order=order) as e:
        got = e.extract_raw(r, d)
    assert np.all(np.isfinite(got))
    _check(got, want, f"{w}x{h} {bits}-bit {order}")


@pytest.mark.parametrize("order", ["avx", "scalar"])
@pytest.mark.parametrize("w,h", [(240, 135), (480, 270), (720, 68), (240, 33), (960, 540)])
def test_wide_tile_kernel_on_noisy_pairs(w, h, order):
    """Planes whose width is a multiple of 240 take the workgroup-wide VIF tiles (vif_stat_wg_kernel: four waves side by
    side, edge strips redone per pixel).  Strong noise makes VIF's sigma^2 = E[x^2] - mu^2 cancellation amplify any
    sample that took the wrong summation order (the reference's two orders are ~5e-5 apart in the features on such
    content), so the 2e-6 bound on the raw sums pins the order of every region: top / bottom rows, the first and last
    8 columns, heights that are not a multiple of the 8-row tiles, one tile per row (w = 240: both edge strips in it)."""
    ref, dis = _clip(w, h, 2, seed=w * 7 + h, dis_noise=25, quant=24)
    want = oracle.clip_atoms(ref, dis, ORDERS[order])
    with vmaf_amd.Engine(w, h, max_batch=2, order=order) as e:
        got = e.extract(ref, dis)
        _check(got, want, f"{w}x{h} {order}")
        for kind, plane in (("vif_ref", ref[1]), ("vif_dis", dis[1])):     # next-scale planes from the wide tiles: bit for bit
            for lvl, want_plane in enumerate(oracle.vif_pyramid(plane + np.float32(-128), ORDERS[order]), start=1):
                assert np.array_equal(e.debug_plane(kind, lvl, 1), want_plane), (kind, lvl)


EXACT_RTOL = 2e-12


@pytest.mark.parametrize("order", ["avx", "scalar"])
@pytest.mark.parametrize("w,h", [(240, 321), (480, 321), (481, 321), (96, 64), (250, 37), (248, 321), (472, 321)])
def test_every_vif_pixel_is_pinned_through_the_fp64_sums(golden_dir, order, w, h):
    """The VIF kernels add their per-pixel fp32 num / den values in fp64; the oracle can add the same values in extended
    precision (oracle.vif_exact_sums).  If every pixel's value is bit-identical, the two sums agree to ~1e-14; a single
    region in the wrong summation order (a few hundred pixels, each off by one fp32 rounding) moves them by >= 1e-11.
    Crops of tests/golden/images/100007 (the pair on which the reference's own two orders end up 3.7e-3 VMAF apart):
    widths 240 / 480 run the workgroup-wide tiles (edge strips, top / bottom rows), 248 / 472 the same with a narrow last
    tile, 481 / 96 / 250 the wave-private tiles
    with their border frame.  The opposite order must miss the bound, or the test would prove nothing."""
    ref = np.fromfile(os.path.join(golden_dir, "images", "100007.yuv"), np.uint8)[:481 * 321].reshape(321, 481)[:h, :w].astype(np.float32)
    dis = np.fromfile(os.path.join(golden_dir, "images", "100007_noisy.yuv"), np.uint8)[:481 * 321].reshape(321, 481)[:h, :w].astype(np.float32)
    ref, dis = np.ascontiguousarray(ref), np.ascontiguousarray(dis)
    off = np.float32(-128)
    want = oracle.vif_exact_sums(ref + off, dis + off, ORDERS[order])
    other = oracle.vif_exact_sums(ref + off, dis + off, ORDERS["scalar" if order == "avx" else "avx"])
    with vmaf_amd.Engine(w, h, order=order) as e:
        e.extract(ref[None], dis[None])
        got = e.debug_sums(0)[:8]
    rel = np.abs(got - want) / np.abs(want)
    assert np.all(rel < EXACT_RTOL), (rel, got, want)
    assert np.max(np.abs(got - other) / np.abs(other)) > 50 * EXACT_RTOL


@pytest.mark.parametrize("order", ["avx", "scalar"])
@pytest.mark.parametrize("w,h,bits", [(1920, 1080, 8), (960, 540, 10), (1280, 720, 8), (720, 203, 8)])
def test_every_vif_pixel_is_pinned_at_the_benchmark_sizes(w, h, bits, order):
    """The same pin (fp64 device sums against the oracle's exact sums of its per-pixel values, 2e-12) at BASELINE.json's
    frame sizes: 1920x1080 and its pyramid run the workgroup-wide tiles, 1280x720 the wave-private ones."""
    r, d = synth.make_clip(w, h, 1, bits=bits, seed=w + 3 * h, dis_noise=12)
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    off = np.float32(-128)
    want = oracle.vif_exact_sums(ref[0] + off, dis[0] + off, ORDERS[order])
    other = oracle.vif_exact_sums(ref[0] + off, dis[0] + off, ORDERS["scalar" if order == "avx" else "avx"])
    with vmaf_amd.Engine(w, h, order=order) as e:
        e.extract(ref, dis)
        got = e.debug_sums(0)[:8]
    assert np.all(np.abs(got - want) / np.abs(want) < EXACT_RTOL), (got, want)
    assert np.max(np.abs(got - other) / np.abs(other)) > 50 * EXACT_RTOL


def test_noisy_image_where_the_two_orders_diverge(golden_dir):
    """tests/golden/images/100007: the reference's own AVX and scalar paths differ by 3.7e-3 VMAF on this pair
    (65.415667 vs 65.412008); the engine has to land on the right one in each mode."""
    ref = np.fromfile(os.path.join(golden_dir, "images", "100007.yuv"), np.uint8)[:481 * 321].reshape(1, 321, 481).astype(np.float32)
    dis = np.fromfile(os.path.join(golden_dir, "images", "100007_noisy.yuv"), np.uint8)[:481 * 321].reshape(1, 321, 481).astype(np.float32)
    got = {}
    for order in ("avx", "scalar"):
        with vmaf_amd.Engine(481, 321, order=order) as e:
            a = e.extract(ref, dis)
        _check(a, oracle.clip_atoms(ref, dis, ORDERS[order]), order)
        got[order] = vmaf_amd.predict_from_atoms(a)[0][0]
    assert abs(got["avx"] - 65.415667) < 1e-4 and abs(got["scalar"] - 65.412008) < 1e-4


@pytest.mark.parametrize("w,h,frames,kw", [(64, 48, 5, {}), (577, 325, 3, {}), (576, 324, 4, dict(cuts=(2,))), (96, 64, 3, dict(dis_noise=25)),
                                           (1920, 1080, 2, {}), (320, 180, 3, dict(bits=10))])
def test_fused_order_stays_inside_the_tolerances(w, h, frames, kw, model_path):
    """Conv order 2 ("fma", opt-in fast mode; not one of the reference's orders): fused multiply-adds in the VIF / motion /
    db2 taps.  Not bit-identical, so it is held to the north-star tolerances against the oracle in the reference's default
    (AVX) order: features 1e-4 absolute, per-frame and mean VMAF 1e-3; the raw sums to 1e-4 relative."""
    bits = kw.pop("bits", 8)
    ref, dis = _clip(w, h, frames, bits=bits, seed=w + h, **kw)
    want = oracle.clip_atoms(ref, dis, oracle.ORDER_AVX)
    with vmaf_amd.Engine(w, h, max_batch=4, order="fma") as e:
        got = e.extract(ref, dis)
    assert not np.array_equal(got, want) or w * h < 64 * 64            # it really is a different rounding
    for k, name in enumerate(vmaf_amd.ATOM_NAMES):
        assert np.all(np.abs(got[:, k] - want[:, k]) <= 1e-4 * np.maximum(1.0, np.abs(want[:, k]))), name
    fg, fw = _feats(got), _feats(want)
    for k in fw:
        assert np.all(np.abs(fg[k] - fw[k]) <= FEATURE_TOL), f"feature {k}: {fg[k]} vs {fw[k]}"
    vg, vw = vmaf_amd.predict_from_atoms(got, model_path)[0], vmaf_amd.predict_from_atoms(want, model_path)[0]
    assert np.all(np.abs(vg - vw) < 1e-3) and abs(vg.mean() - vw.mean()) < 1e-3


def test_fused_order_on_the_noisy_image(golden_dir):
    """On tests/golden/images/100007 the reference's own two orders are 3.7e-3 VMAF apart; the fused order has to stay
    within 1e-3 of the default (AVX) one - it lands at ~1e-4."""
    ref = np.fromfile(os.path.join(golden_dir, "images", "100007.yuv"), np.uint8)[:481 * 321].reshape(1, 321, 481).astype(np.float32)
    dis = np.fromfile(os.path.join(golden_dir, "images", "100007_noisy.yuv"), np.uint8)[:481 * 321].reshape(1, 321, 481).astype(np.float32)
    with vmaf_amd.Engine(481, 321, order="fma") as e:
        a = e.extract(ref, dis)
    assert abs(vmaf_amd.predict_from_atoms(a)[0][0] - 65.415667) < 1e-3


def test_ten_bit_input():
    ref, dis = _clip(320, 180, 3, bits=10, seed=5)
    assert np.any(ref != np.round(ref))            # quarter steps really present
    want = oracle.clip_atoms(ref, dis, oracle.ORDER_AVX)
    with vmaf_amd.Engine(320, 180, max_batch=8) as e:
        got = e.extract(ref, dis)
    _check(got, want, "10-bit")


def test_batch_size_does_not_change_results():
    ref, dis = _clip(160, 96, 9, seed=3, cuts=(4,))
    with vmaf_amd.Engine(160, 96, max_batch=9) as e:
        whole = e.extract(ref, dis)
    with vmaf_amd.Engine(160, 96, max_batch=2) as e:
        pieces = e.extract(ref, dis)
    assert np.array_equal(whole, pieces)
    _check(whole, oracle.clip_atoms(ref, dis, oracle.ORDER_AVX), "cuts")


def test_motion_halo_matches_unsharded_run():
    """Frame-range sharding: a shard seeded with the frame before it reproduces the
    unsharded motion values bit for bit, including across a scene cut at the shard edge."""
    ref, dis = _clip(192, 112, 8, seed=11, cuts=(4,))
    with vmaf_amd.Engine(192, 112, max_batch=8) as e:
        whole = e.extract(ref, dis)
    for start in (3, 4, 5):
        with vmaf_amd.Engine(192, 112, max_batch=8) as e:
            e.motion_seed(ref[start - 1])
            shard = e.extract(ref[start:], dis[start:], continue_sequence=True)
        assert np.array_equal(shard, whole[start:]), f"shard from {start}"


def test_static_and_identical_clip():
    ref, _ = _clip(128, 80, 3, seed=2, static=True)
    want = oracle.clip_atoms(ref, ref, oracle.ORDER_AVX)
    with vmaf_amd.Engine(128, 80) as e:
        got = e.extract(ref, ref)
    _check(got, want, "identical")
    assert np.all(got[:, 10] == 0.0)               # static clip: motion exactly 0
    f = _feats(got)
    assert np.all(np.abs(f["adm2"] - 1.0) < 1e-6)


def test_flat_frames():
    """Reference corner case python/test/feature_test.py:659-719 (flat Y=0 vs Y=10), at reduced size."""
    ref = np.zeros((2, 72, 128), np.float32)
    dis = np.full((2, 72, 128), 10.0, np.float32)
    want = oracle.clip_atoms(ref, dis, oracle.ORDER_AVX)
    with vmaf_amd.Engine(128, 72) as e:
        got = e.extract(ref, dis)
    fg, fw = _feats(got), _feats(want)
    for k in ("adm2", "vif_scale0", "vif_scale1", "vif_scale2", "vif_scale3", "motion2"):
        assert np.all(np.abs(fg[k] - fw[k]) <= FEATURE_TOL), k
    assert np.allclose(got[:, 11], 128 * 72)       # vif_num_scale0 == pixel count (feature_test.py:702)


def test_device_pointer_entry_matches_host_entry():
    torch = pytest.importorskip("torch")
    ref, dis = _clip(256, 144, 4, seed=9)
    with vmaf_amd.Engine(256, 144, max_batch=4) as e:
        host = e.extract(ref, dis)
    tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dis).cuda()
    torch.cuda.synchronize()
    with vmaf_amd.Engine(256, 144, max_batch=4) as e:
        dev = e.extract_device(tr, td, 4)
    assert np.array_equal(host, dev)


def test_svm_predict_matches_oracle(model_path):
    svm = oracle.load_svm_model(model_path + ".model")
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.2, 1.2, (37, 6))
    want = oracle.svm_predict(svm, x)
    got = vmaf_amd.svm_predict(svm["sv"], svm["coef"], svm["gamma"], svm["rho"], x)
    assert np.all(np.abs(got - want) < 1e-10)


@pytest.mark.parametrize("kw", [dict(), dict(phone_model=True), dict(disable_clip=True), dict(pool_method="harmonic_mean"),
                                dict(pool_method="min")])
def test_compute_vmaf_end_to_end(model_path, tmp_path, kw):
    ref, dis = _clip(320, 180, 6, seed=21, cuts=(3,))
    atoms = oracle.clip_atoms(ref, dis, oracle.ORDER_AVX)
    pkl, svm = oracle.load_pkl_model(model_path), oracle.load_svm_model(model_path + ".model")
    per_frame, feats = oracle.vmaf_from_atoms(atoms, pkl, svm, enable_transform=kw.get("phone_model", False),
                                              disable_clip=kw.get("disable_clip", False))
    from oracle.fusion import pool
    want = pool(per_frame, kw.get("pool_method"))
    log = str(tmp_path / "log.json")
    got = vmaf_amd.compute_vmaf_from_arrays(ref, dis, model_path=model_path, log_path=log, log_fmt="json", **kw)
    assert abs(got - want) < 1e-3
    import json
    doc = json.load(open(log))
    assert doc["metrics"] == ["adm2", "motion2", "vif_scale0", "vif_scale1", "vif_scale2", "vif_scale3", "vmaf"]
    assert len(doc["frames"]) == 6
    for i, fr in enumerate(doc["frames"]):
        assert fr["frameNum"] == i
        assert abs(fr["metrics"]["vmaf"] - per_frame[i]) < 1e-3
        for k in ("adm2", "motion2", "vif_scale0", "vif_scale1", "vif_scale2", "vif_scale3"):
            assert abs(fr["metrics"][k] - feats[k][i]) < FEATURE_TOL + 1e-5   # log rounds to 5 decimals


def test_error_paths(model_path):
    with pytest.raises(vmaf_amd.VmafHipError):
        vmaf_amd.Engine(8, 8)
    ref, dis = _clip(64, 48, 1)
    with pytest.raises(vmaf_amd.VmafHipError, match="model"):
        vmaf_amd.compute_vmaf_from_arrays(ref, dis, model_path="/nonexistent.pkl")
    with pytest.raises(vmaf_amd.VmafHipError, match="scale below 1x1"):       # ms_ssim.c:172-177 on a 64x48 frame
        vmaf_amd.compute_vmaf_from_arrays(ref, dis, model_path=model_path, do_ms_ssim=True)
    with vmaf_amd.Engine(64, 48) as e:
        with pytest.raises(vmaf_amd.VmafHipError, match="previous frame"):
            e.extract(ref, dis, continue_sequence=True)


@pytest.mark.parametrize("bits,w,h", [(8, 160, 96), (8, 101, 75), (10, 192, 108), (10, 67, 50)])
def test_raw_luma_ingest_is_bit_identical_to_fp32_ingest(bits, w, h):
    """u8 / u16 luma uploaded as the file holds it and widened on the GPU (file_io.c:86-172 on the
    device) gives the very same atoms as the fp32 planes a read_frame callback would have produced - vector path
    (w % 4 == 0) and scalar path, several batches, and the raw motion halo."""
    r, d = synth.make_clip(w, h, 5, bits=bits, seed=21 + w, cuts=(3,))
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    assert r.dtype == (np.uint8 if bits == 8 else np.uint16)
    with vmaf_amd.Engine(w, h, max_batch=2) as e:
        want = e.extract(ref, dis)
    with vmaf_amd.Engine(w, h, max_batch=2) as e:
        got = e.extract_raw(r, d)
    assert np.array_equal(got, want)
    with vmaf_amd.Engine(w, h, max_batch=4) as e:
        e.motion_seed(r[1])                         # raw halo frame
        shard = e.extract_raw(r[2:], d[2:], continue_sequence=True)
    assert np.array_equal(shard, want[2:])
    _check(got, oracle.clip_atoms(ref, dis, oracle.ORDER_AVX), f"raw {bits}-bit {w}x{h}")


def test_raw_luma_device_pointers_and_errors():
    torch = pytest.importorskip("torch")
    r, d = synth.make_clip(128, 72, 3, seed=4)
    with vmaf_amd.Engine(128, 72, max_batch=3) as e:
        want = e.extract_raw(r, d)
        tr, td = torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda()
        torch.cuda.synchronize()
        got = e.extract_device_raw(tr, td, 3, 1)
        assert np.array_equal(got, want)
        with pytest.raises(vmaf_amd.VmafHipError, match="uint8 or uint16"):
            e.extract_raw(r.astype(np.int32), d.astype(np.int32))
        out = np.zeros((1, 20))
        assert e.L.vmaf_hip_extract_host_raw(e._h, r.ctypes.data, d.ctypes.data, 3, 128, 128 * 72, 1, 0,
                                             out.ctypes.data_as(vmaf_amd.binding._dp)) != 0
        assert "bytes_per_sample" in vmaf_amd.binding._err(e.L)


@pytest.mark.parametrize("order", ["avx", "scalar"])
@pytest.mark.parametrize("w,h,bits,frames", [(64, 48, 8, 3), (101, 75, 8, 2), (250, 130, 10, 2), (496, 288, 8, 2), (576, 324, 8, 3),
                                             (1280, 720, 8, 1), (1920, 1080, 8, 2), (1928, 1090, 10, 1)])
def test_exact_vif_sums_are_the_references_bit_for_bit(w, h, bits, frames, order):
    """vmaf_hip_engine_set_sums_exact (what disable_avx selects in compute_vmaf): the per-pixel num / den values, already
    bit-identical to the reference's, are added as vif_sum_s adds them (vif_tools.c:161-179: fp32 per row, then rows) - the
    eight per-scale VIF atoms and `vif` equal the oracle's (= the compiled reference's, tests/test_oracle_vs_ref.py) exactly,
    in both convolution orders, for both tile kernels, 8 and 10 bit.  The default path's fp64 sums must differ somewhere, or
    the test would mean nothing; everything but VIF is unchanged by the switch."""
    r, d = synth.make_clip(w, h, frames, bits=bits, seed=500 + w)
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    want = oracle.clip_atoms(ref, dis, oracle.ORDER_AVX if order == "avx" else oracle.ORDER_SCALAR)
    with vmaf_amd.Engine(w, h, max_batch=2, order=order, sums_exact=True) as e:
        got = e.extract_raw(r, d)
    with vmaf_amd.Engine(w, h, max_batch=2, order=order) as e:
        dflt = e.extract_raw(r, d)
    assert np.array_equal(got[:, 11:20], want[:, 11:20]), (got[:, 11:20] - want[:, 11:20])
    assert np.array_equal(got[:, :11], dflt[:, :11])
    if w * h >= 250 * 130:
        assert not np.array_equal(dflt[:, 11:19], want[:, 11:19])
    _check(dflt, want, f"default sums {w}x{h}")


@pytest.mark.parametrize("bits,w,h,pad,skew", [(8, 256, 96, 13, 1), (8, 101, 75, 3, 3), (10, 248, 130, 6, 1), (8, 496, 288, 16, 5)])
def test_raw_device_planes_with_padded_rows_and_unaligned_bases(bits, w, h, pad, skew):
    """The raw device entry takes byte strides: rows padded by `pad` samples, frames by a few rows, and the first sample
    `skew` samples into the allocation (an odd byte address for 8-bit input) - the typed buffer loads of VIF scale 0
    (workgroup-wide and wave-private tiles), ADM scale 0 (inner and edge tiles) and motion must deliver the same samples as
    from dense planes: atoms bit-identical."""
    torch = pytest.importorskip("torch")
    frames = 3
    r, d = synth.make_clip(w, h, frames, bits=bits, seed=300 + w)
    bps = r.dtype.itemsize
    with vmaf_amd.Engine(w, h, max_batch=frames) as e:
        want = e.extract_raw(r, d)
    rs, fs = w + pad, (w + pad) * (h + 2)                          # in samples

    def padded(a):
        buf = np.full(skew + frames * fs, 77, dtype=a.dtype)
        view = buf[skew:].reshape(frames, h + 2, w + pad)
        view[:, :h, :w] = a
        return buf
    br, bd = padded(r), padded(d)
    as_t = (lambda a: torch.from_numpy(a)) if bps == 1 else (lambda a: torch.from_numpy(a.view(np.int16)))
    tr, td = as_t(br).cuda(), as_t(bd).cuda()
    torch.cuda.synchronize()
    out = np.zeros((frames, 20))
    with vmaf_amd.Engine(w, h, max_batch=frames) as e:
        rc = e.L.vmaf_hip_extract_device_raw(e._h, tr.data_ptr() + skew * bps, td.data_ptr() + skew * bps, bps, rs * bps, fs * bps,
                                             frames, 0, out.ctypes.data_as(vmaf_amd.binding._dp))
        assert rc == 0, vmaf_amd.binding._err(e.L)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("bits,w,h", [(8, 576, 324), (8, 1280, 720), (10, 577, 325), (8, 200, 180), (8, 1920, 1080),
                                      (8, 1030, 520), (10, 1301, 1281), (8, 2050, 2062)])
def test_psnr_ssim_ms_ssim_match_oracle(bits, w, h):
    """the optional per-frame metrics of combo.c:168-230 (psnr.c, ssim.c, ms_ssim.c) on the GPU against
    the C restatement (itself bit-exact against the compiled reference, tests/test_oracle_vs_ref.py).  Covers SSIM
    decimation factors 1, 2, 3, 4, 5 and 8, odd sizes and the 10-bit PSNR peak.  Reference test tolerance: places=4.
    With all three metrics on, the PSNR sums come from the MS-SSIM level-0 launch (its last tiles take the 10 border
    columns / rows); the PSNR-only engine below takes psnr_kernel and must give the same bits to 1e-9."""
    frames = 2 if w >= 1280 else 3
    r, d = synth.make_clip(w, h, frames, bits=bits, seed=50 + w)
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    with vmaf_amd.Engine(w, h, max_batch=2) as e:
        e.enable_metrics(ten_bit=bits == 10)
        atoms = e.extract(ref, dis)
        got = e.metrics.copy()
        e.extract_raw(r, d)
        assert np.array_equal(e.metrics, got)          # raw ingest: same planes, same metrics
    with vmaf_amd.Engine(w, h, max_batch=2) as e:      # the metrics do not disturb the VMAF atoms
        assert np.array_equal(e.extract(ref, dis), atoms)
        e.enable_metrics(psnr=True, ssim=False, ms_ssim=False, ten_bit=bits == 10)
        e.extract(ref, dis)
        assert np.allclose(e.metrics[:, 0], got[:, 0], rtol=0, atol=1e-9)
    assert got.shape == (frames, vmaf_amd.binding.N_METRICS)
    for f in range(frames):
        want = (oracle.psnr(ref[f], dis[f], bits), oracle.ssim(ref[f], dis[f])[0], oracle.ms_ssim(ref[f], dis[f])[0])
        assert abs(got[f, 0] - want[0]) < 1e-9, (f, got[f], want)
        assert abs(got[f, 1] - want[1]) < 1e-6, (f, got[f], want)
        assert abs(got[f, 2] - want[2]) < 1e-6, (f, got[f], want)


@pytest.mark.parametrize("bits,w,h", [(8, 576, 324), (10, 577, 325), (8, 200, 180), (8, 1280, 720), (8, 1030, 520)])
def test_exact_metric_arithmetic_is_the_references(bits, w, h):
    """vmaf_hip_engine_set_metrics_exact (what disable_avx selects in compute_vmaf): SSIM / MS-SSIM window sums as
    iqa/convolve.c writes them (float product, widened, added) and IEEE sqrt / division per pixel, so every per-pixel
    l, c, s is the reference's bit for bit; what remains is the order of the final double sums over the picture
    (a tree here, a scan there: ~1e-16 relative per sum, and the scores pass through a float).  Held to 1e-9 against the
    oracle - a thousand times closer than the default path's bound, which must differ from it somewhere to make the
    test mean something."""
    r, d = synth.make_clip(w, h, 2, bits=bits, seed=90 + w, dis_noise=9)
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    with vmaf_amd.Engine(w, h, max_batch=2) as e:
        e.enable_metrics(ten_bit=bits == 10, exact=True)
        e.extract(ref, dis)
        exact = e.metrics.copy()
        e.enable_metrics(ten_bit=bits == 10, exact=False)
        e.extract(ref, dis)
        fast = e.metrics.copy()
    for f in range(2):
        want = (oracle.ssim(ref[f], dis[f])[0], oracle.ms_ssim(ref[f], dis[f])[0])
        assert abs(exact[f, 1] - want[0]) < 1e-9 and abs(exact[f, 2] - want[1]) < 1e-9, (f, exact[f], want)
        assert abs(fast[f, 1] - want[0]) < 1e-6 and abs(fast[f, 2] - want[1]) < 1e-6
    assert np.array_equal(exact[:, 0], fast[:, 0])                 # PSNR has one arithmetic


def test_metrics_identical_frames_and_too_small_frames():
    ref, _ = _clip(192, 180, 2, seed=8)
    with vmaf_amd.Engine(192, 180) as e:
        e.enable_metrics()
        e.extract(ref, ref)
        assert np.all(e.metrics[:, 0] == 60.0)         # psnr cap for 8-bit (psnr.c:54, combo.c:174)
        assert np.allclose(e.metrics[:, 1:3], 1.0, atol=1e-6)
    with vmaf_amd.Engine(160, 96) as e:                # 96 -> 48 -> 24 -> 12 -> 6 < 11: "scale below 1x1" (ms_ssim.c:172-177)
        with pytest.raises(vmaf_amd.VmafHipError, match="scale below 1x1"):
            e.enable_metrics()
        e.enable_metrics(ms_ssim=False)


def test_4k_10bit_properties_and_oracle_spot_check():
    """BASELINE.json configs[2] at full size (3840x2160 yuv420p10le): one frame against the oracle, and the
    size-independent properties on a short clip - batch-size independence, shard = whole (motion halo), raw 16-bit
    ingest = fp32 ingest - all bit for bit."""
    w, h = 3840, 2160
    r, d = synth.make_clip(w, h, 3, bits=10, seed=77, cuts=(2,))
    ref, dis = synth.to_luma_f32(r, 10), synth.to_luma_f32(d, 10)
    with vmaf_amd.Engine(w, h, max_batch=3) as e:
        whole = e.extract(ref, dis)
    with vmaf_amd.Engine(w, h, max_batch=1) as e:
        assert np.array_equal(e.extract(ref, dis), whole)
        assert np.array_equal(e.extract_raw(r, d), whole)
    with vmaf_amd.Engine(w, h, max_batch=2) as e:
        e.motion_seed(ref[1])
        assert np.array_equal(e.extract(ref[2:], dis[2:], continue_sequence=True), whole[2:])
    want = oracle.clip_atoms(ref[:2], dis[:2], oracle.ORDER_AVX)
    _check(whole[:2], want, "4K 10-bit")


def test_two_async_calls_in_flight():
    """include/vmaf_hip.h: batch k+1 may be enqueued while batch k runs; results equal the synchronous run, a third
    call in flight is refused, sync_oldest with nothing in flight is an error."""
    torch = pytest.importorskip("torch")
    ref, dis = _clip(256, 144, 6, seed=13, cuts=(3,))
    with vmaf_amd.Engine(256, 144, max_batch=2) as e:
        want = e.extract(ref, dis)
    tr, td = torch.from_numpy(ref).cuda(), torch.from_numpy(dis).cuda()
    torch.cuda.synchronize()
    got = np.zeros((3, 2, 20))
    with vmaf_amd.Engine(256, 144, max_batch=2) as e:
        with pytest.raises(vmaf_amd.VmafHipError, match="no async extract"):
            e.sync_oldest()
        e.extract_device(tr[0:2], td[0:2], 2, out=got[0], sync=False)
        e.extract_device(tr[2:4], td[2:4], 2, continue_sequence=True, out=got[1], sync=False)
        with pytest.raises(vmaf_amd.VmafHipError, match="in flight"):
            e.extract_device(tr[4:6], td[4:6], 2, continue_sequence=True, out=got[2], sync=False)
        with pytest.raises(vmaf_amd.VmafHipError, match="not been synchronised"):
            e.motion_seed(tr[0])
        e.sync_oldest()
        assert np.array_equal(got[0], want[0:2])
        e.extract_device(tr[4:6], td[4:6], 2, continue_sequence=True, out=got[2], sync=False)
        e.sync()
        assert np.array_equal(got.reshape(6, 20), want)


@pytest.mark.parametrize("kind", ["u8", "u16", "f32"])
def test_host_calls_in_flight_overlap_copy_and_compute(kind):
    """Host-pointer calls: the H2D copies of call k+1 go to their own stream and staging set while call k computes
    (vmaf_engine.cpp: s_copy, raw_ref[slot] / stage_ref[slot]).  Two calls in flight from page-locked memory, batch
    after batch, must give bit for bit what the synchronous calls give - for raw 8-bit, raw 16-bit and fp32 planes,
    with a short last batch and a scene cut at a batch edge."""
    from vmaf_amd.binding import PinnedArray
    bits = 10 if kind == "u16" else 8
    w, h, F, B = 320, 180, 11, 3
    r, d = synth.make_clip(w, h, F, bits=bits, seed=77, cuts=(6,))
    if kind == "f32":
        r, d = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    with vmaf_amd.Engine(w, h, max_batch=B) as e:
        want = e.extract(r, d) if kind == "f32" else e.extract_raw(r, d)
    pr, pd = PinnedArray(r.shape, r.dtype), PinnedArray(d.shape, d.dtype)
    try:
        pr.array[...] = r
        pd.array[...] = d
        got = np.zeros((F, 20))
        with vmaf_amd.Engine(w, h, max_batch=B) as e:
            inflight = 0
            for s0 in range(0, F, B):
                n = min(B, F - s0)
                if inflight == 2:
                    e.sync_oldest()
                    inflight -= 1
                e.extract_host_async(pr.array[s0:s0 + n], pd.array[s0:s0 + n], got[s0:s0 + n], continue_sequence=s0 > 0)
                inflight += 1
            e.sync()
            assert np.array_equal(got, want)
            # pageable memory (the copies degrade to synchronous ones) and a second pass over the same engine
            got2 = np.zeros((F, 20))
            for s0 in range(0, F, B):
                n = min(B, F - s0)
                e.extract_host_async(r[s0:s0 + n], d[s0:s0 + n], got2[s0:s0 + n], continue_sequence=s0 > 0)
                if s0:
                    e.sync_oldest()
            e.sync()
            assert np.array_equal(got2, want)
    finally:
        pr.close()
        pd.close()


def test_engines_on_concurrent_host_threads():
    """include/vmaf_hip.h, threading contract: engines are independent objects (own streams, buffers, pending slots; the
    error string is thread-local) and may be created, used and destroyed concurrently.  Four host threads each create
    their own engine, score their own clip five times (sync, raw and pipelined host calls) and destroy the engine, all at
    the same time; every thread gets bit for bit what an engine gives alone.  Daemon threads with bounded joins: a stuck
    thread fails the test instead of holding the process."""
    import threading
    clips = [_clip(320 + 16 * k, 180 + 8 * k, 6, seed=40 + k) for k in range(4)]
    alone = []
    for ref, dis in clips:
        with vmaf_amd.Engine(ref.shape[2], ref.shape[1], max_batch=3) as e:
            alone.append(e.extract(ref, dis))
    got, errs = [None] * 4, []

    def work(k):
        try:
            ref, dis = clips[k]
            for it in range(5):
                with vmaf_amd.Engine(ref.shape[2], ref.shape[1], max_batch=3) as e:
                    if it % 2:
                        out = np.zeros((6, 20))
                        e.extract_host_async(ref[:3], dis[:3], out[:3])
                        e.extract_host_async(ref[3:], dis[3:], out[3:], continue_sequence=True)
                        e.sync()
                        got[k] = out
                    else:
                        got[k] = e.extract(ref, dis)
                if not np.array_equal(got[k], alone[k]):
                    errs.append(f"thread {k} iteration {it}: results differ")
        except Exception as ex:                               # noqa: BLE001 - reported below
            errs.append(repr(ex))

    th = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in th), "a call did not return"
    assert not errs, errs
    for k in range(4):
        assert np.array_equal(got[k], alone[k])


@pytest.mark.parametrize("bits,w,h", [(8, 576, 324), (8, 101, 75), (10, 320, 180)])
def test_ansnr_matches_oracle(bits, w, h):
    """feature/src/ansnr.c through the GPU (the `ansnr` / `anpsnr` atoms of feature/src/all.c) against the C
    restatement, which is bit-exact against the compiled reference (tests/test_oracle_vs_ref.py)."""
    r, d = synth.make_clip(w, h, 3, bits=bits, seed=9 + w)
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    with vmaf_amd.Engine(w, h, max_batch=2) as e:
        e.enable_ansnr(ten_bit=bits == 10)
        e.extract(ref, dis)
        got = e.metrics.copy()
    for f in range(3):
        want = oracle.ansnr(ref[f], dis[f], bits)
        assert abs(got[f, 3] - want[0]) < 1e-4 and abs(got[f, 4] - want[1]) < 1e-4, (f, got[f], want)
    with vmaf_amd.Engine(w, h) as e:
        e.enable_ansnr(ten_bit=bits == 10)
        e.extract(ref[:1], ref[:1])
        want = oracle.ansnr(ref[0], ref[0], bits)        # identical planes still differ after the two different filters
        assert abs(e.metrics[0, 3] - want[0]) < 1e-4 and abs(e.metrics[0, 4] - want[1]) < 1e-4


@pytest.mark.parametrize("order", ["avx", "scalar"])
@pytest.mark.parametrize("w,h", [(576, 324), (577, 325), (200, 66), (1920, 1080)])
def test_intermediate_planes_are_bit_exact(w, h, order):
    """Stage level (the reference's *_OPT_DEBUG_DUMP counterpart): the planes that do touch HBM - the VIF pyramid
    (blur + vif_dec2, vif.c:157-186) and the ADM LL bands (adm.c:241-247) - equal the oracle's BIT FOR BIT in both
    summation orders, border frame included.  (The oracle's convolution is bit-exact against the compiled reference.)"""
    ref, dis = _clip(w, h, 2, seed=w * 3 + h)
    with vmaf_amd.Engine(w, h, max_batch=2, order=order) as e:
        e.extract(ref, dis)
        for f in (0, 1):
            for name, plane in (("ref", ref[f]), ("dis", dis[f])):
                off = plane + np.float32(-128)
                for lvl, want in enumerate(oracle.vif_pyramid(off, ORDERS[order]), start=1):
                    got = e.debug_plane("vif_" + name, lvl, f)
                    assert got.shape == want.shape and np.array_equal(got, want), (name, lvl, f)
                for lvl, want in enumerate(oracle.adm_ll_pyramid(off), start=1):
                    got = e.debug_plane("adm_" + name, lvl, f)
                    assert got.shape == want.shape and np.array_equal(got, want), (name, lvl, f)
