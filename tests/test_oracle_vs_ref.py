"""CPU tests: the oracle's C restatement against the reference's OWN C sources compiled by oracle/Makefile
(oracle/_ref/).  Skipped where oracle/_ref was not built (it needs /root/reference at build time; the
binaries then travel with the repo)."""
import numpy as np
import pytest

import oracle
from oracle import binding as B
from vmaf_amd import synth

pytestmark = pytest.mark.skipif(oracle.ref_lib() is None, reason="oracle/_ref not built (make -C oracle ref)")


@pytest.mark.parametrize("h,w", [(48, 64), (81, 145), (37, 53), (33, 200)])
@pytest.mark.parametrize("avx", [0, 1])
def test_separable_convolution_bit_exact(h, w, avx):
    rng = np.random.default_rng(h * 1000 + w)
    src = rng.integers(0, 256, (h, w)).astype(np.float32) - 128
    filters = [B.vif_filter(s) for s in range(4)] + [B.motion_filter()]
    for f in filters:
        assert np.array_equal(oracle.conv_sep(f, src, avx), oracle.ref_conv_sep(f, src, avx)), (len(f), avx)


@pytest.mark.parametrize("w,h", [(96, 64), (145, 81), (576, 324)])
def test_vif_and_adm_match_compiled_reference(w, h):
    r, d = synth.make_clip(w, h, 1, seed=w)
    r, d = synth.to_luma_f32(r)[0] - 128, synth.to_luma_f32(d)[0] - 128
    for avx in (0, 1):
        o, q = oracle.vif(r, d, avx), oracle.ref_vif(r, d, avx)
        assert o[0] == q[0] and np.array_equal(o[3], q[3])
    q = oracle.ref_adm(r, d)
    o = oracle.adm(r, d, exact_rcp=1)                       # same rcpss + Newton step as adm_tools.c:34-38
    assert np.array_equal(o[3], q[3]) and o[1] == q[1] and o[2] == q[2]
    o = oracle.adm(r, d, exact_rcp=0)                       # portable exact division: <= 3e-7 relative away
    assert np.allclose(o[3], q[3], rtol=3e-7, atol=0)


def test_frame_loop_matches_compiled_reference():
    r, d = synth.make_clip(160, 96, 4, seed=9, cuts=(2,))
    r, d = synth.to_luma_f32(r), synth.to_luma_f32(d)
    for avx in (0, 1):
        got, want = oracle.clip_atoms(r, d, avx), oracle.ref_clip_atoms(r, d, avx)
        assert np.array_equal(got[:, 10:], want[:, 10:])
        assert np.allclose(got[:, :10], want[:, :10], rtol=3e-7, atol=0)


def test_csf_factors_match_survey_values():
    # evaluated 1/Q values
    want = [0.0173815, 0.00589069, 0.0319848, 0.0142991, 0.0433727, 0.0243969, 0.0456734, 0.0313127]
    assert np.allclose(oracle.csf_rfactors(), want, rtol=2e-6)


@pytest.mark.parametrize("w,h,bits", [(576, 324, 8), (200, 180, 8), (577, 325, 10), (1280, 720, 8), (1083, 769, 8),
                                      (1030, 520, 8), (1301, 1281, 10), (2050, 2062, 8)])    # decimation factors 2, 5, 8
def test_psnr_ssim_ms_ssim_bit_exact(w, h, bits):
    """oracle/iqa_oracle.c against psnr.c / ssim.c / ms_ssim.c + iqa/ compiled from the reference: SSIM decimation
    factors 1, 3 (720p, odd-sized kernel), 3 on odd dimensions (sw = w/f + (w&1)), 2, 5 and 8, the 10-bit PSNR peak."""
    r, d = synth.make_clip(w, h, 1, bits=bits, seed=w + h)
    r, d = synth.to_luma_f32(r, bits)[0], synth.to_luma_f32(d, bits)[0]
    assert oracle.psnr(r, d, bits) == oracle.ref_psnr(r, d, bits)
    assert oracle.ssim(r, d) == oracle.ref_ssim(r, d)
    got, want = oracle.ms_ssim(r, d), oracle.ref_ms_ssim(r, d)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    assert oracle.psnr(r, r, bits) == (60.0 if bits == 8 else 72.0)


@pytest.mark.parametrize("w,h,bits", [(576, 324, 8), (101, 75, 8), (320, 180, 10)])
def test_ansnr_bit_exact(w, h, bits):
    r, d = synth.make_clip(w, h, 1, bits=bits, seed=w)
    r, d = synth.to_luma_f32(r, bits)[0], synth.to_luma_f32(d, bits)[0]
    assert oracle.ansnr(r, d, bits) == oracle.ref_ansnr(r, d, bits)
