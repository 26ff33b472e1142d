"""vmaf_hip_extract_device_noalloc (include/vmaf_hip.h): the entry point that allocates nothing, for callers that cannot
call hipMalloc.  Checked through the C-ABI against the existing entry point it is a variant of - an engine's
vmaf_hip_extract_device on the same device-resident frames - bit for bit, on two calls in a row, the second with a
smaller input (the static scratch area keeps the first call's contents), plus the size limit and the argument checks."""
import numpy as np
import pytest

import vmaf_amd
from vmaf_amd import binding, synth

pytestmark = pytest.mark.gpu


def _device_clip(torch, w, h, frames, seed, bits=8):
    r, d = synth.make_clip(w, h, frames, bits=bits, seed=seed)
    ref, dis = synth.to_luma_f32(r, bits), synth.to_luma_f32(d, bits)
    return torch.from_numpy(ref).cuda(), torch.from_numpy(dis).cuda()


def _engine_atoms(tr, td, w, h, n, order="avx"):
    with vmaf_amd.Engine(w, h, max_batch=n, order=order) as e:
        return e.extract_device(tr, td, n)


def test_noalloc_entry_matches_the_engine_on_two_calls_in_a_row():
    torch = pytest.importorskip("torch")
    # first call: close to the limit (two 1080p frames = 28.9 of 32 MiB); second: a smaller, odd-sized input whose
    # buffers land on top of the first call's leftovers; third: the first again (the order of sizes must not matter)
    cases = [(1920, 1080, 2, 5, "avx"), (577, 325, 5, 6, "avx"), (320, 180, 3, 7, "scalar"), (1920, 1080, 2, 5, "avx")]
    for w, h, n, seed, order in cases:
        tr, td = _device_clip(torch, w, h, n, seed)
        torch.cuda.synchronize()
        assert 0 < binding.load_library().vmaf_hip_noalloc_scratch_needed(w, h, n) <= binding.NOALLOC_SCRATCH_BYTES
        want = _engine_atoms(tr, td, w, h, n, order)
        got = binding.extract_device_noalloc(tr, td, w, h, n, order=order)
        assert np.array_equal(got, want), (w, h, n, order, np.abs(got - want).max())
        assert got[0, 10] == 0.0 and np.all(got[1:, 10] > 0.0)        # motion: frame 0 starts the video


def test_noalloc_entry_on_a_caller_stream_and_with_strides():
    torch = pytest.importorskip("torch")
    w, h, n = 200, 66, 3
    tr, td = _device_clip(torch, w, h, n, 21)
    want = _engine_atoms(tr, td, w, h, n)
    # padded rows and frames, as a read_frame-style caller would hold them (stride = 4w rounded up to 32, frame gap)
    rs = (w * 4 + 31) // 32 * 32
    fs = rs * h + 96
    pr = torch.zeros(n * fs // 4, dtype=torch.float32, device="cuda")
    pd = torch.zeros_like(pr)
    for f in range(n):
        pr[f * fs // 4:f * fs // 4 + rs // 4 * h].view(h, rs // 4)[:, :w] = tr[f]
        pd[f * fs // 4:f * fs // 4 + rs // 4 * h].view(h, rs // 4)[:, :w] = td[f]
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    got = binding.extract_device_noalloc(pr, pd, w, h, n, stream=st.cuda_stream, row_stride_bytes=rs, frame_stride_bytes=fs)
    assert np.array_equal(got, want)


def test_noalloc_entry_refuses_what_does_not_fit():
    torch = pytest.importorskip("torch")
    L = binding.load_library()
    assert L.vmaf_hip_noalloc_scratch_needed(3840, 2160, 1) > binding.NOALLOC_SCRATCH_BYTES
    assert L.vmaf_hip_noalloc_scratch_needed(16, 16, 1) == 0 and L.vmaf_hip_noalloc_scratch_needed(64, 48, 0) == 0
    t = torch.zeros(3 * 1080 * 1920, dtype=torch.float32, device="cuda")
    with pytest.raises(vmaf_amd.VmafHipError, match="need .* bytes of scratch"):
        binding.extract_device_noalloc(t, t, 1920, 1080, 3)
    with pytest.raises(vmaf_amd.VmafHipError, match="too small"):
        binding.extract_device_noalloc(t, t, 16, 16, 1)
    with pytest.raises(vmaf_amd.VmafHipError, match="row stride"):
        binding.extract_device_noalloc(t, t, 64, 48, 1, row_stride_bytes=128)
    with pytest.raises(vmaf_amd.VmafHipError, match="null pointer"):
        binding.extract_device_noalloc(0, t, 64, 48, 1)


def test_overlapping_noalloc_calls_are_refused_not_raced():
    """The calls of one GPU share its static scratch area.  Four host threads call at once (ctypes releases the GIL): every
    call either returns the right atoms or is refused with the 'another call' message - never a silently mixed result -
    and at least one goes through."""
    torch = pytest.importorskip("torch")
    import threading
    w, h, n = 640, 360, 4
    tr, td = _device_clip(torch, w, h, n, 33)
    torch.cuda.synchronize()
    want = _engine_atoms(tr, td, w, h, n)
    ok, refused, wrong = [], [], []

    def worker():
        for _ in range(6):
            try:
                got = binding.extract_device_noalloc(tr, td, w, h, n)
                (ok if np.array_equal(got, want) else wrong).append(1)
            except vmaf_amd.VmafHipError as exc:
                (refused if "another call" in str(exc) else wrong).append(str(exc))

    th = [threading.Thread(target=worker) for _ in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not wrong, wrong[:3]
    assert ok and len(ok) + len(refused) == 24
    assert np.array_equal(binding.extract_device_noalloc(tr, td, w, h, n), want)      # the flag is released again
