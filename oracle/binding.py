"""ctypes bindings for the oracle (oracle/liboracle.so) and, when built, for the compiled
reference (oracle/_ref/libvmaf_ref_feature.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORDER_SCALAR, ORDER_AVX = 0, 1

# insertion order of wrapper/src/combo.c:260-364
ATOM_NAMES = (["adm_num", "adm_den"] +
              [f"adm_{k}_scale{s}" for s in range(4) for k in ("num", "den")] +
              ["motion"] +
              [f"vif_{k}_scale{s}" for s in range(4) for k in ("num", "den")] +
              ["vif"])

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_lib = None
_ref = False


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2
    return a


def _ptr(a):
    return a.ctypes.data_as(_fp)


def lib():
    """Load (building if necessary) this repo's C restatement."""
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "liboracle.so")
        srcs = [os.path.join(HERE, f) for f in ("vmaf_oracle.c", "iqa_oracle.c")]
        if not os.path.exists(so) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
            subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
        L = C.CDLL(so)
        L.oracle_conv_sep.argtypes = [_fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_conv_sep.restype = None
        L.oracle_vif.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
        L.oracle_vif_exact_sums.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
        L.oracle_adm.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
        L.oracle_sad.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_sad.restype = C.c_double
        L.oracle_frame.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _dp]
        L.oracle_vif_filter.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.oracle_vif_filter.restype = _fp
        L.oracle_motion_filter.restype = _fp
        L.oracle_adm_csf_rfactors.argtypes = [_fp]
        L.oracle_log2f_approx.argtypes = [C.c_float]
        L.oracle_log2f_approx.restype = C.c_float
        L.oracle_adm_dwt2_ll.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp]
        L.oracle_adm_dwt2_ll.restype = None
        L.oracle_psnr.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.oracle_psnr.restype = C.c_double
        L.oracle_ssim.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, _dp]
        L.oracle_ms_ssim.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.oracle_ssim_scale.argtypes = [C.c_int, C.c_int]
        L.oracle_ansnr.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _dp, _dp]
        _lib = L
    return _lib


def ref_lib():
    """The reference's own C sources, compiled by `make -C oracle ref`.  None when absent."""
    global _ref
    if _ref is False:
        so = os.path.join(HERE, "_ref", "libvmaf_ref_feature.so")
        if not os.path.exists(so):
            _ref = None
        else:
            R = C.CDLL(so)
            R.compute_vif.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
            R.compute_adm.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double]
            R.compute_motion.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
            R.convolution_f32_c_s.argtypes = [_fp, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int]
            R.convolution_f32_c_s.restype = None
            R.ref_set_cpu.argtypes = [C.c_int]
            if hasattr(R, "compute_psnr"):
                R.compute_psnr.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_double]
                R.compute_ssim.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
                R.compute_ms_ssim.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
            if hasattr(R, "compute_ansnr"):
                R.compute_ansnr.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double]
            _ref = R
    return _ref


def ref_exec_path():
    p = os.path.join(HERE, "_ref", "vmafossexec")
    return p if os.path.exists(p) else None


# ------------------------------------------------------------------ oracle wrappers
def vif_filter(scale):
    n = C.c_int()
    p = lib().oracle_vif_filter(scale, C.byref(n))
    return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def motion_filter():
    return np.ctypeslib.as_array(lib().oracle_motion_filter(), shape=(5,)).copy()


def csf_rfactors():
    out = np.zeros(8, np.float32)
    lib().oracle_adm_csf_rfactors(_ptr(out.reshape(1, 8)))
    return out


def conv_sep(filt, src, order=ORDER_SCALAR):
    src = _f32(src)
    filt = np.ascontiguousarray(filt, np.float32)
    h, w = src.shape
    dst = np.empty_like(src)
    lib().oracle_conv_sep(filt.ctypes.data_as(_fp), len(filt), _ptr(src), _ptr(dst), w, h, w, w, order)
    return dst


def vif(ref_off, dis_off, order=ORDER_SCALAR):
    """-> (score, num, den, scores[8]) for already offset (-128) planes."""
    r, d = _f32(ref_off), _f32(dis_off)
    h, w = r.shape
    s, n, dn = C.c_double(), C.c_double(), C.c_double()
    sc = np.zeros(8)
    rc = lib().oracle_vif(_ptr(r), _ptr(d), w, h, w, order, C.byref(s), C.byref(n), C.byref(dn),
                          sc.ctypes.data_as(_dp))
    assert rc == 0
    return s.value, n.value, dn.value, sc


def vif_exact_sums(ref_off, dis_off, order=ORDER_SCALAR):
    """-> [8] sums of the per-pixel VIF num / den values of scales 0..3 in extended precision (planes already offset)."""
    r, d = _f32(ref_off), _f32(dis_off)
    h, w = r.shape
    out = np.zeros(8)
    rc = lib().oracle_vif_exact_sums(_ptr(r), _ptr(d), w, h, w, order, out.ctypes.data_as(_dp))
    assert rc == 0
    return out


def adm(ref_off, dis_off, exact_rcp=0):
    r, d = _f32(ref_off), _f32(dis_off)
    h, w = r.shape
    s, n, dn = C.c_double(), C.c_double(), C.c_double()
    sc = np.zeros(8)
    rc = lib().oracle_adm(_ptr(r), _ptr(d), w, h, w, exact_rcp, C.byref(s), C.byref(n), C.byref(dn),
                          sc.ctypes.data_as(_dp))
    assert rc == 0
    return s.value, n.value, dn.value, sc


def sad(a, b):
    a, b = _f32(a), _f32(b)
    h, w = a.shape
    return lib().oracle_sad(_ptr(a), _ptr(b), w, h, w, w)


def frame(ref_raw, dis_raw, prev_blur, first, order=ORDER_SCALAR):
    r, d = _f32(ref_raw), _f32(dis_raw)
    h, w = r.shape
    atoms = np.zeros(20)
    rc = lib().oracle_frame(_ptr(r), _ptr(d), w, h, w, _ptr(prev_blur), int(first), order,
                            atoms.ctypes.data_as(_dp))
    assert rc == 0
    return atoms


def clip_atoms(ref_frames, dis_frames, order=ORDER_SCALAR):
    """Per-frame loop of wrapper/src/combo.c:147-371 over un-offset luma planes [F,h,w] -> [F,20]."""
    F = len(ref_frames)
    h, w = ref_frames[0].shape
    prev = np.zeros((h, w), np.float32)
    out = np.zeros((F, 20))
    for i in range(F):
        out[i] = frame(ref_frames[i], dis_frames[i], prev, i == 0, order)
    return out


# ------------------------------------------------------------------ compiled-reference wrappers
def _stride32(w):
    b = w * 4
    return b + ((32 - b % 32) % 32)   # ALIGN_CEIL, feature/src/common/alloc.h:24-27


def _padded(a):
    """Copy into a 32-byte aligned, 32-byte-row-stride buffer as combo.c:106-144 allocates."""
    a = _f32(a)
    h, w = a.shape
    sp = _stride32(w) // 4
    raw = np.zeros(h * sp + 16, np.float32)
    off = (-raw.ctypes.data % 32) // 4
    buf = raw[off:off + h * sp].reshape(h, sp)
    buf[:, :w] = a
    return buf, sp * 4


def ref_conv_sep(filt, src, avx):
    R = ref_lib()
    R.ref_set_cpu(1 if avx else 0)
    filt = np.ascontiguousarray(filt, np.float32)
    s, sb = _padded(src)
    h, w = np.asarray(src).shape
    d, _ = _padded(np.zeros((h, w), np.float32))
    t, _ = _padded(np.zeros((h, w + 8), np.float32))
    R.convolution_f32_c_s(filt.ctypes.data_as(_fp), len(filt), _ptr(s), _ptr(d), _ptr(t), w, h, sb // 4, sb // 4)
    return d[:, :w].copy()


def ref_vif(ref_off, dis_off, avx):
    R = ref_lib()
    R.ref_set_cpu(1 if avx else 0)
    r, sb = _padded(ref_off)
    d, _ = _padded(dis_off)
    h, w = np.asarray(ref_off).shape
    s, n, dn = C.c_double(), C.c_double(), C.c_double()
    sc = np.zeros(8)
    rc = R.compute_vif(_ptr(r), _ptr(d), w, h, sb, sb, C.byref(s), C.byref(n), C.byref(dn), sc.ctypes.data_as(_dp))
    assert rc == 0
    return s.value, n.value, dn.value, sc


def ref_adm(ref_off, dis_off):
    R = ref_lib()
    r, sb = _padded(ref_off)
    d, _ = _padded(dis_off)
    h, w = np.asarray(ref_off).shape
    s, n, dn = C.c_double(), C.c_double(), C.c_double()
    sc = np.zeros(8)
    rc = R.compute_adm(_ptr(r), _ptr(d), w, h, sb, sb, C.byref(s), C.byref(n), C.byref(dn),
                       sc.ctypes.data_as(_dp), 0.1)
    assert rc == 0
    return s.value, n.value, dn.value, sc


def ref_frame(ref_raw, dis_raw, prev_blur, first, avx):
    """One iteration of wrapper/src/combo.c:236-364 through the compiled reference.
    -> (atoms[20], blurred ref plane to pass as prev_blur for the next frame)."""
    R = ref_lib()
    R.ref_set_cpu(1 if avx else 0)
    h, w = np.asarray(ref_raw).shape
    out = np.zeros(20)
    r = np.asarray(ref_raw, np.float32) + np.float32(-128)
    d = np.asarray(dis_raw, np.float32) + np.float32(-128)
    _, n, dn, sc = ref_adm(r, d)
    out[0], out[1], out[2:10] = n, dn, sc
    blur = ref_conv_sep(motion_filter(), r, avx)
    if not first:
        a, sb = _padded(prev_blur)
        b, _ = _padded(blur)
        m = C.c_double()
        R.compute_motion(_ptr(a), _ptr(b), w, h, sb, sb, C.byref(m))
        out[10] = m.value
    s, _, _, sc = ref_vif(r, d, avx)
    out[11:19], out[19] = sc, s
    return out, blur


def ref_clip_atoms(ref_frames, dis_frames, avx):
    """The per-frame loop of wrapper/src/combo.c:147-371 driven through the compiled reference."""
    F = len(ref_frames)
    out = np.zeros((F, 20))
    prev = None
    for i in range(F):
        out[i], prev = ref_frame(ref_frames[i], dis_frames[i], prev, i == 0, avx)
    return out


class _SvmNode(C.Structure):
    _fields_ = [("index", C.c_int), ("value", C.c_double)]       # wrapper/src/svm.h:12-16


def ref_svm_predict(model_file, x):
    """svm_load_model + svm_predict of the compiled reference (wrapper/src/svm.cpp:305,468) on rows of x."""
    so = os.path.join(HERE, "_ref", "libref_svm.so")
    if not os.path.exists(so):
        return None
    S = C.CDLL(so)
    S.svm_load_model.restype = C.c_void_p
    S.svm_load_model.argtypes = [C.c_char_p]
    S.svm_predict.restype = C.c_double
    S.svm_predict.argtypes = [C.c_void_p, C.POINTER(_SvmNode)]
    m = S.svm_load_model(model_file.encode())
    assert m, model_file
    x = np.asarray(x, np.float64)
    out = np.zeros(len(x))
    for f in range(len(x)):
        nodes = (_SvmNode * (x.shape[1] + 1))()
        for j in range(x.shape[1]):
            nodes[j].index, nodes[j].value = j + 1, x[f, j]
        nodes[x.shape[1]].index = -1                             # terminator, vmaf.cpp:385-389
        out[f] = S.svm_predict(m, nodes)
    return out


# ---------------------------------------------------------------- optional metrics (psnr.c, ssim.c, ms_ssim.c)
def _peak(bits):
    return (255.0, 60.0) if bits == 8 else (255.75, 72.0)          # combo.c:171-186


def psnr(ref_raw, dis_raw, bits=8):
    r, d = _f32(ref_raw), _f32(dis_raw)
    h, w = r.shape
    return lib().oracle_psnr(_ptr(r), _ptr(d), w, h, w, *_peak(bits))


def ssim(ref_raw, dis_raw):
    """-> (ssim, l, c, s)"""
    r, d = _f32(ref_raw), _f32(dis_raw)
    h, w = r.shape
    out = np.zeros(4)
    if lib().oracle_ssim(_ptr(r), _ptr(d), w, h, w, out.ctypes.data_as(_dp)):
        raise ValueError("image too small for the SSIM window after decimation")
    return tuple(out)


def ms_ssim(ref_raw, dis_raw):
    """-> (ms_ssim, lcs[15])"""
    r, d = _f32(ref_raw), _f32(dis_raw)
    h, w = r.shape
    s = C.c_double()
    lcs = np.zeros(15)
    if lib().oracle_ms_ssim(_ptr(r), _ptr(d), w, h, w, C.byref(s), lcs.ctypes.data_as(_dp)):
        raise ValueError("scale below 1x1")
    return s.value, lcs


def ref_psnr(ref_raw, dis_raw, bits=8):
    R = ref_lib()
    r, sb = _padded(ref_raw)
    d, _ = _padded(dis_raw)
    h, w = np.asarray(ref_raw).shape
    s = C.c_double()
    assert R.compute_psnr(_ptr(r), _ptr(d), w, h, sb, sb, C.byref(s), *_peak(bits)) == 0
    return s.value


def ref_ssim(ref_raw, dis_raw):
    R = ref_lib()
    r, sb = _padded(ref_raw)
    d, _ = _padded(dis_raw)
    h, w = np.asarray(ref_raw).shape
    v = [C.c_double() for _ in range(4)]
    assert R.compute_ssim(_ptr(r), _ptr(d), w, h, sb, sb, *[C.byref(x) for x in v]) == 0
    return tuple(x.value for x in v)


def ref_ms_ssim(ref_raw, dis_raw):
    R = ref_lib()
    r, sb = _padded(ref_raw)
    d, _ = _padded(dis_raw)
    h, w = np.asarray(ref_raw).shape
    s = C.c_double()
    l, c, st = np.zeros(5), np.zeros(5), np.zeros(5)
    assert R.compute_ms_ssim(_ptr(r), _ptr(d), w, h, sb, sb, C.byref(s), l.ctypes.data_as(_dp), c.ctypes.data_as(_dp),
                             st.ctypes.data_as(_dp)) == 0
    return s.value, np.concatenate([l, c, st])


def ansnr(ref_raw, dis_raw, bits=8):
    """-> (ansnr, anpsnr) of the planes after the -128 offset every caller applies (ansnr.c:174-177, all.c:128-129)."""
    r, d = _f32(ref_raw) + np.float32(-128), _f32(dis_raw) + np.float32(-128)
    h, w = r.shape
    a, b = C.c_double(), C.c_double()
    lib().oracle_ansnr(_ptr(r), _ptr(d), w, h, w, *_peak(bits), C.byref(a), C.byref(b))
    return a.value, b.value


def ref_ansnr(ref_raw, dis_raw, bits=8):
    R = ref_lib()
    r, sb = _padded(np.asarray(ref_raw, np.float32) + np.float32(-128))
    d, _ = _padded(np.asarray(dis_raw, np.float32) + np.float32(-128))
    h, w = np.asarray(ref_raw).shape
    a, b = C.c_double(), C.c_double()
    assert R.compute_ansnr(_ptr(r), _ptr(d), w, h, sb, sb, C.byref(a), C.byref(b), *_peak(bits)) == 0
    return a.value, b.value


# ---------------------------------------------------------------- stage-level planes (tests of the GPU intermediates)
def vif_pyramid(plane_off, order=ORDER_SCALAR):
    """The planes compute_vif works on at scales 1..3 (vif.c:157-186): blur with the scale's own filter, then
    vif_dec2 (vif_tools.c:146-159: dst[i][j] = src[2i][2j], i < h/2, j < w/2).  plane_off: after the -128 offset."""
    out, cur = [], _f32(plane_off)
    for s in (1, 2, 3):
        h, w = cur.shape
        blur = conv_sep(vif_filter(s), cur, order)
        cur = np.ascontiguousarray(blur[0:2 * (h // 2):2, 0:2 * (w // 2):2])
        out.append(cur)
    return out


def adm_ll_pyramid(plane_off):
    """The LL bands compute_adm hands from scale to scale (adm.c:241-247), levels 0..2."""
    out, cur = [], _f32(plane_off)
    for _ in range(3):
        h, w = cur.shape
        ll = np.zeros(((h + 1) // 2, (w + 1) // 2), np.float32)
        lib().oracle_adm_dwt2_ll(_ptr(cur), w, h, w, _ptr(ll))
        out.append(ll)
        cur = ll
    return out
