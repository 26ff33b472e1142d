"""ctypes bindings of vmaf_amd/libvmaf_amd.so (include/vmaf_hip.h, include/libvmaf.h).

Fails loudly: a missing library raises VmafHipError at first use; nothing here computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
DEFAULT_MODEL = os.path.join(REPO, "model", "vmaf_v0.6.1.pkl")

# insertion order of the reference's wrapper/src/combo.c:260-364
ATOM_NAMES = (["adm_num", "adm_den"] +
              [f"adm_{k}_scale{s}" for s in range(4) for k in ("num", "den")] +
              ["motion"] +
              [f"vif_{k}_scale{s}" for s in range(4) for k in ("num", "den")] +
              ["vif"])
N_ATOMS = 20
N_METRICS = 5          # psnr, ssim, ms_ssim, ansnr, anpsnr (vmaf_hip_engine_get_metrics)


class VmafHipError(RuntimeError):
    pass


_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
READ_FRAME = C.CFUNCTYPE(C.c_int, _fp, _fp, _fp, C.c_int, C.c_void_p)
READ_FRAME_RAW_AT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p)
_lib = None


def library_path():
    return os.path.join(HERE, "libvmaf_amd.so")


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise VmafHipError(f"{path} is missing: build it with `make -C vmaf_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    L.vmaf_hip_last_error.restype = C.c_char_p
    L.vmaf_hip_device_count.restype = i
    L.vmaf_hip_engine_create.argtypes = [C.POINTER(vp), i, i, i, i]
    L.vmaf_hip_engine_destroy.argtypes = [vp]
    L.vmaf_hip_engine_destroy.restype = None
    L.vmaf_hip_engine_set_conv_order.argtypes = [vp, i]
    for name in ("vmaf_hip_extract_device", "vmaf_hip_extract_host", "vmaf_hip_extract_device_async",
                 "vmaf_hip_extract_host_async"):
        getattr(L, name).argtypes = [vp, vp, vp, sz, sz, i, i, _dp]
    for name in ("vmaf_hip_extract_device_raw", "vmaf_hip_extract_host_raw", "vmaf_hip_extract_device_raw_async",
                 "vmaf_hip_extract_host_raw_async"):
        getattr(L, name).argtypes = [vp, vp, vp, i, sz, sz, i, i, _dp]
    L.vmaf_hip_motion_seed_host_raw.argtypes = [vp, vp, i, sz]
    L.vmaf_hip_motion_seed_device_raw.argtypes = [vp, vp, i, sz]
    L.vmaf_hip_engine_enable_metrics.argtypes = [vp, i, i, i, i]
    L.vmaf_hip_engine_enable_ansnr.argtypes = [vp, i]
    L.vmaf_hip_engine_set_metrics_exact.argtypes = [vp, i]
    L.vmaf_hip_engine_set_sums_exact.argtypes = [vp, i]
    L.vmaf_hip_engine_get_metrics.argtypes = [vp, _dp, i]
    L.vmaf_hip_engine_sync.argtypes = [vp]
    L.vmaf_hip_engine_sync_oldest.argtypes = [vp]
    L.vmaf_hip_engine_debug_read_plane.argtypes = [vp, i, i, i, _fp, sz, C.POINTER(i), C.POINTER(i)]
    L.vmaf_hip_engine_debug_read_sums.argtypes = [vp, i, _dp]
    L.vmaf_hip_host_alloc.argtypes = [sz]
    L.vmaf_hip_host_alloc.restype = vp
    L.vmaf_hip_host_free.argtypes = [vp]
    L.vmaf_hip_host_free.restype = None
    L.vmaf_hip_motion_seed_device.argtypes = [vp, vp, sz]
    L.vmaf_hip_motion_seed_host.argtypes = [vp, vp, sz]
    L.vmaf_hip_noalloc_scratch_needed.argtypes = [i, i, i]
    L.vmaf_hip_noalloc_scratch_needed.restype = sz
    L.vmaf_hip_extract_device_noalloc.argtypes = [i, vp, vp, vp, i, i, sz, sz, i, i, _dp]
    L.vmaf_hip_svm_predict.argtypes = [i, _dp, _dp, i, i, C.c_double, C.c_double, _dp, i, _dp]
    L.vmaf_hip_last_kernel_ms.argtypes = [vp, i]
    L.vmaf_hip_last_kernel_ms.restype = C.c_double
    L.compute_vmaf_noexcept.argtypes = [_dp, C.c_char_p, i, i, READ_FRAME, vp, C.c_char_p, C.c_char_p, C.c_char_p,
                                        i, i, i, i, i, i, i, C.c_char_p, C.c_char_p, i]
    L.compute_vmaf_raw_seekable_noexcept.argtypes = [_dp, C.c_char_p, i, i, C.c_long, READ_FRAME_RAW_AT, vp, C.c_char_p, C.c_char_p, C.c_char_p,
                                                     i, i, i, i, i, i, i, C.c_char_p, C.c_char_p, i]
    L.vmaf_amd_predict_from_atoms_v2.argtypes = [C.c_char_p, _dp, i, i, i, i, _dp, _dp, i, C.c_char_p, i]
    L.vmaf_amd_predict_from_atoms.argtypes = [C.c_char_p, _dp, i, i, i, i, _dp, _dp, C.c_char_p, i]
    L.vmaf_amd_model_feature_names.argtypes = [C.c_char_p, C.POINTER(i), C.c_char_p, i, C.c_char_p, i]
    _lib = L
    return L


def _err(L):
    return (L.vmaf_hip_last_error() or b"").decode(errors="replace")


def device_count():
    return load_library().vmaf_hip_device_count()


class Engine:
    """One GPU, one frame geometry, one video's motion state (vmaf_hip_engine)."""

    def __init__(self, width, height, max_batch=16, device=0, order="avx", sums_exact=False):
        self.L = load_library()
        self.w, self.h, self.max_batch, self.device = width, height, max_batch, device
        h = C.c_void_p()
        if self.L.vmaf_hip_engine_create(C.byref(h), device, width, height, max_batch):
            raise VmafHipError(_err(self.L))
        self._h = h
        self._keep = None
        self._metrics_on = False
        self.metrics = np.zeros((0, N_METRICS))      # of the frames of the last extract / extract_raw call
        if self.L.vmaf_hip_engine_set_conv_order(h, {"avx": 1, "scalar": 0, "fma": 2}[order]):
            raise VmafHipError(_err(self.L))
        # sums_exact: the VIF plane sums in vif_sum_s's own order (fp32 per row, then rows): VIF atoms bit-identical to the reference's
        if sums_exact and self.L.vmaf_hip_engine_set_sums_exact(h, 1):
            raise VmafHipError(_err(self.L))

    def close(self):
        if getattr(self, "_h", None):
            self.L.vmaf_hip_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def enable_metrics(self, psnr=True, ssim=True, ms_ssim=True, ten_bit=False, exact=False):
        """Optional per-frame PSNR / SSIM / MS-SSIM (combo.c:168-230) next to the VMAF atoms; read them from
        self.metrics [F,3] after extract() / extract_raw().  exact: SSIM / MS-SSIM in the reference's own expressions
        (vmaf_hip_engine_set_metrics_exact) instead of exact-product fused sums."""
        if self.L.vmaf_hip_engine_enable_metrics(self._h, int(psnr), int(ssim), int(ms_ssim), int(ten_bit)) or \
                self.L.vmaf_hip_engine_set_metrics_exact(self._h, int(exact)):
            raise VmafHipError(_err(self.L))
        self._metrics_on = bool(psnr or ssim or ms_ssim)

    def enable_ansnr(self, ten_bit=False):
        """ANSNR / ANPSNR (feature/src/ansnr.c) as columns 3 and 4 of self.metrics."""
        if self.L.vmaf_hip_engine_enable_ansnr(self._h, int(ten_bit)):
            raise VmafHipError(_err(self.L))
        self._metrics_on = True

    def _collect_metrics(self, rows, n):
        if self._metrics_on:
            m = np.zeros((n, N_METRICS))
            if self.L.vmaf_hip_engine_get_metrics(self._h, m.ctypes.data_as(_dp), n):
                raise VmafHipError(_err(self.L))
            rows.append(m)

    # -- host arrays ------------------------------------------------------------------------
    def extract(self, ref, dis, continue_sequence=False):
        """ref, dis: float32 numpy [F,h,w] un-offset luma (8-bit units).  -> atoms [F,20]."""
        ref = np.ascontiguousarray(ref, np.float32)
        dis = np.ascontiguousarray(dis, np.float32)
        if ref.ndim == 2:
            ref, dis = ref[None], dis[None]
        F, h, w = ref.shape
        if (h, w) != (self.h, self.w) or dis.shape != ref.shape:
            raise VmafHipError(f"shape {ref.shape}/{dis.shape} does not match engine {self.h}x{self.w}")
        out = np.zeros((F, N_ATOMS))
        rows = []
        done = 0
        while done < F:
            n = min(self.max_batch, F - done)
            if self.L.vmaf_hip_extract_host(self._h, ref[done:].ctypes.data, dis[done:].ctypes.data, w * 4, w * h * 4, n,
                                            1 if (continue_sequence or done) else 0, out[done:].ctypes.data_as(_dp)):
                raise VmafHipError(_err(self.L))
            self._collect_metrics(rows, n)
            done += n
        self.metrics = np.concatenate(rows) if rows else np.zeros((0, N_METRICS))
        return out

    def extract_raw(self, ref, dis, continue_sequence=False):
        """ref, dis: uint8 or uint16 numpy [F,h,w] luma as the YUV file holds it (uint16 = 10-bit words, scored
        as value / 4.0); the widening to fp32 happens on the GPU.  -> atoms [F,20]."""
        ref, dis = np.ascontiguousarray(ref), np.ascontiguousarray(dis)
        if ref.dtype not in (np.uint8, np.uint16) or dis.dtype != ref.dtype:
            raise VmafHipError(f"raw luma must be uint8 or uint16, got {ref.dtype}/{dis.dtype}")
        if ref.ndim == 2:
            ref, dis = ref[None], dis[None]
        F, h, w = ref.shape
        if (h, w) != (self.h, self.w) or dis.shape != ref.shape:
            raise VmafHipError(f"shape {ref.shape}/{dis.shape} does not match engine {self.h}x{self.w}")
        bps = ref.dtype.itemsize
        out = np.zeros((F, N_ATOMS))
        rows = []
        done = 0
        while done < F:
            n = min(self.max_batch, F - done)
            if self.L.vmaf_hip_extract_host_raw(self._h, ref[done:].ctypes.data, dis[done:].ctypes.data, bps, w * bps, w * h * bps, n,
                                                1 if (continue_sequence or done) else 0, out[done:].ctypes.data_as(_dp)):
                raise VmafHipError(_err(self.L))
            self._collect_metrics(rows, n)
            done += n
        self.metrics = np.concatenate(rows) if rows else np.zeros((0, N_METRICS))
        return out

    def extract_device_raw(self, d_ref, d_dis, nframes, bytes_per_sample, continue_sequence=False, out=None, sync=True):
        """Device-resident raw luma (addresses or torch uint8 / int16 tensors), dense rows."""
        pr = d_ref.data_ptr() if hasattr(d_ref, "data_ptr") else int(d_ref)
        pd = d_dis.data_ptr() if hasattr(d_dis, "data_ptr") else int(d_dis)
        rs = self.w * bytes_per_sample
        if out is None:
            out = np.zeros((nframes, N_ATOMS))
        fn = self.L.vmaf_hip_extract_device_raw if sync else self.L.vmaf_hip_extract_device_raw_async
        if fn(self._h, pr, pd, bytes_per_sample, rs, rs * self.h, nframes, 1 if continue_sequence else 0, out.ctypes.data_as(_dp)):
            raise VmafHipError(_err(self.L))
        self._keep = ([self._keep[-1]] if self._keep else []) + [(d_ref, d_dis, out)]   # buffers of the calls in flight
        return out

    def motion_seed_device_raw(self, d_ref_frame, bytes_per_sample):
        """Motion halo from one device-resident raw luma frame (address or torch tensor), dense rows."""
        p = d_ref_frame.data_ptr() if hasattr(d_ref_frame, "data_ptr") else int(d_ref_frame)
        if self.L.vmaf_hip_motion_seed_device_raw(self._h, p, bytes_per_sample, self.w * bytes_per_sample):
            raise VmafHipError(_err(self.L))

    # -- device pointers (torch tensors or raw ints) ----------------------------------------------
    def extract_device(self, d_ref, d_dis, nframes, row_stride_bytes=None, frame_stride_bytes=None,
                       continue_sequence=False, out=None, sync=True):
        """d_ref, d_dis: device addresses (int) or objects with data_ptr() (torch CUDA tensors)."""
        pr = d_ref.data_ptr() if hasattr(d_ref, "data_ptr") else int(d_ref)
        pd = d_dis.data_ptr() if hasattr(d_dis, "data_ptr") else int(d_dis)
        rs = row_stride_bytes or self.w * 4
        fs = frame_stride_bytes or rs * self.h
        if out is None:
            out = np.zeros((nframes, N_ATOMS))
        fn = self.L.vmaf_hip_extract_device if sync else self.L.vmaf_hip_extract_device_async
        if fn(self._h, pr, pd, rs, fs, nframes, 1 if continue_sequence else 0, out.ctypes.data_as(_dp)):
            raise VmafHipError(_err(self.L))
        self._keep = ([self._keep[-1]] if self._keep else []) + [(d_ref, d_dis, out)]   # buffers of the calls in flight
        return out

    def extract_host_async(self, ref, dis, out, continue_sequence=False):
        """One asynchronous call (at most two in flight) on host arrays [n,h,w]: float32 (the read_frame contract) or
        uint8 / uint16 raw luma.  The arrays and `out` [n,20] must stay alive and untouched until the call is
        synchronised (sync_oldest / sync); page-locked memory (host_alloc_array) makes the copies truly asynchronous."""
        n, h, w = ref.shape
        if (h, w) != (self.h, self.w) or dis.shape != ref.shape or ref.dtype != dis.dtype:
            raise VmafHipError(f"shape {ref.shape}/{dis.shape} does not match engine {self.h}x{self.w}")
        if not (ref.flags.c_contiguous and dis.flags.c_contiguous and out.flags.c_contiguous):
            raise VmafHipError("contiguous arrays needed")
        b = ref.dtype.itemsize
        if ref.dtype == np.float32:
            rc = self.L.vmaf_hip_extract_host_async(self._h, ref.ctypes.data, dis.ctypes.data, w * 4, w * h * 4, n,
                                                    1 if continue_sequence else 0, out.ctypes.data_as(_dp))
        elif ref.dtype in (np.uint8, np.uint16):
            rc = self.L.vmaf_hip_extract_host_raw_async(self._h, ref.ctypes.data, dis.ctypes.data, b, w * b, w * h * b, n,
                                                        1 if continue_sequence else 0, out.ctypes.data_as(_dp))
        else:
            raise VmafHipError(f"dtype {ref.dtype} is none of float32, uint8, uint16")
        if rc:
            raise VmafHipError(_err(self.L))
        self._keep = ([self._keep[-1]] if self._keep else []) + [(ref, dis, out)]

    def sync(self):
        if self.L.vmaf_hip_engine_sync(self._h):
            raise VmafHipError(_err(self.L))

    def debug_plane(self, kind, level, frame=0):
        """Stage-level test hook: kind 'vif_ref' | 'vif_dis' | 'adm_ref' | 'adm_dis', level 1..3 -> float32 [h,w]."""
        k = {"vif_ref": 0, "vif_dis": 1, "adm_ref": 2, "adm_dis": 3}[kind]
        buf = np.zeros(self.w * self.h, np.float32)
        w, h = C.c_int(), C.c_int()
        if self.L.vmaf_hip_engine_debug_read_plane(self._h, k, level, frame, buf.ctypes.data_as(_fp), buf.size, C.byref(w), C.byref(h)):
            raise VmafHipError(_err(self.L))
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()

    def debug_sums(self, frame=0):
        """Raw fp64 device sums of one frame of the last synchronised call (include/vmaf_hip_debug.h) -> [33]."""
        out = np.zeros(33)
        if self.L.vmaf_hip_engine_debug_read_sums(self._h, frame, out.ctypes.data_as(_dp)):
            raise VmafHipError(_err(self.L))
        return out

    def sync_oldest(self):
        """Finish the oldest of the (at most two) asynchronous extract calls in flight."""
        if self.L.vmaf_hip_engine_sync_oldest(self._h):
            raise VmafHipError(_err(self.L))

    def motion_seed(self, ref_frame):
        """Blur one ref frame (numpy [h,w] float32, or a device address / torch tensor) as the motion halo."""
        if isinstance(ref_frame, np.ndarray) and ref_frame.dtype in (np.uint8, np.uint16):
            a = np.ascontiguousarray(ref_frame)
            rc = self.L.vmaf_hip_motion_seed_host_raw(self._h, a.ctypes.data, a.dtype.itemsize, self.w * a.dtype.itemsize)
        elif hasattr(ref_frame, "data_ptr") or isinstance(ref_frame, int):
            p = ref_frame.data_ptr() if hasattr(ref_frame, "data_ptr") else ref_frame
            rc = self.L.vmaf_hip_motion_seed_device(self._h, p, self.w * 4)
        else:
            a = np.ascontiguousarray(ref_frame, np.float32)
            rc = self.L.vmaf_hip_motion_seed_host(self._h, a.ctypes.data, self.w * 4)
        if rc:
            raise VmafHipError(_err(self.L))

    def arm_timing(self):
        self.L.vmaf_hip_last_kernel_ms(self._h, 0)

    def last_kernel_ms(self, which=0):
        return self.L.vmaf_hip_last_kernel_ms(self._h, which)


NOALLOC_SCRATCH_BYTES = 32 << 20


def extract_device_noalloc(d_ref, d_dis, width, height, nframes, device=0, order="avx", stream=None,
                           row_stride_bytes=None, frame_stride_bytes=None):
    """vmaf_hip_extract_device_noalloc: atoms [nframes,20] of device-resident fp32 planes without any allocation on the
    library's side (static per-GPU scratch area; see include/vmaf_hip.h for its limit and the no-overlap rule)."""
    L = load_library()
    pr = d_ref.data_ptr() if hasattr(d_ref, "data_ptr") else int(d_ref)
    pd = d_dis.data_ptr() if hasattr(d_dis, "data_ptr") else int(d_dis)
    rs = row_stride_bytes or width * 4
    fs = frame_stride_bytes or rs * height
    out = np.zeros((nframes, N_ATOMS))
    if L.vmaf_hip_extract_device_noalloc(device, stream, pr, pd, width, height, rs, fs, nframes,
                                         {"avx": 1, "scalar": 0, "fma": 2}[order], out.ctypes.data_as(_dp)):
        raise VmafHipError(_err(L))
    return out


class PinnedArray:
    """A numpy array in page-locked host memory (vmaf_hip_host_alloc): H2D copies from it are asynchronous."""

    def __init__(self, shape, dtype):
        self.L = load_library()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = self.L.vmaf_hip_host_alloc(max(n, 1))
        if not self._p:
            raise VmafHipError(_err(self.L))
        self.array = np.ctypeslib.as_array((C.c_ubyte * n).from_address(self._p)).view(dtype).reshape(shape)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self.L.vmaf_hip_host_free(self._p)
            self._p = None

    __del__ = close


def svm_predict(sv, coef, gamma, rho, x, device=0):
    L = load_library()
    sv = np.ascontiguousarray(sv, np.float64)
    coef = np.ascontiguousarray(coef, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(len(x))
    if L.vmaf_hip_svm_predict(device, sv.ctypes.data_as(_dp), coef.ctypes.data_as(_dp), sv.shape[0], sv.shape[1],
                              gamma, rho, x.ctypes.data_as(_dp), len(x), out.ctypes.data_as(_dp)):
        raise VmafHipError(_err(L))
    return out


def model_feature_names(model_path=DEFAULT_MODEL):
    """The model's feature names ("VMAF_feature_adm2_score", ...) in model order, as the library's loader reads them."""
    L = load_library()
    n = C.c_int()
    names = C.create_string_buffer(1 << 16)
    err = C.create_string_buffer(1024)
    if L.vmaf_amd_model_feature_names(model_path.encode(), C.byref(n), names, len(names), err, 1024):
        raise VmafHipError(err.value.decode(errors="replace"))
    out = names.value.decode().split("\n") if n.value else []
    assert len(out) == n.value
    return out


def predict_from_atoms(atoms, model_path=DEFAULT_MODEL, disable_clip=False, enable_transform=False, device=0):
    """atoms [F,20] -> (vmaf [F], features [F,D] in the model's feature order; D from the library)."""
    L = load_library()
    atoms = np.ascontiguousarray(atoms, np.float64)
    F = len(atoms)
    D = len(model_feature_names(model_path))
    vm = np.zeros(F)
    feats = np.zeros((F, D))
    err = C.create_string_buffer(1024)
    if L.vmaf_amd_predict_from_atoms_v2(model_path.encode(), atoms.ctypes.data_as(_dp), F, int(disable_clip),
                                     int(enable_transform), device, vm.ctypes.data_as(_dp), feats.ctypes.data_as(_dp), D,
                                     err, 1024):
        raise VmafHipError(err.value.decode(errors="replace"))
    return vm, feats


def compute_vmaf_from_arrays(ref, dis, fmt="yuv420p", model_path=DEFAULT_MODEL, log_path=None, log_fmt=None,
                             disable_clip=False, enable_transform=False, phone_model=False, pool_method=None,
                             do_psnr=False, do_ssim=False, do_ms_ssim=False, disable_avx=False):
    """Drive compute_vmaf() through its read_frame callback from float32 luma arrays [F,h,w]
    (un-offset, 8-bit units) - the way an external C caller such as FFmpeg's libvmaf filter does."""
    L = load_library()
    ref = np.ascontiguousarray(ref, np.float32)
    dis = np.ascontiguousarray(dis, np.float32)
    F, h, w = ref.shape
    state = {"i": 0}

    def cb(rp, dp_, tp, stride, user):
        i = state["i"]
        if i >= F:
            return 2
        sp = stride // 4
        r = np.ctypeslib.as_array(rp, shape=(h, sp))
        d = np.ctypeslib.as_array(dp_, shape=(h, sp))
        r[:, :w] = ref[i]
        d[:, :w] = dis[i]
        state["i"] = i + 1
        return 0

    score = C.c_double()
    err = C.create_string_buffer(1024)
    enc = lambda s: s.encode() if s is not None else None  # noqa: E731
    rc = L.compute_vmaf_noexcept(C.byref(score), fmt.encode(), w, h, READ_FRAME(cb), None, model_path.encode(), enc(log_path),
                                 enc(log_fmt), int(disable_clip), int(disable_avx), int(enable_transform), int(phone_model), int(do_psnr),
                                 int(do_ssim), int(do_ms_ssim), enc(pool_method), err, 1024)
    if rc:
        raise VmafHipError(err.value.decode(errors="replace"))
    return score.value


def compute_vmaf_seekable_from_arrays(ref, dis, fmt="yuv420p", model_path=DEFAULT_MODEL, log_path=None, log_fmt=None,
                                      disable_clip=False, enable_transform=False, phone_model=False, pool_method=None,
                                      do_psnr=False, do_ssim=False, do_ms_ssim=False, disable_avx=False, fail_at=None):
    """Drive compute_vmaf_raw_seekable() from raw luma arrays [F,h,w] (uint8, or uint16 for the *10le formats): the
    positional callback is called from the library's shard and reader threads (VMAF_HIP_DEVICES, VMAF_HIP_READ_THREADS).
    fail_at = a frame index whose read reports failure (tests)."""
    L = load_library()
    ref, dis = np.ascontiguousarray(ref), np.ascontiguousarray(dis)
    F, h, w = ref.shape
    bps = ref.dtype.itemsize
    seen = []

    def cb(rp, dp_, stride, index, user):
        seen.append(index)
        if index < 0 or index >= F or stride != w * bps or index == fail_at:
            return 1
        C.memmove(rp, ref[index].ctypes.data, w * h * bps)
        C.memmove(dp_, dis[index].ctypes.data, w * h * bps)
        return 0

    score = C.c_double()
    err = C.create_string_buffer(1024)
    enc = lambda s: s.encode() if s is not None else None  # noqa: E731
    rc = L.compute_vmaf_raw_seekable_noexcept(C.byref(score), fmt.encode(), w, h, F, READ_FRAME_RAW_AT(cb), None, model_path.encode(),
                                              enc(log_path), enc(log_fmt), int(disable_clip), int(disable_avx), int(enable_transform),
                                              int(phone_model), int(do_psnr), int(do_ssim), int(do_ms_ssim), enc(pool_method), err, 1024)
    if rc:
        raise VmafHipError(err.value.decode(errors="replace"))
    return score.value, sorted(seen)
