"""vmaf_amd - MI355X-native VMAF v0.6.1 feature extraction + fusion.

The product is the C-ABI shared library vmaf_amd/libvmaf_amd.so (hand-written gfx950 HIP kernels,
include/vmaf_hip.h, and the compute_vmaf() drop-in, include/libvmaf.h).  This package is the thin
Python host mirror over it: ctypes bindings (binding.py), the run_vmaf / run_vmaf_in_batch launchers
(cli.py) and the synthetic-clip generator used by tests and bench.py (synth.py).

There is no CPU fallback: importing works anywhere, but every compute call raises VmafHipError when
the library or a HIP device is missing.
"""
from .binding import (ATOM_NAMES, Engine, VmafHipError, compute_vmaf_from_arrays, device_count, load_library,  # noqa: F401
                      model_feature_names, predict_from_atoms, svm_predict, library_path)

__all__ = ["ATOM_NAMES", "Engine", "VmafHipError", "compute_vmaf_from_arrays", "device_count", "load_library",
           "model_feature_names", "predict_from_atoms", "svm_predict", "library_path"]
