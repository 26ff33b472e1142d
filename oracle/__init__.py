"""CPU oracle for the VMAF v0.6.1 hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker.  The product (vmaf_amd) must never import it.

  oracle.lib()      -> ctypes handle to oracle/liboracle.so  (this repo's C restatement)
  oracle.ref_lib()  -> ctypes handle to oracle/_ref/libvmaf_ref_feature.so (the reference's own
                       C sources compiled by oracle/Makefile) or None when it was not built
"""
from .binding import (lib, ref_lib, ref_exec_path, ORDER_SCALAR, ORDER_AVX, ATOM_NAMES,  # noqa: F401
                      conv_sep, vif, adm, sad, frame, clip_atoms, ref_conv_sep, ref_vif,
                      ref_adm, ref_frame, ref_clip_atoms, ref_svm_predict, csf_rfactors,
                      psnr, ssim, ms_ssim, ref_psnr, ref_ssim, ref_ms_ssim, ansnr, ref_ansnr, vif_pyramid, adm_ll_pyramid, vif_exact_sums)
from .fusion import (load_pkl_model, load_svm_model, svm_predict, vmaf_from_atoms,     # noqa: F401
                     derived_features)
