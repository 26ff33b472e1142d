"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that include/*.h
declares, and fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

import vmaf_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    names = set()
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text):
        n = m.group(1)
        if n.startswith(("vmaf_hip_", "vmaf_amd_", "compute_vmaf")):
            names.add(n)
    return sorted(names)


@pytest.mark.parametrize("header", ["vmaf_hip.h", "vmaf_hip_debug.h", "libvmaf.h"])
def test_every_declared_symbol_is_exported(header):
    lib = ctypes.CDLL(vmaf_amd.library_path())
    names = declared_functions(header)
    assert len(names) >= 2
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/{header} but not exported"


def test_nothing_else_is_exported():
    """Built with -fvisibility=hidden: the dynamic symbol table holds the entry points of include/*.h and nothing of the
    kernel launchers or C++ internals (the __hip_cuid_* markers come from the HIP toolchain)."""
    out = subprocess.run(["nm", "-D", "--defined-only", vmaf_amd.library_path()], capture_output=True, text=True, check=True, timeout=300).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()[1] in "TDB"}
    declared = set(declared_functions("vmaf_hip.h")) | set(declared_functions("vmaf_hip_debug.h")) | set(declared_functions("libvmaf.h"))
    extra = {n for n in exported - declared if not n.startswith("__hip_cuid_")}
    assert not extra, sorted(extra)


def test_headers_compile_as_c():
    for header in ("vmaf_hip.h", "vmaf_hip_debug.h", "libvmaf.h"):
        src = f'#include "{header}"\nint main(void) {{ return 0; }}\n'
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-fsyntax-only"],
                       input=src.encode(), check=True, timeout=300)


def test_compute_vmaf_keeps_the_reference_signature():
    """wrapper/src/libvmaf.h:26 - argument list must stay identical for drop-in linking."""
    text = open(os.path.join(ROOT, "include", "libvmaf.h")).read()
    want = ("double compute_vmaf(char* fmt, int width, int height, int (*read_frame)(float *ref_data, float *main_data, "
            "float *temp_data, int stride_byte, void *user_data), void *user_data, char *model_path, char *log_path, "
            "char *log_fmt, int disable_clip, int disable_avx, int enable_transform, int phone_model, int do_psnr, "
            "int do_ssim, int do_ms_ssim, char *pool_method);")
    assert want in text


def test_no_silent_cpu_fallback():
    if vmaf_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(vmaf_amd.VmafHipError, match="no HIP device"):
        vmaf_amd.Engine(64, 48)
    import numpy as np
    with pytest.raises(vmaf_amd.VmafHipError, match="no HIP device"):
        vmaf_amd.svm_predict(np.zeros((2, 6)), np.zeros(2), 0.04, 0.0, np.zeros((1, 6)))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under vmaf_amd/ (or the C sources) may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vmaf_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M) or "liboracle" in text or "oracle/" in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_built_artefacts_are_present_and_fresh():
    """The library and the two executables are build products, not tracked files: the tests must never run a stale one.
    `make -q` answers whether anything is older than its sources."""
    for name in ("libvmaf_amd.so", "vmafossexec", "vmaf"):
        assert os.path.exists(os.path.join(ROOT, "vmaf_amd", name)), f"vmaf_amd/{name} is missing: run `make -C vmaf_amd/csrc` (or __graft_entry__.build())"
    q = subprocess.run(["make", "-q", "-C", os.path.join(ROOT, "vmaf_amd", "csrc"), "all"], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, "vmaf_amd/csrc has changed since the last build: run `make -C vmaf_amd/csrc`"


def test_vmafossexec_usage_and_exit_codes():
    exe = os.path.join(ROOT, "vmaf_amd", "vmafossexec")
    assert os.path.exists(exe), "vmaf_amd/vmafossexec is missing: run `make -C vmaf_amd/csrc`"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 255 and "Usage:" in p.stderr                     # main.cpp:148-152 returns -1
    p = subprocess.run([exe, "yuv420p", "0", "10", "a", "b", "m"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 255 and "Invalid frame resolution" in p.stderr    # main.cpp:161-166
    p = subprocess.run([exe, "yuv420p", "64", "48", "a", "b", "m", "--log-fmt", "csv"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 255 and "log_fmt must be xml or json" in p.stderr
    p = subprocess.run([exe, "yuv420p", "64", "48", "a", "b", "m", "--pool", "median"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 255 and "pool_method must be" in p.stderr
    p = subprocess.run([exe, "nv12", "64", "48", "a", "b", "m"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "unknown format" in p.stderr                # main.cpp:87-92
    p = subprocess.run([exe, "yuv420p", "64", "48", "/nonexistent_ref", "b", "m"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "fopen ref_path" in p.stderr
