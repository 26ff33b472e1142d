"""Model-file parsers (vmaf_amd/csrc/model.cpp) against malformed input, under AddressSanitizer + UBSan on the CPU.

The reference reads user-supplied model files with ptools and libsvm and answers anything unreadable with an exception
("Error loading model (.pkl)", wrapper/src/vmaf.cpp:88-93; "Error loading SVM model", vmaf.cpp:155-160 / svm.cpp:468-571).
The replacement parsers run inside callers' processes (FFmpeg), so the same inputs must end in a clean error here too:
every truncation of the shipped model files and a seeded set of byte flips / opcode injections."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL = os.path.join(ROOT, "model", "vmaf_v0.6.1.pkl")
EXE = "/tmp/vmaf_amd_model_check_asan_%d" % os.getuid()


@pytest.fixture(scope="module")
def checker():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "vmaf_amd", "csrc"), "model_check_asan", "MODEL_CHECK=" + EXE])
    return EXE


def _run(checker, paths):
    p = subprocess.run([checker] + paths, capture_output=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert p.returncode == 0, p.stderr[-3000:]              # a sanitizer report aborts with a non-zero status
    # one verdict line per file; the loader also prints the reference's "Input model at ... cannot be read" line
    lines = [l for l in p.stdout.decode(errors="replace").splitlines() if l.startswith(("ok ", "error "))]
    assert len(lines) == len(paths)
    return lines


def test_shipped_models_load(checker):
    models = [MODEL] + [os.path.join(ROOT, "tests", "golden", "models", n) for n in sorted(os.listdir(os.path.join(ROOT, "tests", "golden", "models")))
                        if n.endswith(".pkl")]
    lines = _run(checker, models)
    assert lines[0] == "ok 6 211 6"
    assert all(l.startswith("ok ") for l in lines), lines


def _variants(data, rng, n_flips):
    out = []
    step = max(1, len(data) // 60)
    for cut in list(range(0, len(data), step)) + [len(data) - 1]:
        out.append(data[:cut])                                # truncations
    for _ in range(n_flips):
        b = bytearray(data)
        for _ in range(rng.randint(1, 4)):
            i = rng.randrange(len(b))
            b[i] = rng.choice(b"(dlpgSFINsa.0123456789'\n-e:") if rng.random() < 0.7 else rng.randrange(256)
        out.append(bytes(b))
    for op in (b"s", b"a", b"d", b"l", b"(", b"g99\n", b"p\n", b"cos\nsystem\n", b"R", b"."):     # injected opcodes
        for _ in range(6):
            i = data.index(b"\n", rng.randrange(len(data) - 2)) + 1
            out.append(data[:i] + op + (b"" if op.endswith(b"\n") else b"\n") + data[i:])
    return out


def test_malformed_pickles_end_in_an_error_not_a_crash(checker, tmp_path):
    rng = random.Random(20240611)
    data = open(MODEL, "rb").read()
    paths = []
    for k, v in enumerate(_variants(data, rng, 150)):
        p = tmp_path / f"m{k}.pkl"
        p.write_bytes(v)
        shutil.copy(MODEL + ".model", str(p) + ".model")
        paths.append(str(p))
    lines = _run(checker, paths)
    n_err = sum(l.startswith("error ") for l in lines)
    assert n_err >= len(lines) // 3                           # most variants are unreadable; none crashes
    for l in lines:
        assert l.startswith("ok 6 211 6") or l.startswith("error "), l
    # GLOBAL / REDUCE (code execution in a real unpickler) are refused outright
    evil = tmp_path / "evil.pkl"
    evil.write_bytes(b"cos\nsystem\n(S'true'\ntR.")
    shutil.copy(MODEL + ".model", str(evil) + ".model")
    assert _run(checker, [str(evil)])[0].startswith("error Error loading model (.pkl)")


def test_malformed_svm_models_end_in_an_error_not_a_crash(checker, tmp_path):
    rng = random.Random(7)
    data = open(MODEL + ".model", "rb").read()
    paths = []
    for k, v in enumerate(_variants(data, rng, 100)):
        p = tmp_path / f"s{k}.pkl"
        shutil.copy(MODEL, p)
        (tmp_path / f"s{k}.pkl.model").write_bytes(v)
        paths.append(str(p))
    # a huge sparse index must not size the dense SV table
    p = tmp_path / "big.pkl"
    shutil.copy(MODEL, p)
    (tmp_path / "big.pkl.model").write_bytes(data.replace(b" 1:", b" 2000000000:", 1))
    paths.append(str(p))
    p = tmp_path / "nomodel.pkl"
    shutil.copy(MODEL, p)
    paths.append(str(p))
    lines = _run(checker, paths)
    for l in lines:
        assert l.startswith("ok 6 ") or l.startswith("error "), l
    assert lines[-2] == "error Error loading SVM model" and lines[-1] == "error Error loading SVM model"
