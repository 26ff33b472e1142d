"""numpy restatement of the fusion half of the path: derived features, linear rescale, nu-SVR RBF
prediction, de-normalisation, score transform, clip, pooling.  TEST INFRASTRUCTURE ONLY.

Follows wrapper/src/vmaf.cpp:351-529 (derived features .. clip), wrapper/src/svm.cpp:170-247
(RBF k_function, svm_predict_values) and wrapper/src/vmaf.h:59-88 (pooling).
The model files are parsed by a tiny protocol-0 reader that executes nothing from the file.
"""
import math
import re

import numpy as np


def load_pkl_model(path):
    """Parse the pickle protocol-0 text of a VMAF model (model/vmaf_v0.6.1.pkl:1-108) without unpickling.
    Supports exactly the opcodes those files use: ( d l p g S F I N s a . and V."""
    with open(path, "r") as f:
        lines = f.read().split("\n")
    stack, memo, marks = [], {}, []
    i = 0

    def strlit(t):
        t = t.strip()
        if t and t[0] in "'\"" and t[-1] == t[0]:
            return t[1:-1]
        return t

    while i < len(lines):
        ln = lines[i]
        i += 1
        pos = 0
        while pos < len(ln):
            op = ln[pos]
            pos += 1
            if op == "(":
                marks.append(len(stack))
            elif op == "d":
                m = marks.pop(); items = stack[m:]; del stack[m:]
                stack.append({items[k]: items[k + 1] for k in range(0, len(items), 2)})
            elif op == "l":
                m = marks.pop(); items = stack[m:]; del stack[m:]
                stack.append(list(items))
            elif op == "p":
                memo[int(ln[pos:])] = stack[-1]; pos = len(ln)
            elif op == "g":
                stack.append(memo[int(ln[pos:])]); pos = len(ln)
            elif op in "SV":
                stack.append(strlit(ln[pos:])); pos = len(ln)
            elif op == "F":
                stack.append(float(ln[pos:])); pos = len(ln)
            elif op == "I":
                t = ln[pos:]; pos = len(ln)
                stack.append(True if t == "01" else False if t == "00" else int(t))
            elif op == "N":
                stack.append(None)
            elif op == "s":
                v = stack.pop(); k = stack.pop(); stack[-1][k] = v
            elif op == "a":
                v = stack.pop(); stack[-1].append(v)
            elif op == ".":
                return stack[-1]
            else:
                raise ValueError(f"unsupported pickle opcode {op!r} in {path}")
    raise ValueError("pickle stream ended without STOP")


def load_svm_model(path):
    """libsvm text model (wrapper/src/svm.cpp:359-571) -> dict(gamma, rho, coef[l], sv[l,dim] dense)."""
    hdr = {}
    rows = []
    with open(path) as f:
        in_sv = False
        for ln in f:
            if not in_sv:
                parts = ln.split()
                if parts and parts[0] == "SV":
                    in_sv = True
                elif parts:
                    hdr[parts[0]] = parts[1:]
            elif ln.strip():
                rows.append(ln.split())
    assert hdr["svm_type"] == ["nu_svr"] and hdr["kernel_type"] == ["rbf"]
    dim = max(int(t.split(":")[0]) for r in rows for t in r[1:])
    sv = np.zeros((len(rows), dim))
    coef = np.zeros(len(rows))
    for k, r in enumerate(rows):
        coef[k] = float(r[0])
        for t in r[1:]:
            idx, val = t.split(":")
            sv[k, int(idx) - 1] = float(val)       # omitted index == 0 (sparse list, svm.cpp:182-216)
    assert len(rows) == int(hdr["total_sv"][0])
    return dict(gamma=float(hdr["gamma"][0]), rho=float(hdr["rho"][0]), coef=coef, sv=sv)


def svm_predict(svm, x):
    """svm.cpp:229-247: sum_i coef_i * exp(-gamma * |x - sv_i|^2) - rho, accumulated in SV order."""
    x = np.asarray(x, np.float64)
    out = np.zeros(len(x))
    for f in range(len(x)):
        s = 0.0
        for i in range(len(svm["coef"])):
            d2 = 0.0
            for j in range(svm["sv"].shape[1]):
                d = x[f, j] - svm["sv"][i, j]
                d2 += d * d
            s += svm["coef"][i] * math.exp(-svm["gamma"] * d2)
        out[f] = s - svm["rho"]
    return out


_FEATS = ["adm2", "adm_scale0", "adm_scale1", "adm_scale2", "adm_scale3", "motion", "vif_scale0",
          "vif_scale1", "vif_scale2", "vif_scale3", "vif", "motion2"]


def derived_features(atoms):
    """vmaf.cpp:356-379. atoms [F,20] in combo order -> dict of per-frame vectors."""
    a = np.asarray(atoms, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        d = {"adm2": a[:, 0] / a[:, 1], "motion": a[:, 10].copy(), "vif": a[:, 19].copy()}
        for s in range(4):
            d[f"adm_scale{s}"] = a[:, 2 + 2 * s] / a[:, 3 + 2 * s]
            d[f"vif_scale{s}"] = a[:, 11 + 2 * s] / a[:, 12 + 2 * s]
    m = a[:, 10]
    m2 = m.copy()
    m2[:-1] = np.minimum(m[:-1], m[1:])
    d["motion2"] = m2
    return d


def vmaf_from_atoms(atoms, pkl, svm, enable_transform=False, disable_clip=False):
    """vmaf.cpp:391-559 -> (per-frame vmaf, derived-feature dict)."""
    md = pkl["model_dict"]
    assert md["model_type"] == "LIBSVMNUSVR"
    feats = derived_features(atoms)
    names = md["feature_names"]
    F = len(atoms)
    x = np.zeros((F, len(names)))
    for j, nm in enumerate(names):
        m = re.fullmatch(r"VMAF_feature_(.+)_score", nm)
        if not m or m.group(1) not in _FEATS:
            raise ValueError("Unknown feature name")
        v = feats[m.group(1)]
        if md["norm_type"] == "linear_rescale":
            v = md["slopes"][j + 1] * v + md["intercepts"][j + 1]
        x[:, j] = v
    y = svm_predict(svm, x)
    if md["norm_type"] == "linear_rescale":
        y = (y - md["intercepts"][0]) / md["slopes"][0]
    st = md.get("score_transform")
    if enable_transform and st is not None:
        v = np.zeros_like(y)
        if st.get("p0") is not None:
            v = v + st["p0"]
        if st.get("p1") is not None:
            v = v + st["p1"] * y
        if st.get("p2") is not None:
            v = v + st["p2"] * y * y
        if st.get("out_lte_in") == "true":
            v = np.where(v > y, y, v)
        if st.get("out_gte_in") == "true":
            v = np.where(v < y, y, v)
        y = v
    clip = md.get("score_clip")
    if not disable_clip and clip is not None:
        y = np.where(y < clip[0], clip[0], np.where(y > clip[1], clip[1], y))
    return y, feats


def pool(v, method=None):
    """vmaf.h:59-88"""
    v = np.asarray(v, np.float64)
    if method == "min":
        return float(v.min())
    if method == "harmonic_mean":
        return float(1.0 / np.mean(1.0 / (v + 1.0)) - 1.0)
    s = 0.0
    for e in v:
        s += e
    return s / len(v)
