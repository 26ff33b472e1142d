#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_round.sh (gpurun_out/<tag>/) into the tracked summaries:

    profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py` (verbatim)
    profiles/<tag>_hbm_traffic.csv    per kernel: FETCH_SIZE, WRITE_SIZE per launch (KB as reported) and the
                                      corrected bytes (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md "HBM")
    profiles/<tag>_bench.json         the bench line of the same build
    profiles/traffic.json             what bench.py reads for roofline.traffic (dominant kernel, bytes/launch)
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]


def counters(path, counter):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    bench = os.path.join(src, "bench.json")
    frames = 30
    if os.path.exists(bench):
        lines = [l for l in open(bench) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"{tag}_bench.json"), "w").write(lines[-1])
            frames = json.loads(lines[-1])["roofline"]["frames_per_launch"]
    fetch = counters(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = counters(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write), key=lambda k: -fetch.get(k, (0, 0))[0]):
        f, n = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        rows.append({"kernel": k, "launches_sampled": n, "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                     "read_bytes_corrected": int(2 * f * 1024), "write_bytes": int(w * 1024),
                     "hbm_bytes_per_launch": int((2 * f + w) * 1024)})
    if rows:
        with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w", newline="") as fh:
            wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
            wr.writeheader()
            wr.writerows(rows)
    met = glob.glob(os.path.join(src, "stats_metrics", "**", "*kernel_stats.csv"), recursive=True)
    if met:
        shutil.copy(met[0], os.path.join(dst, f"{tag}_kernel_stats_metrics.csv"))
    # the metric kernels launch at several plane sizes (MS-SSIM levels): one row per (kernel, grid), counters and the duration
    # of the same dispatches; issue per clock = wave-level VALU instructions / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)
    mfiles = glob.glob(os.path.join(src, "pmc_metrics", "**", "*counter_collection.csv"), recursive=True)
    if mfiles:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(mfiles[0])):
            k = short(r["Kernel_Name"])
            if not any(x in k for x in ("ssim", "lpf", "box", "psnr")):
                continue
            key = (k, int(r["Grid_Size"]))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[key]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        cols = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]
        with open(os.path.join(dst, f"{tag}_metrics_counters.csv"), "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(["kernel", "grid_threads", "launches", "us_under_counters"] + cols + ["valu_wave_insts_per_cycle_per_simd", "lds_conflict_frac"])
            for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["us"])):
                m = {c: sum(v[c]) / len(v[c]) if v[c] else 0.0 for c in cols}
                per_clk = m["SQ_INSTS_VALU"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8) if m["GRBM_GUI_ACTIVE"] else 0.0
                wr.writerow([k, g, len(v["SQ_INSTS_VALU"]), round(sum(v["us"]) / len(v["us"]), 1)] + [int(m[c]) for c in cols] +
                            [round(per_clk, 3), round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_WAVE_CYCLES"], 4) if m["SQ_WAVE_CYCLES"] else ""])
    ser = glob.glob(os.path.join(src, "stats_serial", "**", "*kernel_stats.csv"), recursive=True)
    dur = {}
    if ser:
        shutil.copy(ser[0], os.path.join(dst, f"{tag}_kernel_stats_serial.csv"))
        for r in csv.DictReader(open(ser[0])):
            dur[short(r["Name"])] = float(r["AverageNs"])
    names = ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES",
             "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
    cnt = {n: counters(os.path.join(src, "pmc_valu"), n) for n in names}
    kernels = sorted(set(k for n in names for k in cnt[n]), key=lambda k: -dur.get(k, 0))
    if kernels:
        with open(os.path.join(dst, f"{tag}_valu_counters.csv"), "w", newline="") as fh:
            wr = csv.writer(fh)
            # valu_issue_frac: wave-level VALU instructions x 2 cycles (a wave64 fp32 op occupies a SIMD for 2 cycles on
            # gfx950: 157 TFLOP/s = 1024 SIMDs x 32 lanes x 2 flop x 2.4 GHz) over the cycles the 1024 SIMDs offer
            # during the kernel's exclusive duration at the 2.4 GHz peak engine clock
            wr.writerow(["kernel", "exclusive_us"] + names + ["valu_issue_frac_at_2.4GHz"])
            for k in kernels:
                row = [cnt[n].get(k, (0.0, 0))[0] for n in names]
                d = dur.get(k)
                frac = round(row[0] * 2 / (1024 * d * 1e-9 * 2.4e9), 3) if d else ""
                wr.writerow([k, round(d / 1e3, 1) if d else ""] + [int(x) for x in row] + [frac])
    # the dominant kernel = VIF scale 0: vif_stat_wg_kernel<17, true, 1> at widths that are multiples of 240 (1080p, 4K),
    # else vif_stat_avx_kernel<17, true>
    # (third template argument 1 = the reference's AVX order, the default; 2 = the opt-in fused order of bench.py's fast_mode leg)
    # from round 3 on the kernels that read the picture carry its sample type: "unsigned char" = bench.py's default (raw
    # 8-bit luma resident in HBM), "float" = the fp32 planes of its f32_input leg
    input_kind = "raw"
    dom = [r for r in rows if r["kernel"] == "vif_stat_wg_kernel<17, true, 1, unsigned char>"]
    if not dom:
        input_kind = "f32"
        dom = [r for r in rows if r["kernel"] in ("vif_stat_wg_kernel<17, true, 1>", "vif_stat_wg_kernel<17, true, 1, float>")] or \
            [r for r in rows if r["kernel"].startswith("vif_stat_avx_kernel<17")]
    if dom:
        valu = sum(cnt["SQ_INSTS_VALU"].get(r["kernel"], (0.0, 0))[0] for r in dom)
        json.dump({"source": f"profiles/{tag}_hbm_traffic.csv, profiles/{tag}_valu_counters.csv",
                   "kernels": [r["kernel"] for r in dom], "frames_per_launch": frames, "width": 1920, "height": 1080, "input": input_kind,
                   "hbm_bytes_per_launch": sum(r["hbm_bytes_per_launch"] for r in dom),
                   "valu_wave_insts_per_launch": int(valu),
                   "exclusive_us": round(sum(dur.get(r["kernel"], 0.0) for r in dom) / 1e3, 1),
                   "note": "FETCH_SIZE (KB) x2 [gfx950 correction] + WRITE_SIZE (KB), separate --pmc passes; "
                           "SQ_INSTS_VALU = wave-level VALU instructions"},
                  open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
