#!/bin/bash
# Runs ON THE GPU BOX: per-kernel durations at small batches (does a batch whose planes fit the 256 MB Infinity Cache make
# the second and third reader of the input planes - ADM scale 0, motion - faster?).  bash tools/small_batch_profile.sh
set -e
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
for n in 6 12 30; do
  out=$root/gpurun_out/small_batch/n$n
  mkdir -p "$out"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --distinct $n --repeat 1 --steps 200 --warmup 10 > "$out/log.txt" 2>&1
  python3 - "$out" $n <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2])
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Name"])
    k = m.group(1) if m else r["Name"][:30]
    if any(x in k for x in ("<17, true, 1>", "adm_scale_kernel<true, false>", "motion_kernel<1>", "<9, false, 1>")):
        print("  n=%-3d %-36s avg %8.1f us = %6.2f us/frame" % (n, k, float(r["AverageNs"]) / 1e3, float(r["AverageNs"]) / 1e3 / n))
PY
done
