#!/bin/bash
# Runs ON THE GPU BOX: exclusive durations of the optional-metric kernels (bench.py --metrics, one stream).  bash tools/profile_metrics.sh <tag>
set -e
tag=${1:-met}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
export VMAF_HIP_SERIAL=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_metrics" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --metrics --steps 6 > "$out/stats_metrics.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/stats_metrics/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd_rocclr_\w+)", r["Name"])
    print("  %-36s calls %4s avg %9.1f us total %9.1f us" % (m.group(1) if m else r["Name"][:36], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
