#!/bin/bash
# Runs ON THE GPU BOX: one --pmc pass over the one-stream bench, per-kernel averages printed.
#   bash tools/pmc_kernel.sh <tag> "<counters>" [bench args]
set -e
tag=$1; ctrs=$2; shift 2 || true
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
export VMAF_HIP_SERIAL=1
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pmc" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --steps 2 --warmup 1 "$@" > "$out/pmc.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
    acc[m.group(1) if m else r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "rocclr" in k or "elementwise" in k: continue
    print("%-36s" % k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())), "n=%d" % len(next(iter(d.values()))))
PY
