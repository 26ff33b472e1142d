#!/bin/bash
# Runs ON THE GPU BOX: exclusive (one-stream) kernel durations of bench.py for one conv order.
#   bash tools/profile_order.sh fma   -> gpurun_out/order_<order>/
set -e
order=${1:-fma}
root=$(pwd)
out=$root/gpurun_out/order_$order
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
export VMAF_HIP_SERIAL=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_serial" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --order "$order" --steps 10 > "$out/stats_serial.log" 2>&1
cd "$root"
