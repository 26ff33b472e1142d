#!/bin/bash
# Runs ON THE GPU BOX: stall / wait counters of the serial bench, one pass.  bash tools/pmc_stall.sh <order>
set -e
order=${1:-avx}
root=$(pwd)
out=$root/gpurun_out/stall_$order
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
export VMAF_HIP_SERIAL=1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d "$out/pmc" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --order "$order" --steps 2 --warmup 1 > "$out/pmc.log" 2>&1
cd "$root"
