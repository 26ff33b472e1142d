#!/bin/bash
# Runs ON THE GPU BOX: one short bench line + exclusive kernel durations on one stream.  bash tools/quick_perf.sh <tag> [bench args]
set -e
tag=${1:-quick}; shift || true
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline --no-end-to-end --min-seconds 0 --steps 100 "$@" > "$out/bench.json" 2> "$out/bench.err"
python3 -c "
import json,sys
l=json.loads(open('$out/bench.json').read().strip().split('\n')[-1])
print('value', l['value'], 'ms/step', l['ms_per_step'], 'launch_ms', l['roofline']['launch_ms'], 'fast', l.get('fast_mode',{}).get('value'))"
bash tools/profile_serial.sh "$tag" "$@"
