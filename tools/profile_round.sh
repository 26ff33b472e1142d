#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_round.sh r01
# 1. bench.py default run (the judged command)                  -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command       -> gpurun_out/<tag>/stats/
# 3. FETCH_SIZE and WRITE_SIZE in separate --pmc passes         -> gpurun_out/<tag>/pmc_fetch, pmc_write
#    (short run: the counters are per launch, 2 steps are enough)
# tools/summarize_profiles.py then turns these into the tracked files under profiles/.
set -e
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 -u bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o vm -- python3 "$root/bench.py" --no-cpu-baseline > "$out/stats.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --steps 2 --warmup 1 > "$out/pmc_write.log" 2>&1
cd "$root"
tail -1 "$out/bench.json"
