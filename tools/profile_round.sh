#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_round.sh r01
# 1. bench.py default run (the judged command)                  -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command       -> gpurun_out/<tag>/stats/
# 3. FETCH_SIZE and WRITE_SIZE in separate --pmc passes         -> gpurun_out/<tag>/pmc_fetch, pmc_write
#    (short run: the counters are per launch, 2 steps are enough)
# 4. one-stream run: exclusive kernel durations + VALU/LDS counters  -> gpurun_out/<tag>/stats_serial, pmc_valu
# 5. bench.py --metrics, one stream, kernel trace + issue counters  -> gpurun_out/<tag>/stats_metrics, pmc_metrics
# tools/summarize_profiles.py then turns these into the tracked files under profiles/.
set -e
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 -u bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 > "$out/stats.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --steps 2 --warmup 1 > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --steps 2 --warmup 1 > "$out/pmc_write.log" 2>&1
# 4. VALU / LDS issue counters and exclusive kernel durations, one stream (VMAF_HIP_SERIAL=1)
export VMAF_HIP_SERIAL=1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_valu" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --steps 2 --warmup 1 > "$out/pmc_valu.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_serial" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --steps 60 --warmup 10 > "$out/stats_serial.log" 2>&1
# 5. the optional metrics (PSNR / SSIM / MS-SSIM) next to the features, one stream: exclusive durations of their kernels
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_metrics" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --metrics --steps 30 --warmup 10 > "$out/stats_metrics.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_metrics" -o vm -- python3 "$root/bench.py" --no-cpu-baseline --no-end-to-end --min-seconds 0 --metrics --steps 3 --warmup 2 > "$out/pmc_metrics.log" 2>&1
unset VMAF_HIP_SERIAL
cd "$root"
tail -1 "$out/bench.json"
