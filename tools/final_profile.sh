#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: the bench line, the per-kernel trace and the
# two PMC passes that profiles/<tag>_* are copied from.   usage: bash tools/final_profile.sh r01_final
set -e -o pipefail
TAG=${1:-r01_final}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json" | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --skip-optimizer --steps 200 > "$OUT/stats.log" 2>&1
echo stats done
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" --no-cpu-baseline --skip-optimizer --steps 5 --warmup 2 > "$OUT/fetch.log" 2>&1
echo fetch done
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" --no-cpu-baseline --skip-optimizer --steps 5 --warmup 2 > "$OUT/write.log" 2>&1
echo write done
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" --no-cpu-baseline --skip-optimizer --steps 5 --warmup 2 > "$OUT/sq.log" 2>&1
echo sq done
cd "$ROOT"
python3 tools/pmc.py "$OUT/sq" > "$OUT/pmc_sq_summary.txt" || true
python3 tools/traffic.py "$OUT/fetch" "$OUT/write" "$OUT/traffic.json" "$TAG"
python3 tools/kstats.py "$OUT/stats" 220 > "$OUT/kernel_summary.txt"
tail -3 "$OUT/kernel_summary.txt"
