#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: the bench line, the per-kernel trace and the
# PMC passes that profiles/<tag>_* are copied from.   usage: bash tools/final_profile.sh r02_final [C2]
set -e -o pipefail
TAG=${1:-r03_final}
WL=${2:-C2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
BENCH="python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --skip-optimizer --skip-stage-times"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 200 > "$OUT/stats.log" 2>&1
echo stats done
pmc() {  # name, counters...
    local name=$1; shift
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH --steps 5 --warmup 2 > "$OUT/$name.log" 2>&1
    echo "$name done"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pmc sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES
cd "$ROOT"
python3 tools/pmc.py "$OUT/sq1" "$OUT/sq2" > "$OUT/pmc_sq_summary.txt" || true
python3 tools/traffic.py "$OUT" "$OUT/counters.json" "$TAG" "$WL"
# the bench line last, with the counters of THIS build in place so that roofline.traffic is filled in
mkdir -p "$ROOT/profiles" && cp "$OUT/counters.json" "$ROOT/profiles/${TAG}_counters.json"
timeout -k 10 400 python3 bench.py --workload $WL > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json" | cut -c1-300
python3 tools/kstats.py "$OUT/stats" 220 "$OUT/bench.json" > "$OUT/kernel_summary.txt"
tail -3 "$OUT/kernel_summary.txt"
