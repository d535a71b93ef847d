#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: the secondary measurements profiles/<tag>_*
# quote beside tools/final_profile.sh's: other workloads, N bases, the one-rank all-reduce rehearsal,
# eval / trainer / small-batch probes, in-kernel stamps, the matrix-pipe probe.
#   usage: bash tools/extra_profiles.sh r03_final
TAG=${1:-r03_final}
OUT=gpurun_out/${TAG}_extra
mkdir -p "$OUT"
B="python3 bench.py --no-cpu-baseline"
for wl in C3 C4 C5; do timeout -k 10 300 $B --workload $wl --steps 100 --warmup 10 2> /dev/null | tail -1 > "$OUT/bench_$wl.json"; echo "$wl done"; done
timeout -k 10 300 $B --workload C4 --scaling strong --steps 30 --warmup 5 2> /dev/null | tail -1 > "$OUT/bench_C4_strong.json"; echo strong done
timeout -k 10 300 $B --n-frac 0.02 2> /dev/null | tail -1 > "$OUT/bench_nfrac.json"; echo nfrac done
EXPLAINN_BENCH_FORCE_SYNC=1 timeout -k 10 300 $B 2> /dev/null | tail -1 > "$OUT/bench_sync1.json"; echo sync1 done
timeout -k 10 300 python3 tools/eval_probe.py 2>&1 | grep -v "^W20\|amdgpu.ids" > "$OUT/eval_probe.txt"; echo eval done
timeout -k 10 300 python3 tools/trainer_probe.py 2>&1 | grep -v "^W20\|amdgpu.ids" > "$OUT/trainer_probe.txt"; echo trainer done
timeout -k 10 300 python3 tools/c1_probe.py 2>&1 | grep -v "^W20\|amdgpu.ids" > "$OUT/small_batch_probe.txt"; echo small done
[ -x tools/_bin/stampbench ] && timeout -k 10 120 tools/_bin/stampbench > "$OUT/stampbench.txt" 2>&1; echo stamps done
[ -x tools/_bin/cpm_stamp ] && timeout -k 10 120 tools/_bin/cpm_stamp > "$OUT/cpm_stamp.txt" 2>&1; echo cpm done
[ -x tools/_bin/mfma_probe ] && timeout -k 10 120 tools/_bin/mfma_probe > "$OUT/mfma_probe.txt" 2>&1; echo mfma done
for f in "$OUT"/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
