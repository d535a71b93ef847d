#!/bin/bash
# A/B of experiment knobs on the GPU box: each line = one bench run (ms_per_step of the C2 step)
run() { echo -n "$* : "; env "$@" python3 bench.py --no-cpu-baseline --skip-optimizer --steps 400 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['gpu_ms_per_step_hip_events'])"; }
for v in "$@"; do run $v; done
