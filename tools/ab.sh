#!/bin/bash
# A/B of library builds on the GPU box: each argument = one library file (or "base" for the in-tree
# one); prints ms_per_step and the per-stage times of the C2 step
for v in "$@"; do
  if [ "$v" = base ]; then unset EXPLAINN_HIP_LIB; else export EXPLAINN_HIP_LIB="$PWD/$v"; fi
  echo -n "$v : "
  python3 bench.py --no-cpu-baseline --skip-optimizer --steps 400 --warmup 30 2>&1 | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline'].get('kernels') or {}
print(d['ms_per_step'], ' '.join('%s %.1f' % (n, v['us']) for n, v in sorted(k.items())))"
done
