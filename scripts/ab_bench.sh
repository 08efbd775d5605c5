#!/bin/bash
# A/B of environment settings on one box: usage ab_bench.sh "VAR=1" "VAR=0 OTHER=2" ...  (each run: bench.py --steps 20, ms per step)
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-40s %.3f ms  %s' % ('$cfg', d['ms_per_step'], {k: round(v, 2) for k, v in d['phases_ms_per_step'].items()}))"
  done
done
