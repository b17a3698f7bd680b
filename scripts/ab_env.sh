#!/bin/bash
# A/B of environment settings on ONE box: bash scripts/ab_env.sh "A=1" "RCGAN_X=0 RCGAN_Y=2" ...  -> ms per iteration, three interleaved rounds
run() { env $1 python bench.py --no-cpu-baseline --steps 30 --warmup 5 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2 3; do
  for s in "$@"; do echo "$s  $(run "$s")"; done
done
