#!/bin/bash
# One-box sweep of launch-policy knobs (environment defaults of the kernels' launchers): ms per iteration, two rounds.
#   bash scripts/sweep_knobs.sh "A=1" "RCGAN_X=..." ...
run() { env $1 python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2; do
  for s in "$@"; do echo "$s  $(run "$s")"; done
done
