#!/bin/bash
# The nine-tap filter-gradient kernel as built and with one part of its loop removed (scripts/build_p8_ablate.sh n1 n2 n3 first)
export WGRAD_SETS="${WGRAD_SETS:-generator 256-ch: 32x32 alone,critic 128-ch: 32x32 alone}"
echo "== as built"; python3 scripts/bench_wgrad_group.py 64 2>&1 | grep -v amdgpu.ids
for k in n1 n2 n3; do
  case $k in n1) d="no LDS-DMA after the prologue";; n2) d="no MFMAs";; n3) d="no LDS-DMA, no MFMAs (fragment reads + barriers)";; esac
  [ -f scripts/probes/_bin/librcgan_abl$k.so ] || continue
  echo "== $k: $d"; RCGAN_LIB_PATH=scripts/probes/_bin/librcgan_abl$k.so python3 scripts/bench_wgrad_group.py 64 2>&1 | grep -v amdgpu.ids
done
