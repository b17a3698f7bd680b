#!/bin/bash
# The three-tap filter-gradient kernel as built and with one part of its loop removed (scripts/build_p8_ablate.sh w1 w2 w4 w8 w3 w6 first)
export WGRAD_SETS="${WGRAD_SETS:-generator 256-ch: 32x32 alone,critic 128-ch: 32x32 alone,three-tap only}"
echo "== as built"; python3 scripts/bench_wgrad_group.py 64 2>&1 | grep -v amdgpu.ids
for k in w1 w2 w4 w8 w3 w6; do
  case $k in w1) d="no LDS-DMA after the prologue";; w2) d="no MFMAs";; w4) d="fragment reads of stage 0 only";; w8) d="no ReLU / edge masks";;
             w3) d="no LDS-DMA, no MFMAs (fragment reads + barriers)";; w6) d="no MFMAs, stage-0 reads (LDS-DMA + barriers)";; esac
  [ -f scripts/probes/_bin/librcgan_abl$k.so ] || continue
  echo "== $k: $d"; RCGAN_LIB_PATH=scripts/probes/_bin/librcgan_abl$k.so python3 scripts/bench_wgrad_group.py 64 2>&1 | grep -v amdgpu.ids
done
