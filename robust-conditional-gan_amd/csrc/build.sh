#!/bin/bash
# Build the C-ABI libraries for gfx950 in-tree (the .so files travel to the GPU box with the snapshot):
#   ../librcgan_hip.so      16-bit activations = bf16   (default)
#   ../librcgan_hip_f16.so  16-bit activations = fp16   (same sources, -DRCGAN_HALF_FP16=1)
# usage: build.sh [bf16|f16|all]   (default all)
set -e
cd "$(dirname "$0")"
SRCS="api.hip comm.hip conv_direct.hip conv_small.hip conv_mfma.hip conv_mfma8.hip conv_mfma8h.hip conv_wgrad9.hip conv_trunk.hip conv_rf.hip conv_image.hip elementwise.hip bn.hip sn.hip loss.hip"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
WHAT=${1:-all}

build_one() {   # $1 = object dir, $2 = output, $3 = extra flags
  mkdir -p "$1"
  pids=()
  for s in $SRCS; do
    o=$1/${s%.hip}.o
    stale=0
    if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ build.sh -nt "$o" ]; then stale=1; fi
    for h in *.h ../../include/*.h; do        # every header: an object is rebuilt when ANY of them is newer
      if [ "$stale" = 0 ] && [ "$h" -nt "$o" ]; then stale=1; fi
    done
    if [ "$stale" = 1 ]; then
      $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC $3 -c "$s" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]}"; do wait $p; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC $1/*.o -o "$2"
  echo "built $2"
}

if [ "$WHAT" = "bf16" ] || [ "$WHAT" = "all" ]; then build_one _obj ../librcgan_hip.so ""; fi
if [ "$WHAT" = "f16" ] || [ "$WHAT" = "all" ]; then build_one _obj_f16 ../librcgan_hip_f16.so "-DRCGAN_HALF_FP16=1"; fi
