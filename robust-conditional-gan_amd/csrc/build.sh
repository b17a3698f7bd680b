#!/bin/bash
# Build librcgan_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
OUT=../librcgan_hip.so
SRCS="api.hip conv_direct.hip conv_small.hip conv_mfma.hip conv_mfma8.hip conv_image.hip elementwise.hip bn.hip sn.hip loss.hip"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p _obj
pids=()
for s in $SRCS; do
  o=_obj/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ conv_mfma.h -nt "$o" ] || [ mfma_util.h -nt "$o" ] || [ ../../include/rcgan_hip.h -nt "$o" ]; then
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC _obj/*.o -o $OUT
echo "built $OUT"
