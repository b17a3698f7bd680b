#!/bin/bash
# Builds librcgan_probe{1,2,3}.so: the product library with ONE leg of the forward MFMA conv kernel removed
# (1: no MFMA, 2: no LDS fragment reads, 3: no global->LDS loads) to see which leg bounds it.
# Results are wrong by construction; use with RCGAN_LIB_PATH=... python scripts/bench_conv.py
set -e
cd "$(dirname "$0")/../../robust-conditional-gan_amd/csrc"
mkdir -p _obj_probe
for p in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRCGAN_PROBE=$p -c conv_mfma.hip -o _obj_probe/conv_mfma_$p.o &
done
wait
for p in 1 2 3; do
  objs=$(ls _obj/*.o | grep -v conv_mfma.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj_probe/conv_mfma_$p.o -o ../../gpurun_probe_lib$p.so
done
echo done
