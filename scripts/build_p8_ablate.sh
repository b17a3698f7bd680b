#!/bin/bash
# Timing-only ablation builds of the 256x256 kernel's ping-pong K loop: scripts/probes/_bin/librcgan_abl<k>.so for k in "$@"
# (bit 1 = no LDS-DMA issue, 2 = no fragment reads, 4 = no MFMAs, 8 = no barriers, 16 = no pixel DMA; h<k>: the halo-patch kernel's H8_ABLATE bits; w<k>: the three-tap filter-gradient kernel's WG3_ABLATE bits).  Run with RCGAN_LIB_PATH=<that file> RCGAN_P8_PP=1.
set -e
cd "$(dirname "$0")/../robust-conditional-gan_amd/csrc"
OUT=../../scripts/probes/_bin
mkdir -p $OUT _obj_probe
for k in "$@"; do
  if [ "${k#n}" != "$k" ]; then      # n<k>: the nine-tap filter-gradient kernel (conv_wgrad9.hip, -DWG9_ABLATE: 1 = no LDS-DMA after the prologue, 2 = no MFMAs)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWG9_ABLATE=${k#n} -c conv_wgrad9.hip -o _obj_probe/conv_wgrad9_abl$k.o
    objs=$(ls _obj/*.o | grep -v "conv_wgrad9.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj_probe/conv_wgrad9_abl$k.o -o $OUT/librcgan_abl$k.so
  elif [ "${k#w}" != "$k" ]; then      # w<k>: the three-tap filter-gradient kernel (conv_mfma.hip, -DWG3_ABLATE)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWG3_ABLATE=${k#w} -c conv_mfma.hip -o _obj_probe/conv_mfma_abl$k.o
    objs=$(ls _obj/*.o | grep -v "conv_mfma.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj_probe/conv_mfma_abl$k.o -o $OUT/librcgan_abl$k.so
  elif [ "${k#h}" != "$k" ]; then      # h<k>: the halo-patch kernel (conv_mfma8h.hip, -DH8_ABLATE)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH8_ABLATE=${k#h} -c conv_mfma8h.hip -o _obj_probe/conv_mfma8h_abl$k.o
    objs=$(ls _obj/*.o | grep -v conv_mfma8h.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj_probe/conv_mfma8h_abl$k.o -o $OUT/librcgan_abl$k.so
  else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DP8_ABLATE=$k -c conv_mfma8.hip -o _obj_probe/conv_mfma8_abl$k.o
    objs=$(ls _obj/*.o | grep -v "conv_mfma8.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj_probe/conv_mfma8_abl$k.o -o $OUT/librcgan_abl$k.so
  fi
  echo built $OUT/librcgan_abl$k.so
done
