#!/bin/bash
# Build ablation variants of the library (conv.hip compiled with -DP2PHD_ABL_*) into pix2pixhdaudiosr_amd/abl/ for
# tools/trunk_only.py / tools/layer_table.py runs with P2PHD_LIB=...  (what binds the gather-conv main loop: MFMA issue,
# LDS fragment reads or the LDS-DMA stream).  Usage: tools/ablate_gconv.sh NAME "-DFLAG ..." [NAME2 "-D..."]...
set -e
cd "$(dirname "$0")/../pix2pixhdaudiosr_amd/csrc"
make -j8 >/dev/null
mkdir -p ../abl build/abl
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function $flags -c conv.hip -o build/abl/conv_$name.o
  objs=$(ls build/*.o | grep -v '/conv.o')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/abl/conv_$name.o -o ../abl/libp2phd_$name.so
  echo built abl/libp2phd_$name.so
done
