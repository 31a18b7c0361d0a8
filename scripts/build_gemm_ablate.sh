#!/bin/bash
# builds vers_amd/lib/variants/libvers_hip_abl<bits>.so: the product library with kmeans.hip compiled with -DVERS_X3W_ABLATE=<bits>
# (gemm.hip.h: timing experiments on the assign contraction; results are WRONG with any bit set)
set -e
cd "$(dirname "$0")/.."
mkdir -p vers_amd/lib/variants vers_amd/build/variants
python -m vers_amd.build > /dev/null
pids=()
for b in "$@"; do
  /opt/rocm/bin/hipcc -DVERS_X3W_ABLATE=$b -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function -c vers_amd/csrc/kmeans.hip -o vers_amd/build/variants/kmeans_abl$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
for b in "$@"; do
  objs=$(ls vers_amd/build/*.o | grep -v "/kmeans.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o vers_amd/lib/variants/libvers_hip_abl$b.so $objs vers_amd/build/variants/kmeans_abl$b.o
done
ls vers_amd/lib/variants/
