#!/bin/bash
# builds vers_amd/lib/variants/libvers_hip_<tag>.so with extra compiler flags for the index's translation units
# (vers_amd/csrc/ivf_*.hip; compile-time A/B: VERS_LIB_PATH selects the variant)
# usage: build_variant.sh <tag> <flags...>      (run where hipcc is: the container; the .so travels with the snapshot)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p vers_amd/lib/variants vers_amd/build/variants
python -m vers_amd.build > /dev/null
pids=()
for src in vers_amd/csrc/ivf_*.hip; do
  name=$(basename "$src" .hip)
  /opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wall -Wno-unused-function -c "$src" -o vers_amd/build/variants/${name}_$tag.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=$(ls vers_amd/build/*.o | grep -v "/ivf_")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o vers_amd/lib/variants/libvers_hip_$tag.so $objs vers_amd/build/variants/ivf_*_$tag.o
echo vers_amd/lib/variants/libvers_hip_$tag.so
