#!/usr/bin/env bash
# A/B builds of the library with compile-time switches of one translation unit:
#   scripts/build_variant.sh <name> <file.hip> <-Dflags...>   ->  build_exp/lib_<name>.so  (select with ANNLITE_HIP_LIB)
set -eu
cd "$(dirname "$0")/../annlite_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../build_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -c $src -o ../../build_exp/${name}_${src%.hip}.o
objs=""
for f in capi scan scan_qfilter scan_q8 scan_prep seed_mfma graph graph_build ivf lut codec; do
  if [ "$f.hip" = "$src" ]; then objs="$objs ../../build_exp/${name}_$f.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../build_exp/lib_$name.so
echo built build_exp/lib_$name.so
