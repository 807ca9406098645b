#!/usr/bin/env bash
# A/B build of the WHOLE library with compile-time switches:  scripts/build_variant_all.sh <name> <-Dflags...>  ->  build_exp/lib_<name>.so
set -eu
cd "$(dirname "$0")/../annlite_amd/csrc"
name=$1; shift
mkdir -p ../../build_exp/$name
objs=""
for f in capi scan scan_qfilter scan_q8 scan_prep seed_mfma graph ivf lut codec; do
  extra=""; case $f in scan_q8|scan_qfilter|scan_prep) extra="-mllvm -amdgpu-atomic-optimizer-strategy=None";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=off $extra "$@" -c $f.hip -o ../../build_exp/$name/$f.o &
  objs="$objs ../../build_exp/$name/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../build_exp/lib_$name.so
echo built build_exp/lib_$name.so
