#!/bin/bash
# A/B builds of the HIP library: scripts/build_variant.sh <name> "<extra hipcc flags>" -> okvis_amd/lib_variants/<name>/ (git-ignored; travels
# to the GPU box).  Select it with OKVIS_AMD_LIB_DIR=okvis_amd/lib_variants/<name>.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
V=okvis_amd/lib_variants/$name
mkdir -p $V/obj
C=okvis_amd/csrc
for u in ba_capi store_capi dist_capi fe_capi; do
  if [ $u = ba_capi ] || [ ! -f $V/obj/$u.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $C/$u.hip -o $V/obj/$u.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $V/obj/ba_capi.o $V/obj/store_capi.o $V/obj/dist_capi.o $V/obj/fe_capi.o -ldl -o $V/libokvis_amd_ba.so
g++ -std=c++17 -O2 -fPIC -shared $C/host/estimator.cpp $C/host/estimator_capi.cpp $C/host/replay.cpp $C/host/okvis_config.cpp -o $V/libokvis_amd_estimator.so -L$V -lokvis_amd_ba '-Wl,-rpath,$ORIGIN'
g++ -std=c++17 -O2 $C/host/replay_main.cpp -o $V/okvis_amd_replay -L$V -lokvis_amd_estimator -lokvis_amd_ba '-Wl,-rpath,$ORIGIN'
echo built $V
