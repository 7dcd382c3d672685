#!/bin/bash
# Host side under AddressSanitizer + UndefinedBehaviorSanitizer, no GPU needed: the C-ABI library's host code (index build, window
# container, argument checks; --cuda-host-only: no device code, a zeroed stand-in for the fat binary) and the estimator library
# are compiled with clang's sanitizers into $OUT (default /tmp/okvis_amd_san) and the CPU test-suite runs against them.
# The product build is not touched: the tests pick the libraries up through OKVIS_AMD_LIB_DIR.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/okvis_amd_san}
mkdir -p $OUT
CLANG=/opt/rocm/lib/llvm/bin/clang++
HIPCC=/opt/rocm/bin/hipcc
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
for f in ba_capi store_capi dist_capi fe_capi; do
  $HIPCC --offload-arch=gfx950 --cuda-host-only $SAN -std=c++17 -fPIC -w -c $R/okvis_amd/csrc/$f.hip -o $OUT/$f.o
done
: > $OUT/stub.c
for f in ba_capi store_capi dist_capi fe_capi; do
  for sym in $(nm $OUT/$f.o | awk '/ U __hip_fatbin/ {print $2}'); do
    echo "const char $sym[4096] __attribute__((aligned(4096))) = {0};" >> $OUT/stub.c
  done
done
gcc -c -fPIC $OUT/stub.c -o $OUT/stub.o
$HIPCC $SAN -shared -fPIC $OUT/ba_capi.o $OUT/store_capi.o $OUT/dist_capi.o $OUT/fe_capi.o $OUT/stub.o -ldl -o $OUT/libokvis_amd_ba.so
$CLANG $SAN -std=c++17 -fPIC -shared -w $R/okvis_amd/csrc/host/estimator.cpp $R/okvis_amd/csrc/host/estimator_capi.cpp \
  $R/okvis_amd/csrc/host/replay.cpp -o $OUT/libokvis_amd_estimator.so -L$OUT -lokvis_amd_ba -Wl,-rpath,'$ORIGIN'
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $R
OKVIS_AMD_LIB_DIR=$OUT LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:log_path=$OUT/asan UBSAN_OPTIONS=print_stacktrace=0:log_path=$OUT/ubsan \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
