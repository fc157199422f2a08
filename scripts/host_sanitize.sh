#!/bin/bash
# The host side under sanitizers (SURVEY.md §5 "race detection"; the reference's races: pc_preprocessing_main.cpp:330-336 vs
# :134-157). Builds, in /tmp/cm_san (nothing lands in the tree):
#   * the oracle's C++ with -fsanitize=address,undefined and with -fsanitize=thread, and runs its self-check;
#   * libcloudmerge_hip.so with its HOST code (cm_api.cpp) under -fsanitize=thread (the gfx950 kernels are built as usual:
#     GPU-side sanitizers are not available on this pool), and host_tests + the node shell against it, also under TSan.
# Without a GPU (build container): host_tests' CPU part runs (config, PCD, PointCloud2 code; cm_create must fail loudly).
# With a GPU (gpurun -- bash scripts/host_sanitize.sh gpu): the whole host_tests run — subscriber thread against the loop
# thread, submits of growing clouds against cm_merged_copy / cm_ground_copy — under ThreadSanitizer.
# usage: scripts/host_sanitize.sh [gpu]
set -e
MODE=${1:-cpu}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/cm_san
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
# ---- oracle
for SAN in "address,undefined" "thread"; do
  g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fsanitize=$SAN -fno-omit-frame-pointer -pthread \
      -o $OUT/libcm_oracle_${SAN%%,*}.so oracle/cm_oracle.cpp
  g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=$SAN -fno-omit-frame-pointer -pthread -I oracle -o $OUT/oracle_driver_${SAN%%,*} \
      oracle/san_driver.cpp -L$OUT -lcm_oracle_${SAN%%,*} -Wl,-rpath,$OUT
  echo -n "oracle under -fsanitize=$SAN: "; $OUT/oracle_driver_${SAN%%,*}
done
# ---- HIP library: kernels as usual, the C-ABI host file under TSan
CSRC=cloud_merger_amd/csrc
FLAGS="--offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -I $CSRC"
for f in cm_kernels cm_kernels_v2 cm_kernels_v3 cm_kernels_v4 cm_kernels_ground; do
  /opt/rocm/bin/hipcc $FLAGS -c $CSRC/$f.hip -o $OUT/$f.o
done
/opt/rocm/bin/hipcc $FLAGS -fsanitize=thread -fno-omit-frame-pointer -c $CSRC/cm_api.cpp -o $OUT/cm_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=thread -o $OUT/libcloudmerge_hip.so $OUT/cm_kernels.o $OUT/cm_kernels_v2.o \
    $OUT/cm_kernels_v3.o $OUT/cm_kernels_v4.o $OUT/cm_kernels_ground.o $OUT/cm_api.o -Wl,-rpath,/opt/rocm/lib
echo "libcloudmerge_hip.so built with cm_api.cpp under -fsanitize=thread"
HOST=cloud_merger_amd/host
/opt/rocm/llvm/bin/clang++ -O1 -g -std=c++17 -pthread -fsanitize=thread -fno-omit-frame-pointer -ffp-contract=off \
    -o $OUT/host_tests_tsan $HOST/host_tests.cpp $HOST/merger_node.cpp $HOST/pcd_io.cpp -L$OUT -lcloudmerge_hip \
    -Wl,-rpath,$OUT -Wl,-rpath,/opt/rocm/lib
cat > $OUT/tsan.supp <<SUPP
# the ROCm runtime is not instrumented: its own worker threads and signal handling are not ours to judge
called_from_lib:libamdhip64.so
called_from_lib:libhsa-runtime64.so
race:libamdhip64.so
race:libhsa-runtime64.so
SUPP
export TSAN_OPTIONS="suppressions=$OUT/tsan.supp halt_on_error=0 exitcode=66 report_signal_unsafe=0"
mkdir -p $OUT/tmp
if [ "$MODE" = gpu ]; then $OUT/host_tests_tsan $OUT/tmp gpu; else $OUT/host_tests_tsan $OUT/tmp; fi
echo "host_tests under ThreadSanitizer ($MODE): rc $?"
