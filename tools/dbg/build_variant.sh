#!/bin/bash
# Builds an A/B variant of the library: the device translation unit compiled with extra flags, linked with the product's other objects.
#   tools/dbg/build_variant.sh keepchain -DCFR_TEXT_KEEP_CHAIN=1      -> tools/dbg/libcfr_hip_keepchain.so   (tools/dbg/ab_keep.sh)
#   tools/dbg/build_variant.sh teambr    -DCFR_TEAM_BEST_BRANCHES=1   -> tools/dbg/libcfr_hip_teambr.so      (tools/dbg/ab_team_best.sh)
#   tools/dbg/build_variant.sh nt        -DCFR_GATHER_NT=1            -> tools/dbg/libcfr_hip_nt.so          (tools/dbg/ab_nt.sh, ab_reprobe.sh)
# The variants are never loaded by the product; the .so files are git-ignored.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT/centrifuger_amd/csrc
make > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result "$@" -c -o /tmp/cfr_device_$NAME.o cfr_device.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/tools/dbg/libcfr_hip_$NAME.so cfr_index.o cfr_tail.o cfr_dust.o cfr_capi.o /tmp/cfr_device_$NAME.o cfr_build.o cfr_build_sa.o -lpthread
ls -la $ROOT/tools/dbg/libcfr_hip_$NAME.so
