#!/bin/bash
# A/B builds of libnefii_hip.so with extra -D switches: tools/ab_build.sh NAME "-DFOO -DBAR" -> build_ab/libnefii_NAME.so
# (only nefii_tracer.hip is recompiled; select with NEFII_LIB_PATH=build_ab/libnefii_NAME.so).  build_ab/ is git-ignored.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build_ab
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
/opt/rocm/bin/hipcc $F "$@" -c nefii_amd/csrc/nefii_tracer.hip -o build_ab/tracer_$NAME.o 2>&1 | grep -v "packed-fp32-ops" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/libnefii_$NAME.so nefii_amd/csrc/nefii_mlp.o build_ab/tracer_$NAME.o nefii_amd/csrc/nefii_shading.o nefii_amd/csrc/nefii_probe.o
rm -f build_ab/tracer_$NAME.o
echo built build_ab/libnefii_$NAME.so
