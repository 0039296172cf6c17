#!/bin/bash
# Type-check and link the plugin's backend class against the stand-in headers (SDL2/glm are not in
# this image) and build a tiny headless driver around it: tools/crt_bench.
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
OUT=$ROOT/build; mkdir -p $OUT
CORE=$ROOT/chameleonrt_amd/libcrt_hip_core.so
/opt/rocm/bin/hipcc -std=c++17 -O2 -fPIC -DCRT_HIP_STANDIN -D__HIP_PLATFORM_AMD__ -I$ROOT/include -I$HERE -I/opt/rocm/include \
    -x c++ $HERE/render_hip.cpp $ROOT/tools/crt_bench.cpp -x none -o $OUT/crt_bench \
    $CORE -L/opt/rocm/lib -lrccl -lamdhip64 -pthread -Wl,-rpath,$(dirname $CORE) -Wl,-rpath,/opt/rocm/lib
echo "shim ok: $OUT/crt_bench"
