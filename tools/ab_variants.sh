#!/bin/bash
# A/B builds of one kernel file: tools/ab_variants.sh <file.hip> "<flags A>" "<flags B>" ... -> gpurun_out-free variant
# libraries spaln_amd/libspdp_hip.<i>.so (selected with SPDP_LIB=<path> by spaln_amd/engine.py), for timing runs on the GPU box
set -e
cd "$(dirname "$0")/../spaln_amd/csrc"
F=$1; shift
make >/dev/null
i=0
for fl in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $fl -c -o build/ab_$i.o $F
    objs=$(ls build/*.o | grep -v "build/ab_" | grep -v "build/${F%.hip}.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libspdp_hip.$i.so $objs build/ab_$i.o
    echo "variant $i: $fl"
    i=$((i+1))
done
