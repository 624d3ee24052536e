#!/bin/bash
# Developer aid: build ablation variants of libusc3d_hip.so into build/ablate/<name>.so
# usage: tools/build_ablate.sh name "-DUSC_ABLATE_A -DUSC_ABLATE_B" [source.hip]   (default source: spconv.hip;
#        e.g. tools/build_ablate.sh tritime "-DUSC_TRI_TIMING" ncut.hip); run with USC3D_LIB=build/ablate/<name>.so
set -e
cd "$(dirname "$0")/../unscene3d_amd/csrc"
mkdir -p ../../build/ablate
name=$1; flags=$2; src=${3:-spconv.hip}
make >/dev/null
obj=/tmp/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $flags -c $src -o $obj
others=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -o ../../build/ablate/$name.so
echo built build/ablate/$name.so
