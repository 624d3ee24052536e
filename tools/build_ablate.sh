#!/bin/bash
# Developer aid: build ablation variants of libusc3d_hip.so into build/ablate/<name>.so
# usage: tools/build_ablate.sh name "-DUSC_ABLATE_A -DUSC_ABLATE_B"
set -e
cd "$(dirname "$0")/../unscene3d_amd/csrc"
mkdir -p ../../build/ablate
name=$1; flags=$2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $flags -c spconv.hip -o /tmp/spconv_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC coords.o /tmp/spconv_$name.o spconv_sorted.o rows.o points.o ncut.o misc.o decoder.o attention.o -o ../../build/ablate/$name.so
echo built build/ablate/$name.so
