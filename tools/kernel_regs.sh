#!/bin/bash
# VGPR / scratch use of the kernels of one HIP source (device-only compile; no GPU needed)
# usage: bash tools/kernel_regs.sh spconv.hip [name filter] [extra compiler flags]
set -e
src=$1; filt=${2:-.}; shift; shift || true
T=$(mktemp -d)
cd "$(dirname "$0")/../unscene3d_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on --cuda-device-only "$@" -c "$src" -o "$T/dev.co" 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$T/dev.co" --output="$T/dev.elf" --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$T/dev.elf" | grep -E "\.name:|\.vgpr_count|\.sgpr_count|private_segment_fixed_size|\.agpr_count|group_segment_fixed_size" | \
  awk '/\.name:/{name=$2} /agpr_count/{a=$3} /group_segment/{l=$2} /private_segment/{p=$2} /sgpr_count/{s=$2} /\.vgpr_count/{print name, "vgpr", $2, "sgpr", s, "scratch", p, "lds", l}' | grep -E "$filt" | c++filt | cut -c1-160
rm -rf "$T"
