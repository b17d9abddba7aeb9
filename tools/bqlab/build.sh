#!/bin/bash
# Builds a lab program against the in-tree libpn2hip.so: tools/bqlab/build.sh lab1
set -e
here="$(cd "$(dirname "$0")" && pwd)"
repo="$(cd "$here/../.." && pwd)"
pkg="$repo/khairil_tum-facade_semantic_segmentation_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function \
  -I "$repo/include" -I "$pkg/csrc" -I "$here" "$here/$1.hip" -o "$here/$1.bin" -L "$pkg" -lpn2hip -Wl,-rpath,'$ORIGIN/../../khairil_tum-facade_semantic_segmentation_amd' "${@:2}"
echo "built $here/$1.bin"
