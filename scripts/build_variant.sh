#!/bin/bash
# Build a diagnostic variant of the library next to the default one.
# usage: scripts/build_variant.sh SUFFIX [-DNAME=VALUE ...]   ->  implementation_phd_lab_vision_amd/libr50hipSUFFIX.so
# (select it at run time with R50_LIB=<path>); an empty SUFFIX rebuilds the default library.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SUF="$1"; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" \
    -o "$ROOT/implementation_phd_lab_vision_amd/libr50hip${SUF}.so" "$ROOT/implementation_phd_lab_vision_amd/csrc/r50_abi.hip"
echo "built libr50hip${SUF}.so"
