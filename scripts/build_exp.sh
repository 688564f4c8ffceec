#!/bin/bash
# usage: scripts/build_exp.sh NAME [-DFLAG ...]  -> constriction_amd/lib/exp_NAME.so (experimental build for scripts/ab.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value "$@" constriction_amd/csrc/*.hip -o constriction_amd/lib/exp_$name.so
echo built exp_$name.so
