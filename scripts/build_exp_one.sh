#!/bin/bash
# usage: scripts/build_exp_one.sh NAME FILE.hip [-DFLAG ...] -> constriction_amd/lib/exp_NAME.so: the current objects of build/obj with ONE
# translation unit recompiled under the flags (CSRC_DIR=<copy of csrc with other generated .inc files>: from there) (seconds instead of the minutes of scripts/build_exp.sh); for AB_LIB=... runs
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
stem=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-function "$@" -c ${CSRC_DIR:-constriction_amd/csrc}/$stem.hip -o /tmp/exp_${name}_$stem.o
objs=$(ls build/obj/*.o | grep -v "/$stem.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/exp_${name}_$stem.o -ldl -o constriction_amd/lib/exp_$name.so
echo built exp_$name.so
