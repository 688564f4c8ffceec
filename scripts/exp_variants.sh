#!/bin/bash
# Timing experiments on a generated loop: builds constriction_amd/lib/variants/<name>.so for each "name:ENV=1,..." argument
# (results of such variants are wrong -- they only show what a wait or a block of instructions costs).
# usage: scripts/exp_variants.sh <generator.py> <kernel.hip> base: nolgkm:GEN_NO_LGKM=1 ...
set -e
cd "$(dirname "$0")/.."
gen=$1; src=$2; shift 2
mkdir -p constriction_amd/lib/variants build/exp
objs=$(ls build/obj/*.o | grep -v "/$(basename ${src%.hip}).o")
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( for e in ${envs//,/ }; do export $e; done; python $gen >/dev/null )
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c $src -o build/exp/$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/exp/$name.o -ldl -o constriction_amd/lib/variants/$name.so
  echo "built $name"
done
python $gen >/dev/null
