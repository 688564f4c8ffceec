#!/bin/bash
# Like exp_variants.sh, but every variant is generated and compiled in its own copy of csrc/, all of them in parallel.
# (a variable EXP_CFLAGS=-DSOMETHING in a variant's list is passed to hipcc)
# usage: scripts/exp_variants_par.sh <generator.py> <kernel.hip> name:ENV=1,ENV2=x ...   -> constriction_amd/lib/variants/<name>.so
set -e
cd "$(dirname "$0")/.."
gen=$1; src=$2; shift 2
mkdir -p constriction_amd/lib/variants build/exp
objs=$(ls build/obj/*.o | grep -v "/$(basename ${src%.hip}).o")
pids=()
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  (
    d=build/exp/src_$name; rm -rf $d; mkdir -p $d/constriction_amd $d/include
    cp -r constriction_amd/csrc $d/constriction_amd/csrc; cp include/*.h $d/include/
    for e in ${envs//,/ }; do export $e; done
    GEN_CSRC=$PWD/$d/constriction_amd/csrc python $gen >/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXP_CFLAGS -c $d/constriction_amd/csrc/$(basename $src) -o build/exp/$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/exp/$name.o -ldl -o constriction_amd/lib/variants/$name.so
    echo "built $name"
  ) &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done; exit $rc
