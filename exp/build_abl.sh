#!/bin/bash
# experiment builds of the library: exp/build_abl.sh <name> <file.hip> "<-D flags>"  ->  exp/libs/lib_<name>.so
# (recompiles ONE kernel file with the flags and relinks it with the product's other objects)
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; file=$2; flags=$3
mkdir -p $R/exp/libs /tmp/abl_$name
cd $R/het_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DHET_GIT_SHA=\"abl-$name\" -Wno-unused-result $flags -c $file -o /tmp/abl_$name/obj.o
objs=$(ls build/*.o | grep -v "build/${file%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/exp/libs/lib_$name.so $objs /tmp/abl_$name/obj.o
echo built exp/libs/lib_$name.so
