#!/bin/bash
# A/B builds of the library: recompile ONE source with extra flags and link it with the other objects of the regular build
#   tools/build_variant.sh NAME SOURCE.hip "-DFLAG=.. ..."   ->  passion_amd/lib/libpassion_hip_NAME.so   (use: PASSION_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../passion_amd/csrc"
name=$1; src=$2; flags=$3
make -s -j8 ../lib/libpassion_hip.so
mkdir -p build/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I. -I../../include -Wno-unused-result -ffp-contract=off $flags -x hip -c $src -o build/var_$name/$src.o
objs=$(ls build/*.o | grep -v "build/$src.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/var_$name/$src.o -o ../lib/libpassion_hip_$name.so
echo "built passion_amd/lib/libpassion_hip_$name.so"
