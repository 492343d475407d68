#!/bin/bash
# builds phnet_amd/lib/exp_<name>.so: conv.hip recompiled with the given -D flags, every other object from the normal build
# usage: tests/tools/build_variant.sh name -DFLAG1 -DFLAG2 ...   (select it with PHNET_LIB=phnet_amd/lib/exp_<name>.so)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p phnet_amd/lib/obj_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function "$@" -c phnet_amd/csrc/conv.hip -o phnet_amd/lib/obj_exp/conv_$name.o
objs=$(ls phnet_amd/lib/obj/*.o | grep -v '/conv.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o phnet_amd/lib/exp_$name.so $objs phnet_amd/lib/obj_exp/conv_$name.o
echo built phnet_amd/lib/exp_$name.so
