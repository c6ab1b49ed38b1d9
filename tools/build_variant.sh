#!/bin/bash
# Development helper: build a variant of libfusg.so with extra -D flags on the halo kernels only.
# usage: tools/build_variant.sh NAME -DFLAG...   ->  future_urban_scene_generation_amd/libfusg_NAME.so
set -e
cd "$(dirname "$0")/../future_urban_scene_generation_amd/csrc"
name=$1; shift
tmp=$(mktemp -d)
for f in conv_halo_128 conv_halo_64 conv_halo_32; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off "$@" -c $f.hip -o $tmp/$f.o 2>$tmp/$f.log &
done
wait
objs=$(ls *.o | grep -v conv_halo_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmp/conv_halo_*.o -o ../libfusg_$name.so
rm -rf $tmp
echo built libfusg_$name.so
