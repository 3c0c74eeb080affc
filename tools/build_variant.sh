#!/bin/bash
# A/B builds of the same ABI: tools/build_variant.sh <name> <extra hipcc flags...>  ->  build/variants/lib<name>.so  (use with CAVIAR_HIP_LIB=...)
name=$1; shift
mkdir -p build/variants/obj_$name
for f in circuitmap_amd/csrc/*.hip circuitmap_amd/csrc/*.cpp; do
  b=$(basename ${f%.*})
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result "$@" -c $f -o build/variants/obj_$name/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib$name.so build/variants/obj_$name/*.o && echo built build/variants/lib$name.so
