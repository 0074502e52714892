#!/bin/bash
# Build kernel variants of librt_amd.so (extra -D flags) into rustraytracer_amd/csrc/build/variants/<name>.so
# usage: tools/variants.sh name1 "-DFOO=1 -DBAR=2" name2 "..."      (here, before gpurun; select with RT_AMD_LIB)
cd "$(dirname "$0")/../rustraytracer_amd/csrc" || exit 1
make -s >/dev/null || exit 1
mkdir -p build/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function --offload-arch=gfx950 -munsafe-fp-atomics $flags -c abi.hip -o build/variants/$name.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/variants/$name.so build/variants/$name.o $(ls build/host/*.o build/bvh_build.o build/bvh_gpu.o build/env_dist.o) -Wl,-rpath,/opt/rocm/lib || exit 1
  echo "built $name ($flags)"
done
