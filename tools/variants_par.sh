#!/bin/bash
# Like tools/variants.sh, but the variants (and the base library) compile side by side.
# usage: tools/variants_par.sh name1 "-DFOO=1" name2 "-DBAR=2" ...   -> rustraytracer_amd/csrc/build/variants/<name>.so
cd "$(dirname "$0")/../rustraytracer_amd/csrc" || exit 1
mkdir -p build/variants build/host
# host objects + generated f32 sources first (cheap), then abi.hip for the base and every variant in parallel
make -s build/f32/kernels.hip build/bvh_gpu.o $(for f in host/rr_host host/scenes host/procedural host/host_api bvh_build env_dist; do echo build/$f.o; done) >/dev/null || exit 1
(make -s >/dev/null 2>build/variants/base.log; echo "base rc=$?") &
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function --offload-arch=gfx950 -munsafe-fp-atomics"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $flags -c abi.hip -o build/variants/$name.o 2>build/variants/$name.log &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/variants/$name.so build/variants/$name.o build/host/*.o build/bvh_build.o build/bvh_gpu.o build/env_dist.o -Wl,-rpath,/opt/rocm/lib 2>>build/variants/$name.log;
    echo "built $name ($flags) rc=$?" ) &
done
wait
