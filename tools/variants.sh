#!/bin/bash
# Build kernel variants of librt_amd.so (extra -D flags) into rustraytracer_amd/csrc/build/variants/<name>.so
# usage: tools/variants.sh name1 "-DFOO=1 -DBAR=2" name2 "..."      (here, before gpurun; select with RT_AMD_LIB)
cd "$(dirname "$0")/../rustraytracer_amd/csrc" || exit 1
mkdir -p build/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  make -s -j8 OUT=build/variants/$name.so BUILD=build/var_$name EXTRA="$flags" 2>&1 | grep -E "error" && exit 1
  rm -rf build/var_$name
  echo "built $name ($flags)"
done
