#!/bin/bash
# usage: build_variant.sh NAME "-DFLAG ..."  -> /tmp/oriana_variants/liboriana_NAME.so (analysis builds of the library;
# select one with ORIANA_HIP_LIB=...).  Never used by the package itself.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=/tmp/oriana_variants/$1
mkdir -p $O
for f in pack passes updates dense dense_mfma dense_f32 metrics stateless; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $2 -c $R/oriana_amd/csrc/$f.hip -o $O/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/oriana_variants/liboriana_$1.so $O/*.o
echo /tmp/oriana_variants/liboriana_$1.so
