#!/bin/bash
# usage: build_variant_file.sh NAME FILE.hip "-DFLAG ..."  -> scratch/variants/liboriana_NAME.so: the library with csrc/FILE.hip
# rebuilt with the given flags (analysis builds; select one with ORIANA_HIP_LIB=...).  Never used by the package itself.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/scratch/variants
mkdir -p $O
B=$(basename $2 .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $3 -c $R/oriana_amd/csrc/$2 -o $O/${B}_$1.o
OBJS=$(ls $R/oriana_amd/csrc/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/liboriana_$1.so $OBJS $O/${B}_$1.o
rm -f $O/${B}_$1.o
echo $O/liboriana_$1.so
