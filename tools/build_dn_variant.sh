#!/bin/bash
# usage: build_dn_variant.sh NAME "-DFLAG ..."  -> scratch/variants/liboriana_NAME.so: the library with csrc/dense_pass.hip
# rebuilt with the given flags (analysis builds; select one with ORIANA_HIP_LIB=...).  Never used by the package itself.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/scratch/variants
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $2 -c $R/oriana_amd/csrc/dense_pass.hip -o $O/dense_pass_$1.o
OBJS=$(ls $R/oriana_amd/csrc/*.o | grep -v dense_pass.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/liboriana_$1.so $OBJS $O/dense_pass_$1.o
rm -f $O/dense_pass_$1.o
echo $O/liboriana_$1.so
