#!/bin/bash
# kernel resource summary of one .hip source: name, VGPRs, spills, occupancy   (tools/kres.sh oriana_amd/csrc/x.hip [filter])
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
  awk '/Function Name/{n=$5} / VGPRs:/{v=$4} /AGPRs:/{a=$4} /VGPRs Spill/{s=$5} /Occupancy/{o=$5} /LDS Size/{print n, "vgpr", v, "agpr", a, "spill", s, "occ", o}' |
  c++filt | grep -E "${2:-.}"
