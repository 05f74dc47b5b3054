#!/bin/bash
# Per-kernel register / spill / LDS report of one HIP source (hipcc remarks): tools/kres.sh samrs_amd/csrc/encoder_kernels.hip [filter]
f=$1; pat=${2:-.}
cd "$(dirname "$f")" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC ${KRES_FLAGS:-} -Rpass-analysis=kernel-resource-usage -c "$(basename "$f")" -o /dev/null 2>&1 |
  sed -n 's/.*remark: *\(.*\) \[-Rpass-analysis.*/\1/p' |
  awk '/Function Name:/ {if (n) print n, v, a, s, o, l; n=$3; v=a=s=o=l=""}
       /^ *VGPRs:/ {v="vgpr=" $2} /AGPRs:/ {a="agpr=" $2} /VGPRs Spill:/ {s="spill=" $3} /Occupancy/ {o="occ=" $3} /LDS Size/ {l="lds=" $4}
       END {print n, v, a, s, o, l}' | while read n rest; do echo "$(echo $n | c++filt | cut -c1-110) $rest"; done | grep -E "$pat"
