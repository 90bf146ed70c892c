#!/bin/sh
# developer helper: register / LDS / occupancy figures of every kernel as the compiler reports them (no GPU needed)
#   tools/kernel_resources.sh > profiles/roundN_kernel_resources.txt
cd "$(dirname "$0")/../vvdec_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-scalarize-global-loads=false -w -c -x hip vvr_kernels.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 |
awk '/Function Name|remark:.* Name:/ { name=$NF; sub(/\[.*/, "", name); n=$0; sub(/.*Name: /, "", n); sub(/ \[.*/, "", n); name=n }
     /TotalSGPRs:/ { s=$0; sub(/.*TotalSGPRs: /, "", s); sub(/ .*/, "", s) }
     / VGPRs:/ { v=$0; sub(/.* VGPRs: /, "", v); sub(/ .*/, "", v) }
     /ScratchSize/ { sc=$0; sub(/.*: /, "", sc); sub(/ .*/, "", sc) }
     /Occupancy/ { o=$0; sub(/.*: /, "", o); sub(/ .*/, "", o) }
     /LDS Size/ { l=$0; sub(/.*: /, "", l); sub(/ .*/, "", l); printf "%-72s SGPR %3s  VGPR %3s  scratch %s  waves/SIMD %s  LDS %6s B\n", name, s, v, sc, o, l }' | c++filt | sort
