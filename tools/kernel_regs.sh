#!/bin/bash
# Register / spill / LDS / occupancy summary of the gfx950 kernels of one source file (recompiles it with
# -Rpass-analysis=kernel-resource-usage, device side only):   tools/kernel_regs.sh aux_kernels.hip [name filter]
src=${1:?source file under gops_amd/csrc}; pat=${2:-.}
cd "$(dirname "$0")/../gops_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -I../../include -Xclang -target-feature -Xclang -packed-fp32-ops ${EXTRA} --cuda-device-only -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' | awk -v pat="$pat" '
  /Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /VGPRs Spill/ {sp=$NF} /ScratchSize/ {scr=$NF}
  /Occupancy/ {occ=$NF} /LDS Size/ { if (name ~ pat) printf "%s vgpr %s agpr %s spill %s scratch %s occ %s lds %s\n", name, v, a, sp, scr, occ, $NF }' | c++filt | sed 's/(.*) vgpr/ vgpr/'
