#!/bin/bash
# register / scratch / occupancy of every kernel in one source file of the library, from the compiler's own report
# usage: tools/kernel_regs.sh raytracer_amd/csrc/kernels_trace.hip [name-filter]
src=$1; pat=${2:-.}
cd "$(dirname "$0")/../raytracer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-slp-vectorize -Wno-unused-function --cuda-device-only -c "$(basename "$src")" -x hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk -v pat="$pat" '/Function Name:/ {name=$(NF-1)} /VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /SGPRs:/ {s=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /VGPR Spill/ {vs=$(NF-1)} /SGPR Spill/ {ss=$(NF-1)}
       /Occupancy \[waves\/SIMD\]/ {o=$(NF-1)} /LDS Size/ { if (name ~ pat) printf "%-44s vgpr %3s agpr %3s (spill %3s) sgpr %3s (spill %3s) scratch %5s B  occupancy %s  lds %s\n", substr(name,1,44), v, a, vs, s, ss, sc, o, $(NF-1) }'
