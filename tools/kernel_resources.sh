#!/bin/bash
# Register and scratch use of every kernel in rt_device.hip as the compiler reports it (no GPU needed).
# Usage: bash tools/kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/../raytracing_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I../../include "$@" \
    -Rpass-analysis=kernel-resource-usage -c rt_device.hip -o /tmp/rt_device_res.o 2>/tmp/rt_device_res.txt
awk '/Function Name:/ {name=$(NF-1)} /TotalSGPRs:/ {s=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {occ=$(NF-1)}
     /SGPRs Spill/ {sp=$(NF-1)} /LDS Size/ {printf "%-100s sgpr %3s (spilt %s) vgpr %3s scratch %5s waves/SIMD %s\n", name, s, sp, v, sc, occ}' /tmp/rt_device_res.txt | sed 's/_ZN12_GLOBAL__N_1[0-9]*//' | sort -u
