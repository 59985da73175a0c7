#!/bin/bash
# kernel resource usage of one csrc file: tools/kres.sh explorer_kernels.hip [name filter]
cd "$(dirname "$0")/../gnn-motion-planning_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - - | grep -E "${2:-.}" \
  | sed 's/Function Name: //; s/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/; s/LDS Size \[bytes\/block\]/lds/'
