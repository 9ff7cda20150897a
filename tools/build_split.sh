#!/bin/bash
# compile spc_spatial_split.hip alone with the kernel resource remarks (registers, spills) in one line per kernel
cd /root/repo/spectral_cube_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 -mllvm -unroll-threshold=1000000 -Rpass-analysis=kernel-resource-usage -c spc_spatial_split.hip -o build/spc_spatial_split.o 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|ScratchSize" | grep -A4 "${1:-ILi4E}\|error" | grep -v "^--" | paste - - - - - | sed 's/spc_spatial_split.hip:[0-9]*:[0-9]*: remark://g; s/\[-Rpass-analysis=kernel-resource-usage\]//g' | awk '{print $3, $5,$6, $7,$8,$9,$10, $11,$12,$13,$14,$15,$16}'
