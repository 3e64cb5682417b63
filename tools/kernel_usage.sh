#!/bin/bash
# usage: tools_kernel_usage.sh file.hip [extra hipcc flags] -- one line per kernel: vgpr sgpr scratch occupancy lds
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
 grep -E "Function Name|VGPRs:|TotalSGPRs|ScratchSize|Occupancy|LDS Size" | paste - - - - - - | \
 sed -E 's/.*Function Name: ([^ ]+).*TotalSGPRs: ([0-9]+).*VGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*Occupancy \[waves\/SIMD\]: ([0-9]+).*LDS Size \[bytes\/block\]: ([0-9]+).*/\1 sgpr=\2 vgpr=\3 scratch=\4 occ=\5 lds=\6/' | c++filt
