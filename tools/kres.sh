#!/bin/bash
# compact per-kernel resource usage of one .hip file: bash tools/kres.sh libstb_amd/csrc/fill_ck.hip
# columns: kernel, SGPRs, VGPRs, scratch bytes per lane
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|TotalSGPRs|Occupancy|ScratchSize|LDS Size" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - - - |
  awk '{print $3, "sgpr", $5, "vgpr", $7, "scratch", $10}'
