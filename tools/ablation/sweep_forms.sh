#!/bin/bash
# which fill form wins at which batch size (N = M = 10000 unless given): bash tools/sweep_forms.sh [N] > gpurun_out/sweep_forms.txt
N=${1:-10000}
for D in 1 2 3 4 6 8 12 16 24 32; do
  timeout -k 10 300 python tools/ab_ck.py $N $D "ck@STB_CK_WG_PER_CU=1@STB_CK_C=2,ck@STB_CK_WG_PER_CU=1@STB_CK_C=4,ck@STB_CK_WG_PER_CU=2@STB_CK_C=4,chain,pc" 2 2>&1 | grep -v amdgpu.ids | sed 's/ M=[0-9]*//' | awk '{print $1, $2, $3, $(NF-4), $(NF-3), $(NF-1), $NF}'
done
