set -e
T="timeout -k 10 120 python tools/timeline_hb.py"
echo "== D1 P4 C2 grid96"; TL_BLOCKS=0 STB_HB_GRID=96 STB_HB_P=4 $T 10000 1 gpurun_out/t6.txt | sed -n '2,4p;26,40p'
echo "== D8 P4"; TL_BLOCKS=0 TL_BINS=0,8,16,24,32,40 STB_HB_P=4 $T 10000 8 gpurun_out/t8.txt | sed -n '2,4p;14,100p'
