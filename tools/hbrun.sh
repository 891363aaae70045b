set -e
for N in 1000 2000 4000; do
for D in 1 3 8; do
timeout -k 10 200 python tools/ab_ck.py $N $D "hb,hb@STB_HB_C=4,hb@STB_HB_GRID=64,chain,auto" 2
done; done
timeout -k 10 200 python tools/ab_ck.py 4000 32 "hb,hb@STB_HB_C=2,ck,chain,auto" 2
timeout -k 10 200 python tools/ab_ck.py 4000 64 "hb,ck,pc,auto" 2
timeout -k 10 200 python tools/ab_ck.py 10000 24 "hb,ck,pc" 2
timeout -k 10 200 python tools/ab_ck.py 10000 32 "hb,ck,pc" 2
timeout -k 10 200 python tools/ab_ck.py 20000 1 "hb,hb@STB_HB_C=4,ck,chain,auto" 2
timeout -k 10 200 python tools/ab_ck.py 20000 4 "hb,ck,pc,auto" 2
timeout -k 10 200 python tools/ab_ck.py 10000 3 "hb,hb@STB_HB_C=4,ck" 2 2000
