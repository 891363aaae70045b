set -e
timeout -k 10 200 python tools/ab_ck.py 1000 16 "hb,chain" 2
timeout -k 10 200 python tools/ab_ck.py 1000 32 "hb,chain" 2
timeout -k 10 200 python tools/ab_ck.py 1000 256 "hb,chain,pc" 2
timeout -k 10 200 python tools/ab_ck.py 600 1000 "hb,chain,pc" 2
timeout -k 10 200 python tools/ab_ck.py 2000 40 "hb,chain,pc" 2
timeout -k 10 200 python tools/ab_ck.py 2000 128 "hb,chain,pc" 2
timeout -k 10 200 python tools/ab_ck.py 3000 100 "hb,chain,pc" 2 200
timeout -k 10 200 python tools/ab_ck.py 50000 4 "hb,chain,pc" 2 100
timeout -k 10 200 python tools/ab_ck.py 700 8 "hb,chain" 2
timeout -k 10 200 python tools/ab_ck.py 512 1 "hb,chain" 2
