import numpy as np, sys
S={};W=[]
for line in open(sys.argv[1]):
    f=line.split()
    if f[0]=="G": geo=list(map(int,f[1:]))
    elif f[0]=="S": S[int(f[1])]=np.array(list(map(int,f[2:])),dtype=np.int64)
    else: W.append(tuple(map(int,f[1:])))
C,P,JW,NBK,RB,TP,G,D=geo
print("geo",geo)
t0=min(s[0] for s in S.values() if s[0])
for jw in (0,1,4,16,40,JW-1):
    s=S[jw]; b=[x for x in s[1:NBK+1] if x]
    print("strip",jw,"start %.1f end %.1f ns/row %.1f"%((s[0]-t0)*0.01,(s[NBK+1]-t0)*0.01,np.median(np.diff(b))*10/(RB*8) if len(b)>2 else -1))
Wa=np.array([w for w in W if w[2] and w[4]],dtype=np.int64)
print("tiles",len(Wa),"wait med %.1f compute med %.1f p90 %.1f sum compute ms %.2f"%(np.median(Wa[:,3]-Wa[:,2])*0.01,np.median(Wa[:,4]-Wa[:,3])*0.01,np.percentile(Wa[:,4]-Wa[:,3],90)*0.01,(Wa[:,4]-Wa[:,3]).sum()*1e-5))
print("last tile %.1f last spine %.1f"%((Wa[:,4].max()-t0)*0.01,(max(s[NBK+1] for s in S.values())-t0)*0.01))
