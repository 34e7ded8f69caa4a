import sys, numpy as np
f=sys.argv[1]
t=np.fromfile(f,dtype=np.uint64).reshape(-1,16).astype(np.int64)
A,B,C=t[:,0:4],t[:,4:8],t[:,8:12]
hA,hB,hC=A[:,2]>0,B[:,2]>0,C[:,2]>0
t0=min(A[hA,0].min(),B[hB,0].min())
us=lambda x:(x-t0)/100.0
ev=[]
for X,h,k in((A,hA,'b'),(B,hB,'x')):
    for s,e,_,blk in X[h]: ev.append((int(blk),us(s),us(e),k))
ev.sort()
nb=max(b for b,_,_,_ in ev)+1
busy=np.zeros(nb); first=np.full(nb,1e9); last=np.zeros(nb); gaps=np.zeros(nb); prev_end={}
for b,s,e,k in ev:
    busy[b]+=e-s; first[b]=min(first[b],s); last[b]=max(last[b],e)
    if b in prev_end: gaps[b]+=max(0,s-prev_end[b])
    prev_end[b]=e
print("blocks",nb,"busy mean %.0f  last-end q10 %.0f q50 %.0f q90 %.0f max %.0f; gaps mean %.0f max %.0f"%(busy.mean(),*np.quantile(last,[.1,.5,.9]),last.max(),gaps.mean(),gaps.max()))
# utilisation over time: number of busy blocks in 250us bins
T=int(last.max()//250)+1
occ=np.zeros(T)
for b,s,e,k in ev:
    for i in range(int(s//250),int(e//250)+1):
        lo=max(s,i*250); hi=min(e,(i+1)*250)
        if hi>lo and i<T: occ[i]+=(hi-lo)/250
print("busy blocks per 250us bin:", " ".join("%d"%x for x in occ))
