import sys, numpy as np
t=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,16).astype(np.int64)
A,B,C=t[:,0:4],t[:,4:8],t[:,8:12]
hasA,hasB,hasC=A[:,2]>0,B[:,2]>0,C[:,2]>0
t0=A[hasA,0].min()
us=lambda x:(x-t0)/100.0
for nm,X,h in(("bound",A,hasA),("exact",B,hasB),("tier1",C,hasC)):
    d=(X[h,1]-X[h,0])/100
    print("%-6s n=%5d start %6.0f..%6.0f end q50 %6.0f q99 %6.0f max %6.0f | dur mean %5.0f q50 %5.0f q90 %5.0f q99 %5.0f max %5.0f | sum %7.0f WG-ms"%(nm,h.sum(),us(X[h,0]).min(),us(X[h,0]).max(),*np.quantile(us(X[h,1]),[.5,.99]),us(X[h,1]).max(),d.mean(),*np.quantile(d,[.5,.9,.99]),d.max(),d.sum()/1e3))
# tier-1 detail: queued time = end of the exact task of that episode in tier 0 (B end) ; start in tier1
q=us(B[hasC,1]); s=us(C[hasC,0]); e=us(C[hasC,1]); d=e-s
o=np.argsort(e)[::-1][:12]
print("last tier-1 episodes to finish: queued, started, waited, ran, ended")
for k in o: print("   %6.0f %6.0f %6.0f %6.0f %6.0f"%(q[k],s[k],s[k]-q[k],d[k],e[k]))
print("tier-1 wait q50 %.0f q90 %.0f ; ran q50 %.0f q90 %.0f max %.0f; first start %.0f"%(np.median(s-q),np.quantile(s-q,.9),np.median(d),np.quantile(d,.9),d.max(),s.min()))
# the first launch's last tasks
eb=us(B[hasB,1]); sb=us(B[hasB,0]); db=eb-sb
o=np.argsort(eb)[::-1][:8]
print("last tier-0 exact tasks: start, dur, end, retries")
r=(t[hasB,14]>>8)
for k in o: print("   %6.0f %6.0f %6.0f %d"%(sb[k],db[k],eb[k],r[k]))
print("tier-0 tasks handed out until %.0f us; busy WG-ms %.0f of %.0f available until first-launch end" % (max(sb.max(),us(A[hasA,0]).max()), ((A[hasA,1]-A[hasA,0]).sum()+(B[hasB,1]-B[hasB,0]).sum())/1e5, 1024*eb.max()/1e3))
