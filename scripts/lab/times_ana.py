import sys, numpy as np
for f in sys.argv[1:]:
    t=np.fromfile(f,dtype=np.uint64).reshape(-1,16).astype(np.int64)
    A,B,C=t[:,0:4],t[:,4:8],t[:,8:12]
    hasA,hasB,hasC=A[:,2]>0,B[:,2]>0,C[:,2]>0
    t0=min(A[hasA,0].min() if hasA.any() else 1<<62, B[hasB,0].min() if hasB.any() else 1<<62)
    us=lambda x:(x-t0)/100.0
    print(f)
    for nm,X,h in(("bound",A,hasA),("exact",B,hasB),("tier1",C,hasC)):
        if not h.any(): continue
        d=(X[h,1]-X[h,0])/100
        print("  %-6s n=%5d start %6.0f..%6.0f endmax %6.0f dur mean %5.0f q50 %5.0f q90 %5.0f q99 %5.0f max %5.0f sum(ms) %7.0f"%(nm,h.sum(),us(X[h,0]).min(),us(X[h,0]).max(),us(X[h,1]).max(),d.mean(),*np.quantile(d,[.5,.9,.99]),d.max(),d.sum()/1e3))
    print("  prepass nodes mean %.0f"%t[:,13].mean(), "bounded %.3f"%t[:,14].mean())
