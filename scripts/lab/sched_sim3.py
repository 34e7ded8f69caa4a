"""Replay of a step from the per-task stamps (scripts/lab/times_dump.py -> gpurun_out/sched/t.bin): exact-task order by what the bounding pass knows
(huge or no bound first), units handed over to the second window once fewer than thr tasks are left, repeated passes moved to the second window.
usage: python scripts/lab/sched_sim3.py"""
import sys, numpy as np, heapq
t=np.fromfile('gpurun_out/sched/t.bin',dtype=np.uint64).reshape(-1,16); ti=t.astype(np.int64)
A,B,C=ti[:,0:4],ti[:,4:8],ti[:,8:12]
N=len(t)
bd=(A[:,1]-A[:,0])/100.0
hasB=B[:,2]>0; hasC=C[:,2]>0
ed=np.where(hasB,(B[:,1]-B[:,0])/100.0,0.0)
cd=np.where(hasC,(C[:,1]-C[:,0])/100.0,0.0)
U=t[:,13].copy().view(np.float64)
retries=(ti[:,14]>>8)
def sim(key=None, K=0, left_thr=0, rm=False, ncu=256, lat=60.0, heavy=None, verbose=False):
    ed2=ed.copy(); cd2=cd.copy(); goes=hasC.copy()
    if rm:
        r=(retries>0)&hasB&~hasC
        ed2[r]=0.4*ed[r]; cd2[r]=0.3*ed[r]; goes=goes|r
    xs=[e for e in (np.argsort(-key,kind='stable') if key is not None else range(N)) if hasB[e]]
    tasks=[('b',e) for e in range(N)]+[('x',e) for e in xs]
    nt=len(tasks); nxt=0
    free=[(0.0,s) for s in range(ncu*4)]; heapq.heapify(free)
    bdone=np.zeros(N); disc=np.full(N,np.inf); slot_end=np.zeros(ncu*4)
    while nxt<nt:
        tm,s=heapq.heappop(free)
        cu=s//4
        if cu>=ncu-K and (nt-nxt)<left_thr:
            slot_end[s]=tm; continue
        kind,e=tasks[nxt]; nxt+=1
        if kind=='b':
            bdone[e]=tm+bd[e]; heapq.heappush(free,(bdone[e],s))
            if goes[e] and not hasB[e]: disc[e]=bdone[e]
        else:
            st=max(tm,bdone[e]); en=st+ed2[e]; heapq.heappush(free,(en,s))
            if goes[e]: disc[e]=en
    while free:
        tm,s=heapq.heappop(free); slot_end[s]=max(slot_end[s],tm)
    t0end=slot_end.max()
    cu_free=slot_end.reshape(-1,4).max(axis=1)+lat
    cus=[(x,i) for i,x in enumerate(cu_free)]; heapq.heapify(cus)
    pending=set(np.nonzero(goes)[0].tolist()); t1end=0.0
    hv=heavy if heavy is not None else (cd2>700)
    while pending:
        tm,c=heapq.heappop(cus)
        avail=[e for e in pending if disc[e]<=tm]
        if not avail:
            heapq.heappush(cus,(min(disc[e] for e in pending),c)); continue
        avail.sort(key=lambda e:(not hv[e],disc[e]))
        e=avail[0]; pending.discard(e); t1end=max(t1end,tm+cd2[e]); heapq.heappush(cus,(tm+cd2[e],c))
    return t0end,t1end
print("plain                         : %.0f %.0f"%sim())
print("rm only                       : %.0f %.0f"%sim(rm=True))
cls=((U>=1e4)|~np.isfinite(U)|(U>1e300))*1.0
for nm,key in (("plain order",None),("U>=1e4 first",cls),("oracle overflow first",hasC*1.0),("oracle monsters first",(cd>900)*1.0)):
    for K,thr in ((0,0),(8,4500),(16,4500),(16,4000),(24,4500),(24,4000),(32,4500),(32,4000)):
        a=sim(key,K,thr,rm=False); b=sim(key,K,thr,rm=True)
        print("%-24s K=%2d thr=%4d:  no-rm tier0 %.0f tier1 %.0f  |  rm tier0 %.0f tier1 %.0f"%((nm,K,thr)+a+b))
