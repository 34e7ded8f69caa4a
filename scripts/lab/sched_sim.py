"""Replay of one step's schedule from the per-task stamps of the instrumented build (scripts/lab/mk_times.py, times_dump.py):
bounding tasks, exact tasks and second-window searches with their measured durations under other task orders / reserved units.
usage: python scripts/lab/sched_sim.py <times.bin>"""
import sys, numpy as np, heapq
f=sys.argv[1]
t=np.fromfile(f,dtype=np.uint64).reshape(-1,16); ti=t.astype(np.int64)
A,B,C=ti[:,0:4],ti[:,4:8],ti[:,8:12]
N=len(t); t0=A[:,0].min()
bd=(A[:,1]-A[:,0])/100.0
hasB=B[:,2]>0; hasC=C[:,2]>0
ed=np.where(hasB,(B[:,1]-B[:,0])/100.0,0.0)
cd=np.where(hasC,(C[:,1]-C[:,0])/100.0,0.0)
span=(t[:,12]&0xffffffff).astype(np.int64); bn=(t[:,12]>>32).astype(np.int64)
n8=(ti[:,15]&0xfffff); n16=(ti[:,15]>>40)&0xfffff
retries=(ti[:,14]>>8)
print("actual: tier0 end %.0f tier1 start %.0f..%.0f end %.0f"%(((B[hasB,1]-t0)/100).max(),((C[hasC,0]-t0)/100).min(),((C[hasC,0]-t0)/100).max(),((C[hasC,1]-t0)/100).max()))
# bound overflow (no exact task): episodes with hasC and not hasB -> discovered at bound end
def sim(key=None, R=0, ncu=256, t1_latency=60.0, heavy=None, verbose=False, speed1=1.0):
    nslot=(ncu-R)*4
    free=[(0.0,s) for s in range(nslot)]; heapq.heapify(free)
    bdone=np.zeros(N)
    order=np.arange(N)
    for e in order:
        tm,s=heapq.heappop(free); bdone[e]=tm+bd[e]; heapq.heappush(free,(bdone[e],s))
    # exact tasks: in 'key' order (static list; claims in list order, waits for the bound if not ready)
    lst=list(np.argsort(-key,kind='stable')) if key is not None else list(range(N))
    disc=np.full(N,np.inf)   # time an overflow entry is queued
    for e in range(N):
        if hasC[e] and not hasB[e]: disc[e]=bdone[e]
    eend=np.zeros(N)
    for e in lst:
        if not hasB[e]: continue
        tm,s=heapq.heappop(free); st=max(tm,bdone[e]); eend[e]=st+ed[e]; heapq.heappush(free,(eend[e],s))
        if hasC[e]: disc[e]=eend[e]
    slot_end=np.zeros(nslot)
    while free:
        tm,s=heapq.heappop(free); slot_end[s]=tm
    t0end=slot_end.max()
    # a tier-0 workgroup leaves when it finds no task: at its last task's end.  CU free = max of its 4 slots
    cu_free=slot_end.reshape(-1,4).max(axis=1)+t1_latency
    cus=[(x,i) for i,x in enumerate(cu_free)]+[(0.0,1000+i) for i in range(R)]
    heapq.heapify(cus)
    ents=sorted([e for e in range(N) if hasC[e]],key=lambda e:disc[e])
    pending=set(ents)
    t1end=0.0; starts=[]
    # event loop: each CU when free takes the best available entry (heavy class first, FIFO), or waits for the next discovery
    hv=heavy if heavy is not None else np.zeros(N,bool)
    while pending:
        tm,c=heapq.heappop(cus)
        avail=[e for e in pending if disc[e]<=tm]
        if not avail:
            nxt=min(disc[e] for e in pending)
            heapq.heappush(cus,(nxt,c)); continue
        avail.sort(key=lambda e:(not hv[e],disc[e]))
        e=avail[0]; pending.discard(e)
        d=cd[e]*speed1
        starts.append(tm); t1end=max(t1end,tm+d); heapq.heappush(cus,(tm+d,c))
    return t0end,t1end,np.median(starts) if starts else 0
# heavy class emulation: cd large ~ we do not know the key; use actual: top by cd>700 as heavy (approx prio_thr)
heavy=cd>700
print("sim current           : tier0 end %.0f, tier1 end %.0f (median start %.0f)"%sim(None,0,heavy=heavy))
for R in (8,16,32):
    print("sim reserve R=%2d      : tier0 end %.0f, tier1 end %.0f (median start %.0f)"%((R,)+sim(None,R,heavy=heavy)))
print("oracle LPT exact      : tier0 end %.0f, tier1 end %.0f (median start %.0f)"%sim(ed,0,heavy=heavy))
print("oracle: overflow first: tier0 end %.0f, tier1 end %.0f (median start %.0f)"%sim(hasC*1000.0+ed,0,heavy=heavy))
for nm,k in (("bn",bn.astype(float)),("span",span.astype(float)),("bn*span",bn*span.astype(float)),("span>600",(span>600)*1.0),("bn>9000",(bn>9000)*1.0)):
    print("LPT by %-14s : tier0 end %.0f, tier1 end %.0f (median start %.0f)"%((nm,)+sim(k,0,heavy=heavy)))
    for R in (8,16):
        print("   + reserve R=%2d      : tier0 end %.0f, tier1 end %.0f (median start %.0f)"%((R,)+sim(k,R,heavy=heavy)))
print("overflow predicted by span>600: recall %.2f precision %.2f (n=%d)"%((span[hasC]>600).mean(),hasC[span>600].mean(),(span>600).sum()))
print("overflow predicted by bn>9000: recall %.2f precision %.2f (n=%d)"%((bn[hasC]>9000).mean(),hasC[bn>9000].mean(),(bn>9000).sum()))

# ---- exact-task order combined with units handed over to the second window while the first launch runs (K units stop taking tasks once fewer
# than left_thr tasks are left): replayed with oracle knowledge of the overflows and with what the bounding pass knows
span=(t[:,12]&0xffffffff).astype(np.int64); bn=(t[:,12]>>32).astype(np.int64)
def sim2(key, K=0, left_thr=0, ncu=256, lat=60.0):
    xs=[e for e in np.argsort(-key,kind='stable') if hasB[e]]
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
            if hasC[e] and not hasB[e]: disc[e]=bdone[e]
        else:
            st=max(tm,bdone[e]); en=st+ed[e]; heapq.heappush(free,(en,s))
            if hasC[e]: disc[e]=en
    while free:
        tm,s=heapq.heappop(free); slot_end[s]=max(slot_end[s],tm)
    t0end=slot_end.max()
    cu_free=slot_end.reshape(-1,4).max(axis=1)+lat
    cus=[(x,i) for i,x in enumerate(cu_free)]; heapq.heapify(cus)
    pending=set(np.nonzero(hasC)[0].tolist()); t1end=0.0
    while pending:
        tm,c=heapq.heappop(cus)
        avail=[e for e in pending if disc[e]<=tm]
        if not avail:
            heapq.heappush(cus,(min(disc[e] for e in pending),c)); continue
        avail.sort(key=lambda e:(not heavy[e],disc[e]))
        e=avail[0]; pending.discard(e); t1end=max(t1end,tm+cd[e]); heapq.heappush(cus,(tm+cd[e],c))
    return t0end,t1end
print("---- exact-task order + hand-over")
for nm,key in (("oracle overflow first",hasC*1.0),("oracle heavy-tier1 first",cd),("oracle LPT total",ed+cd),("pred span>600",(span>600)*1.0),("pred span",span.astype(float))):
    for K,thr in ((0,0),(24,3000),(32,3000),(48,3000),(32,2400)):
        print("%-26s K=%2d thr=%4d: tier0 end %.0f tier1 end %.0f"%((nm,K,thr)+sim2(key,K,thr)))
