import numpy as np, sys
f=np.load(sys.argv[1])
tl,used=f['tl'],f['used']
waves=np.nonzero(used)[0]
rec=np.concatenate([tl[w,:used[w]] for w in waves]).astype(np.int64)
wait0,got,done,meta=rec[:,0],rec[:,1],rec[:,2],rec[:,3]
count=meta&0xFFFFFFFF; path=(meta>>32)&0xFF; pq=((meta>>56)&0xFF)/4.0
t0=wait0.min()
start=(got-t0)/100.0; dur=(done-got)/100.0; end=(done-t0)/100.0
print("units %d, waves %d, span %.1f us"%(len(rec),len(waves),end.max()))
edges=list(range(0,int(end.max())+20,20))
for a,b in zip(edges[:-1],edges[1:]):
    m=(start>=a)&(start<b)
    if m.any():
        print("  start in [%3d,%3d) us: %5d units, mean tuples %5.0f, dur mean %5.1f p10 %5.1f p50 %5.1f p90 %5.1f max %5.1f; paths %s"%(a,b,m.sum(),count[m].mean(),dur[m].mean(),*np.percentile(dur[m],[10,50,90]),dur[m].max(),np.bincount(path[m]).tolist()))
big=count>=512
if big.any():
    print("big units: ns/tuple/wave %.1f"%(1e3*dur[big].sum()/count[big].sum()))
