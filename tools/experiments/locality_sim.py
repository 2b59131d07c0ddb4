"""Simulation (numpy, no device): how many DISTINCT partner reads a window of consecutive candidates touches in file order and after
ordering the rows of the candidate matrix by a one-hop / two-hop min-partner key."""
import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
import bench
reads,cand,cfg,st=bench.build_workload("c3-lite",0)
r1=cand["read1"].astype(np.int64); r2=cand["read2"].astype(np.int64)
n=r1.size; R=reads.n_reads
print(n, R)
def distinct_ratio(order, W):
    a=r2[order]; m=(n//W)*W
    a=a[:m].reshape(-1,W)
    a=np.sort(a,axis=1)
    d=(np.diff(a,axis=1)!=0).sum(axis=1)+1
    return d.mean()/W
ident=np.arange(n)
# key per row: min(r1, min r2 over the row) ; also symmetric: min over all partners (as read2 of others too)
key=np.arange(R)
np.minimum.at(key, r1, r2)
np.minimum.at(key, r2, r1)
# iterate once more (two-hop) for larger clusters
key2=key.copy()
np.minimum.at(key2, r1, key[r2]); np.minimum.at(key2, r2, key[r1])
for name,k in (("file order",None),("key1",key),("key2",key2)):
    if k is None: order=ident
    else: order=np.lexsort((r2, r1, k[r1]))
    print(name, [round(distinct_ratio(order,W),4) for W in (16384, 65536, 262144, 1048576)], "clusters", None if k is None else np.unique(k).size)
