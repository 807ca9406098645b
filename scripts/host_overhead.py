import os, sys, time, torch
sys.path.insert(0,'/root/repo')
import torch.distributed as dist
from annlite_amd import Metric, PQCodec, PQFlatGpuIndex
from annlite_amd.sharded import ShardedPQIndex
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29588')
dist.init_process_group('nccl', rank=0, world_size=1)
dev=torch.device('cuda',0); torch.cuda.set_device(0)
N,D,M,B,k=20000,128,16,1024,10   # tiny table: GPU time per step is small, the host cost shows
x=torch.randn(N,D,device=dev)
codec=PQCodec(dim=D,n_subvectors=M,n_clusters=256,metric=Metric.EUCLIDEAN,n_init=1); codec.seed=1; codec.fit(x[:8192],iter=3)
index=PQFlatGpuIndex(dim=D,metric=Metric.EUCLIDEAN,pq_codec=codec,initial_size=N)
index.add_with_ids(x, torch.arange(N,device=dev))
sh=ShardedPQIndex(index,row_base=0)
q=torch.randn(B,D,device=dev)
for mode in ('nogather','gather'):
    if mode=='gather': os.environ['ANNLITE_FORCE_GATHER']='1'
    for _ in range(5): sh.search_batch_async(q,k).result(wait=False)
    torch.cuda.synchronize()
    t=time.perf_counter(); n=200
    for _ in range(n): sh.search_batch_async(q,k).result(wait=False)
    host=time.perf_counter()-t
    torch.cuda.synchronize(); tot=time.perf_counter()-t
    print(mode,'host us/step',host/n*1e6,'total us/step',tot/n*1e6)
import cProfile,pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(200): sh.search_batch_async(q,k).result(wait=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
