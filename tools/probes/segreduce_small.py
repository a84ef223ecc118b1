import sys, torch, time
sys.path.insert(0,'/root/repo')
from hsg_amd import ops
dev=torch.device('cuda:0')
for n,d,P in [(12544,128,128),(12544,16,128),(12544,128,16),(3000,128,64),(50000,128,128)]:
    x=torch.randn(n,d,device=dev); lab=torch.randint(0,P,(n,),device=dev)
    for _ in range(3): ops.segment_reduce(x,lab,P,1)
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.segment_reduce(x,lab,P,1)
    b.record(); torch.cuda.synchronize()
    print(n,d,P,'%.1f us per call'%(a.elapsed_time(b)/20*1e3))
