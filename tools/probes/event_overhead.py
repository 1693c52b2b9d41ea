import torch, statistics
x=torch.zeros(16,device='cuda')
torch.cuda.synchronize()
def run(n):
    evs=[torch.cuda.Event(enable_timing=True) for _ in range(2*n)]
    for i in range(n):
        evs[2*i].record(); x.zero_(); evs[2*i+1].record()
    torch.cuda.synchronize()
    return [evs[2*i].elapsed_time(evs[2*i+1])*1e3 for i in range(n)]
run(10)
v=run(50); print('tiny kernel between events: median %.2f us min %.2f'%(statistics.median(v),min(v)))
# with a big kernel queued before (GPU busy) 
y=torch.randn(8192,8192,device='cuda')
z=y@y
v=[]
for _ in range(10):
    z=y@y
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); x.zero_(); e1.record()
    torch.cuda.synchronize(); v.append(e0.elapsed_time(e1)*1e3)
print('after a busy kernel: median %.2f us'%statistics.median(v))
