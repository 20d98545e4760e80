import torch, time
torch.cuda.init()
s=torch.cuda.Stream()
x=torch.zeros(1<<20,device='cuda')
with torch.cuda.stream(s):
    for _ in range(10): x.add_(1)
    evs=[]
    for i in range(200):
        x.add_(1)
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(s); b.record(s)
        evs.append((a,b))
        x.add_(1)
s.synchronize()
d=[a.elapsed_time(b)*1e3 for a,b in evs]
d.sort()
print('null event pair: median %.2f us, min %.2f, max %.2f'%(d[len(d)//2],d[0],d[-1]))
