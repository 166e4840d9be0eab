import torch, time, os
x = torch.randn(8192, 8192, device='cuda')
small = torch.zeros(16384, dtype=torch.float64, device='cuda')
def work():
    for _ in range(40): y = x @ x
for mode in ('cpu','event_spin','sync','cpu'):
    torch.cuda.synchronize()
    t0=time.perf_counter(); work(); torch.cuda.synchronize(); tw=time.perf_counter()-t0
    t0=time.perf_counter(); work()
    if mode=='cpu': h = small.cpu()
    elif mode=='sync': torch.cuda.synchronize(); h=small.cpu()
    else:
        ev=torch.cuda.Event(); ev.record()
        while not ev.query(): pass
        h = small.cpu()
    t1=time.perf_counter()-t0
    print(mode, 'work %.1f ms, work+readback %.1f ms'%(tw*1e3, t1*1e3))
