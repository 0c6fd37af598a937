import torch, time, subprocess
dev=torch.device('cuda:0')
big=torch.empty(1966080000//4, dtype=torch.float32, device=dev)  # the 1.97 GB of the cfg4 observation stream of 16 steps
big.fill_(1.0); torch.cuda.synchronize()
for rep in range(2):
    t0=time.perf_counter()
    for _ in range(10): big.fill_(2.0)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print('torch fill 1.97 GB: %.3f ms %.2f TB/s'%(dt*1e3, big.numel()*4/dt/1e12), flush=True)
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): big.fill_(3.0)
e1.record(); torch.cuda.synchronize()
print('torch fill (events): %.3f ms'%(e0.elapsed_time(e1)/10))
z=torch.zeros_like(big); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): big.zero_()
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
print('torch zero_: %.3f ms %.2f TB/s'%(dt*1e3, big.numel()*4/dt/1e12))
