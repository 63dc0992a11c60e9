import torch, time, numpy as np
n=240_000_000
d=torch.empty(n,dtype=torch.uint8,device='cuda')
h=torch.empty(n,dtype=torch.uint8)
hp=torch.empty(n,dtype=torch.uint8,pin_memory=True)
for name,dst in (('pageable',h),('pinned',hp)):
    for r in range(3):
        torch.cuda.synchronize(); t=time.perf_counter(); dst.copy_(d); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(name, '%.1f ms  %.1f GB/s'%(dt*1e3, n/dt/1e9))
a=np.empty(n,dtype=np.uint8)
t=time.perf_counter(); a[:]=hp.numpy(); print('memcpy pinned->numpy %.1f ms'%((time.perf_counter()-t)*1e3))
t=time.perf_counter(); b=np.empty(n,dtype=np.uint8); b[:]=1; print('alloc+touch %.1f ms'%((time.perf_counter()-t)*1e3))
t=time.perf_counter(); x=torch.empty(n,dtype=torch.uint8,pin_memory=True); print('pinned alloc %.1f ms'%((time.perf_counter()-t)*1e3))
