import torch, time, numpy as np
a = np.random.randint(0,255,100_000_000,dtype=np.uint8)
t = torch.from_numpy(a)
for i in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); d=t.to('cuda'); torch.cuda.synchronize(); print("H2D pageable 100MB %.1f ms" % ((time.perf_counter()-t0)*1e3))
p = t.pin_memory()
for i in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); d=p.to('cuda'); torch.cuda.synchronize(); print("H2D pinned 100MB %.1f ms" % ((time.perf_counter()-t0)*1e3))
for i in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); h=d[:32_000_000].cpu(); torch.cuda.synchronize(); print("D2H pageable 32MB %.1f ms" % ((time.perf_counter()-t0)*1e3))
t0=time.perf_counter(); x=torch.empty(7_000_000_000,dtype=torch.uint8,device='cuda'); torch.cuda.synchronize(); print("alloc 7GB %.1f ms" % ((time.perf_counter()-t0)*1e3))
