import torch, time
dev="cuda:0"
n=256*1024*1024  # floats = 1 GiB
a=torch.empty(n,device=dev); b=torch.empty(n,device=dev)
def timeit(f, nbytes, name, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/it
    print("%-28s %.3f ms  %.0f GB/s"%(name, ms, nbytes/ms/1e6))
timeit(lambda: a.fill_(1.0), n*4, "fill (write only)")
timeit(lambda: b.copy_(a), n*8, "copy (read+write)")
timeit(lambda: a.sum(), n*4, "sum (read only)")
timeit(lambda: torch.add(a,b,out=b), n*12, "add (2 read + 1 write)")
c=torch.empty(n//6,device=dev)
timeit(lambda: torch.nn.functional.relu(a[:n//6], inplace=False), n//6*8, "relu 1:1 small (178MB each)")
