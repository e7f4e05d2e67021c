import torch, time
dev="cuda"
n=1<<29  # 2 GiB floats
a=torch.empty(n,device=dev); b=torch.empty(n,device=dev)
def t(f,reps=5):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps
w=t(lambda: a.fill_(1.0)); print("fill  %.2f TB/s"%(n*4/w/1e12))
c=t(lambda: b.copy_(a)); print("copy  %.2f TB/s (read+write)"%(2*n*4/c/1e12))
r=t(lambda: a.sum()); print("sum   %.2f TB/s"%(n*4/r/1e12))
