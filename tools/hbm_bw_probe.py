import torch, time
dev=torch.device('cuda:0')
n=2*1024**3
x=torch.empty(n, dtype=torch.float32, device=dev); y=torch.empty(n, dtype=torch.float32, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps
print('fill  %.2f TB/s' % (n*4/t(lambda: x.fill_(1.0))/1e12))
print('copy  %.2f TB/s (read+write)' % (2*n*4/t(lambda: y.copy_(x))/1e12))
print('sum   %.2f TB/s' % (n*4/t(lambda: x.sum())/1e12))
print('add   %.2f TB/s (2r+1w)' % (3*n*4/t(lambda: torch.add(x,y,out=y))/1e12))
