import torch
dev="cuda:0"
def ev(fn,steps=50):
    for _ in range(5): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/steps*1e3
for mb in (30,60,120,480,1920):
    x=torch.empty(mb*1024*1024//4,device=dev); y=torch.empty_like(x)
    f=ev(lambda: x.fill_(1.0)); c=ev(lambda: y.copy_(x))
    print(f"{mb} MiB fill {f:.1f} us = {mb*1.048576/f*1e3/1e3:.2f} TB/s ; copy {c:.1f} us = {2*mb*1.048576/c:.2f} TB/s (r+w)")
