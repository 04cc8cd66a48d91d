import torch
dev='cuda'
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(True); e1=torch.cuda.Event(True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
n=3*1024**3//4
x=torch.empty(n,device=dev); y=torch.empty(n,device=dev)
print("fill 3GB: %.2f TB/s"%(n*4/t(lambda: x.fill_(1.0))/1e9))
print("copy 3GB: %.2f TB/s (read+write)"%(2*n*4/t(lambda: y.copy_(x))/1e9))
print("read-sum 3GB: %.2f TB/s"%(n*4/t(lambda: x.sum())/1e9))
xb=torch.empty(n*2,device=dev,dtype=torch.bfloat16)
print("fill bf16 3GB: %.2f TB/s"%(n*4/t(lambda: xb.fill_(1.0))/1e9))
