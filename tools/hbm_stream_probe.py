import torch,time
x=torch.randn(512*1024*1024//1, device='cuda')  # 2 GiB
def t(fn,it=10):
    fn(); torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it
ms=t(lambda: x.sum()); print("sum read   %.3f ms  %.2f TB/s"%(ms, x.numel()*4/ms/1e9))
y=torch.empty_like(x)
ms=t(lambda: y.copy_(x)); print("copy r+w   %.3f ms  %.2f TB/s (total)"%(ms, 2*x.numel()*4/ms/1e9))
ms=t(lambda: y.fill_(1.0)); print("fill write %.3f ms  %.2f TB/s"%(ms, x.numel()*4/ms/1e9))
ms=t(lambda: x.max()); print("max read   %.3f ms  %.2f TB/s"%(ms, x.numel()*4/ms/1e9))
