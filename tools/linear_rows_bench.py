import torch, sys, os
sys.path.insert(0, os.getcwd())
from situation3d_amd.blip2 import linear_rows
dev='cuda'
def t(fn, it=50):
    for _ in range(5): fn()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    g=torch.cuda.CUDAGraph()
    st=torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(it): fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it*1e3
for (m,k,n) in [(128,768,2048),(512,768,2048)]:
    lin=torch.nn.Linear(k,n).to(dev); x=torch.randn(m,k,device=dev,requires_grad=True); go=torch.randn(m,n,device=dev)
    def a():
        y=linear_rows(x,lin); y.backward(go); x.grad=None; lin.weight.grad=None; lin.bias.grad=None
    def b():
        y=lin(x); y.backward(go); x.grad=None; lin.weight.grad=None; lin.bias.grad=None
    print(m,k,n,"hand-written %.1f us  library %.1f us"%(t(a),t(b)))
