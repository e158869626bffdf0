import sys, torch
sys.path.insert(0, ".")
from sglang_amd import kernels as K
dev = torch.device("cuda:0")
lg = (torch.randn((64, 128256), device=dev) * 2).to(torch.bfloat16)
def gt(fn, reps=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("argmax bf16 [64,128256] us:", gt(lambda: K.argmax(lg)))
print("x8 in one graph, per call:", gt(lambda: [K.argmax(lg) for _ in range(8)]) / 8)
