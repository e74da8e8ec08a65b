import sys, time; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from nnpops_amd import capi
dev = 'cuda:0'
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (M, K, N) in [(1333, 1008, 2048), (1333, 2048, 1008), (667, 1008, 1536), (10000, 1008, 2048), (1333, 256, 192)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / np.sqrt(K); b = torch.randn(N, device=dev)
    pl = capi.split_planes(w); out = torch.empty(M, N, device=dev)
    ref = torch.nn.functional.celu((a.double() @ w.double().t()) + b.double(), alpha=0.1)
    capi.gemm_split(a, pl, bias=b, out=out); err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    us_mine = t(lambda: capi.gemm_split(a, pl, bias=b, out=out))
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N}: {us_mine:.1f} us ({fl/us_mine/1e6:.1f} TFLOP/s) err {err:.1e}")
