"""Achieved HBM rate of the per-channel kernels on one map (algorithmic bytes / time)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip
N, H, W, C = (int(v) for v in sys.argv[1:5])
G = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dt = torch.bfloat16
x = torch.randn(N, H, W, C, device='cuda').to(dt); dz = torch.randn(N, H, W, C, device='cuda').to(dt)
res = torch.randn(N, H, W, C, device='cuda').to(dt); z = torch.empty_like(x)
gamma = torch.ones(C, device='cuda'); beta = torch.zeros(C, device='cuda')
mean, rstd = hip.norm_stats(x, G, 1e-5)
nbytes = x.numel() * 2
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
for name, fn, nb in (
    ('stats (1R)', lambda: hip.norm_stats(x, G, 1e-5), 1),
    ('apply lrelu (1R+1W)', lambda: hip.norm_apply(x, mean, rstd, gamma, beta, None, z, 'lrelu', 0.2), 2),
    ('apply relu+res (2R+1W)', lambda: hip.norm_apply(x, mean, rstd, gamma, beta, res, z, 'relu', 0.0), 3),
    ('backward lrelu (reduce 2R + apply 2R+1W)', lambda: hip.norm_backward(dz, x, mean, rstd, gamma, beta, None, 'lrelu', 0.2), 5),
    ('backward relu+res (3R + 3R+2W)', lambda: hip.norm_backward(dz, x, mean, rstd, gamma, beta, res, 'relu', 0.0, want_dres=True), 8),
    ('channel_sum (1R)', lambda: hip.channel_sum(x), 1)):
    s = t(fn)
    print(f'{name:45s} {s*1e6:8.1f} us   {nb*nbytes/s/1e12:6.2f} TB/s')
