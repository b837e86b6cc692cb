"""Normalisation kernels on HBM-cold maps: cycles through enough distinct tensors that nothing is left in the 256 MB Infinity Cache
(tools/norm_bench.py loops on one map, which stays cache-resident).  Run under rocprofv3 --kernel-trace --stats for per-kernel times:
    python tools/norm_cold.py N H W C [G]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip
N, H, W, C = (int(v) for v in sys.argv[1:5])
G = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dt = torch.bfloat16
nbytes = N * H * W * C * 2
sets = max(3, int(700e6 / (3 * nbytes)) + 1)
xs = [torch.randn(N, H, W, C, device='cuda').to(dt) for _ in range(sets)]
dzs = [torch.randn(N, H, W, C, device='cuda').to(dt) for _ in range(sets)]
zs = [torch.empty_like(xs[0]) for _ in range(sets)]
gamma = torch.ones(C, device='cuda'); beta = torch.zeros(C, device='cuda')
for it in range(3 * sets):
    i = it % sets
    mean, rstd = hip.norm_stats(xs[i], G, 1e-5)
    hip.norm_apply(xs[i], mean, rstd, gamma, beta, None, zs[i], 'lrelu', 0.2)
for it in range(3 * sets):
    i = it % sets
    hip.norm_backward(dzs[i], xs[i], mean, rstd, gamma, beta, None, 'lrelu', 0.2)
torch.cuda.synchronize()
print(f'{sets} sets of {nbytes / 1e6:.1f} MB maps')
