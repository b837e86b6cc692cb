"""Micro-benchmark of one conv shape through the C ABI (fwd / dgrad / wgrad), for rocprofv3 --pmc runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip
N, H, W, Ci, Co, k, s, p = (int(v) for v in sys.argv[1:9])
which = sys.argv[9] if len(sys.argv) > 9 else 'fwd'
iters = int(sys.argv[10]) if len(sys.argv) > 10 else 20
dt = torch.bfloat16
x = torch.randn(N, H, W, Ci, device='cuda').to(dt)
w = torch.randn(Co, Ci, k, k, device='cuda') * 0.05
wf, wb = hip.pack_weight(w, dt)
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
y = torch.empty(N, Ho, Wo, Co, device='cuda', dtype=dt)
gy = torch.randn(N, Ho, Wo, Co, device='cuda').to(dt)
dx = torch.empty_like(x)
if os.environ.get('RDFC_BENCH_ZERO') == '1':      # all-zero operands: no switching power in the matrix pipe -> shows what the clock costs
    x.zero_(); gy.zero_(); w.zero_(); wf, wb = hip.pack_weight(w, dt)
d = hip.conv_desc(x.shape, Co, k, s, p, ldx=Ci, ldy=Co, dtype=hip.dt(x))
def run():
    if which == 'fwd': hip.conv_forward(d, x, wf, None, y)
    elif which == 'dgrad': hip.conv_dgrad(d, gy, wb, None, dx)
    else: hip.conv_wgrad(d, x, gy)
for _ in range(3): run()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(iters): run()
torch.cuda.synchronize(); dtm = (time.perf_counter() - t) / iters
fl = 2.0 * N * Ho * Wo * Co * Ci * k * k
print(f'{which} N{N} {H}x{W} {Ci}->{Co} k{k}s{s}p{p}: {dtm*1e6:.1f} us  {fl/dtm/1e12:.1f} TFLOP/s')
