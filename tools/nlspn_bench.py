"""Micro-benchmark of the NLSPN propagation kernels alone (B x 1 x H x W, T steps, random offsets of +-`spread` pixels):
HIP-event time of the forward chain and of the backward (chain + coefficient pass).  `python tools/nlspn_bench.py [B H W T spread]`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rdfc_gan_amd import hip

B, H, W, T = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 228, 304, 18)
spread = float(sys.argv[5]) if len(sys.argv) > 5 else 1.5
torch.manual_seed(0)
dev = 'cuda'
feat0 = torch.randn(B, 1, H, W, device=dev)
off = (torch.rand(B, 18, H, W, device=dev) * 2 - 1) * spread
off[:, 8:10] = 0
aff = torch.randn(B, 9, H, W, device=dev) * 0.1
wt = torch.ones(9, device=dev)
gout = torch.randn(B, 1, H, W, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


feats = hip.nlspn_propagate(feat0, off, aff, wt, None, T)
px = B * H * W
t_f = timed(lambda: hip.nlspn_propagate(feat0, off, aff, wt, None, T))
t_b1 = timed(lambda: hip.nlspn_propagate_backward(feats, off, aff, wt, gout, False))
t_b2 = timed(lambda: hip.nlspn_propagate_backward(feats, off, aff, wt, gout, True))
print(f'B{B} {H}x{W} T{T} spread {spread}: forward {t_f:.0f} us ({t_f / T:.1f} per step, {116.0 * px / (t_f / T) / 1e6:.2f} TB/s of 116 B/px); '
      f'backward chain {t_b1:.0f} us ({t_b1 / T:.1f} per step); coefficient pass {t_b2 - t_b1:.0f} us '
      f'({(T * 8 + 216.0) * px / max(t_b2 - t_b1, 1e-3) / 1e6:.2f} TB/s of {T * 8 + 216} B/px)')
