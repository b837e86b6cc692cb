"""What the epilogue addends of the input-gradient launches cost (rdfc_conv2d_dgrad_add / _add2: the residual junctions of the BasicBlocks
and the skip connections behind the stride-2 stages): us per launch without / with one / with two addends, alone on the chip, f16.
python tools/addend_bench.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rdfc_gan_amd import hip

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = torch.float16
SHAPES = [(16, 228, 304, 64, 64, 3, 1, 1), (16, 114, 152, 128, 128, 3, 1, 1), (16, 57, 76, 256, 256, 3, 1, 1), (16, 228, 304, 64, 128, 3, 2, 1),
          (16, 114, 152, 128, 256, 3, 2, 1), (16, 228, 304, 64, 192, 3, 2, 1)]


def timed(fn):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for (N, H, W, Ci, Co, k, s, p) in SHAPES:
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = torch.randn(Co, Ci, k, k, device='cuda') * 0.05
    wf, wb = hip.pack_weight(w, dt)
    gy = torch.randn(N, Ho, Wo, Co, device='cuda').to(dt)
    dx = torch.empty(N, H, W, Ci, device='cuda', dtype=dt)
    a1, a2 = torch.randn_like(dx), torch.randn_like(dx)
    d = hip.conv_desc(dx.shape, Co, k, s, p, ldx=Ci, ldy=Co, dtype=hip.dt(dx))
    t0 = timed(lambda: hip.conv_dgrad(d, gy, wb, None, dx))
    kern = hip.lib().rdfc_last_kernel().decode()
    t1 = timed(lambda: hip.conv_dgrad(d, gy, wb, None, dx, addend=a1))
    t2 = timed(lambda: hip.conv_dgrad(d, gy, wb, None, dx, addend=a1, addend2=a2))
    mb = dx.numel() * 2 / 1e6
    print(f'dgrad N{N} {H}x{W} {Ci}<-{Co} k{k}s{s} [{kern}]: plain {t0:.1f} us | + addend {t1:.1f} | + two {t2:.1f}   (an addend is {mb:.0f} MB = {mb / 5.0e3:.1f} us at 5 TB/s)', flush=True)
