import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from rdfc_gan_amd import hip
dt = torch.bfloat16
def run(N, H, W, Ci, Co, k, s, p, which):
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Ci, device='cuda').to(dt)
    w = (torch.randn(Co, Ci, k, k, device='cuda') * (Ci * k * k) ** -0.5).to(dt).float()
    wf, wb = hip.pack_weight(w, dt)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    gy = torch.randn(N, Ho, Wo, Co, device='cuda').to(dt)
    d = hip.conv_desc(x.shape, Co, k, s, p, ldx=Ci, ldy=Co, dtype=hip.dt(x))
    if which == 'fwd':
        y = torch.empty(N, Ho, Wo, Co, device='cuda', dtype=dt)
        hip.conv_forward(d, x, wf, None, y)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, s, p).permute(0, 2, 3, 1)
        got = y.float()
    else:
        dx = torch.empty_like(x)
        hip.conv_dgrad(d, gy, wb, None, dx)
        ref = F.conv_transpose2d(gy.float().permute(0, 3, 1, 2), w, None, s, p).permute(0, 2, 3, 1)
        got = dx.float()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(which, (N, H, W, Ci, Co, k, s, p), hip.lib().rdfc_last_kernel().decode(), 'rel err', err)
    if err > 0.02:
        bad = ((got - ref).abs() > 0.05 * ref.abs().max()).nonzero()
        print('  bad count', len(bad), 'of', got.numel(), 'first', bad[:6].tolist())
        print('  got', got.flatten()[:6].tolist(), 'ref', ref.flatten()[:6].tolist())
for c in [(2, 10, 12, 128, 1, 3, 1, 1, 'fwd'), (2, 9, 9, 512, 1, 4, 1, 1, 'fwd'), (1, 20, 22, 64, 3, 7, 1, 0, 'fwd'),
          (2, 10, 12, 1, 16, 3, 1, 1, 'dgrad'), (1, 20, 22, 1, 64, 7, 1, 0, 'dgrad'), (1, 4, 4, 64, 1, 1, 1, 0, 'fwd'), (1, 4, 4, 64, 1, 3, 1, 1, 'fwd'),
          (1, 4, 4, 8, 1, 3, 1, 1, 'fwd'), (1, 4, 4, 512, 1, 3, 1, 1, 'fwd')]:
    run(*c)
