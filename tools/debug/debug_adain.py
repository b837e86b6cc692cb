import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch, math, torch.nn.functional as F
from rdfc_gan_amd import functional as Fh, hip
Fh.set_compute_dtype(torch.float32)
for N in (2, 1):
  for (C, H, W) in ((192, 30, 38), (64, 8, 10)):
    torch.manual_seed(0)
    x = torch.randn(N, C, H, W); s = torch.randn(N, C, H, W); w = torch.randn(2*C, C); b = torch.randn(2*C); cot = torch.randn(N, C, H, W)
    xr, sr, wr, br = (t.double().requires_grad_(True) for t in (x, s, w, b))
    gb = F.linear(sr.permute(0,2,3,1), wr*math.sqrt(2.0/C), br).permute(0,3,1,2)
    out = gb[:, :C]*F.instance_norm(xr) + gb[:, C:]
    (out*cot.double()).sum().backward()
    xg, sg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, s, w, b))
    o = Fh.to_nchw(Fh.wadain(Fh.to_pm(xg), Fh.to_pm(sg), wg, bg))
    (o*cot.cuda()).sum().backward()
    e = lambda a, r: float((a.cpu().double()-r).norm()/r.norm())
    print(N, C, H, W, 'out', e(o, out), 'dx', e(xg.grad, xr.grad), 'ds', e(sg.grad, sr.grad), 'dw', e(wg.grad, wr.grad), 'db', e(bg.grad, br.grad))
    # direct kernel check, no autograd
    x_pm = x.permute(0,2,3,1).contiguous().cuda(); gb_pm = gb.detach().float().permute(0,2,3,1).contiguous().cuda(); go = cot.permute(0,2,3,1).contiguous().cuda()
    mean, rstd = hip.norm_stats(x_pm, N, 1e-5)
    dx, dgb = hip.adain_backward(go, x_pm, mean, rstd, gb_pm)
    print('   direct dx', e(dx.permute(0,3,1,2), xr.grad))
