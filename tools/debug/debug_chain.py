import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch, math, torch.nn.functional as F
from rdfc_gan_amd import functional as Fh, hip
Fh.set_compute_dtype(torch.float32)
for N in (2, 1):
    torch.manual_seed(0)
    C, H, W = 64, 8, 10
    a = torch.randn(N, C, H, W); bsk = torch.randn(N, C, H, W); s = torch.randn(N, 2*C, H, W); w = torch.randn(4*C, 2*C); b = torch.randn(4*C); cot = torch.randn(N, 2*C, H, W)
    ar, br_, sr, wr, bbr = (t.double().requires_grad_(True) for t in (a, bsk, s, w, b))
    x = torch.cat([ar, br_], 1)
    gb = F.linear(sr.permute(0,2,3,1), wr*math.sqrt(2.0/(2*C)), bbr).permute(0,3,1,2)
    out = gb[:, :2*C]*F.instance_norm(x) + gb[:, 2*C:]
    (out*cot.double()).sum().backward()
    ag, bg, sg, wg, bbg = (t.cuda().requires_grad_(True) for t in (a, bsk, s, w, b))
    apm, bpm = Fh.to_pm(ag), Fh.to_pm(bg)
    xg = Fh.crop_cat(apm, bpm)
    captured = {}
    xg.register_hook(lambda g: captured.__setitem__('dcat', (g.shape, g.stride(), g.is_contiguous(), g.detach().clone())))
    apm.register_hook(lambda g: captured.__setitem__('da', (g.shape, g.stride(), g.is_contiguous(), g.detach().clone())))
    o = Fh.to_nchw(Fh.wadain(xg, Fh.to_pm(sg), wg, bbg))
    (o*cot.cuda()).sum().backward()
    e = lambda a_, r: float((a_.cpu().double()-r).norm()/r.norm())
    print(N, 'da', e(ag.grad, ar.grad), 'db', e(bg.grad, br_.grad), 'ds', e(sg.grad, sr.grad))
    for k, v in captured.items():
        print('   ', k, v[0], v[1], v[2])
    dcat = captured['dcat'][3].permute(0,3,1,2).cpu().double()
    print('    dcat vs ref', e(dcat[:, :C], ar.grad), e(dcat[:, C:], br_.grad), ' da-hook vs ref', e(captured['da'][3].permute(0,3,1,2), ar.grad))
