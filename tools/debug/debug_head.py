import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch, torch.nn as nn, torch.nn.functional as F
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd.layers import conv_bn_relu, run_conv
Fh.set_compute_dtype(torch.float32)
torch.manual_seed(0)
for cout in (64, 32):
    N, H, W = 2, 60, 76
    blk = conv_bn_relu(128, cout, 3, 1, 1)
    head = nn.Conv2d(cout + 64, 1, 3, 1, 1)
    with torch.no_grad():
        blk[1].weight.normal_(1, 0.1); blk[1].bias.normal_(0, 0.1)
    x = torch.randn(N, 128, H, W).abs() * 0.7; skip = torch.randn(N, 64, H, W); cot = torch.randn(N, 1, H, W)
    # reference fp64
    import copy
    rb, rh = copy.deepcopy(blk).double(), copy.deepcopy(head).double()
    xr, sr = x.double().requires_grad_(True), skip.double().requires_grad_(True)
    y = F.leaky_relu(F.batch_norm(F.conv2d(xr, rb[0].weight, None, 1, 1), None, None, rb[1].weight, rb[1].bias, True), 0.2)
    o = torch.sigmoid(F.conv2d(torch.cat([y, sr], 1), rh.weight, rh.bias, 1, 1))
    (o * cot.double()).sum().backward()
    # fp32 torch
    fb, fh = copy.deepcopy(blk), copy.deepcopy(head)
    xf = x.clone().requires_grad_(True)
    yf = F.leaky_relu(F.batch_norm(F.conv2d(xf, fb[0].weight, None, 1, 1), None, None, fb[1].weight, fb[1].bias, True), 0.2)
    of = torch.sigmoid(F.conv2d(torch.cat([yf, skip], 1), fh.weight, fh.bias, 1, 1))
    (of * cot).sum().backward()
    # HIP
    blk.cuda().train(); head.cuda()
    xg = x.cuda().requires_grad_(True); sg = skip.cuda().requires_grad_(True)
    xp, sp = Fh.to_pm(xg), Fh.to_pm(sg)
    yg = blk(xp)
    og = Fh.to_nchw(run_conv(Fh.crop_cat(yg, sp), head, None, nn.Sigmoid(), out_f32=True))
    (og * cot.cuda()).sum().backward()
    def e(a, b): return float((a.cpu().double() - b).norm() / b.norm())
    print(f'cout={cout}: out {e(og, o):.1e}')
    for name, g, f32, r in (('conv.w', blk[0].weight.grad, fb[0].weight.grad, rb[0].weight.grad), ('bn.w', blk[1].weight.grad, fb[1].weight.grad, rb[1].weight.grad),
                       ('bn.b', blk[1].bias.grad, fb[1].bias.grad, rb[1].bias.grad), ('head.w', head.weight.grad, fh.weight.grad, rh.weight.grad),
                       ('x', xg.grad, xf.grad, xr.grad), ('skip', sg.grad, None, sr.grad)):
        print(f'   {name:8s} hip-vs-f64 {e(g, r):.2e}   torchf32-vs-f64 {e(f32, r) if f32 is not None else float("nan"):.2e}')
