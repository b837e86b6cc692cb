import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch, torch.nn.functional as F
from cases import gen_inputs, out_weights
from detweights import det_state_dict
from oracle import nets as onets
from rdfc_gan_amd import functional as Fh, generator, hip
Fh.set_compute_dtype(torch.float32)
B = int(os.environ.get('DBG_B', 1))
net, ref = generator.RDFGenerator(pretrained_on_imagenet=False), onets.RDFGenerator()
sd = det_state_dict(net.state_dict(), 0)
net.load_state_dict(sd); ref.load_state_dict(sd); net.cuda().train(); ref.double().train()
P, O = [], []
orig_bwd = Fh._WAdaIN.backward
def bwd(ctx, dout):
    x, style, w, gb, mean, rstd = ctx.saved_tensors
    res = orig_bwd(ctx, dout)
    P.append(dict(x=x.detach().clone(), style=style.detach().clone(), gb=gb.detach().clone(), dout=dout.detach().clone(), dx=res[0].detach().clone(), ds=res[1].detach().clone()))
    return res
Fh._WAdaIN.backward = staticmethod(bwd)
def fh(mod, inp, out):
    rec = dict(x=inp[0].detach(), style=inp[1].detach())
    inp[0].register_hook(lambda g: rec.__setitem__('dx_total', g.detach().clone()))
    out.register_hook(lambda g: rec.__setitem__('dout', g.detach().clone()))
    O.append(rec)
for i in (1, 2, 3, 4):
    getattr(ref, f'fuse_layer{i}').register_forward_hook(fh)
inputs = gen_inputs('g_a2b', B, 60, 76)
def go(m, ins, dev, dtype):
    ins = [t.to(dev).to(dtype).requires_grad_(True) for t in ins]
    o = m(*ins); loss = 0
    for k, v in o.items():
        loss = loss + (v * out_weights('dbg', k, v.shape).to(dev).to(dtype)).sum()
    loss.backward()
go(net, inputs, 'cuda', torch.float32); go(ref, inputs, 'cpu', torch.float64)
e = lambda a, r: float((a.cpu().double().permute(0,3,1,2) - r).norm() / r.norm())
for p, o in zip(P, reversed(O)):     # product backward order is fuse4..fuse1, oracle forward order fuse1..fuse4
    print(tuple(p['x'].shape), 'x', e(p['x'], o['x']), 'style', e(p['style'], o['style']), 'dout', e(p['dout'], o['dout']), 'dx vs oracle total-dx', e(p['dx'], o['dx_total']),
          '|dx|', float(p['dx'].norm()), '|oracle dx_total|', float(o['dx_total'].norm()))

def manual(dout, x, gb):   # all NCHW double
    C = x.shape[1]
    mu = x.mean((2, 3), keepdim=True); var = x.var((2, 3), unbiased=False, keepdim=True); rs = (var + 1e-5).rsqrt(); xh = (x - mu) * rs
    dxh = dout * gb[:, :C]
    return rs * (dxh - dxh.mean((2, 3), keepdim=True) - xh * (dxh * xh).mean((2, 3), keepdim=True)), var
p, o = P[0], O[-1]
nchw = lambda t: t.cpu().double().permute(0, 3, 1, 2)
fuse = ref.fuse_layer4
gb_o = fuse.style.linear(o['style'].permute(0, 2, 3, 1)).permute(0, 3, 1, 2).detach()
dx_o_manual, var_o = manual(o['dout'], o['x'], gb_o)
dx_p_manual, var_p = manual(nchw(p['dout']), nchw(p['x']), nchw(p['gb']))
rel = lambda a, b: float((a - b).norm() / b.norm())
print('oracle autograd dx vs manual(oracle inputs):', rel(o['dx_total'], dx_o_manual))
print('product dx vs manual(product inputs):', rel(nchw(p['dx']), dx_p_manual))
print('manual(product inputs) vs manual(oracle inputs):', rel(dx_p_manual, dx_o_manual))
print('gb product vs oracle', rel(nchw(p['gb']), gb_o))
print('min channel var', float(var_o.min()), 'channels with var<1e-6:', int((var_o < 1e-6).sum()), 'of', var_o.numel())
d = (dx_p_manual - dx_o_manual).pow(2).sum((0, 2, 3)); t = dx_o_manual.pow(2).sum((0, 2, 3))
idx = torch.argsort(d, descending=True)[:8]
print('worst channels', [(int(i), float(d[i]), float(t[i]), float(var_o.flatten()[i])) for i in idx])
