"""Debug aid: gradient wrt the OUTPUT of every fused block, HIP (f32) vs CPU oracle (f64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch, torch.nn as nn
from cases import gen_inputs, out_weights
from detweights import det_state_dict
from oracle import nets as onets
from rdfc_gan_amd import functional as Fh, layers
from rdfc_gan_amd.generator import RDFGenerator, AdaptiveInstanceNorm

Fh.set_compute_dtype(torch.float32)
net, ref = RDFGenerator(pretrained_on_imagenet=False), onets.RDFGenerator()
sd = det_state_dict(net.state_dict(), 0)
net.load_state_dict(sd); ref.load_state_dict(sd)
net.cuda().train(); ref.double().train()
inputs = gen_inputs('g_a2b', int(os.environ.get('DBG_B', 2)), 60, 76)
G, R, GO, RO = {}, {}, {}, {}
def hook_mod(m, name, store, ostore, pm):
    def fh(mod, inp, out):
        if torch.is_tensor(out) and out.requires_grad:
            ostore[name] = out.detach()
            out.register_hook(lambda g, n=name: store.__setitem__(n, g.detach().clone()))
    m.register_forward_hook(fh)
names = {}
for name, m in net.named_modules():
    if isinstance(m, (layers.ConvSeq, layers.BasicBlock, AdaptiveInstanceNorm)):
        hook_mod(m, name, G, GO, True); names[name] = 1
for name, m in ref.named_modules():
    if name in names:
        hook_mod(m, name, R, RO, False)
def go(m, ins, dev, dtype):
    ins = [t.to(dev).to(dtype).requires_grad_(True) for t in ins]
    o = m(*ins); loss = 0
    for k, v in o.items():
        loss = loss + (v * out_weights('dbg', k, v.shape).to(dev).to(dtype)).sum()
    loss.backward()
go(net, inputs, 'cuda', torch.float32); go(ref, inputs, 'cpu', torch.float64)
for name in names:
    if name in G and name in R:
        g = G[name].cpu().double(); r = R[name]
        if g.dim() == 4 and g.shape != r.shape: g = g.permute(0, 3, 1, 2)
        o = GO[name].cpu().double(); ro = RO[name]
        if o.dim() == 4 and o.shape != ro.shape: o = o.permute(0, 3, 1, 2)
        d = (g - r).abs()
        print(f'{name:50s} out L2rel {float((o-ro).norm()/ro.norm()):.1e}  dOut L2rel {float((g-r).norm()/r.norm()):.2e}  frac(|err|>1e-3*max) {float((d > 1e-3*r.abs().max()).double().mean()):.2e}')
