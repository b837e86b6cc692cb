import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch
from cases import gen_inputs, out_weights
from detweights import det_state_dict
from oracle import nets as onets
from rdfc_gan_amd import functional as Fh, generator
Fh.set_compute_dtype(torch.float32)
B = int(os.environ.get('DBG_B', 1))
net, ref = generator.RDFGenerator(pretrained_on_imagenet=False), onets.RDFGenerator()
sd = det_state_dict(net.state_dict(), 0)
net.load_state_dict(sd); ref.load_state_dict(sd)
net.cuda().train(); ref.double().train()
G, R = {}, {}
def wrap(fn, store, tagbase):
    cnt = [0]
    def f(a, b):
        i = cnt[0]; cnt[0] += 1
        out = fn(a, b)
        for nm, t in (('out', out), ('fd', a), ('fe', b)):
            if t.requires_grad:
                t.register_hook(lambda g, k=f'{i}.{nm}': store.setdefault(k, []).append((tuple(g.shape), g.stride(), g.detach().clone())))
        return out
    return f
Fh_crop = Fh.crop_cat
Fh.crop_cat = wrap(Fh_crop, G, 'g')
onets._crop_cat = wrap(onets._crop_cat, R, 'r')
inputs = gen_inputs('g_a2b', B, 60, 76)
def go(m, ins, dev, dtype):
    ins = [t.to(dev).to(dtype).requires_grad_(True) for t in ins]
    o = m(*ins); loss = 0
    for k, v in o.items():
        loss = loss + (v * out_weights('dbg', k, v.shape).to(dev).to(dtype)).sum()
    loss.backward()
go(net, inputs, 'cuda', torch.float32); go(ref, inputs, 'cpu', torch.float64)
for k in sorted(G, key=lambda s: (int(s.split('.')[0]), s)):
    if k not in R: continue
    for j, (gs, rs) in enumerate(zip(G[k], R[k])):
        g = gs[2].cpu().double(); r = rs[2]
        if g.shape != r.shape: g = g.permute(0, 3, 1, 2)
        print(f'{k:8s}#{j} shape {gs[0]} stride {gs[1]}  L2rel {float((g - r).norm() / r.norm()):.2e}')
