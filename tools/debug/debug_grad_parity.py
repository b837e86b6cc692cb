"""Debug aid (GPU box): per-parameter gradient error of the HIP RDFGenerator (f32 mode) against the
CPU oracle on the same weights / inputs, in registration order -- localises a backward bug."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch
from cases import gen_inputs, out_weights, NLSPN_CFG
from detweights import det_state_dict
from oracle import nets as onets
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd.generator import RDFGenerator, ResnetGenerator

which = sys.argv[1] if len(sys.argv) > 1 else 'a2b'
Fh.set_compute_dtype(torch.float32)
if which == 'a2b':
    net, ref = RDFGenerator(pretrained_on_imagenet=False), onets.RDFGenerator()
    B_, H_, W_ = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (2, 60, 76)))
    inputs = gen_inputs('g_a2b', B_, H_, W_)
else:
    net, ref = ResnetGenerator(1, 3, 64, 'IN2d'), onets.ResnetGenerator(1, 3, 64, 'IN2d')
    inputs = (gen_inputs('g_b2a', 2, 60, 76)[1],)
sd = det_state_dict(net.state_dict(), 0)
net.load_state_dict(sd); ref.load_state_dict(sd)
net.cuda().train(); ref.double().train()

def go(m, ins, dev, dtype):
    ins = [t.to(dev).to(dtype).requires_grad_(True) for t in ins]
    o = m(*ins)
    o = o if isinstance(o, dict) else {'out': o}
    loss = 0
    for k, v in o.items():
        loss = loss + (v * out_weights('dbg', k, v.shape).to(dev).to(dtype)).sum()
    loss.backward()
    return ins, o
gi, go_ = go(net, inputs, 'cuda', torch.float32)
ri, ro = go(ref, inputs, 'cpu', torch.float64)
for k in ro:
    print(f'out {k:22s} maxabs {float((go_[k].cpu().double() - ro[k]).abs().max()):.2e}')
rp = dict(ref.named_parameters())
for k, p in net.named_parameters():
    if p.grad is None:
        continue
    a, b = p.grad.cpu().double(), rp[k].grad
    print(f'{k:70s} L2rel {float((a - b).norm() / (b.norm() + 1e-30)):.2e}  |ref| {float(b.norm()):.2e}')
for i, (a, b) in enumerate(zip(gi, ri)):
    if a.grad is not None and b.grad is not None:
        print(f'grad_in.{i} L2rel {float((a.grad.cpu().double() - b.grad).norm() / b.grad.norm()):.2e}')
