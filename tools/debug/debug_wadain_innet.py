import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch, torch.nn.functional as F
from cases import gen_inputs, out_weights
from detweights import det_state_dict
from rdfc_gan_amd import functional as Fh, generator, hip
Fh.set_compute_dtype(torch.float32)
B = int(os.environ.get('DBG_B', 1))
net = generator.RDFGenerator(pretrained_on_imagenet=False)
net.load_state_dict(det_state_dict(net.state_dict(), 0)); net.cuda().train()
orig_bwd = Fh._WAdaIN.backward
def bwd(ctx, dout):
    x, style, w, gb, mean, rstd = ctx.saved_tensors
    res = orig_bwd(ctx, dout)
    N, H, W, C = x.shape
    with torch.enable_grad():
        xr = x.detach().double().permute(0,3,1,2).requires_grad_(True)
        gbr = gb.detach().double().permute(0,3,1,2)
        out = gbr[:, :C]*F.instance_norm(xr) + gbr[:, C:]
        (out*dout.detach().double().permute(0,3,1,2)).sum().backward()
    m2 = x.detach().double().mean((1,2)); 
    print(f'wadain bwd: x{tuple(x.shape)} stride {x.stride()} dout stride {dout.stride()} contiguous {dout.is_contiguous()} gb stride {gb.stride()} '
          f'mean err {float((mean.double()-m2).abs().max()):.1e}  dx err {float((res[0].double().permute(0,3,1,2)-xr.grad).norm()/xr.grad.norm()):.2e}')
    return res
Fh._WAdaIN.backward = staticmethod(bwd)
inputs = [t.cuda().requires_grad_(True) for t in gen_inputs('g_a2b', B, 60, 76)]
o = net(*inputs); loss = 0
for k, v in o.items():
    loss = loss + (v * out_weights('dbg', k, v.shape).cuda()).sum()
loss.backward()
