"""cProfile of the host side of ONE train step enqueued into an idle queue (after warm-up): which Python / binding calls the
enqueue time goes to.  `python tools/host_profile.py [n_rows]`"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from rdfc_gan_amd import functional as Fh
from rdfc_gan_amd.rdfc_gan import RDFCGAN

dev = torch.device('cuda', 0)
Fh.set_compute_dtype(torch.float16)
m = RDFCGAN(dev, False, bench.model_cfg(False), is_train=True)
m.train()
batches = [bench.synth_batch(8, 228, 304, 1234 + i, dev) for i in range(2)]


def step(i):
    m.set_input(batches[i % 2])
    return m.optimize_parameters(sync_losses=False)


for i in range(5):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
if os.environ.get('BACKWARD', '0') == '1':
    # the autograd engine runs the backward functions on its own thread, which the profiler of this thread does not see: profile
    # _ConvBlock.backward (two thirds of the backward's host time) from the inside instead
    prb = cProfile.Profile()
    fn = Fh._ConvBlock.backward

    def profiled(ctx, *g):
        return prb.runcall(fn, ctx, *g)
    Fh._ConvBlock.backward = staticmethod(profiled)
    for i in range(4):
        step(i)
        torch.cuda.synchronize()
    pstats.Stats(prb).sort_stats('tottime').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
    sys.exit(0)
for i in range(4):
    pr.enable()
    step(i)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
